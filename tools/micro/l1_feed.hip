// How many bytes per clock a CU takes in through its vector-memory path (L2 -> L1 -> registers) with every CU busy:
// the number behind the costing of split-precision Winograd products in DESIGN section 7 (the filter-domain operands
// of wino44.hip come through this path: every matrix instruction takes a fresh 4 / 16 bytes per lane of them).
//   table   1.5 MB per workgroup slot, the same for all (L2-resident after the first pass; far larger than the 32 KB L1)
//           or 16 KB (L1-resident)
//   launch  one 512-thread workgroup per CU (256), each wave streams 16 bytes per lane with 8 loads in flight
//   output  bytes per shader clock and CU (s_memtime deltas of the slowest workgroup), and GB/s over the wall clock
// build: hipcc -O3 --offload-arch=gfx950 l1_feed.hip -o l1_feed
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int W>  // bytes per lane and load: 16, 8 or 4
__global__ void __launch_bounds__(512) feed_kernel(const float* __restrict__ table, size_t table_floats, int iters,
                                                   float* sink, long long* cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int FL = W / 4;
    const size_t stride = (size_t)8 * 64 * FL;  // floats a workgroup's 8 waves cover per step
    size_t off = ((size_t)wave * 64 + lane) * FL;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* p = table + off;
            if (W == 16) acc += *reinterpret_cast<const f4*>(p);
            else if (W == 8) { acc.x += p[0]; acc.y += p[1]; }
            else acc.x += p[0];
            off += stride;
            if (off >= table_floats) off -= table_floats;
        }
    }
    const long long t1 = clock64();
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int W>
static void run(const char* what, const float* table, size_t floats, float* sink, long long* dcyc) {
    const int blocks = 256, iters = 2000;
    feed_kernel<W><<<blocks, 512>>>(table, floats, 50, sink, dcyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    feed_kernel<W><<<blocks, 512>>>(table, floats, iters, sink, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> cyc(blocks);
    hipMemcpy(cyc.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    const long long worst = *std::max_element(cyc.begin(), cyc.end());
    const double bytes_per_wg = (double)iters * 8 * 512 * W;
    std::printf("%-34s %2d B/lane: %6.1f B per s_memtime tick and CU (slowest workgroup), %7.1f GB/s per CU, %6.2f TB/s chip\n",
                what, W, bytes_per_wg / (double)worst, bytes_per_wg / (ms * 1e6), bytes_per_wg * blocks / (ms * 1e9));
}

int main() {
    const size_t big = (size_t)3 * 128 * 1024, small = 4096;  // floats: 1.5 MB, 16 KB
    float *tb, *sink;
    long long* dcyc;
    hipMalloc(&tb, big * 4);
    hipMemset(tb, 0, big * 4);
    hipMalloc(&sink, 64);
    hipMalloc(&dcyc, 256 * sizeof(long long));
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    int wall = 0;
    hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    std::printf("shader clock %d kHz, s_memtime %d kHz\n", clk, wall);
    run<16>("1.5 MB table (L2-resident)", tb, big, sink, dcyc);
    run<8>("1.5 MB table (L2-resident)", tb, big, sink, dcyc);
    run<4>("1.5 MB table (L2-resident)", tb, big, sink, dcyc);
    run<16>("16 KB table (L1-resident)", tb, small, sink, dcyc);
    run<8>("16 KB table (L1-resident)", tb, small, sink, dcyc);
    run<4>("16 KB table (L1-resident)", tb, small, sink, dcyc);
    return 0;
}
