// buffer_load_dword ... lds on gfx950: a row of floats goes global -> LDS without passing registers; lanes whose
// byte offset falls outside the descriptor's num_records (negative offsets included) must leave ZERO in LDS.
//   hipcc -O2 --offload-arch=gfx950 -o buf_lds buf_lds.hip && ./buf_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void row_to_lds(const float* base, int bytes, unsigned lds_addr, int voff) {
    const unsigned long long b = (unsigned long long)base;
    const i4 r = {(int)(unsigned)b, (int)((b >> 32) & 0xffffu), bytes, 0x00020000};
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r) : "memory");
}
__global__ void k(const float* __restrict__ a, float* __restrict__ o, int W, int shift) {
    __shared__ float t[4][64];
    const int lane = threadIdx.x;
    for (int r = 0; r < 4; ++r) t[r][lane] = -1.f;
    __syncthreads();
    row_to_lds(a, W * 4, (unsigned)(size_t)t[0], (lane - shift) * 4);            // left edge: lanes < shift out
    row_to_lds(a, W * 4, (unsigned)(size_t)t[1], (W - 60 + lane) * 4);           // right edge: lanes >= 60 out
    row_to_lds(a, 0, (unsigned)(size_t)t[2], lane * 4);                          // empty descriptor: all out
    row_to_lds(a + W, W * 4, (unsigned)(size_t)t[3], lane * 4);                  // second row, in range
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int r = 0; r < 4; ++r) o[r * 64 + lane] = t[r][lane];
}
int main() {
    const int W = 200, shift = 3;
    std::vector<float> h(2 * W);
    for (int i = 0; i < 2 * W; ++i) h[i] = 1.f + i;
    float *a, *o;
    hipMalloc(&a, h.size() * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, o, W, shift);
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const float e0 = l < shift ? 0.f : h[l - shift];
        const float e1 = l >= 60 ? 0.f : h[W - 60 + l];
        const float e2 = 0.f, e3 = h[W + l];
        if (r[l] != e0 || r[64 + l] != e1 || r[128 + l] != e2 || r[192 + l] != e3) {
            if (bad < 8) printf("lane %d: got %g %g %g %g want %g %g %g %g\n", l, r[l], r[64 + l], r[128 + l], r[192 + l], e0, e1, e2, e3);
            ++bad;
        }
    }
    printf(bad ? "buf_lds: %d lanes differ\n" : "buf_lds: ok (%d)\n", bad);
    return bad != 0;
}
