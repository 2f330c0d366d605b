// Micro-benchmark (development): B operands re-read from LDS for every tile of a v_mfma_f32_32x32x2_f32 stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int KS = 22;

template <int LDSREAD, int NST, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(const float* tab, float* out, int tiles) {
    __shared__ float X[44 * 128 * 2];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 44 * 128 * 2; e += 64 * WAVES) X[e] = tab[e & 4095];
    __syncthreads();
    float a[KS], b[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a[s] = tab[s * 64 + lane]; b[s] = tab[(KS + s) * 64 + lane]; }
    float sum = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const float* xp = X + (t & 7) * 32 + (lane & 31) + (lane >> 5) * 256;
        asm volatile("" : "+v"(xp));
        if (LDSREAD == 1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) b[s] = xp[s * 512];
        }
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (LDSREAD == 2) {
            // reads of the NEXT tile's operands interleaved one per matrix instruction
            float bn[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                bn[s] = xp[s * 512];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], acc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) b[s] = bn[s];
        } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], acc, 0, 0, 0);
        }
        float* wb = out + ((size_t)(blockIdx.x * WAVES + (threadIdx.x >> 6)) * 64) * 1024 + (size_t)(t & 15) * 4096;
#pragma unroll
        for (int g = 0; g < NST; ++g) wb[g * 256 + (lane >> 4) * 64 + (lane & 15)] = acc[g & 15];
        if (NST == 0) sum += acc[0] + acc[5];
        __builtin_amdgcn_sched_barrier(0);
    }
    out[(size_t)(blockIdx.x * 64 * WAVES + threadIdx.x) * 1024 + 1000] = sum;
}

template <int LDSREAD, int NST, int WAVES>
void run(const float* tab, float* out) {
    const int tiles = 400;
    const int grid = 256;  // one workgroup per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<LDSREAD, NST, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, tab, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<LDSREAD, NST, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, tab, out, tiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)tiles * KS * WAVES / 4;
    printf("waves/CU %2d, LDS B reads %d, stores %2d: %.3f ms, %6.1f cyc/MFMA/SIMD\n", WAVES, LDSREAD, NST, ms, ms * 1e6 / per_simd * 2.4);
}

int main() {
    float *tab, *out;
    (void)hipMalloc(&tab, 64 * 4 * 4096); (void)hipMalloc(&out, (size_t)256 * 1024 * 1024 * 4);
    (void)hipMemset(tab, 0, 64 * 4 * 4096);
    run<0, 0, 8>(tab, out); run<1, 0, 8>(tab, out); run<2, 0, 8>(tab, out); run<2, 16, 8>(tab, out); run<2, 0, 16>(tab, out);
    return 0;
}
