// Micro-benchmark (development): v_mfma_f32_32x32x2_f32 chains with distinct operand registers per step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int KS, int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
k(const float* tab, float* out, int tiles) {
    const int lane = threadIdx.x & 63;
    float a[KS], b[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a[s] = tab[s * 64 + lane]; b[s] = tab[(KS + s) * 64 + lane]; }
    float sum = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const float* tp = tab;
        asm volatile("" : "+s"(tp));
        float an[KS];
        if (MODE >= 1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) an[s] = tp[(2 * KS + (t & 7) * KS + s) * 64 + lane];
        }
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        if (MODE >= 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_logf(fmaf(acc[r], acc[r], 1e-12f));
        } else {
            sum += acc[0] + acc[5];
        }
        if (MODE >= 1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) a[s] = an[s];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int KS, int MODE>
void run(const float* tab, float* out) {
    const int tiles = 600;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KS, MODE>), dim3(512), dim3(512), 0, 0, tab, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KS, MODE>), dim3(512), dim3(512), 0, 0, tab, out, tiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)tiles * KS * 4;  // 2 workgroups x 8 waves per CU = 4 waves per SIMD
    printf("KS %d mode %d: %.3f ms, %.1f cycles per MFMA per SIMD @2.4GHz\n", KS, MODE, ms, ms * 1e6 / per_simd * 2.4);
}

int main() {
    float *tab, *out;
    (void)hipMalloc(&tab, 64 * 4 * 512); (void)hipMalloc(&out, 512 * 512 * 4);
    (void)hipMemset(tab, 0, 64 * 4 * 512);
    run<22, 0>(tab, out); run<22, 1>(tab, out); run<22, 2>(tab, out);
    return 0;
}
