// Micro-benchmark (development): issue rate of v_mfma_f32_32x32x2_f32 chains by waves per SIMD and
// independent accumulators per wave.  Build: hipcc -O3 --offload-arch=gfx950 mfma_f32_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float a0, float b0) {
    f16v acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int waves_per_simd, float* out) {
    const int threads = 256 * waves_per_simd;  // one workgroup per CU: waves_per_simd waves on each SIMD
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * CHAINS * waves_per_simd;
    printf("chains %d waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz), %.1f TF/s\n", CHAINS,
           waves_per_simd, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4,
           mfma_per_simd * 1024 * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    for (int w = 1; w <= 4; ++w) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
