// Micro-benchmark (development): issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950 by waves per SIMD.
// Per wave: 16 independent accumulators (scalar form) or 8 independent accumulator pairs (packed form), the same
// 16 FMAs per lane per round either way.  Build: hipcc -O3 --offload-arch=gfx950 fma_rate.hip -o fma_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <bool PACKED>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float x0, float y0) {
    const float x = x0 + 1e-9f * threadIdx.x, y = y0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    if (PACKED) {
        f2 acc[8];
        for (int c = 0; c < 8; ++c) acc[c] = f2{0.f, (float)c};
        const f2 xv = {x, x * 1.0001f}, yv = {y, y * 0.5f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = __builtin_elementwise_fma(acc[c], xv, yv);
            }
        }
        for (int c = 0; c < 8; ++c) s += acc[c].x + acc[c].y;
    } else {
        float acc[16];
        for (int c = 0; c < 16; ++c) acc[c] = (float)c;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int c = 0; c < 16; ++c) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[c]) : "v"(x), "v"(y));  // plain fmaf is SLP-packed by hipcc -O3
            }
        }
        for (int c = 0; c < 16; ++c) s += acc[c];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<unsigned long long*>(out + 256 * 1024)[0] = t1 - t0;
}

template <bool PACKED>
void run(int waves_per_simd, float* out) {
    const int threads = 256 * waves_per_simd;
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PACKED>, dim3(256), dim3(threads), 0, 0, out, 10, 0.999f, 0.001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<PACKED>, dim3(256), dim3(threads), 0, 0, out, iters, 0.999f, 0.001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc = 0;
    (void)hipMemcpy(&cyc, out + 256 * 1024, 8, hipMemcpyDeviceToHost);
    const double fma_per_lane = (double)iters * 8 * 16;
    const double instr = PACKED ? fma_per_lane / 2 : fma_per_lane;  // wave-instructions per wave
    printf("%s waves/SIMD %d: %.3f ms, wave 0: %.2f cycles per instruction per wave, %.2f per SIMD; %.1f TFLOP/s (clock %.2f GHz)\n",
           PACKED ? "v_pk_fma_f32" : "v_fma_f32   ", waves_per_simd, ms, (double)cyc / instr, (double)cyc / instr / waves_per_simd,
           fma_per_lane * 2 * 64 * waves_per_simd * 4 * 256 / (ms * 1e-3) / 1e12, (double)cyc / (ms * 1e6));
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 1024 * 4 + 64);
    for (int w = 1; w <= 4; ++w) { run<false>(w, out); run<true>(w, out); }
    return 0;
}
