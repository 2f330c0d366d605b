// Micro-benchmark (development): one ds_read_b32 per v_mfma_f32_32x32x2_f32 with 1, 2 or 4 independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int KS = 24;

template <int CH, int WAVES, int READS>
__global__ void __launch_bounds__(64 * WAVES) k(const float* tab, float* out, int tiles) {
    __shared__ float X[48 * 128 * 2];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 48 * 128 * 2; e += 64 * WAVES) X[e] = tab[e & 4095];
    __syncthreads();
    float a[KS], b[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a[s] = tab[s * 64 + lane]; b[s] = tab[(KS + s) * 64 + lane]; }
    float sum = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const float* xp = X + (t & 7) * 32 + (lane & 31) + (lane >> 5) * 256;
        asm volatile("" : "+v"(xp));
        f16v acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        float bn[KS];
        // KS k-steps, each feeding CH independent accumulators (CH row tiles sharing the B operand);
        // READS LDS reads per k-step
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (READS >= 1) bn[s] = xp[s * 512];
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[(s + c) % KS], acc[c], 0, 0, 0);
            if (READS >= 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CH, 0);
        }
        if (READS >= 1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) b[s] = bn[s];
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) sum += acc[c][0] + acc[c][5];
        __builtin_amdgcn_sched_barrier(0);
    }
    out[(size_t)(blockIdx.x * 64 * WAVES + threadIdx.x)] = sum;
}

template <int CH, int WAVES, int READS>
void run(const float* tab, float* out) {
    const int tiles = 400;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CH, WAVES, READS>), dim3(256), dim3(64 * WAVES), 0, 0, tab, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<CH, WAVES, READS>), dim3(256), dim3(64 * WAVES), 0, 0, tab, out, tiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)tiles * KS * CH * WAVES / 4;
    printf("accumulators %d, waves/CU %2d, LDS reads per k-step %d (%.2f per MFMA): %6.1f cyc/MFMA/SIMD\n", CH, WAVES, READS,
           (double)READS / CH, ms * 1e6 / per_simd * 2.4);
}

int main() {
    float *tab, *out;
    (void)hipMalloc(&tab, 64 * 4 * 4096); (void)hipMalloc(&out, (size_t)256 * 1024 * 4);
    (void)hipMemset(tab, 0, 64 * 4 * 4096);
    run<1, 8, 0>(tab, out); run<1, 8, 1>(tab, out); run<2, 8, 1>(tab, out); run<3, 8, 1>(tab, out); run<4, 8, 1>(tab, out); run<6, 8, 1>(tab, out);
    run<2, 16, 1>(tab, out); run<4, 16, 1>(tab, out);
    return 0;
}
