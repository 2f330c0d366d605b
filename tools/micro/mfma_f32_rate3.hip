// Micro-benchmark (development): what slows a v_mfma_f32_32x32x2_f32 stream -- per-tile loads, copies, stores
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int KS = 22, KP = 24;

// MODE bits: 1 = dword prefetch loads, 2 = dwordx4 prefetch loads, 4 = consume prefetch (ping-pong, no copies),
//            8 = 4 float4 stores per tile, 16 = 48 VALU ops per tile (epilogue)
template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
k(const float* tab, float* out, int tiles) {
    const int lane = threadIdx.x & 63;
    float fa[2][KP], b[KS];
#pragma unroll
    for (int s = 0; s < KP; ++s) { fa[0][s] = tab[s * 64 + lane]; fa[1][s] = tab[(64 + s) * 64 + lane]; }
#pragma unroll
    for (int s = 0; s < KS; ++s) b[s] = tab[(KS + s) * 64 + lane];
    float sum = 0.f;
    float* o = out + (size_t)(blockIdx.x * 512 + threadIdx.x) * 64;
#pragma unroll 2
    for (int t = 0; t < tiles; ++t) {
        const float* tp = tab;
        asm volatile("" : "+s"(tp));
        const int cur = t & 1;
        float tmp[KP];
        if (MODE & 1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) tmp[s] = tp[((t & 7) * KP + s) * 64 + lane];
        }
        if (MODE & 2) {
            const f4* t4 = reinterpret_cast<const f4*>(tp + ((size_t)(t & 7) * 64 + lane) * KP);
#pragma unroll
            for (int j = 0; j < KP / 4; ++j) { const f4 v = t4[j]; tmp[4*j] = v.x; tmp[4*j+1] = v.y; tmp[4*j+2] = v.z; tmp[4*j+3] = v.w; }
        }
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], b[s], acc, 0, 0, 0);
        if (MODE & 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(__builtin_amdgcn_logf(fmaf(acc[r], acc[r], 1e-12f)), 0.3f, 0.1f);
        }
        if (MODE & 8) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f4 w = {acc[4*g], acc[4*g+1], acc[4*g+2], acc[4*g+3]};
                *reinterpret_cast<f4*>(o + (size_t)(t & 3) * 16 + 4 * g) = w;
            }
        } else {
            sum += acc[0] + acc[5];
        }
        if (MODE & 3) {
            if (MODE & 4) {
#pragma unroll
                for (int s = 0; s < KS; ++s) fa[cur ^ 1][s] = tmp[s];
            } else {
                sum += tmp[0] + tmp[KS - 1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] += sum;
}

template <int MODE>
void run(const float* tab, float* out) {
    const int tiles = 600;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(512), 0, 0, tab, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(512), 0, 0, tab, out, tiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)tiles * KS * 4;
    printf("mode %2d: %.3f ms, %.1f cycles per MFMA per SIMD @2.4GHz\n", MODE, ms, ms * 1e6 / per_simd * 2.4);
}

int main() {
    float *tab, *out;
    (void)hipMalloc(&tab, 64 * 4 * 4096); (void)hipMalloc(&out, (size_t)512 * 512 * 64 * 4);
    (void)hipMemset(tab, 0, 64 * 4 * 4096);
    run<0>(tab, out); run<1>(tab, out); run<2>(tab, out); run<1 | 4>(tab, out); run<2 | 4>(tab, out);
    run<8>(tab, out); run<16>(tab, out); run<2 | 4 | 8 | 16>(tab, out);
    return 0;
}
