// Micro-benchmark (development): cost of vector-memory instructions inside a v_mfma_f32_32x32x2_f32 stream
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int KS = 22;

// NST stores of WIDTH dwords per tile (private 4 KB region per thread: L2-resident), NLD loads of 4 dwords
template <int NST, int WIDTH, int NLD, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(const float* tab, float* out, int tiles) {
    const int lane = threadIdx.x & 63;
    float a[KS], b[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { a[s] = tab[s * 64 + lane]; b[s] = tab[(KS + s) * 64 + lane]; }
    float sum = 0.f;
    float* o = out + (size_t)(blockIdx.x * 64 * WAVES + threadIdx.x) * 1024;
    for (int t = 0; t < tiles; ++t) {
        const float* tp = tab;
        asm volatile("" : "+s"(tp));
        f4 ld[NLD > 0 ? NLD : 1];
        const f4* t4 = reinterpret_cast<const f4*>(tp + ((size_t)(t & 7) * 64 + lane) * 24);
#pragma unroll
        for (int j = 0; j < NLD; ++j) ld[j] = t4[j];
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
        float* ot = o + (t & 15) * 64;
#pragma unroll
        for (int g = 0; g < NST; ++g) {
            if (WIDTH == 4) { f4 w = {acc[(4*g)&15], acc[(4*g+1)&15], acc[(4*g+2)&15], acc[(4*g+3)&15]}; *reinterpret_cast<f4*>(ot + 4 * g) = w; }
            if (WIDTH == 5) {  // wave-contiguous: the 64 lanes of one store cover 1 KB
                f4 w = {acc[(4*g)&15], acc[(4*g+1)&15], acc[(4*g+2)&15], acc[(4*g+3)&15]};
                float* wb = out + ((size_t)(blockIdx.x * WAVES + (threadIdx.x >> 6)) * 64) * 1024 + (size_t)(t & 15) * 4096;
                *reinterpret_cast<f4*>(wb + g * 256 + 4 * lane) = w;
            }
            if (WIDTH == 2) { f2 w = {acc[(2*g)&15], acc[(2*g+1)&15]}; *reinterpret_cast<f2*>(ot + 2 * g) = w; }
            if (WIDTH == 1) ot[g] = acc[g & 15];
        }
        if (NST == 0) sum += acc[0] + acc[5];
#pragma unroll
        for (int j = 0; j < NLD; ++j) a[(4 * j) % KS] += ld[j].x * 1e-30f + ld[j].w * 1e-30f;
        __builtin_amdgcn_sched_barrier(0);
    }
    out[(size_t)(blockIdx.x * 64 * WAVES + threadIdx.x) * 1024 + 1000] = sum + a[0];
}

template <int NST, int WIDTH, int NLD, int WAVES>
void run(const float* tab, float* out) {
    const int tiles = 400;
    const int grid = 256 * 16 / WAVES;  // 16 waves per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NST, WIDTH, NLD, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, tab, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NST, WIDTH, NLD, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, tab, out, tiles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)tiles * KS * 4;
    const double cyc = ms * 1e6 / per_simd * 2.4;
    printf("stores %2d x %d dw, loads %d x 4 dw: %.3f ms, %6.1f cyc/MFMA/SIMD, extra per tile per wave %.0f cyc\n", NST, WIDTH, NLD, ms, cyc,
           (cyc - 66.0) * KS);
}

int main() {
    float *tab, *out;
    (void)hipMalloc(&tab, 64 * 4 * 4096); (void)hipMalloc(&out, (size_t)256 * 1024 * 1024 * 4);
    (void)hipMemset(tab, 0, 64 * 4 * 4096);
    run<0, 4, 0, 8>(tab, out);
    run<4, 4, 0, 8>(tab, out); run<4, 5, 0, 8>(tab, out); run<2, 5, 0, 8>(tab, out); run<8, 5, 0, 8>(tab, out);
    run<4, 5, 0, 4>(tab, out); run<4, 4, 0, 4>(tab, out);
    return 0;
}
