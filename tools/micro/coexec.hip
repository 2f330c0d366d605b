// Micro-benchmark (development): what runs BESIDE a stream of v_mfma_f32_32x32x2_f32 on the same SIMD when the
// other work lives in a DIFFERENT wave (role-specialised waves), against the same work fused into the matrix wave.
//
// One 512-thread workgroup per CU (8 waves: waves 0-3 land one per SIMD, waves 4-7 likewise).  Roles:
//   M  matrix wave: per tile 22 x v_mfma_f32_32x32x2_f32 (the 10 -> 14 composite of the round-3 wpt3_deep_kernel, K = 44)
//   E  epilogue wave: per tile, for 16 values per lane, log2(v*v + eps)*k1 + k0, four 4x4 quad transposes over DPP
//      and four 16-byte stores (each instruction = two 512-byte runs of whole lines, streaming to HBM)
//   V  the epilogue's vector arithmetic only (no stores);  S  the stores only
//   P  packed-FMA wave: per item 7 ds_read_b128 + 48 v_pk_fma_f32 + 2 ds_write_b64 (one output pair of both
//      children in wpt3_top_kernel, 24 taps)
//   F  fused: matrix tile followed by its epilogue in the same wave (what the round-3 wpt3_deep_kernel did; retired in round 4 for wpt4.hip)
// A configuration gives the role of waves 0-3 and of waves 4-7 ('-' = the waves exit at once).
// Build: hipcc -O3 --offload-arch=gfx950 coexec.hip -o coexec ; run on an MI355X.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int KS = 22;

enum Role { R_NONE = 0, R_M, R_E, R_V, R_S, R_P, R_F, R_M2 };

__device__ __forceinline__ void quad_transpose4(float& v0, float& v1, float& v2, float& v3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    auto xchg = [](float keep_lo, float keep_hi, bool hi, int two, float& out_lo, float& out_hi) {
        const float send = hi ? keep_lo : keep_hi;
        const int si = __builtin_bit_cast(int, send);
        const int ri = two ? __builtin_amdgcn_mov_dpp(si, 0x4E, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(si, 0xB1, 0xF, 0xF, true);
        const float recv = __builtin_bit_cast(float, ri);
        out_lo = hi ? recv : keep_lo;
        out_hi = hi ? keep_hi : recv;
    };
    float a0, a1, a2, a3;
    xchg(v0, v1, b0, 0, a0, a1);
    xchg(v2, v3, b0, 0, a2, a3);
    xchg(a0, a2, b1, 1, v0, v2);
    xchg(a1, a3, b1, 1, v1, v3);
}

template <bool VALU, bool STORE>
__device__ __forceinline__ void epilogue(f16v acc, float* __restrict__ o, int lane, float eps, float k1, float k0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v0 = acc[4 * g], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
        f4 w;
        if (VALU) {
            quad_transpose4(v0, v1, v2, v3, lane);
            w.x = fmaf(__builtin_amdgcn_logf(fmaf(v0, v0, eps)), k1, k0);
            w.y = fmaf(__builtin_amdgcn_logf(fmaf(v1, v1, eps)), k1, k0);
            w.z = fmaf(__builtin_amdgcn_logf(fmaf(v2, v2, eps)), k1, k0);
            w.w = fmaf(__builtin_amdgcn_logf(fmaf(v3, v3, eps)), k1, k0);
        } else {
            w.x = v0; w.y = v1; w.z = v2; w.w = v3;
        }
        if (STORE) {
            // lanes 0-31: one 512-byte run, lanes 32-63: another one 64 KB further (two time rows of the features)
            *reinterpret_cast<f4*>(o + g * 256 + (lane >> 5) * 16384 + (lane & 31) * 4) = w;
        } else {
            asm volatile("" ::"v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));
        }
    }
}

template <int L>
struct Window {
    static constexpr int NV = (L + 2 + 3) / 4;
    float w[4 * NV];
    __device__ __forceinline__ void load(const float* __restrict__ src) {
        const f4* s4 = reinterpret_cast<const f4*>(src);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f4 x = s4[v];
            w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w;
        }
    }
    // the same dot product as 24 plain v_fma_f32 in two chains (inline asm: hipcc -O3 SLP-packs fmaf pairs)
    template <int O>
    __device__ __forceinline__ float dot_scalar(const float* taps) const {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int t = 0; t < L / 2; ++t) {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(w[2 * t + 2 * O]), "s"(taps[2 * t]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(w[2 * t + 1 + 2 * O]), "s"(taps[2 * t + 1]));
        }
        return a0 + a1;
    }
    template <int O>
    __device__ __forceinline__ float dot(const float* taps) const {
        f2 acc = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < L / 2; ++t) {
            const f2 tp = {taps[2 * t], taps[2 * t + 1]};
            const f2 xv = {w[2 * t + 2 * O], w[2 * t + 1 + 2 * O]};
            acc = __builtin_elementwise_fma(tp, xv, acc);
        }
        return acc.x + acc.y;
    }
};

struct Taps { float lo[24], hi[24]; };

template <int ROLE_A, int ROLE_B, int PRIO_B = 0, bool SCALAR_FMA = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
k(const float* __restrict__ tab, float* __restrict__ out, int tiles, size_t wave_stride, const Taps tp,
  unsigned long long* __restrict__ stamps) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 2048];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    for (int i = threadIdx.x; i < 8 * 2048; i += 512) lds[i] = tab[i & 4095];
    __syncthreads();
    if (role == R_NONE) return;
    if (PRIO_B > 0 && wave >= 4) __builtin_amdgcn_s_setprio(PRIO_B);
    // diagnostic stamps (their own buffer; no output depends on them): shader clock, 100 MHz wall clock, HW_ID
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    float* o = out + ((size_t)blockIdx.x * 8 + wave) * wave_stride;
    const float eps = 1e-12f, k1 = 0.69314718f, k0 = 0.1f;
    if (role == R_M || role == R_F) {
        float a[KS], b[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { a[s] = tab[s * 64 + lane]; b[s] = tab[(KS + s) * 64 + lane]; }
        for (int t = 0; t < tiles; ++t) {
            f16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
            if (role == R_F) {
                epilogue<true, true>(acc, o + (size_t)t * 1024, lane, eps, k1, k0);
            } else {
                asm volatile("" ::"v"(acc));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (role == R_M2) {
        // two independent accumulator chains interleaved in one wave (half the tiles each pass does two)
        float a[KS], b[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { a[s] = tab[s * 64 + lane]; b[s] = tab[(KS + s) * 64 + lane]; }
        for (int t = 0; t < tiles; t += 2) {
            f16v acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s], a[s], acc1, 0, 0, 0);
            }
            asm volatile("" ::"v"(acc0), "v"(acc1));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (role == R_E || role == R_V || role == R_S) {
        f16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = tab[r * 64 + lane] + 1.5f;
        for (int t = 0; t < tiles; ++t) {
            asm volatile("" : "+v"(acc));
            if (role == R_E) epilogue<true, true>(acc, o + (size_t)t * 1024, lane, eps, k1, k0);
            if (role == R_V) epilogue<true, false>(acc, o, lane, eps, k1, k0);
            if (role == R_S) epilogue<false, true>(acc, o + (size_t)t * 1024, lane, eps, k1, k0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (role == R_P) {
        // per "tile" 8 items per lane (the vector levels' share of a deep tile is set by the caller through `tiles`)
        float* mine = lds + wave * 2048;
        for (int t = 0; t < tiles; ++t) {
#pragma unroll 1
            for (int it = 0; it < 4; ++it) {
                Window<24> win;
                win.load(mine + 4 * lane + 64 * it);
                const float ca0 = SCALAR_FMA ? win.dot_scalar<0>(tp.lo) : win.dot<0>(tp.lo);
                const float cd0 = SCALAR_FMA ? win.dot_scalar<0>(tp.hi) : win.dot<0>(tp.hi);
                const float ca1 = SCALAR_FMA ? win.dot_scalar<1>(tp.lo) : win.dot<1>(tp.lo);
                const float cd1 = SCALAR_FMA ? win.dot_scalar<1>(tp.hi) : win.dot<1>(tp.hi);
                *reinterpret_cast<f2*>(mine + 1024 + 2 * lane + 128 * (it & 1)) = f2{ca0, ca1};
                *reinterpret_cast<f2*>(mine + 1024 + 512 + 2 * lane + 128 * (it & 1)) = f2{cd0, cd1};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (lane == 0) {
        unsigned long long* st = stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
        st[0] = __builtin_amdgcn_s_memtime() - t0;
        st[1] = __builtin_amdgcn_s_memrealtime() - r0;
        st[2] = hwid;
        st[3] = role;
    }
}

template <int A, int B, int PRIO_B = 0, bool SCALAR_FMA = false>
double run(const char* name, const float* tab, float* out, int tiles, size_t wave_stride, double mfma_waves, double epi_waves) {
    Taps tp;
    for (int i = 0; i < 24; ++i) { tp.lo[i] = 0.01f * (i + 1); tp.hi[i] = (i & 1) ? -0.02f * i : 0.02f * i; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    static unsigned long long* stamps = nullptr;
    if (!stamps) (void)hipMalloc(&stamps, 256 * 8 * 4 * 8);
    (void)hipMemset(stamps, 0, 256 * 8 * 4 * 8);
    hipLaunchKernelGGL((k<A, B, PRIO_B, SCALAR_FMA>), dim3(256), dim3(512), 0, 0, tab, out, 8, wave_stride, tp, stamps);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<A, B, PRIO_B, SCALAR_FMA>), dim3(256), dim3(512), 0, 0, tab, out, tiles, wave_stride, tp, stamps);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double us_per_tile = best * 1e3 / tiles;
    printf("%-34s %8.3f ms  %7.3f us per tile-slot", name, best, us_per_tile);
    if (mfma_waves > 0) printf("  matrix: %5.1f cyc/MFMA/SIMD @2.4GHz", best * 1e-3 * 2.4e9 / (tiles * KS * mfma_waves / 4.0));
    if (epi_waves > 0) printf("  stores: %5.2f TB/s", 256.0 * epi_waves * tiles * 4096.0 / (best * 1e-3) / 1e12);
    printf("\n");
    {
        static unsigned long long h[256 * 8 * 4];
        (void)hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
        // block 0's waves: SIMD of each wave, shader cycles per tile, clock = cycles / wall
        printf("      block 0:");
        for (int w = 0; w < 8; ++w) {
            const unsigned long long* st = h + w * 4;
            if (!st[3]) { printf(" [w%d -]", w); continue; }
            printf(" [w%d simd%llu cu%llu %.0f cyc/tile %.2f GHz]", w, (st[2] >> 4) & 3, (st[2] >> 8) & 15,
                   (double)st[0] / tiles, st[1] ? (double)st[0] / (double)st[1] * 0.1 : 0.0);
        }
        // distinct (se, sh, cu) triples over all blocks = CUs in use
        int used = 0; static bool seen[1 << 16];
        for (int i = 0; i < (1 << 16); ++i) seen[i] = false;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w) {
                const unsigned long long* st = h + ((size_t)b * 8 + w) * 4;
                if (!st[3]) continue;
                const unsigned key = (unsigned)((st[2] >> 8) & 0xFF) | (unsigned)(b % 8) << 8;  // cu/sh/se bits + XCD slot
                if (!seen[key]) { seen[key] = true; ++used; }
            }
        printf("  distinct CUs (by HW_ID bits 8-15 x XCD slot): %d\n", used);
    }
    return best;
}

int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 1000;
    const size_t wave_stride = (size_t)tiles * 1024 + 32768;  // floats per wave's output region
    float *tab, *out;
    (void)hipMalloc(&tab, 64 * 4 * 4096);
    const size_t out_floats = (size_t)256 * 8 * wave_stride;
    if (hipMalloc(&out, out_floats * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    float* h = (float*)malloc(64 * 4096 * 4);
    srand(1);
    for (int i = 0; i < 64 * 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(tab, h, 64 * 4096 * 4, hipMemcpyHostToDevice);
    printf("tiles per wave %d; a tile = 22 MFMA 32x32x2 f32 (1408 cycles at the pipe rate) / 16 outputs per lane\n", tiles);
    run<R_M, R_NONE>("M  -   matrix, 1 wave/SIMD", tab, out, tiles, wave_stride, 4, 0);
    run<R_M, R_M>("M  M   matrix, 2 waves/SIMD", tab, out, tiles, wave_stride, 8, 0);
    run<R_M2, R_NONE>("M2 -   matrix, 2 chains in 1 wave", tab, out, tiles, wave_stride, 4, 0);
    run<R_M2, R_M2>("M2 M2  2 chains x 2 waves/SIMD", tab, out, tiles, wave_stride, 8, 0);
    run<R_NONE, R_E>("-  E   epilogue, 1 wave/SIMD", tab, out, tiles, wave_stride, 0, 4);
    run<R_E, R_E>("E  E   epilogue, 2 waves/SIMD", tab, out, tiles, wave_stride, 0, 8);
    run<R_NONE, R_V>("-  V   epilogue arithmetic only", tab, out, tiles, wave_stride, 0, 0);
    run<R_NONE, R_S>("-  S   stores only, 1 wave/SIMD", tab, out, tiles, wave_stride, 0, 4);
    run<R_S, R_S>("S  S   stores only, 2 waves/SIMD", tab, out, tiles, wave_stride, 0, 8);
    run<R_M, R_E>("M  E   matrix beside epilogue", tab, out, tiles, wave_stride, 4, 4);
    run<R_M, R_V>("M  V   matrix beside arithmetic", tab, out, tiles, wave_stride, 4, 0);
    run<R_M, R_S>("M  S   matrix beside stores", tab, out, tiles, wave_stride, 4, 4);
    run<R_F, R_NONE>("F  -   fused, 1 wave/SIMD", tab, out, tiles, wave_stride, 4, 4);
    run<R_F, R_F>("F  F   fused, 2 waves/SIMD (2x work)", tab, out, tiles, wave_stride, 8, 8);
    run<R_NONE, R_P>("-  P   packed FMA, 1 wave/SIMD", tab, out, tiles, wave_stride, 0, 0);
    run<R_P, R_P>("P  P   packed FMA, 2 waves/SIMD", tab, out, tiles, wave_stride, 0, 0);
    run<R_M, R_P>("M  P   matrix beside packed FMA", tab, out, tiles, wave_stride, 4, 0);
    printf("-- waves 4-7 at s_setprio 3 --\n");
    run<R_M, R_E, 3>("M  E^  matrix beside epilogue", tab, out, tiles, wave_stride, 4, 4);
    run<R_M, R_V, 3>("M  V^  matrix beside arithmetic", tab, out, tiles, wave_stride, 4, 0);
    run<R_M, R_S, 3>("M  S^  matrix beside stores", tab, out, tiles, wave_stride, 4, 4);
    run<R_M, R_M, 3>("M  M^  matrix, 2 waves/SIMD", tab, out, tiles, wave_stride, 8, 0);
    run<R_F, R_F, 3>("F  F^  fused, 2 waves/SIMD (2x work)", tab, out, tiles, wave_stride, 8, 8);
    run<R_M, R_P, 3>("M  P^  matrix beside packed FMA", tab, out, tiles, wave_stride, 4, 0);
    printf("-- plain v_fma_f32 instead of v_pk_fma_f32 --\n");
    run<R_NONE, R_P, 0, true>("-  Ps  scalar FMA, 1 wave/SIMD", tab, out, tiles, wave_stride, 0, 0);
    run<R_P, R_P, 0, true>("Ps Ps  scalar FMA, 2 waves/SIMD", tab, out, tiles, wave_stride, 0, 0);
    run<R_M, R_P, 3, true>("M  Ps^ matrix beside scalar FMA", tab, out, tiles, wave_stride, 4, 0);
    return 0;
}
