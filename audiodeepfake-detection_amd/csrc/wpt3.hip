// Wavelet-packet front end, third generation (reference src/audiofakedetect/wavelet_math.py:167-263,
// :380-382; same contract as wpt.hip).
//
// Two kernels, both without any per-tap index arithmetic:
//
//   wpt3_top_kernel   levels 1..Ks (Ks <= 8) on the vector ALU.  Workgroup = (frame, level-1 half).
//       Every node lives in LDS time-major WITH ITS REFLECT EXTENSION MATERIALISED: the producer of a
//       node writes each coefficient to its own slot and, for the L-2 coefficients next to a border,
//       also to the mirrored pad slot.  An output pair is then a plain dot product of 24 contiguous
//       samples: L/2 aligned ds_read_b64 + L packed FMAs (the pair (x[2t], x[2t+1]) against the
//       reversed tap pairs of both filters), lanes run along the output index (conflict-free), no
//       border case anywhere.  The last level runs lanes along the nodes so that its stores -- the
//       features [B][C][T][P] or the level-8 hand-off [B][n8][256] -- are contiguous in packets.
//
//   wpt3_deep_kernel  levels 9..14 of the level-14 transform of 1 s frames on the matrix cores
//       (v_mfma_f32_32x32x2_f32: exact fp32 products).  An analysis step is the same linear map for
//       every node of a level, children = A_k parent, so a level is the GEMM Y = A_k X with the nodes
//       as columns.  Three phases per workgroup (32 level-8 nodes of one frame, 8 waves):
//         8 -> 9   A [2 n9 x n8], banded row tiles, 32 columns
//         9 -> 10  A [2 n10 x n9], 64 columns
//         10 -> 14 ONE composite matrix C [16 n14 x n10] = the product of the four level matrices along
//                  each of the 16 filter paths (deep nodes are 24..44 samples long: the composite has
//                  fewer entries than the four steps it replaces, 16 896 against 21 800 products per
//                  level-10 node for coif4), 128 columns, rows ordered (time, packet) so that a lane's
//                  four consecutive accumulator registers are four neighbouring packets -> float4 stores.
//       Matrices are built on the host in double precision from the taps, once per wavelet and device.
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kMaxTaps = 32;
constexpr int kTopThreads = 1024;
constexpr int kTopLdsFloats = 40960;  // 163 840 B: all of a CU's LDS
constexpr int kKsMax = 8;

struct Epi {
    unsigned flags;
    float power, eps, k1, k0, mean, inv_std, sgn_neg, sgn_pos;
};

__device__ __noinline__ float pow_log_slow3(float v, float power, float eps) {
    return logf(powf(fabsf(v), power) + eps);
}

// log(|v|^power + eps) and (x - mean) / std.  power == 2: v*v + eps >= 1e-12 is a normal float, the bare
// v_log_f32 (log2, ~1 ulp) needs no denormal pre-scaling; ln 2, 1/std and -mean/std are folded into one FMA.
__device__ __forceinline__ float epi_value(float v, const Epi& e) {
    if (e.flags & AFD_WPT_LOG) {
        if (e.power == 2.0f) return fmaf(__builtin_amdgcn_logf(fmaf(v, v, e.eps)), e.k1, e.k0);
        v = pow_log_slow3(v, e.power, e.eps);
    }
    if (e.flags & AFD_WPT_NORM) v = (v - e.mean) * e.inv_std;
    return v;
}

// ------------------------------------------------------------------------------------------------
// top levels
// ------------------------------------------------------------------------------------------------
struct T3Params {
    const float* x;
    float* dst;  // features (final) or the level-Ks hand-off image
    int B, N, Ks, final_;
    int n[kKsMax + 1];
    int off[kKsMax + 1];    // LDS offset (floats) of level k's image; sample 0 of node 0 at off + L - 2
    int pitch[kKsMax + 1];  // node pitch (floats): even, pitch / 2 odd
    unsigned magic[kKsMax + 1];  // floor(2^32 / n[k]) + 1
    Epi e;
    float rlo[kMaxTaps], rhi[kMaxTaps];  // taps reversed: rlo[t] = dec_lo[L - 1 - t]
};

// one coefficient of a child node of length n: its own slot and the pad slots that mirror it
template <int L>
__device__ __forceinline__ void put(float* node, int i, int n, float v) {
    constexpr int PAD = L - 2;
    node[i] = v;
    if (i >= 1 && i <= PAD) node[-i] = v;
    const int j = n - 1 - i;
    if (j >= 1 && j <= PAD + (n & 1)) node[n - 1 + j] = v;
}

template <int L>
__device__ __forceinline__ void dot_both(const T3Params& p, const float* __restrict__ src, float& ca, float& cd) {
    const f2* s2 = reinterpret_cast<const f2*>(src);
    f2 accA = {0.f, 0.f}, accD = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < L / 2; ++t) {
        const f2 xv = s2[t];
        const f2 tl = {p.rlo[2 * t], p.rlo[2 * t + 1]};
        const f2 th = {p.rhi[2 * t], p.rhi[2 * t + 1]};
        accA = __builtin_elementwise_fma(tl, xv, accA);
        accD = __builtin_elementwise_fma(th, xv, accD);
    }
    ca = accA.x + accA.y;
    cd = accD.x + accD.y;
}

template <int L, bool HI>
__device__ __forceinline__ float dot_one(const T3Params& p, const float* __restrict__ src) {
    const f2* s2 = reinterpret_cast<const f2*>(src);
    f2 acc = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < L / 2; ++t) {
        const f2 tp = {HI ? p.rhi[2 * t] : p.rlo[2 * t], HI ? p.rhi[2 * t + 1] : p.rlo[2 * t + 1]};
        acc = __builtin_elementwise_fma(tp, s2[t], acc);
    }
    return acc.x + acc.y;
}

template <int L>
__global__ void __launch_bounds__(kTopThreads) wpt3_top_kernel(const T3Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PAD = L - 2;
    const int tid = threadIdx.x;
    // the two halves of a frame are B workgroups apart: with B % 8 == 0 they share an XCD's L2
    const int b = blockIdx.x % p.B;
    const int h = blockIdx.x / p.B;

    // ---- frame -> level-0 image (with both reflect pads) ----
    {
        const float* xg = p.x + (size_t)b * p.N;
        float* X0 = lds + p.off[0] + PAD;
        if ((p.N & 1) == 0) {
            const float2* xv = reinterpret_cast<const float2*>(xg);  // frames are 8-byte aligned
            const int n2 = p.N >> 1;
            for (int base = 0; base < n2; base += kTopThreads * 8) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base + u * kTopThreads + tid;
                    v[u] = i < n2 ? xv[i] : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = base + u * kTopThreads + tid;
                    if (i < n2) *reinterpret_cast<float2*>(X0 + 2 * i) = v[u];
                }
            }
        } else {
            for (int i = tid; i < p.N; i += kTopThreads) X0[i] = xg[i];
        }
        if (tid >= 1 && tid <= PAD) X0[-tid] = xg[tid];
        const int j = tid - 64;
        if (j >= 1 && j <= PAD + (p.N & 1)) X0[p.N - 1 + j] = xg[p.N - 1 - j];
    }
    __syncthreads();

    // ---- level 1: this workgroup's child of the frame (h = 0 low-pass, 1 high-pass) ----
    if (p.Ks == 1) return;  // not built: the caller keeps one-level transforms on the first-generation kernel
    {
        const float* src = lds + p.off[0];
        float* node = lds + p.off[1] + PAD;
        const int n1 = p.n[1];
        if (h) {
            for (int i = tid; i < n1; i += kTopThreads) put<L>(node, i, n1, dot_one<L, true>(p, src + 2 * i));
        } else {
            for (int i = tid; i < n1; i += kTopThreads) put<L>(node, i, n1, dot_one<L, false>(p, src + 2 * i));
        }
    }
    __syncthreads();

    // ---- levels 2 .. Ks-1: both children of every node, lanes along the output index ----
    for (int k = 2; k < p.Ks; ++k) {
        const int Mp = 1 << (k - 2);  // parents (nodes of level k-1 in this half)
        const int nk = p.n[k];
        const int total = Mp * nk;
        const float* src0 = lds + p.off[k - 1];
        float* dst0 = lds + p.off[k] + PAD;
        const int pin = p.pitch[k - 1], pout = p.pitch[k];
        const unsigned magic = p.magic[k];
        for (int idx = tid; idx < total; idx += kTopThreads) {
            const int q = (int)__umulhi((unsigned)idx, magic);
            const int i = idx - q * nk;
            float ca, cd;
            dot_both<L>(p, src0 + q * pin + 2 * i, ca, cd);
            // odd-frequency parents list their children (d, a)
            const int par = (k == 2) ? h : (q & 1);
            put<L>(dst0 + (2 * q + par) * pout, i, nk, ca);
            put<L>(dst0 + (2 * q + 1 - par) * pout, i, nk, cd);
        }
        __syncthreads();
    }

    // ---- level Ks: lanes along the nodes, results leave the chip packet-contiguous ----
    {
        const int k = p.Ks;
        const int logM = k - 2;  // parents in this half: 2^(k-2)
        const int Mp = 1 << logM;
        const int nk = p.n[k];
        const int total = nk << logM;
        const float* src0 = lds + p.off[k - 1];
        const int pin = p.pitch[k - 1];
        const size_t P = (size_t)1 << k;
        const size_t chan = (size_t)nk * P;
        const int nch = (p.final_ && (p.e.flags & AFD_WPT_SIGN)) ? 2 : 1;
        float* outb = p.dst + (size_t)b * nch * chan;
        for (int idx = tid; idx < total; idx += kTopThreads) {
            const int q = idx & (Mp - 1);
            const int i = idx >> logM;
            float ca, cd;
            dot_both<L>(p, src0 + q * pin + 2 * i, ca, cd);
            const int par = (k == 2) ? h : (q & 1);
            f2 v;
            v.x = par ? cd : ca;
            v.y = par ? ca : cd;
            const size_t o = (size_t)i * P + 2 * ((size_t)(h << logM) + q);
            if (!p.final_) {
                *reinterpret_cast<f2*>(outb + o) = v;
            } else {
                f2 r;
                r.x = epi_value(v.x, p.e);
                r.y = epi_value(v.y, p.e);
                *reinterpret_cast<f2*>(outb + o) = r;
                if (p.e.flags & AFD_WPT_SIGN) {
                    f2 sg;
                    sg.x = v.x < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                    sg.y = v.y < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                    *reinterpret_cast<f2*>(outb + chan + o) = sg;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// deep levels on the matrix cores
// ------------------------------------------------------------------------------------------------
constexpr int kDeepWaves = 8;
constexpr int kDeepThreads = kDeepWaves * 64;
constexpr int kGroup = 32;  // level-8 nodes per workgroup

constexpr int refl_c(int j, int n) {
    j = j < 0 ? -j : j;
    return j >= n ? 2 * (n - 1) - j : j;
}

// node lengths of the standard 1 s frame (N = 22 050) at levels 8..14
template <int L> struct Shape3;
template <> struct Shape3<24> { static constexpr int L = 24; static constexpr int n[7] = {109, 66, 44, 33, 28, 25, 24}; };
template <> struct Shape3<10> { static constexpr int L = 10; static constexpr int n[7] = {95, 52, 30, 19, 14, 11, 10}; };
template <> struct Shape3<16> { static constexpr int L = 16; static constexpr int n[7] = {101, 58, 36, 25, 20, 17, 16}; };
template <int L> struct HasShape3 { static constexpr bool value = false; };
template <> struct HasShape3<24> { static constexpr bool value = true; };
template <> struct HasShape3<10> { static constexpr bool value = true; };
template <> struct HasShape3<16> { static constexpr bool value = true; };

// one stepwise level (J = 0: 8 -> 9, J = 1: 9 -> 10): 32-row tiles of A and the band of columns they touch
template <class SH, int J> struct Band3 {
    static constexpr int n_in = SH::n[J], n_out = SH::n[J + 1];
    static constexpr int KS = (n_in + 1) / 2;        // k-steps of the full matrix (two columns each)
    static constexpr int T = (2 * n_out + 31) / 32;  // 32-row tiles
    static constexpr int band(int t, bool hi) {
        int lo_c = n_in, hi_c = 0;
        for (int i = 16 * t; i < 16 * t + 16 && i < n_out; ++i)
            for (int m = 0; m < SH::L; ++m) {
                const int c = refl_c(2 * i + 1 - m, n_in);
                lo_c = c < lo_c ? c : lo_c;
                hi_c = c > hi_c ? c : hi_c;
            }
        return hi ? hi_c : lo_c;
    }
    static constexpr int steps(int t) { return (band(t, true) - (band(t, false) & ~1)) / 2 + 1; }
    static constexpr int max_steps() {
        int m = 0;
        for (int t = 0; t < T; ++t) m = steps(t) > m ? steps(t) : m;
        return m;
    }
    static constexpr int KSB = max_steps();  // k-steps issued per tile
    static constexpr int kstart(int t) {
        const int s0 = band(t, false) / 2;
        return s0 + KSB > KS ? KS - KSB : s0;
    }
};

template <class SH> struct Deep3 {
    using P1 = Band3<SH, 0>;
    using P2 = Band3<SH, 1>;
    static constexpr int n10 = SH::n[2], n14 = SH::n[6];
    static constexpr int KS3 = (n10 + 1) / 2;
    static constexpr int T3 = 16 * n14 / 32;
    static_assert((16 * n14) % 32 == 0, "composite rows fill whole tiles");
    static_assert(P1::T <= kDeepWaves && 2 * P2::T <= kDeepWaves, "one task per wave in the stepwise phases");
    // LDS images, position-major: X8 [R8][32] and X10 [R10][128] share region A, X9 [R9][64] is region B
    static constexpr int R8 = 2 * P1::KS, R9 = 2 * P2::KS, R10 = 2 * KS3;
    static constexpr int A_FLOATS = (R8 * 32 > R10 * 128) ? R8 * 32 : R10 * 128;
    static constexpr int B_FLOATS = R9 * 64;
    static constexpr int off1 = 0;
    static constexpr int off2 = off1 + P1::T * P1::KSB * 64;
    static constexpr int off3 = off2 + P2::T * P2::KSB * 64;
    static constexpr int tab_floats = off3 + T3 * KS3 * 64;
};

struct D3Params {
    const float* ws;
    const float* tab;
    float* out;
    short kst1[kDeepWaves], kst2[kDeepWaves];
    Epi e;
};

// stepwise level for one wave: row tile mt of A (fragments a[]) times the 32 columns [32 nt, 32 nt + 32) of
// src (row stride 1 << LOGS); children written position-major to dst (row stride 1 << LOGD)
template <class LV, int LOGS, int LOGD>
__device__ __forceinline__ void step_tile(const float (&a)[LV::KSB], int mt, int kst, int nt,
                                          const float* __restrict__ src, float* __restrict__ dst, int lane) {
    constexpr int S = 1 << LOGS;
    const int half = lane >> 5, col = lane & 31;
    const float* bp = src + ((size_t)kst * 2 + half) * S + nt * 32 + col;
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < LV::KSB; ++st)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], bp[st * 2 * S], acc, 0, 0, 0);
    // D: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); rows (2i, 2i+1) = (cA[i], cD[i])
    const int n = nt * 32 + col;
    int i0 = mt * 16 + half * 2;
    asm volatile("" : "+v"(i0));
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const int i = i0 + (r >> 2) * 4 + ((r & 3) >> 1);
        if (i < LV::n_out) {
            // odd-frequency parents list their children (d, a); values pinned so that the select is
            // not turned into a dynamic vector index
            float e0 = acc[r], e1 = acc[r + 1];
            asm volatile("" : "+v"(e0), "+v"(e1));
            f2 v;
            v.x = (col & 1) ? e1 : e0;
            v.y = (col & 1) ? e0 : e1;
            *reinterpret_cast<f2*>(dst + ((size_t)i << LOGD) + 2 * n) = v;
        }
    }
}

template <int KS>
__device__ __forceinline__ void load_frags(float (&a)[KS], const float* tab, int mt, int lane) {
    const float* t = tab + (size_t)mt * KS * 64 + lane;
#pragma unroll
    for (int st = 0; st < KS; ++st) a[st] = t[st * 64];
}

template <class SH>
__global__ void __launch_bounds__(kDeepThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
wpt3_deep_kernel(const D3Params p) {
    using D = Deep3<SH>;
    using P1 = typename D::P1;
    using P2 = typename D::P2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int groups = 256 / kGroup;
    const int b = blockIdx.x / groups;
    const int grp = blockIdx.x - b * groups;
    float* XA = lds;
    float* XB = lds + D::A_FLOATS;
    // rows past a node's length meet zero matrix columns but must hold finite numbers
    for (int e = tid; e < D::A_FLOATS + D::B_FLOATS; e += kDeepThreads) lds[e] = 0.f;
    // fragments of the first phase are on their way while the level-8 nodes arrive
    float a1[P1::KSB];
    const int mt1 = wave < P1::T ? wave : 0;
    load_frags<P1::KSB>(a1, p.tab + D::off1, mt1, lane);
    __syncthreads();
    {
        constexpr int n8 = SH::n[0];
        const float* wsb = p.ws + (size_t)b * n8 * 256 + grp * kGroup;
        for (int e = tid; e < n8 * kGroup; e += kDeepThreads) {
            const int pos = e >> 5, j = e & 31;
            XA[e] = wsb[(size_t)pos * 256 + j];
        }
    }
    float a2[P2::KSB];
    const int mt2 = wave % P2::T;
    load_frags<P2::KSB>(a2, p.tab + D::off2, mt2, lane);
    __syncthreads();
    // 8 -> 9: one row tile per wave, 32 columns -> 64
    if (wave < P1::T) step_tile<P1, 5, 6>(a1, mt1, p.kst1[mt1], 0, XA, XB, lane);
    __syncthreads();
    // X8 is dead: clear what the level-10 image leaves untouched but the composite's k-loop reads
    if (D::R10 > SH::n[2]) {
        for (int e = tid; e < 128; e += kDeepThreads) XA[(D::R10 - 1) * 128 + e] = 0.f;
    }
    // 9 -> 10: (row tile, column tile) per wave, 64 columns -> 128
    if (wave < 2 * P2::T) step_tile<P2, 6, 7>(a2, mt2, p.kst2[mt2], wave / P2::T, XB, XA, lane);
    __syncthreads();
    // 10 -> 14: wave = (column tile, half of the composite's row tiles); B fragments stay in registers
    {
        constexpr int KS3 = D::KS3, T3 = D::T3, TH = (T3 + 1) / 2;
        const int ct = wave & 3, part = wave >> 2;
        const int half = lane >> 5, col = lane & 31;
        float bf[KS3];
#pragma unroll
        for (int s = 0; s < KS3; ++s) bf[s] = XA[(2 * s + half) * 128 + ct * 32 + col];
        const int rt0 = part * TH;
        const float* tb = p.tab + D::off3 + lane;
        const size_t P = 16384;
        const size_t chan = (size_t)D::n14 * P;
        const int nch = (p.e.flags & AFD_WPT_SIGN) ? 2 : 1;
        const int n = ct * 32 + col;           // level-10 node inside the group
        const bool odd = n & 1;                // its frequency index is odd: packets come out reversed
        float* outc = p.out + (size_t)b * nch * chan + (size_t)16 * (grp * 128 + n);
        float a[KS3];
#pragma unroll
        for (int s = 0; s < KS3; ++s) a[s] = tb[((size_t)rt0 * KS3 + s) * 64];
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            const int rt = rt0 + r;
            if (rt < T3) {
                float an[KS3];
                if (r + 1 < TH && rt + 1 < T3) {
#pragma unroll
                    for (int s = 0; s < KS3; ++s) an[s] = tb[((size_t)(rt + 1) * KS3 + s) * 64];
                }
                f16v acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                for (int s = 0; s < KS3; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bf[s], acc, 0, 0, 0);
                // tile rows = (time 2 rt + {0, 1}) x (packet offset 0..15); this lane: register group g
                // holds time 2 rt + (g >> 1), packets 8 (g & 1) + 4 half + {0..3}
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int t = 2 * rt + (g >> 1);
                    const int f0 = 8 * (g & 1) + 4 * half;
                    const float v0 = acc[4 * g], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
                    float* o = outc + (size_t)t * P + (odd ? 12 - f0 : f0);
                    f4 w;
                    w.x = epi_value(odd ? v3 : v0, p.e);
                    w.y = epi_value(odd ? v2 : v1, p.e);
                    w.z = epi_value(odd ? v1 : v2, p.e);
                    w.w = epi_value(odd ? v0 : v3, p.e);
                    *reinterpret_cast<f4*>(o) = w;
                    if (p.e.flags & AFD_WPT_SIGN) {
                        f4 sg;
                        sg.x = (odd ? v3 : v0) < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                        sg.y = (odd ? v2 : v1) < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                        sg.z = (odd ? v1 : v2) < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                        sg.w = (odd ? v0 : v3) < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                        *reinterpret_cast<f4*>(o + chan) = sg;
                    }
                }
                if (r + 1 < TH && rt + 1 < T3) {
#pragma unroll
                    for (int s = 0; s < KS3; ++s) a[s] = an[s];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host: matrices in double precision, fragment tables, cache
// ------------------------------------------------------------------------------------------------
struct Mat {
    int rows = 0, cols = 0;
    std::vector<double> a;
    double& at(int r, int c) { return a[(size_t)r * cols + c]; }
    double at(int r, int c) const { return a[(size_t)r * cols + c]; }
};

int child_len3(int n, int L) { return (n + L - 2 + (n & 1)) / 2; }

// A [2 n_out x n_in]: row 2i + c = filter c (0 = dec_lo, 1 = dec_hi) placed for output i, i.e. taps m at
// the positions refl(2i + 1 - m) of the whole-sample reflect extension (folded back onto the node)
Mat level_matrix(const float* lo, const float* hi, int L, int n_in) {
    Mat m;
    const int n_out = child_len3(n_in, L);
    m.rows = 2 * n_out;
    m.cols = n_in;
    m.a.assign((size_t)m.rows * m.cols, 0.0);
    for (int i = 0; i < n_out; ++i)
        for (int t = 0; t < L; ++t) {
            const int c = refl_c(2 * i + 1 - t, n_in);
            m.at(2 * i, c) += (double)lo[t];
            m.at(2 * i + 1, c) += (double)hi[t];
        }
    return m;
}

// rows c, c + 2, ... of A: the map parent -> child through filter c
Mat select_filter(const Mat& a, int c) {
    Mat s;
    s.rows = a.rows / 2;
    s.cols = a.cols;
    s.a.resize((size_t)s.rows * s.cols);
    for (int r = 0; r < s.rows; ++r)
        for (int k = 0; k < s.cols; ++k) s.at(r, k) = a.at(2 * r + c, k);
    return s;
}

Mat matmul(const Mat& x, const Mat& y) {
    Mat z;
    z.rows = x.rows;
    z.cols = y.cols;
    z.a.assign((size_t)z.rows * z.cols, 0.0);
    for (int r = 0; r < x.rows; ++r)
        for (int k = 0; k < x.cols; ++k) {
            const double v = x.at(r, k);
            if (v == 0.0) continue;
            for (int c = 0; c < y.cols; ++c) z.at(r, c) += v * y.at(k, c);
        }
    return z;
}

// fragment (row tile rt, k-step s, lane l) = A[32 rt + (l & 31)][2 (kst + s) + (l >> 5)]
void put_fragments(std::vector<float>& tab, int off, const Mat& a, int tiles, int ksb, const int* kst) {
    for (int rt = 0; rt < tiles; ++rt)
        for (int s = 0; s < ksb; ++s)
            for (int l = 0; l < 64; ++l) {
                const int row = 32 * rt + (l & 31), col = 2 * ((kst ? kst[rt] : 0) + s) + (l >> 5);
                const double v = (row < a.rows && col < a.cols) ? a.at(row, col) : 0.0;
                tab[(size_t)off + ((size_t)rt * ksb + s) * 64 + l] = (float)v;
            }
}

struct TableEntry {
    int device, L;
    float lo[kMaxTaps], hi[kMaxTaps];
    float* dev;
};
std::vector<TableEntry>& table_cache() {
    static std::vector<TableEntry> v;
    return v;
}
std::mutex& table_mutex() {
    static std::mutex m;
    return m;
}

template <class SH>
int get_tables(const float* lo, const float* hi, hipStream_t stream, const float** out_tab, short* kst1, short* kst2) {
    using D = Deep3<SH>;
    constexpr int L = SH::L;
    int kst1i[kDeepWaves] = {0}, kst2i[kDeepWaves] = {0};
    for (int t = 0; t < D::P1::T; ++t) kst1[t] = (short)(kst1i[t] = D::P1::kstart(t));
    for (int t = 0; t < D::P2::T; ++t) kst2[t] = (short)(kst2i[t] = D::P2::kstart(t));
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: hipGetDevice failed");
    std::lock_guard<std::mutex> guard(table_mutex());
    for (const TableEntry& e : table_cache())
        if (e.device == dev && e.L == L && !memcmp(e.lo, lo, L * sizeof(float)) && !memcmp(e.hi, hi, L * sizeof(float))) {
            *out_tab = e.dev;
            return AFD_OK;
        }
    // level matrices 8->9 .. 13->14
    Mat lv[6];
    for (int j = 0; j < 6; ++j) {
        lv[j] = level_matrix(lo, hi, L, SH::n[j]);
        if (lv[j].rows != 2 * SH::n[j + 1]) return afd::fail(AFD_ERR_ARG, "wpt: node length table mismatch");
    }
    std::vector<float> tab((size_t)D::tab_floats, 0.f);
    put_fragments(tab, D::off1, lv[0], D::P1::T, D::P1::KSB, kst1i);
    put_fragments(tab, D::off2, lv[1], D::P2::T, D::P2::KSB, kst2i);
    // composite 10 -> 14: packet offset f (for an even-frequency level-10 node) has Gray-ordered path bits
    // b11..b14 (MSB first); the filter taken at a level is c = b ^ (bit of the level above), c11 = b11
    Mat comp;
    comp.rows = 16 * SH::n[6];
    comp.cols = SH::n[2];
    comp.a.assign((size_t)comp.rows * comp.cols, 0.0);
    for (int f = 0; f < 16; ++f) {
        const int bits[4] = {(f >> 3) & 1, (f >> 2) & 1, (f >> 1) & 1, f & 1};
        Mat m = select_filter(lv[2], bits[0]);
        for (int j = 1; j < 4; ++j) m = matmul(select_filter(lv[2 + j], bits[j] ^ bits[j - 1]), m);
        for (int t = 0; t < SH::n[6]; ++t)
            for (int k = 0; k < comp.cols; ++k) comp.at(16 * t + f, k) = m.at(t, k);
    }
    put_fragments(tab, D::off3, comp, D::T3, D::KS3, nullptr);
    float* dptr = nullptr;
    hipError_t e = hipMalloc(&dptr, tab.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dptr, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);  // `tab` is freed on return; once per wavelet and device
    if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: matrix table upload: %s", hipGetErrorString(e));
    TableEntry ent{};
    ent.device = dev;
    ent.L = L;
    memcpy(ent.lo, lo, L * sizeof(float));
    memcpy(ent.hi, hi, L * sizeof(float));
    ent.dev = dptr;
    table_cache().push_back(ent);
    *out_tab = dptr;
    return AFD_OK;
}

// LDS plan of the top kernel; false when a level pair does not fit
bool plan_top(T3Params& p, int L) {
    const int PAD = L - 2;
    long size[kKsMax + 1];
    for (int k = 0; k <= p.Ks; ++k) {
        int pitch = p.n[k] + 2 * PAD + 2;
        while (pitch % 4 != 2) ++pitch;  // even, and pitch / 2 odd: node-strided b64 reads hit distinct banks
        p.pitch[k] = pitch;
        const long nodes = k == 0 ? 1 : (1L << (k - 1));
        size[k] = nodes * pitch;
        p.magic[k] = (unsigned)(0x100000000ULL / (unsigned)p.n[k]) + 1u;
    }
    // images alternate between the two ends of the carve; the last level is never stored
    for (int k = 0; k < p.Ks; ++k) {
        if (size[k] > kTopLdsFloats) return false;
        if (k + 1 < p.Ks && size[k] + size[k + 1] > kTopLdsFloats) return false;
        p.off[k] = (k & 1) ? (int)((kTopLdsFloats - size[k]) & ~1L) : 0;
    }
    p.off[p.Ks] = 0;
    return true;
}

template <int L>
int launch3(T3Params& p, const float* dec_lo, const float* dec_hi, float* out, void* ws, int level, int t_len,
            hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt3_top_kernel<L>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kTopLdsFloats * 4);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
    const int C = (p.e.flags & AFD_WPT_SIGN) ? 2 : 1;
    afd::ScopedTiming timing(AFD_K_WPT, 4.0 * p.B * ((double)p.N + (double)C * t_len * (double)(1L << level)), stream);
    if (level <= kKsMax) {
        p.dst = out;
        p.final_ = 1;
        hipLaunchKernelGGL(wpt3_top_kernel<L>, dim3((unsigned)p.B * 2), dim3(kTopThreads), (size_t)kTopLdsFloats * 4,
                           stream, p);
        return afd::check_launch("wpt3_top_kernel");
    }
    if constexpr (HasShape3<L>::value) {
        using SH = Shape3<L>;
        using D = Deep3<SH>;
        static bool dattr = false;
        constexpr size_t lds = (size_t)(D::A_FLOATS + D::B_FLOATS) * 4;
        if (!dattr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt3_deep_kernel<SH>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: hipFuncSetAttribute: %s", hipGetErrorString(e));
            dattr = true;
        }
        D3Params q{};
        int rc = get_tables<SH>(dec_lo, dec_hi, stream, &q.tab, q.kst1, q.kst2);
        if (rc != AFD_OK) return rc;
        p.dst = static_cast<float*>(ws);
        p.final_ = 0;
        hipLaunchKernelGGL(wpt3_top_kernel<L>, dim3((unsigned)p.B * 2), dim3(kTopThreads), (size_t)kTopLdsFloats * 4,
                           stream, p);
        q.ws = static_cast<const float*>(ws);
        q.out = out;
        q.e = p.e;
        hipLaunchKernelGGL(wpt3_deep_kernel<SH>, dim3((unsigned)p.B * (256 / kGroup)), dim3(kDeepThreads), lds, stream, q);
        return afd::check_launch("wpt3 kernels");
    }
    return 1;
}

}  // namespace

namespace afd {

// returns AFD_OK, an error, or 1 = "not this generation's case" (the caller falls back to wpt2 / wpt)
int wpt3_forward(const float* x, int B, int N, const float* dec_lo, const float* dec_hi, int L, int level,
                 unsigned flags, float power, float eps, float mean, float std, float sign_mean, float sign_std,
                 float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
    if (getenv("AFD_WPT_NO_V3")) return 1;
    if (L < 4 || L > kMaxTaps || (L & 1) || level < 2) return 1;
    if (level > kKsMax && level != 14) return 1;
    T3Params p{};
    p.x = x;
    p.B = B;
    p.N = N;
    p.Ks = level < kKsMax ? level : kKsMax;
    p.n[0] = N;
    for (int k = 1; k <= p.Ks; ++k) {
        const int prev = p.n[k - 1];
        if (L - 2 + (prev & 1) >= prev) return 1;  // reflect pad must be shorter than the node
        p.n[k] = child_len3(prev, L);
    }
    int t_len = p.n[p.Ks];
    if (level == 14) {
        int deep[7];
        deep[0] = p.n[8];
        for (int j = 1; j < 7; ++j) {
            if (L - 2 + (deep[j - 1] & 1) >= deep[j - 1]) return 1;
            deep[j] = child_len3(deep[j - 1], L);
        }
        t_len = deep[6];
        auto matches = [&](const int* n) {
            for (int j = 0; j < 7; ++j)
                if (deep[j] != n[j]) return false;
            return true;
        };
        bool ok = false;
        if (L == 24) ok = matches(Shape3<24>::n);
        if (L == 10) ok = matches(Shape3<10>::n);
        if (L == 16) ok = matches(Shape3<16>::n);
        if (!ok) return 1;
        const size_t need = (size_t)B * p.n[8] * 256 * sizeof(float);
        if (!ws || ws_bytes < need) return afd::fail(AFD_ERR_WORKSPACE, "wpt: workspace of %zu bytes needed", need);
    }
    if (!plan_top(p, L)) return 1;
    if ((long)B * 2 > 0x7fffffffL) return 1;
    Epi& e = p.e;
    e.flags = flags;
    e.power = power;
    e.eps = eps;
    e.mean = mean;
    e.inv_std = (float)(1.0 / (double)(std == 0.f ? 1.f : std));
    const bool norm = flags & AFD_WPT_NORM;
    e.k1 = (float)(0.6931471805599453 * (norm ? 1.0 / (double)std : 1.0));
    e.k0 = norm ? (float)(-(double)mean / (double)std) : 0.f;
    e.sgn_neg = norm ? (-1.f - sign_mean) / sign_std : -1.f;
    e.sgn_pos = norm ? (1.f - sign_mean) / sign_std : 1.f;
    for (int m = 0; m < L; ++m) {
        p.rlo[m] = dec_lo[L - 1 - m];
        p.rhi[m] = dec_hi[L - 1 - m];
    }
    switch (L) {
        case 4: return launch3<4>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 6: return launch3<6>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 8: return launch3<8>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 10: return launch3<10>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 16: return launch3<16>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 24: return launch3<24>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        default: return 1;
    }
}

}  // namespace afd
