// Wavelet-packet front end, second generation: sliding-window packed-FMA filter bank.
//
// Same contract as wpt.hip (reference src/audiofakedetect/wavelet_math.py:167-263,
// :380-382); the arithmetic is reorganised around one device routine:
//
//   work unit  = (parent node q, output range [i0, i1)) of one level.  A thread keeps the L
//                taps' worth of parent samples in registers and slides the window by two per
//                output pair: 2 LDS loads + L packed FMAs (v_pk_fma_f32 computes the
//                approximation and the detail output together: (lo[m], hi[m]) * (x, x)).
//   layout     = nodes are position-major in LDS, src[pos * M + q] (M nodes of the level):
//                lanes run over nodes -> conflict-free reads, children written as one
//                8-byte store, and the last level's store to HBM is contiguous in packets.
//   indices    = for M >= 64 every lane of a wave works on the same output range, so the
//                reflect index arithmetic is scalar; units that do not touch a node border
//                use plain pointer increments.
//
// Two kernels:
//   wpt2_top   workgroup = (frame, level-2 subtree): frame -> LDS, the two filters on the path
//              to the subtree root, then breadth-first to level Ks <= 8.  If the transform
//              ends there the epilogue (log-power / sign / normalise) writes the features,
//              otherwise the level-Ks nodes go to a workspace [B][n_Ks][2^Ks].
//   wpt2_deep  workgroup = (frame, G level-Ks nodes): levels Ks+1..level in ~44 KB of LDS
//              (3 workgroups per CU), epilogue, store.
// Extra HBM traffic of the split: 2 * 4 * 2^Ks * n_Ks bytes per frame (coif4: +13 %).
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kMaxLevel = 16;
constexpr int kMaxTaps = 32;
constexpr int kTopThreads = 1024;
constexpr int kDeepThreads = 256;
constexpr int kTopLdsFloats = 40000;
constexpr int kKsMax = 8;
constexpr int kMinOut = 6;  // outputs per work unit, at least

struct W2Params {
    const float* x;
    float* out;
    float* ws;
    int B, N, L, level, K1, Ks, G;
    int offA, offB, deepFloats;  // deep kernel LDS carve (floats)
    // MFMA deep kernel (levels Ks+1 .. Ks+6): per level j the k-steps, the 32-row tiles of the
    // level matrix, the offset of its fragment table; LDS carve in floats
    int mf, mfKs[6], mfTiles[6], mfOff[6], mfTab, mfX, mfY, mfC13;
    int n[kMaxLevel + 1];
    unsigned flags;
    float power, eps, mean, std, inv_std;
    float sgn_neg, sgn_pos;  // the sign channel's two values, normalised with its own statistics
    float lo[kMaxTaps], hi[kMaxTaps];
    float rlo[kMaxTaps], rhi[kMaxTaps];  // taps reversed: rlo[t] = lo[L-1-t]
};

enum { MODE_PAIR = 0, MODE_ONE = 1, MODE_FINAL = 2 };

struct Sink {
    float* dst;     // MODE_PAIR / MODE_ONE: LDS destination
    int sel;        // MODE_ONE: 0 = approximation, 1 = detail
    float* outb;    // MODE_FINAL: &out[b][0][0][0]
    size_t P;       // MODE_FINAL: packets of the stored level
    size_t chan;    // MODE_FINAL: T * P (offset of the sign channel)
    unsigned flags;
};

// LDS position padding: one slot after every 32 positions.  Threads that own different
// output chunks of the same node read positions 2*len*c + t; without the skew those strides
// are multiples of 16 or 32 words and every lane of a read hits the same one or two banks.
__device__ __forceinline__ constexpr int padpos(int pos) { return pos + (pos >> 5); }

__device__ __forceinline__ int refl_clamp(int j, int n) {
    j = j < 0 ? -j : j;
    j = j >= n ? 2 * (n - 1) - j : j;
    j = j < 0 ? 0 : j;
    return j >= n ? n - 1 : j;
}

// power == 2 fast path: log(v*v + eps) through the hardware log2 (v_log_f32, ~1 ulp in log2,
// i.e. < 2e-6 absolute on these features); other powers take the precise library path.
// |v|^power for power != 2: ~150 instructions of powf/logf -- kept out of line so that the
// unrolled filter bodies (one epilogue per output) stay small enough for the instruction cache
__device__ __noinline__ float pow_log_slow(float v, float power, float eps) {
    return logf(powf(fabsf(v), power) + eps);
}

__device__ __forceinline__ float epilogue_value(float v, unsigned flags, float power, float eps, float mean,
                                                float inv_std) {
    if (flags & AFD_WPT_LOG) {
        if (power == 2.0f) {
            // v*v + eps >= 1e-12 is a normal float: the bare v_log_f32 (log2) needs no
            // denormal pre-scaling; ~1 ulp of log2, < 2e-6 absolute on these features
            v = __builtin_amdgcn_logf(fmaf(v, v, eps)) * 0.6931471805599453f;
        } else {
            v = pow_log_slow(v, power, eps);
        }
    }
    if (flags & AFD_WPT_NORM) v = (v - mean) * inv_std;
    return v;
}

__device__ __forceinline__ float epilogue2(float v, const W2Params& p, unsigned flags) {
    return epilogue_value(v, flags, p.power, p.eps, p.mean, p.inv_std);
}

__device__ __forceinline__ void emit(const W2Params& p, const Sink& s, int mode, int i, int q, int M2,
                                     int F, f2 acc) {
    if (mode == MODE_ONE) {
        s.dst[padpos(i)] = s.sel ? acc.y : acc.x;
        return;
    }
    const bool par = F & 1;  // odd-frequency parents list their children (d, a)
    f2 v;
    v.x = par ? acc.y : acc.x;
    v.y = par ? acc.x : acc.y;
    if (mode == MODE_PAIR) {
        *reinterpret_cast<f2*>(s.dst + padpos(i) * M2 + 2 * q) = v;
    } else {
        const unsigned o = (unsigned)i * (unsigned)s.P + 2u * (unsigned)F;  // < 2^31 per frame
        f2 r;
        r.x = epilogue2(v.x, p, s.flags);
        r.y = epilogue2(v.y, p, s.flags);
        *reinterpret_cast<f2*>(s.outb + o) = r;
        if (s.flags & AFD_WPT_SIGN) {
            f2 sg;
            sg.x = v.x < 0.f ? p.sgn_neg : p.sgn_pos;
            sg.y = v.y < 0.f ? p.sgn_neg : p.sgn_pos;
            *reinterpret_cast<f2*>(s.outb + s.chan + o) = sg;
        }
    }
}

// Register window of W = L + 2*kAhead samples: position (rel. to the unit start) r lives in
// register r % W.  Output step j uses registers (2j + t) % W, t < L, and afterwards refills
// the two oldest registers with positions 2j + W, 2j + W + 1 -- those are first consumed
// kAhead + 1 steps later, which hides the LDS latency of the refill (a window of exactly L
// samples consumes a refill in the very next step and stalls on it).
constexpr int kAhead = 0;

// The window is held as W/2 register PAIRS (f2): pair k = positions (ps + 2k, ps + 2k + 1).
// Sliding by one output moves by exactly one pair, and the packed FMA runs ALONG the taps:
//   accA += (lo[L-1-t], lo[L-2-t]) * (x[t], x[t+1]),  cA = accA.x + accA.y   (same for hi)
// so both operands are natural aligned pairs (taps in SGPR pairs, samples in VGPR pairs) and
// no broadcast copies are needed (a (lo,hi)*(x,x) formulation costs one v_mov per FMA here).
template <int L, int ST>
__device__ __forceinline__ f2 window_dot(const W2Params& p, const f2 (&w)[(L + 2 * kAhead) / 2]) {
    constexpr int WP = (L + 2 * kAhead) / 2;
    f2 accA = {0.f, 0.f}, accD = {0.f, 0.f};
#pragma unroll
    for (int t2 = 0; t2 < L / 2; ++t2) {
        const f2 tl = {p.rlo[2 * t2], p.rlo[2 * t2 + 1]};
        const f2 th = {p.rhi[2 * t2], p.rhi[2 * t2 + 1]};
        const f2 x = w[(t2 + ST) % WP];
        accA = __builtin_elementwise_fma(tl, x, accA);
        accD = __builtin_elementwise_fma(th, x, accD);
    }
    f2 r;
    r.x = accA.x + accA.y;
    r.y = accD.x + accD.y;
    return r;
}

template <int L, int ST, bool INTERIOR>
struct Steps {
    static __device__ __forceinline__ void run(const W2Params& p, const Sink& s, int mode,
                                               const float* __restrict__ col, int M, int n_in,
                                               int ps, int i, int i0, int i1, int q, int F,
                                               f2 (&w)[(L + 2 * kAhead) / 2]) {
        constexpr int W = L + 2 * kAhead;
        // units are whole or half blocks of W/2 outputs: a unit that ends here skips the second half
        if ((W / 2) % 2 == 0 && ST == W / 4 && i + ST >= i1) return;
        const f2 acc = window_dot<L, ST>(p, w);
        if (i + ST < i1) emit(p, s, mode, i + ST, q, 2 * M, F, acc);
        const int pn = ps + 2 * (i + ST - i0) + W;  // absolute position of the refill
        f2 nw;
        if (INTERIOR) {
            nw.x = col[padpos(pn) * M];
            nw.y = col[padpos(pn + 1) * M];
        } else {
            nw.x = col[padpos(refl_clamp(pn, n_in)) * M];
            nw.y = col[padpos(refl_clamp(pn + 1, n_in)) * M];
        }
        w[ST % (W / 2)] = nw;
        if (ST + 1 < W / 2)
            Steps<L, (ST + 1 < W / 2 ? ST + 1 : 0), INTERIOR>::run(p, s, mode, col, M, n_in, ps, i, i0, i1, q, F, w);
    }
};

// outputs [i0, i1) of parent column `col` (stride M between positions).  The W/2 rotation
// states times L packed FMAs are ~6 KB of code per path: each kernel has exactly ONE call site
// (the level loops pass the mode at run time) so the body exists once per kernel.
template <int L>
__device__ __forceinline__ void process_unit(const W2Params& p, const Sink& s, int mode,
                                             const float* __restrict__ col, int M, int n_in, int i0,
                                             int i1, int q, int F) {
    constexpr int W = L + 2 * kAhead;
    f2 w[W / 2];
    const int ps = 2 * i0 + 2 - L;  // position of the first window sample (even)
    const int nblk = (i1 - i0 + W / 2 - 1) / (W / 2);
    const int iend = i0 + nblk * (W / 2);  // outputs are computed in blocks of W/2
    if (ps >= 0 && ps + 2 * (iend - i0) + W + 1 < n_in) {
        // interior: no reflection anywhere in this unit (look-ahead refills included)
#pragma unroll
        for (int t = 0; t < W / 2; ++t) {
            w[t].x = col[padpos(ps + 2 * t) * M];
            w[t].y = col[padpos(ps + 2 * t + 1) * M];
        }
        for (int i = i0; i < i1; i += W / 2)
            Steps<L, 0, true>::run(p, s, mode, col, M, n_in, ps, i, i0, i1, q, F, w);
    } else {
#pragma unroll
        for (int t = 0; t < W / 2; ++t) {
            w[t].x = col[padpos(refl_clamp(ps + 2 * t, n_in)) * M];
            w[t].y = col[padpos(refl_clamp(ps + 2 * t + 1, n_in)) * M];
        }
        for (int i = i0; i < i1; i += W / 2)
            Steps<L, 0, false>::run(p, s, mode, col, M, n_in, ps, i, i0, i1, q, F, w);
    }
}

constexpr int refl_c(int j, int n) {
    j = j < 0 ? -j : j;
    return j >= n ? 2 * (n - 1) - j : j;
}

template <int L, int NIN, int I>
struct DenseOut {
    static __device__ __forceinline__ void run(const W2Params& p, const Sink& s, int mode, int q, int M2,
                                               int F, int ilo, int ihi, const float (&x)[NIN]) {
        constexpr int NOUT = (NIN + L - 2 + (NIN & 1)) / 2;
        if (I >= ilo && I < ihi) {  // wave-uniform
            f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
            for (int m = 0; m < L; m += 2) {
                const f2 t0 = {p.lo[m], p.hi[m]};
                const f2 t1 = {p.lo[m + 1], p.hi[m + 1]};
                const float x0 = x[refl_c(2 * I + 1 - m, NIN)];
                const float x1 = x[refl_c(2 * I - m, NIN)];
                const f2 xx0 = {x0, x0};
                const f2 xx1 = {x1, x1};
                acc0 = __builtin_elementwise_fma(t0, xx0, acc0);
                acc1 = __builtin_elementwise_fma(t1, xx1, acc1);
            }
            emit(p, s, mode, I, q, M2, F, acc0 + acc1);
        }
        if (I + 1 < NOUT) DenseOut<L, NIN, (I + 1 < NOUT ? I + 1 : 0)>::run(p, s, mode, q, M2, F, ilo, ihi, x);
    }
};

// Deep levels of the standard 22 050-sample frame: the parent node (NIN <= 33 samples) lives in
// registers, lanes run over nodes, every reflect index is a compile-time constant: exactly
// L packed FMAs per output pair, no index arithmetic.  Waves split the outputs when there are
// fewer parents than threads.
template <int L, int NIN>
__device__ __forceinline__ void dense_level(const W2Params& p, const Sink& s, int mode,
                                            const float* __restrict__ src, int logM, int Fb) {
    constexpr int NOUT = (NIN + L - 2 + (NIN & 1)) / 2;
    const int M = 1 << logM;
    int parts = (int)blockDim.x >> logM;
    if (parts < 1) parts = 1;
    const int per = (NOUT + parts - 1) / parts;
    for (int u = threadIdx.x; u < (parts << logM); u += blockDim.x) {
        const int q = u & (M - 1);
        const int part = __builtin_amdgcn_readfirstlane(u >> logM);
        const int ilo = part * per;
        int ihi = ilo + per;
        if (ihi > NOUT) ihi = NOUT;
        float x[NIN];
#pragma unroll
        for (int j = 0; j < NIN; ++j) x[j] = src[padpos(j) * M + q];
        DenseOut<L, NIN, 0>::run(p, s, mode, q, 2 * M, Fb + q, ilo, ihi, x);
    }
}

// dense path for the node lengths of N = 22050 at levels >= 12 (haar, sym5, coif4)
template <int L>
__device__ __forceinline__ bool try_dense(const W2Params& p, const Sink& s, int mode,
                                          const float* __restrict__ src, int logM, int n_in, int Fb) {
    if ((1 << logM) < 64) return false;
    if (L == 24) {
        if (n_in == 33) { dense_level<L, 33>(p, s, mode, src, logM, Fb); return true; }
        if (n_in == 28) { dense_level<L, 28>(p, s, mode, src, logM, Fb); return true; }
        if (n_in == 25) { dense_level<L, 25>(p, s, mode, src, logM, Fb); return true; }
    }
    if (L == 10) {
        if (n_in == 19) { dense_level<L, 19>(p, s, mode, src, logM, Fb); return true; }
        if (n_in == 14) { dense_level<L, 14>(p, s, mode, src, logM, Fb); return true; }
        if (n_in == 11) { dense_level<L, 11>(p, s, mode, src, logM, Fb); return true; }
    }
    if (L == 2) {
        if (n_in == 11) { dense_level<L, 11>(p, s, mode, src, logM, Fb); return true; }
        if (n_in == 6) { dense_level<L, 6>(p, s, mode, src, logM, Fb); return true; }
        if (n_in == 3) { dense_level<L, 3>(p, s, mode, src, logM, Fb); return true; }
    }
    return false;
}

// one level: parents src[pos * M + q] (M = 1 << logM nodes of n_in samples) -> children
template <int L>
__device__ __forceinline__ void level_units(const W2Params& p, const Sink& s, int mode,
                                            const float* __restrict__ src, int logM, int n_in,
                                            int n_out, int Fb) {
    const int M = 1 << logM;
    const int nthreads = blockDim.x;
    int C = nthreads >> logM;  // chunks per parent so that every thread has a unit
    if (C < 1) C = 1;
    constexpr int blk = (L + 2 * kAhead) / 2;  // outputs come in blocks of W/2 (or half blocks)
    constexpr int unit = (blk % 2 == 0 && blk >= 8) ? blk / 2 : blk;
    const int maxC = (n_out + kMinOut - 1) / kMinOut;  // at least kMinOut outputs per unit (window fill amortised)
    if (C > maxC) C = maxC;
    int len = (n_out + C - 1) / C;
    len = ((len + unit - 1) / unit) * unit;
    const int units = C << logM;
    for (int u = threadIdx.x; u < units; u += nthreads) {
        const int c = u >> logM;
        const int q = u & (M - 1);
        const int i0 = c * len;
        int i1 = i0 + len;
        if (i1 > n_out) i1 = n_out;
        if (i0 < i1) process_unit<L>(p, s, mode, src + q, M, n_in, i0, i1, q, Fb + q);
    }
}

template <int L>
__global__ void __launch_bounds__(kTopThreads) wpt2_top_kernel(const W2Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    // the 2^K1 workgroups of a frame are B apart: with B % 8 == 0 they share an XCD's L2
    const int b = blockIdx.x % p.B;
    const int j1 = blockIdx.x / p.B;

    const float* xg = p.x + (size_t)b * p.N;
    // frame -> LDS with 8 loads in flight per thread (a load -> store -> load chain costs one
    // full HBM/L2 latency per element and dominated the whole transform)
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
                if (i < n2) {
                    const int a = padpos(2 * i);  // 2i and 2i+1 never straddle a pad slot
                    lds[a] = v[u].x;
                    lds[a + 1] = v[u].y;
                }
            }
        }
    } else {
        for (int base = 0; base < p.N; base += kTopThreads * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * kTopThreads + tid;
                v[u] = i < p.N ? xg[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * kTopThreads + tid;
                if (i < p.N) lds[padpos(i)] = v[u];
            }
        }
    }
    __syncthreads();

    float* cur = lds;
    bool at_bottom = true;
    int n_cur = p.N;
    Sink s{};
    // levels 1..K1: the filters that lead to frequency index j1 (Gray code, MSB = level 1),
    // one child per level; then breadth-first.  One call site for every level (code size).
    const int g = j1 ^ (j1 >> 1);
    int logM = 0;
    int Fb = j1;
    for (int k = 1; k <= p.Ks; ++k) {
        const int n_out = p.n[k];
        int mode;
        if (k <= p.K1) {
            mode = MODE_ONE;
            s.dst = at_bottom ? (lds + kTopLdsFloats - (padpos(n_out) + 2)) : lds;
            s.sel = (g >> (p.K1 - k)) & 1;
        } else if (k == p.Ks) {
            mode = MODE_FINAL;
            const bool last = (k == p.level);
            s.P = (size_t)1 << k;
            s.chan = (size_t)n_out * s.P;
            s.flags = last ? p.flags : 0u;
            const size_t nch = (last && (p.flags & AFD_WPT_SIGN)) ? 2 : 1;
            s.outb = (last ? p.out : p.ws) + (size_t)b * nch * s.chan;
        } else {
            mode = MODE_PAIR;
            s.dst = at_bottom ? (lds + kTopLdsFloats - (((padpos(n_out) + 1) << (logM + 1)) & ~1)) : lds;
        }
        level_units<L>(p, s, mode, cur, logM, n_cur, n_out, Fb);
        if (k == p.Ks) break;
        __syncthreads();
        cur = s.dst;
        at_bottom = !at_bottom;
        n_cur = n_out;
        if (k > p.K1) {
            ++logM;
            Fb *= 2;
        }
    }
}

template <int L>
__global__ void __launch_bounds__(kDeepThreads) wpt2_deep_kernel(const W2Params p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int groups = (1 << p.Ks) / p.G;
    const int b = blockIdx.x / groups;
    const int grp = blockIdx.x - b * groups;
    const int F0 = grp * p.G;
    const int nKs = p.n[p.Ks];
    const size_t PKs = (size_t)1 << p.Ks;
    // input nodes, position-major: in[pos * G + j]
    const float* wsb = p.ws + (size_t)b * nKs * PKs + F0;
    int logG = 31 - __clz(p.G);
    for (int e = tid; e < (nKs << logG); e += kDeepThreads) {
        const int pos = e >> logG;
        const int j = e & (p.G - 1);
        lds[(padpos(pos) << logG) + j] = wsb[(size_t)pos * PKs + j];
    }
    __syncthreads();
    const float* cur = lds;
    int n_cur = nKs;
    int logM = logG;
    int Fb = F0;
    bool useA = true;
    Sink s{};
    for (int k = p.Ks + 1; k <= p.level; ++k) {
        const int n_out = p.n[k];
        int mode = MODE_PAIR;
        if (k == p.level) {
            mode = MODE_FINAL;
            s.P = (size_t)1 << k;
            s.chan = (size_t)n_out * s.P;
            s.flags = p.flags;
            const size_t nch = (p.flags & AFD_WPT_SIGN) ? 2 : 1;
            s.outb = p.out + (size_t)b * nch * s.chan;
        } else {
            s.dst = lds + (useA ? p.offA : p.offB);
        }
        if (!try_dense<L>(p, s, mode, cur, logM, n_cur, Fb))
            level_units<L>(p, s, mode, cur, logM, n_cur, n_out, Fb);
        if (k == p.level) break;
        __syncthreads();
        cur = s.dst;
        useA = !useA;
        n_cur = n_out;
        ++logM;
        Fb *= 2;
    }
}

// ---------------------------------------------------------------------------------------------
// Deep levels on the matrix cores.
//
// One analysis step is the same linear map for every node of a level: children = A_k * parent,
// A_k [2 n_out x n_in] with row 2i + c = filter c (0 = lo, 1 = hi) placed for output i and the
// reflect extension folded back onto the columns it mirrors.  From level 9 on the rows are mostly
// dense (L = 24 taps over 25..66 columns), so a level is the GEMM
//     Y [2 n_out x nodes] = A_k [2 n_out x n_in] * X [n_in x nodes]
// on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation): no per-tap addressing, no
// reflect index arithmetic, the nodes (position-major in LDS, as everywhere in this file) are the
// N dimension.  The D fragment holds rows (2i, 2i+1) = (cA[i], cD[i]) of one node in neighbouring
// accumulator registers: one 8-byte store puts the two children side by side in the next level.
//
// workgroup = (frame, 32 neighbouring level-8 nodes), 8 waves.  Levels 9-11 for all 32 subtrees
// (32 / 64 / 128 columns), then four passes of 64 level-11 nodes each through levels 12-14
// (128 / 256 columns; level 14 goes to HBM through the epilogue) -- ~75 KB of LDS, 2 workgroups
// per CU.  A wave owns one 32-row tile of A_k (fragments in registers, loaded once per level
// from the table `wpt2_matrices_kernel` wrote) and walks the 32-column tiles.
// ---------------------------------------------------------------------------------------------
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kMfWaves = 8;
constexpr int kMfThreads = kMfWaves * 64;
constexpr int kMfG = 32;       // level-Ks nodes per workgroup

// what the table kernel needs: taps, node lengths, and per level the banded k-range of each tile
struct MfTabParams {
    float lo[kMaxTaps], hi[kMaxTaps];
    int L;
    int n[7];
    int ksb[6], tiles[6], off[6];
    short kst[6][kMfWaves];
};

// entry (row 2i + c, column col) of the level matrix: the taps m with refl(2i + 1 - m) == col
__device__ __forceinline__ float level_matrix_entry(const float* lo, const float* hi, int L, int i, int c,
                                                    int col, int n_in, int n_out) {
    if (i >= n_out || col >= n_in) return 0.f;
    const float* f = c ? hi : lo;
    float v = 0.f;
    int m = 2 * i + 1 - col;  // position col itself
    if (m >= 0 && m < L) v += f[m];
    m = 2 * i + 1 + col;  // position -col, mirrored at the left border
    if (col > 0 && m < L) v += f[m];
    m = 2 * i + 1 - (2 * (n_in - 1) - col);  // position 2(n-1) - col, mirrored at the right border
    if (col < n_in - 1 && m >= 0 && m < L) v += f[m];
    return v;
}

// fragment table: for level j, tile mt, k-step s, lane l:
//   A_j[32 mt + (l & 31)][2 (kst[j][mt] + s) + (l >> 5)],   s < ksb[j] (the tile's band)
__global__ void __launch_bounds__(64) wpt2_matrices_kernel(const MfTabParams p, float* __restrict__ tab) {
    __shared__ float lo[kMaxTaps], hi[kMaxTaps];
    const int lane = threadIdx.x;
    if (lane < kMaxTaps) {
        lo[lane] = lane < p.L ? p.lo[lane] : 0.f;
        hi[lane] = lane < p.L ? p.hi[lane] : 0.f;
    }
    __syncthreads();
    int e = blockIdx.x;
    int j = 0;
    while (j < 5 && e >= p.tiles[j] * p.ksb[j]) {
        e -= p.tiles[j] * p.ksb[j];
        ++j;
    }
    const int mt = e / p.ksb[j], st = e - mt * p.ksb[j];
    const int row = 32 * mt + (lane & 31), col = 2 * (p.kst[j][mt] + st) + (lane >> 5);
    tab[p.off[j] + e * 64 + lane] = level_matrix_entry(lo, hi, p.L, row >> 1, row & 1, col, p.n[j], p.n[j + 1]);
}

// what the deep kernel needs of the plan (a small kernel argument: the full W2Params costs
// ~250 scalar registers here)
struct MfParams {
    const float* ws;
    const float* tab;
    float* out;
    int off[6];                 // fragment-table offset of each level
    short kst[6][kMfWaves];     // first k-step of each row tile's band
    int X, Y, C13, level;
    unsigned flags;
    float power, eps, mean, inv_std, sgn_neg, sgn_pos;
};

struct MfSink {
    float* outb;     // &out[b][0][0][0] + first packet of this workgroup's pass
    size_t P, chan;  // packets per row, offset of the sign channel
};

// Node lengths of the standard 1 s frame (N = 22 050) at levels 8..14: with the shapes known at
// compile time the k-loops are straight-line code (every LDS offset an immediate, the reads of a
// tile issued ahead of its MFMAs).  Other geometries stay on the vector kernel above.
template <int L> struct MfShape;
template <> struct MfShape<24> { static constexpr int L = 24; static constexpr int n[7] = {109, 66, 44, 33, 28, 25, 24}; };
template <> struct MfShape<10> { static constexpr int L = 10; static constexpr int n[7] = {95, 52, 30, 19, 14, 11, 10}; };
template <> struct MfShape<16> { static constexpr int L = 16; static constexpr int n[7] = {101, 58, 36, 25, 20, 17, 16}; };

template <class SH, int J> struct MfLevel {
    static constexpr int n_in = SH::n[J], n_out = SH::n[J + 1];
    static constexpr int KS = (n_in + 1) / 2;           // k-steps of the dense matrix (two columns each)
    static constexpr int T = (2 * n_out + 31) / 32;     // 32-row tiles of the level matrix
    static constexpr int G = kMfWaves / T;              // waves per tile
    // a row tile holds outputs 16 t .. 16 t + 15: its non-zero columns are a band of ~L + 30
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
    static constexpr int KSB = max_steps();             // k-steps issued per tile
    static constexpr int kstart(int t) {
        const int s0 = band(t, false) / 2;
        return s0 + KSB > KS ? KS - KSB : s0;
    }
};

// tiles of one level for one wave: A fragments of row tile mt, column tiles nt0, nt0 + G, ...
// src: parents, position-major with row stride 1 << LOGS; dst: children with row stride 1 << LOGD
template <class LV, int LOGS, int LOGD, int NTILES, bool FINAL>
__device__ __forceinline__ void mfma_level(const MfParams& p, const float (&a)[LV::KSB], int mt, int kst,
                                           int nt0, const float* __restrict__ src, float* __restrict__ dst,
                                           const MfSink& fs, int lane) {
    constexpr int S = 1 << LOGS;
    const int half = lane >> 5, col = lane & 31;
    src += (size_t)kst * 2 * S;
#pragma unroll
    for (int t = 0; t < (NTILES + LV::G - 1) / LV::G; ++t) {
        const int nt = nt0 + t * LV::G;
        if (nt < NTILES) {
            f16v acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* bp = src + half * S + nt * 32 + col;
#pragma unroll
            for (int st = 0; st < LV::KSB; ++st)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], bp[st * 2 * S], acc, 0, 0, 0);
            // D: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); rows (2i, 2i+1) = (cA, cD)
            const int n = nt * 32 + col;
            // rows of this lane: i = i0 + {0, 1, 4, 5, 8, 9, 12, 13}; the compare masks are rebuilt per
            // tile (kept across the kernel they would fill the scalar registers)
            int i0 = mt * 16 + half * 2;
            asm volatile("" : "+v"(i0));
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int i = i0 + (r >> 2) * 4 + ((r & 3) >> 1);
                if (i < LV::n_out) {
                    // odd-frequency parents list their children (d, a).  The values are pinned first:
                    // a select between acc[r] and acc[r + 1] is otherwise turned into a dynamic
                    // vector index (a 16-way compare / select chain per element)
                    float e0 = acc[r], e1 = acc[r + 1];
                    asm volatile("" : "+v"(e0), "+v"(e1));
                    f2 v;
                    v.x = (col & 1) ? e1 : e0;
                    v.y = (col & 1) ? e0 : e1;
                    if (!FINAL) {
                        *reinterpret_cast<f2*>(dst + ((size_t)i << LOGD) + 2 * n) = v;
                    } else {
                        const size_t o = (size_t)i * fs.P + 2 * n;
                        f2 q;
                        q.x = epilogue_value(v.x, p.flags, p.power, p.eps, p.mean, p.inv_std);
                        q.y = epilogue_value(v.y, p.flags, p.power, p.eps, p.mean, p.inv_std);
                        *reinterpret_cast<f2*>(fs.outb + o) = q;
                        if (p.flags & AFD_WPT_SIGN) {
                            f2 sg;
                            sg.x = v.x < 0.f ? p.sgn_neg : p.sgn_pos;
                            sg.y = v.y < 0.f ? p.sgn_neg : p.sgn_pos;
                            *reinterpret_cast<f2*>(fs.outb + fs.chan + o) = sg;
                        }
                    }
                }
            }
        }
    }
}

template <int KS>
__device__ __forceinline__ void load_fragments(float (&a)[KS], const float* tab, int mt, int lane) {
    const float* t = tab + (size_t)mt * KS * 64 + lane;
#pragma unroll
    for (int st = 0; st < KS; ++st) a[st] = t[st * 64];
}

template <class SH>
__global__ void __launch_bounds__(kMfThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
wpt2_deep_mfma_kernel(const MfParams p) {
    using L0 = MfLevel<SH, 0>;
    using L1 = MfLevel<SH, 1>;
    using L2 = MfLevel<SH, 2>;
    using L3 = MfLevel<SH, 3>;
    using L4 = MfLevel<SH, 4>;
    using L5 = MfLevel<SH, 5>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int groups = 256 / kMfG;
    const int b = blockIdx.x / groups;
    const int grp = blockIdx.x - b * groups;
    float* X = lds;
    float* Y = lds + p.X;
    // every float the k-loops can touch must be finite (rows past n_in meet zero matrix columns)
    for (int e = tid; e < p.X + p.Y; e += kMfThreads) lds[e] = 0.f;
    __syncthreads();
    {
        constexpr int n8 = SH::n[0];
        const float* wsb = p.ws + (size_t)b * n8 * 256 + grp * kMfG;
        for (int e = tid; e < n8 * kMfG; e += kMfThreads) {
            const int pos = e >> 5, j = e & 31;
            X[e] = wsb[(size_t)pos * 256 + j];
        }
    }
    __syncthreads();
    MfSink fs{};
    // levels 9 .. 11 over all 32 subtrees: 32 -> 64 -> 128 -> 256 columns
    if (wave < L0::G * L0::T) {
        float a[L0::KSB];
        load_fragments<L0::KSB>(a, p.tab + p.off[0], wave % L0::T, lane);
        mfma_level<L0, 5, 6, 1, false>(p, a, wave % L0::T, p.kst[0][wave % L0::T], wave / L0::T, X, Y, fs, lane);
    }
    __syncthreads();
    if (wave < L1::G * L1::T) {
        float a[L1::KSB];
        load_fragments<L1::KSB>(a, p.tab + p.off[1], wave % L1::T, lane);
        mfma_level<L1, 6, 7, 2, false>(p, a, wave % L1::T, p.kst[1][wave % L1::T], wave / L1::T, Y, X, fs, lane);
    }
    __syncthreads();
    if (wave < L2::G * L2::T) {
        float a[L2::KSB];
        load_fragments<L2::KSB>(a, p.tab + p.off[2], wave % L2::T, lane);
        mfma_level<L2, 7, 8, 4, false>(p, a, wave % L2::T, p.kst[2][wave % L2::T], wave / L2::T, X, Y, fs, lane);
    }
    __syncthreads();
    // keep the fragment loads of the last three levels below the wide levels (register pressure)
    asm volatile("" ::: "memory");
    // levels 12 .. 14 in four passes of 64 level-11 nodes: 64 -> 128 -> 256 columns -> HBM
    float a3[L3::KSB], a4[L4::KSB], a5[L5::KSB];
    load_fragments<L3::KSB>(a3, p.tab + p.off[3], wave % L3::T, lane);
    load_fragments<L4::KSB>(a4, p.tab + p.off[4], wave % L4::T, lane);
    load_fragments<L5::KSB>(a5, p.tab + p.off[5], wave % L5::T, lane);
    const int k3 = p.kst[3][wave % L3::T], k4 = p.kst[4][wave % L4::T], k5 = p.kst[5][wave % L5::T];
    float* C12 = X;
    float* C13 = X + p.C13;
    fs.P = (size_t)1 << p.level;
    fs.chan = (size_t)SH::n[6] * fs.P;
    const size_t nch = (p.flags & AFD_WPT_SIGN) ? 2 : 1;
    float* outb = p.out + (size_t)b * nch * fs.chan + (size_t)grp * kMfG * 64;
    for (int c = 0; c < 4; ++c) {
        if (wave < L3::G * L3::T)
            mfma_level<L3, 8, 7, 2, false>(p, a3, wave % L3::T, k3, wave / L3::T, Y + c * 64, C12, fs, lane);
        __syncthreads();
        if (wave < L4::G * L4::T)
            mfma_level<L4, 7, 8, 4, false>(p, a4, wave % L4::T, k4, wave / L4::T, C12, C13, fs, lane);
        __syncthreads();
        fs.outb = outb + c * 512;
        if (wave < L5::G * L5::T)
            mfma_level<L5, 8, 0, 8, true>(p, a5, wave % L5::T, k5, wave / L5::T, C13, nullptr, fs, lane);
        // the next pass overwrites C12 only after every wave left this pass's first two levels;
        // C13 is rewritten after the barrier that follows the next pass's first level
    }
}

template <int L> struct MfHas { static constexpr bool value = false; };
template <> struct MfHas<24> { static constexpr bool value = true; };
template <> struct MfHas<10> { static constexpr bool value = true; };
template <> struct MfHas<16> { static constexpr bool value = true; };

template <int L>
bool mf_shape_matches(const W2Params& p) {
    if constexpr (MfHas<L>::value) {
        for (int j = 0; j < 7; ++j)
            if (p.n[8 + j] != MfShape<L>::n[j]) return false;
        return true;
    }
    return false;
}

template <int L>
int launch_deep_mfma(const W2Params& p, hipStream_t stream) {
    if constexpr (MfHas<L>::value) {
        static bool mf_attr = false;
        if (!mf_attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt2_deep_mfma_kernel<MfShape<L>>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: hipFuncSetAttribute: %s", hipGetErrorString(e));
            mf_attr = true;
        }
        using SH = MfShape<L>;
        float* tab = p.ws + (size_t)p.B * p.n[p.Ks] * ((size_t)1 << p.Ks);
        MfTabParams tp{};
        MfParams q{};
        for (int m = 0; m < L; ++m) { tp.lo[m] = p.lo[m]; tp.hi[m] = p.hi[m]; }
        tp.L = L;
        for (int j = 0; j < 7; ++j) tp.n[j] = SH::n[j];
        int off = 0;
        auto level = [&](int j, int ksb, int tiles, auto kstart) {
            tp.ksb[j] = ksb; tp.tiles[j] = tiles; tp.off[j] = q.off[j] = off;
            for (int t = 0; t < tiles; ++t) tp.kst[j][t] = q.kst[j][t] = (short)kstart(t);
            off += tiles * ksb * 64;
        };
        level(0, MfLevel<SH, 0>::KSB, MfLevel<SH, 0>::T, MfLevel<SH, 0>::kstart);
        level(1, MfLevel<SH, 1>::KSB, MfLevel<SH, 1>::T, MfLevel<SH, 1>::kstart);
        level(2, MfLevel<SH, 2>::KSB, MfLevel<SH, 2>::T, MfLevel<SH, 2>::kstart);
        level(3, MfLevel<SH, 3>::KSB, MfLevel<SH, 3>::T, MfLevel<SH, 3>::kstart);
        level(4, MfLevel<SH, 4>::KSB, MfLevel<SH, 4>::T, MfLevel<SH, 4>::kstart);
        level(5, MfLevel<SH, 5>::KSB, MfLevel<SH, 5>::T, MfLevel<SH, 5>::kstart);
        if (off > p.mfTab) return afd::fail(AFD_ERR_WORKSPACE, "wpt: fragment table larger than planned");
        hipLaunchKernelGGL(wpt2_matrices_kernel, dim3((unsigned)(off / 64)), dim3(64), 0, stream, tp, tab);
        q.ws = p.ws; q.tab = tab; q.out = p.out;
        q.X = p.mfX; q.Y = p.mfY; q.C13 = p.mfC13; q.level = p.level;
        q.flags = p.flags; q.power = p.power; q.eps = p.eps; q.mean = p.mean; q.inv_std = p.inv_std;
        q.sgn_neg = p.sgn_neg; q.sgn_pos = p.sgn_pos;
        hipLaunchKernelGGL(wpt2_deep_mfma_kernel<MfShape<L>>, dim3((unsigned)p.B * (256 / kMfG)), dim3(kMfThreads),
                           (size_t)(p.mfX + p.mfY) * 4, stream, q);
    }
    return AFD_OK;
}

int child_len2(int n, int L) { return (n + L - 2 + (n & 1)) / 2; }

// returns 0, or 1 when this geometry is left to the first-generation kernel
int make_plan2(W2Params& p) {
    p.n[0] = p.N;
    for (int k = 1; k <= p.level; ++k) {
        const int prev = p.n[k - 1];
        if (p.L - 2 + (prev & 1) >= prev) return 1;
        p.n[k] = child_len2(prev, p.L);
    }
    auto pad = [](long n) { return n + (n >> 5) + 2; };
    if (pad(p.N) > kTopLdsFloats) return 1;
    // K1 = levels walked as a single path before the breadth-first part (2^K1 workgroups per frame).
    // K1 = 1: each workgroup owns a level-1 subtree -- 25 % fewer filter instructions than K1 = 2 (the
    // path levels compute both filters and keep one) and one round of workgroups at B = 128:
    // coif4 l14 front end 231 -> 182 us, sym5 l14 145 -> 106 us.  K1 = 2 remains the fallback when the
    // breadth-first levels of a level-1 subtree do not fit the LDS carve.
    p.K1 = p.level >= 3 ? 1 : p.level - 1;
    if (getenv("AFD_WPT_K1") && p.level >= 3) p.K1 = atoi(getenv("AFD_WPT_K1")) == 2 ? 2 : 1;
    for (int attempt = 0; attempt < 2; ++attempt) {
        bool fits = true;
        auto szk = [&](int k) { return (long)(1L << (k - p.K1)) * pad(p.n[k]); };
        const int ks = p.level < kKsMax ? p.level : kKsMax;
        for (int k = p.K1 + 1; k < ks; ++k) fits = fits && szk(k - 1) + szk(k) <= kTopLdsFloats;
        if (ks > p.K1 + 1) fits = fits && szk(ks - 1) <= kTopLdsFloats;
        if (fits || p.K1 == 2 || p.level < 3) break;
        p.K1 = 2;
    }
    for (int k = 1; k <= p.K1; ++k)
        if (pad(p.n[k - 1]) + pad(p.n[k]) > kTopLdsFloats) return 1;
    p.Ks = p.level < kKsMax ? p.level : kKsMax;
    auto size_at = [&](int k) { return (long)(1L << (k - p.K1)) * pad(p.n[k]); };
    for (int k = p.K1 + 1; k < p.Ks; ++k)
        if (size_at(k - 1) + size_at(k) > kTopLdsFloats) return 1;
    if (p.Ks > p.K1 + 1 && size_at(p.Ks - 1) > kTopLdsFloats) return 1;
    p.G = 1;
    p.offA = p.offB = p.deepFloats = 0;
    if (p.level > p.Ks) {
        int best = 0;
        for (int pass = 0; pass < 2 && best == 0; ++pass)
        for (int G = 64; G >= 1; G >>= 1) {
            if (G > (1 << p.Ks)) continue;
            long a = 0, bsz = 0;
            bool useA = true;
            for (int k = p.Ks + 1; k <= p.level - 1; ++k) {
                const long sz = (long)G * (1L << (k - p.Ks)) * pad(p.n[k]);
                if (useA) a = sz > a ? sz : a; else bsz = sz > bsz ? sz : bsz;
                useA = !useA;
            }
            const long in = (long)G * pad(p.n[p.Ks]);
            const long total = ((in + 3) & ~3L) + ((a + 3) & ~3L) + ((bsz + 3) & ~3L);
            const long budget = pass == 0 ? 13000 : 38000;  // 52 KB -> 3 workgroups per CU
            if (total <= budget) {
                best = G;
                p.offA = (int)((in + 3) & ~3L);
                p.offB = p.offA + (int)((a + 3) & ~3L);
                p.deepFloats = (int)total;
                break;
            }
        }
        if (best == 0) return 1;
        p.G = best;
    }
    // matrix-core path for the six levels below Ks = 8 (the level-14 transforms of 1 s frames)
    p.mf = 0;
    if (p.Ks == 8 && p.level == 14 && p.L > 2 && !getenv("AFD_WPT_NO_MFMA")) {
        bool ok = true;
        int off = 0;
        for (int j = 0; j < 6; ++j) {
            p.mfKs[j] = (p.n[8 + j] + 1) / 2;
            p.mfTiles[j] = (2 * p.n[9 + j] + 31) / 32;
            p.mfOff[j] = off;
            off += p.mfTiles[j] * p.mfKs[j] * 64;
            ok = ok && p.mfTiles[j] <= kMfWaves;
        }
        p.mfTab = off;
        auto mx = [](int a, int b) { return a > b ? a : b; };
        // buffers hold 2 * k-steps rows (the k-loop's reach) of the level's column count
        p.mfC13 = 2 * p.mfKs[4] * 128;
        p.mfX = mx(mx(2 * p.mfKs[0] * 32, 2 * p.mfKs[2] * 128), p.mfC13 + 2 * p.mfKs[5] * 256);
        p.mfY = mx(2 * p.mfKs[1] * 64, 2 * p.mfKs[3] * 256);
        // children written by a level must fit the rows its reader allocates
        for (int j = 0; j < 5; ++j) ok = ok && p.n[9 + j] <= 2 * p.mfKs[j + 1];
        if (ok && (size_t)(p.mfX + p.mfY) * 4 <= 80 * 1024) p.mf = 1;
    }
    return 0;
}

template <int L>
int launch2(const W2Params& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt2_top_kernel<L>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kTopLdsFloats * 4);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt2_deep_kernel<L>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const int C = (p.flags & AFD_WPT_SIGN) ? 2 : 1;
    afd::ScopedTiming timing(AFD_K_WPT, 4.0 * p.B * ((double)p.N + (double)C * p.n[p.level] * (double)(1L << p.level)), stream);
    hipLaunchKernelGGL(wpt2_top_kernel<L>, dim3((unsigned)p.B << p.K1), dim3(kTopThreads),
                       (size_t)kTopLdsFloats * 4, stream, p);
    if (p.level > p.Ks && p.mf && mf_shape_matches<L>(p)) {
        const int rc = launch_deep_mfma<L>(p, stream);
        if (rc != AFD_OK) return rc;
    } else if (p.level > p.Ks) {
        const unsigned groups = (1u << p.Ks) / p.G;
        hipLaunchKernelGGL(wpt2_deep_kernel<L>, dim3((unsigned)p.B * groups), dim3(kDeepThreads),
                           (size_t)p.deepFloats * 4, stream, p);
    }
    return afd::check_launch("wpt2 kernels");
}

}  // namespace

namespace afd {

bool wpt2_preferred(int L, int level) {
    if (getenv("AFD_WPT_V1")) return false;
    if (getenv("AFD_WPT_V2")) return true;
    return level >= 11 || L >= 12 || L == 2;
}

size_t wpt2_workspace_bytes(int B, int N, int L, int level) {
    if (!wpt2_preferred(L, level)) return 0;
    W2Params p{};
    p.B = B; p.N = N; p.L = L; p.level = level;
    if (make_plan2(p) != 0 || level <= p.Ks) return 0;
    return ((size_t)B * p.n[p.Ks] * ((size_t)1 << p.Ks) + (p.mf ? (size_t)p.mfTab : 0)) * sizeof(float);
}

// returns AFD_OK, an error, or 1 = "not handled here, use the first-generation kernel"
int wpt2_forward(const float* x, int B, int N, const float* dec_lo, const float* dec_hi, int L,
                 int level, unsigned flags, float power, float eps, float mean, float std, float sign_mean,
                 float sign_std, float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
    W2Params p{};
    p.x = x; p.out = out; p.ws = static_cast<float*>(ws);
    p.B = B; p.N = N; p.L = L; p.level = level;
    p.flags = flags; p.power = power; p.eps = eps; p.mean = mean; p.std = std;
    p.inv_std = (float)(1.0 / (double)(std == 0.f ? 1.f : std));
    p.sgn_neg = (flags & AFD_WPT_NORM) ? (-1.f - sign_mean) / sign_std : -1.f;
    p.sgn_pos = (flags & AFD_WPT_NORM) ? (1.f - sign_mean) / sign_std : 1.f;
    for (int m = 0; m < L; ++m) {
        p.lo[m] = dec_lo[m];
        p.hi[m] = dec_hi[m];
        p.rlo[m] = dec_lo[L - 1 - m];
        p.rhi[m] = dec_hi[L - 1 - m];
    }
    if (make_plan2(p) != 0) return 1;
    if (level > p.Ks) {
        const size_t need = ((size_t)B * p.n[p.Ks] * ((size_t)1 << p.Ks) + (p.mf ? (size_t)p.mfTab : 0)) * sizeof(float);
        if (!ws || ws_bytes < need) return afd::fail(AFD_ERR_WORKSPACE, "wpt: workspace of %zu bytes needed", need);
    }
    switch (L) {
        case 2: return launch2<2>(p, stream);
        case 4: return launch2<4>(p, stream);
        case 6: return launch2<6>(p, stream);
        case 8: return launch2<8>(p, stream);
        case 10: return launch2<10>(p, stream);
        case 16: return launch2<16>(p, stream);
        case 24: return launch2<24>(p, stream);
        default: return 1;
    }
}

}  // namespace afd
