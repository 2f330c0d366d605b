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

struct W2Params {
    const float* x;
    float* out;
    float* ws;
    int B, N, L, level, K1, Ks, G;
    int offA, offB, deepFloats;  // deep kernel LDS carve (floats)
    int n[kMaxLevel + 1];
    unsigned flags;
    float power, eps, mean, std, inv_std;
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

__device__ __forceinline__ float epilogue2(float v, const W2Params& p, unsigned flags) {
    if (flags & AFD_WPT_LOG) {
        if (p.power == 2.0f) {
            // v*v + eps >= 1e-12 is a normal float: the bare v_log_f32 (log2) needs no
            // denormal pre-scaling; ~1 ulp of log2, < 2e-6 absolute on these features
            v = __builtin_amdgcn_logf(fmaf(v, v, p.eps)) * 0.6931471805599453f;
        } else {
            v = pow_log_slow(v, p.power, p.eps);
        }
    }
    if (flags & AFD_WPT_NORM) v = (v - p.mean) * p.inv_std;
    return v;
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
            sg.x = v.x < 0.f ? -1.f : 1.f;
            sg.y = v.y < 0.f ? -1.f : 1.f;
            if (s.flags & AFD_WPT_NORM) {
                sg.x = (sg.x - p.mean) * p.inv_std;
                sg.y = (sg.y - p.mean) * p.inv_std;
            }
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
    const int maxC = (n_out + 7) >> 3;  // at least 8 outputs per unit (window fill amortised)
    if (C > maxC) C = maxC;
    int len = (n_out + C - 1) / C;
    {
        constexpr int blk = (L + 2 * kAhead) / 2;  // outputs come in blocks of W/2
        len = ((len + blk - 1) / blk) * blk;
    }
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
    p.K1 = p.level >= 3 ? 2 : p.level - 1;
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
    if (p.level > p.Ks) {
        const unsigned groups = (1u << p.Ks) / p.G;
        hipLaunchKernelGGL(wpt2_deep_kernel<L>, dim3((unsigned)p.B * groups), dim3(kDeepThreads),
                           (size_t)p.deepFloats * 4, stream, p);
    }
    return afd::check_launch("wpt2 kernels");
}

}  // namespace

namespace afd {

size_t wpt2_workspace_bytes(int B, int N, int L, int level) {
    if (level < 11 && !getenv("AFD_WPT_V2")) return 0;
    W2Params p{};
    p.B = B; p.N = N; p.L = L; p.level = level;
    if (make_plan2(p) != 0 || level <= p.Ks) return 0;
    return (size_t)B * p.n[p.Ks] * ((size_t)1 << p.Ks) * sizeof(float);
}

// returns AFD_OK, an error, or 1 = "not handled here, use the first-generation kernel"
int wpt2_forward(const float* x, int B, int N, const float* dec_lo, const float* dec_hi, int L,
                 int level, unsigned flags, float power, float eps, float mean, float std, float* out,
                 void* ws, size_t ws_bytes, hipStream_t stream) {
    W2Params p{};
    p.x = x; p.out = out; p.ws = static_cast<float*>(ws);
    p.B = B; p.N = N; p.L = L; p.level = level;
    p.flags = flags; p.power = power; p.eps = eps; p.mean = mean; p.std = std;
    p.inv_std = (float)(1.0 / (double)(std == 0.f ? 1.f : std));
    for (int m = 0; m < L; ++m) {
        p.lo[m] = dec_lo[m];
        p.hi[m] = dec_hi[m];
        p.rlo[m] = dec_lo[L - 1 - m];
        p.rhi[m] = dec_hi[L - 1 - m];
    }
    if (make_plan2(p) != 0) return 1;
    if (level > p.Ks) {
        const size_t need = (size_t)B * p.n[p.Ks] * ((size_t)1 << p.Ks) * sizeof(float);
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
