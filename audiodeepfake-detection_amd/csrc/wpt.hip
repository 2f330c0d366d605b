// Wavelet-packet front end for gfx950: the whole packet tree of one frame stays in LDS.
//
// Replaces the per-node F.pad + F.conv1d recursion of ptwt.WaveletPacket, the 2^level-way
// stack, log-power, sign channel, permute and Normalize of the reference
// (src/audiofakedetect/wavelet_math.py:167-263, :380-382) by ONE launch:
//
//   workgroup (b, j1)  = frame b, level-K1 node j1 (frequency order)      grid = B * 2^K1
//   stage 1  frame -> LDS; walk the K1 filters on the path to node j1 (one child per level)
//   stage 2  breadth-first levels K1+1..K2 of that subtree, ping-pong between the two ends
//            of the LDS arena (nodes kept in frequency order at every level)
//   stage 3  levels K2+1..level in chunks of G level-K2 nodes (the deep levels outgrow LDS
//            when taken breadth-first over the whole subtree)
//   final    the last level is never stored in LDS: lanes run over parent nodes, each lane
//            writes its two children (adjacent packets) as one float2 -> rows of the
//            [B][C][T][P] output are written in contiguous 512 B runs per wave.
//
// HBM traffic: the frame is read 2^K1 times (L2 / Infinity-Cache hits after the first),
// the features are written once.  Algorithmic bytes per frame = 4 * (N + C*P*T).
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>

namespace {

constexpr int kMaxLevel = 16;
constexpr int kMaxTaps = 128;  // coif17, the longest discrete wavelet of pywt, has 102 taps
constexpr int kThreads = 512;
constexpr int kLdsFloats = 40000;  // 160 000 B of the 163 840 B a workgroup may claim

struct WptParams {
    const float* x;
    float* out;
    int B, N, L, level, K1, K2, G;
    int n[kMaxLevel + 1];
    unsigned flags;
    float power, eps, mean, std;
    float sgn_neg, sgn_pos;  // the sign channel's two values, already normalised with its own statistics
    float lo[kMaxTaps], hi[kMaxTaps];
};

__device__ __forceinline__ int reflect_idx(int j, int n) {
    j = j < 0 ? -j : j;
    return j >= n ? 2 * (n - 1) - j : j;
}

// cA[i], cD[i] of one node: sum_m taps[m] * xe[2i+1-m], xe = whole-sample reflect extension
template <int LT>
__device__ __forceinline__ void analysis_pair(const float* __restrict__ x, int n, int i, int L,
                                              const WptParams& p, float& sa, float& sd) {
    const int j0 = 2 * i + 1;
    float a = 0.f, d = 0.f;
    if (j0 - (L - 1) >= 0 && j0 < n) {
        if (LT > 0) {
#pragma unroll
            for (int m = 0; m < LT; ++m) {
                const float v = x[j0 - m];
                a = fmaf(p.lo[m], v, a);
                d = fmaf(p.hi[m], v, d);
            }
        } else {
            for (int m = 0; m < L; ++m) {
                const float v = x[j0 - m];
                a = fmaf(p.lo[m], v, a);
                d = fmaf(p.hi[m], v, d);
            }
        }
    } else {
        if (LT > 0) {
#pragma unroll
            for (int m = 0; m < LT; ++m) {
                const float v = x[reflect_idx(j0 - m, n)];
                a = fmaf(p.lo[m], v, a);
                d = fmaf(p.hi[m], v, d);
            }
        } else {
            for (int m = 0; m < L; ++m) {
                const float v = x[reflect_idx(j0 - m, n)];
                a = fmaf(p.lo[m], v, a);
                d = fmaf(p.hi[m], v, d);
            }
        }
    }
    sa = a;
    sd = d;
}

__device__ __forceinline__ float epilogue(float v, const WptParams& p) {
    if (p.flags & AFD_WPT_LOG) {
        const float a = fabsf(v);
        const float pw = (p.power == 2.0f) ? a * a : powf(a, p.power);
        v = logf(pw + p.eps);
    }
    if (p.flags & AFD_WPT_NORM) v = (v - p.mean) / p.std;
    return v;
}

// one level, parents [M][n_in] -> children [2M][n_out], both in frequency order
template <int LT>
__device__ __forceinline__ void level_step(const WptParams& p, const float* src, int n_in, int M,
                                           int Fb, float* dst, int n_out, int L) {
    const int total = M * n_out;
    for (int w = threadIdx.x; w < total; w += kThreads) {
        const int q = w / n_out;
        const int i = w - q * n_out;
        float sa, sd;
        analysis_pair<LT>(src + q * n_in, n_in, i, L, p, sa, sd);
        const int par = (Fb + q) & 1;  // odd-frequency parents list (d, a)
        dst[(2 * q + par) * n_out + i] = sa;
        dst[(2 * q + 1 - par) * n_out + i] = sd;
    }
}

// last level: children go straight to HBM; lanes run over parents so packets are contiguous
template <int LT>
__device__ __forceinline__ void final_step(const WptParams& p, const float* src, int n_in, int M,
                                           int Fb, int b, int L) {
    const int T = p.n[p.level];
    const size_t P = (size_t)1 << p.level;
    const int C = (p.flags & AFD_WPT_SIGN) ? 2 : 1;
    const int total = M * T;
    const int shift = 31 - __clz(M);
    float* outb = p.out + (size_t)b * C * T * P;
    for (int w = threadIdx.x; w < total; w += kThreads) {
        const int i = w >> shift;
        const int q = w & (M - 1);
        float sa, sd;
        analysis_pair<LT>(src + q * n_in, n_in, i, L, p, sa, sd);
        const int F = Fb + q;
        const bool par = F & 1;
        const float v0 = par ? sd : sa;
        const float v1 = par ? sa : sd;
        const size_t o = (size_t)i * P + 2 * (size_t)F;
        float2 r;
        r.x = epilogue(v0, p);
        r.y = epilogue(v1, p);
        *reinterpret_cast<float2*>(outb + o) = r;
        if (C == 2) {
            float2 s;
            s.x = v0 < 0.f ? p.sgn_neg : p.sgn_pos;
            s.y = v1 < 0.f ? p.sgn_neg : p.sgn_pos;
            *reinterpret_cast<float2*>(outb + (size_t)T * P + o) = s;
        }
    }
}

template <int LT>
__global__ void __launch_bounds__(kThreads) wpt_fused_kernel(const WptParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int L = LT > 0 ? LT : p.L;
    const int tid = threadIdx.x;
    // the 2^K1 workgroups of a frame are B apart: with B % 8 == 0 they share an XCD's L2
    const int b = blockIdx.x % p.B;
    const int j1 = blockIdx.x / p.B;

    const float* xg = p.x + (size_t)b * p.N;
    // 8 loads in flight per thread (a load -> store chain pays one memory latency per element)
    for (int base = 0; base < p.N; base += kThreads * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kThreads + tid;
            v[u] = i < p.N ? xg[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kThreads + tid;
            if (i < p.N) lds[i] = v[u];
        }
    }
    __syncthreads();

    float* cur = lds;
    bool at_bottom = true;
    int n_cur = p.N;
    // stage 1: the K1 filters that lead to frequency index j1 (Gray code, MSB = level 1)
    const int g = j1 ^ (j1 >> 1);
    for (int k = 1; k <= p.K1; ++k) {
        const int n_out = p.n[k];
        float* dst = at_bottom ? (lds + kLdsFloats - n_out) : lds;
        const bool use_hi = (g >> (p.K1 - k)) & 1;
        for (int i = tid; i < n_out; i += kThreads) {
            float sa, sd;
            analysis_pair<LT>(cur, n_cur, i, L, p, sa, sd);
            dst[i] = use_hi ? sd : sa;
        }
        __syncthreads();
        cur = dst;
        at_bottom = !at_bottom;
        n_cur = n_out;
    }

    // stage 2: breadth-first inside LDS
    int M = 1;
    int Fb = j1;
    int k = p.K1 + 1;
    const int k2 = p.level < p.K2 ? p.level : p.K2;
    for (; k <= k2; ++k) {
        const int n_out = p.n[k];
        if (k == p.level) {
            final_step<LT>(p, cur, n_cur, M, Fb, b, L);
            return;
        }
        float* dst = at_bottom ? (lds + kLdsFloats - 2 * M * n_out) : lds;
        level_step<LT>(p, cur, n_cur, M, Fb, dst, n_out, L);
        __syncthreads();
        cur = dst;
        at_bottom = !at_bottom;
        n_cur = n_out;
        M *= 2;
        Fb *= 2;
    }

    // stage 3: chunks of G level-K2 nodes, scratch = the arena minus the level-K2 nodes
    float* free_lo = at_bottom ? (cur + M * n_cur) : lds;
    float* free_hi = at_bottom ? (lds + kLdsFloats) : cur;
    for (int c0 = 0; c0 < M; c0 += p.G) {
        const float* src = cur + c0 * n_cur;
        int n_in = n_cur;
        int Mc = p.G;
        int Fc = Fb + c0;
        bool s_bottom = true;
        for (int kk = k; kk <= p.level; ++kk) {
            const int n_out = p.n[kk];
            if (kk == p.level) {
                final_step<LT>(p, src, n_in, Mc, Fc, b, L);
                break;
            }
            float* dst = s_bottom ? free_lo : (free_hi - 2 * Mc * n_out);
            level_step<LT>(p, src, n_in, Mc, Fc, dst, n_out, L);
            __syncthreads();
            src = dst;
            s_bottom = !s_bottom;
            n_in = n_out;
            Mc *= 2;
            Fc *= 2;
        }
        __syncthreads();
    }
}

int child_len(int n, int L) { return (n + L - 2 + (n & 1)) / 2; }

// One analysis step of long frames straight from global memory (afd_wpt_analysis_step): thread = output position of one
// frame, both children.  For frames whose packet tree does not fit LDS the host splits level 1 (2, ...) off with this
// kernel and hands the children -- frames of half the length -- to the LDS-resident kernels.
struct StepParams {
    const float* x;
    float* ca;
    float* cd;
    int B, N, L, n_out;
    float lo[kMaxTaps], hi[kMaxTaps];
};

__global__ void __launch_bounds__(256) wpt_step_kernel(const StepParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= p.n_out) return;
    const float* x = p.x + (size_t)b * p.N;
    const int j0 = 2 * i + 1;
    float a = 0.f, d = 0.f;
    if (j0 - (p.L - 1) >= 0 && j0 < p.N) {
        for (int m = 0; m < p.L; ++m) {
            const float v = x[j0 - m];
            a = fmaf(p.lo[m], v, a);
            d = fmaf(p.hi[m], v, d);
        }
    } else {
        for (int m = 0; m < p.L; ++m) {
            const float v = x[reflect_idx(j0 - m, p.N)];
            a = fmaf(p.lo[m], v, a);
            d = fmaf(p.hi[m], v, d);
        }
    }
    p.ca[(size_t)b * p.n_out + i] = a;
    p.cd[(size_t)b * p.n_out + i] = d;
}

// fills n[], K1, K2, G; returns 0 or an error
int make_plan(WptParams& p) {
    p.n[0] = p.N;
    for (int k = 1; k <= p.level; ++k) {
        const int prev = p.n[k - 1];
        if (p.L - 2 + (prev & 1) >= prev)
            return afd::fail(AFD_ERR_UNSUPPORTED,
                             "wpt: reflect pad %d >= node length %d at level %d",
                             p.L - 2 + (prev & 1), prev, k);
        p.n[k] = child_len(prev, p.L);
    }
    p.K1 = p.level >= 3 ? 2 : p.level - 1;
    // stage 1 capacity
    for (int k = 1; k <= p.K1; ++k)
        if (p.n[k - 1] + p.n[k] > kLdsFloats)
            return afd::fail(AFD_ERR_UNSUPPORTED, "wpt: frame of %d samples does not fit LDS", p.N);
    if (p.N > kLdsFloats) return afd::fail(AFD_ERR_UNSUPPORTED, "wpt: frame too long");
    // stage 2: deepest level whose parent+child sets fit together
    auto size_at = [&](int k) { return (long)(1L << (k - p.K1)) * p.n[k]; };
    int K2 = p.K1;
    for (int k = p.K1 + 1; k <= p.level && k <= 8; ++k) {
        if (k == p.level) {  // final level is not stored
            K2 = k;
            break;
        }
        if (size_at(k - 1) + size_at(k) > kLdsFloats) break;
        K2 = k;
    }
    p.K2 = K2;
    p.G = 1;
    if (p.level > K2) {
        if (K2 == p.K1 && size_at(K2) > kLdsFloats)
            return afd::fail(AFD_ERR_UNSUPPORTED, "wpt: level-%d node set does not fit LDS", K2);
        const long free_floats = kLdsFloats - size_at(K2);
        const int M2 = 1 << (K2 - p.K1);
        int G = M2;
        for (; G >= 1; G >>= 1) {
            bool ok = true;
            for (int kk = K2 + 1; kk <= p.level - 1 && ok; ++kk) {
                long need = (long)G * (1L << (kk - K2)) * p.n[kk];
                if (kk + 1 <= p.level - 1) need += (long)G * (1L << (kk + 1 - K2)) * p.n[kk + 1];
                if (need > free_floats) ok = false;
            }
            if (ok) break;
        }
        if (G < 1) return afd::fail(AFD_ERR_UNSUPPORTED, "wpt: deep levels do not fit LDS");
        p.G = G;
    }
    return AFD_OK;
}

template <int LT>
int launch(const WptParams& p, hipStream_t stream) {
    static afd::PerDeviceOnce attr_set;
    const size_t lds_bytes = (size_t)kLdsFloats * sizeof(float);
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt_fused_kernel<LT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds_bytes);
        if (e != hipSuccess)
            return afd::fail(AFD_ERR_HIP, "wpt: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set.mark();
    }
    const unsigned grid = (unsigned)p.B << p.K1;
    const int C = (p.flags & AFD_WPT_SIGN) ? 2 : 1;
    afd::ScopedTiming timing(AFD_K_WPT, 4.0 * p.B * ((double)p.N + (double)C * p.n[p.level] * (double)(1L << p.level)), stream);
    hipLaunchKernelGGL(wpt_fused_kernel<LT>, dim3(grid), dim3(kThreads), lds_bytes, stream, p);
    return afd::check_launch("wpt_fused_kernel");
}

}  // namespace

extern "C" int afd_wpt_out_len(int N, int L, int level) {
    if (N < 2 || L < 2 || (L & 1) || level < 0 || level > kMaxLevel) return -1;
    int n = N;
    for (int k = 0; k < level; ++k) {
        if (L - 2 + (n & 1) >= n) return -1;
        n = child_len(n, L);
    }
    return n;
}

namespace afd {
int wpt_haar14_forward(const float* x, int B, int N, const float* dec_lo, int L, int level,
                       unsigned flags, float power, float eps, float mean, float std, float sign_mean,
                       float sign_std, float* out, hipStream_t stream);
size_t wpt3_workspace_bytes(int B, int N, int L, int level);
int wpt3_forward(const float* x, int B, int N, const float* dec_lo, const float* dec_hi, int L,
                 int level, unsigned flags, float power, float eps, float mean, float std, float sign_mean,
                 float sign_std, float* out, void* ws, size_t ws_bytes, hipStream_t stream);
}  // namespace afd

extern "C" int afd_wpt_analysis_step(const float* x, int B, int N, const float* dec_lo, const float* dec_hi, int L,
                                     float* ca, float* cd, afd_stream_t stream) {
    if (!x || !dec_lo || !dec_hi || !ca || !cd) return afd::fail(AFD_ERR_ARG, "wpt step: null pointer");
    if (B < 1 || B > 65535 || N < 2) return afd::fail(AFD_ERR_ARG, "wpt step: bad shape B=%d N=%d", B, N);
    if (L < 2 || L > kMaxTaps || (L & 1)) return afd::fail(AFD_ERR_ARG, "wpt step: filter length %d", L);
    if (L - 2 + (N & 1) >= N) return afd::fail(AFD_ERR_UNSUPPORTED, "wpt step: reflect pad %d >= frame length %d", L - 2 + (N & 1), N);
    StepParams p{};
    p.x = x; p.ca = ca; p.cd = cd; p.B = B; p.N = N; p.L = L;
    p.n_out = child_len(N, L);
    for (int m = 0; m < L; ++m) {
        p.lo[m] = dec_lo[m];
        p.hi[m] = dec_hi[m];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    afd::ScopedTiming timing(AFD_K_WPT, 4.0 * B * ((double)N + 2.0 * p.n_out), s);
    hipLaunchKernelGGL(wpt_step_kernel, dim3((p.n_out + 255) / 256, B), dim3(256), 0, s, p);
    return afd::check_launch("wpt_step_kernel");
}

extern "C" size_t afd_wpt_workspace_bytes(int B, int N, int L, int level) {
    if (B < 1 || N < 2 || L < 2 || (L & 1) || L > kMaxTaps || level < 1 || level > kMaxLevel) return 0;
    return afd::wpt3_workspace_bytes(B, N, L, level);
}

extern "C" int afd_wpt_forward(const float* x, int B, int N, const float* dec_lo,
                               const float* dec_hi, int L, int level, unsigned flags, float power,
                               float eps, float mean, float std, float sign_mean, float sign_std,
                               float* out, void* ws, size_t ws_bytes, afd_stream_t stream) {
    if (!x || !out || !dec_lo || !dec_hi) return afd::fail(AFD_ERR_ARG, "wpt: null pointer");
    if (B < 1 || N < 2) return afd::fail(AFD_ERR_ARG, "wpt: bad shape B=%d N=%d", B, N);
    if (L < 2 || L > kMaxTaps || (L & 1))
        return afd::fail(AFD_ERR_ARG, "wpt: filter length %d not in {2,4,..,%d}", L, kMaxTaps);
    if (level < 1 || level > kMaxLevel) return afd::fail(AFD_ERR_ARG, "wpt: level %d", level);
    if ((flags & AFD_WPT_SIGN) && !(flags & AFD_WPT_LOG))
        return afd::fail(AFD_ERR_ARG, "wpt: AFD_WPT_SIGN needs AFD_WPT_LOG");
    if ((flags & AFD_WPT_NORM) && (std == 0.f || ((flags & AFD_WPT_SIGN) && sign_std == 0.f)))
        return afd::fail(AFD_ERR_ARG, "wpt: std == 0");
    if ((long)B << (level >= 3 ? 2 : level - 1) > 0x7fffffffL)
        return afd::fail(AFD_ERR_ARG, "wpt: batch too large");
    if (!getenv("AFD_WPT_NO_HAAR")) {
        // Haar, level 14, 22 050-sample frames: dedicated add/subtract network (wpt_haar.hip)
        const int rch = afd::wpt_haar14_forward(x, B, N, dec_lo, L, level, flags, power, eps, mean, std,
                                                sign_mean, sign_std, out, static_cast<hipStream_t>(stream));
        if (rch != 1) return rch;
    }
    // third generation (wpt3.hip): padded-node vector kernel for levels <= 8, matrix-core composite for the
    // level-14 transforms of 1 s frames; 1 = not its case
    {
        const int rc3 = afd::wpt3_forward(x, B, N, dec_lo, dec_hi, L, level, flags, power, eps, mean, std,
                                          sign_mean, sign_std, out, ws, ws_bytes, static_cast<hipStream_t>(stream));
        if (rc3 != 1) return rc3;
    }
    // everything else -- 2 taps, 34..128 taps (coif6..17, db17..38, sym17..20, dmey), levels 9..13, other frame lengths,
    // banks without a lattice at level 14 -- on the single-launch kernel below
    WptParams p{};
    p.x = x;
    p.out = out;
    p.B = B;
    p.N = N;
    p.L = L;
    p.level = level;
    p.flags = flags;
    p.power = power;
    p.eps = eps;
    p.mean = mean;
    p.std = std;
    p.sgn_neg = (flags & AFD_WPT_NORM) ? (-1.f - sign_mean) / sign_std : -1.f;
    p.sgn_pos = (flags & AFD_WPT_NORM) ? (1.f - sign_mean) / sign_std : 1.f;
    for (int m = 0; m < L; ++m) {
        p.lo[m] = dec_lo[m];
        p.hi[m] = dec_hi[m];
    }
    int rc = make_plan(p);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (L) {
        case 2: return launch<2>(p, s);
        case 4: return launch<4>(p, s);
        case 10: return launch<10>(p, s);
        case 16: return launch<16>(p, s);
        case 24: return launch<24>(p, s);
        default: return launch<0>(p, s);
    }
}
