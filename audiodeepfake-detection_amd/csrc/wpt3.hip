// Wavelet-packet front end, third generation (reference src/audiofakedetect/wavelet_math.py:167-263,
// :380-382; same contract as wpt.hip).
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
//   Level 14 of 1 s frames: levels 1..8 here (the level-8 nodes leave as a hand-off image [B][n8][256]), levels
//   9..14 in wpt4.hip (lattice form on the vector ALU; it replaced the matrix-core composite of rounds 2-3).
#include "wpt_shared.h"

#include <cstdlib>
#include <type_traits>

namespace {

using namespace afd::wptc;

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a template argument
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// ------------------------------------------------------------------------------------------------
// top levels
// ------------------------------------------------------------------------------------------------
struct T3Params {
    const float* x;
    float* dst;  // features (final) or the level-Ks hand-off image
    int B, N, Ks;
    int n[kKsMax + 1];
    int off[kKsMax + 1];    // LDS offset (floats) of level k's image; sample 0 of node 0 at off + L - 2
    int pitch[kKsMax + 1];  // node pitch (floats): pitch / 4 odd (16-byte aligned nodes, node-strided reads conflict-free)
    unsigned magic[kKsMax + 1];  // floor(2^32 / m) + 1, m = (n[k] + 1) / 2 output pairs per node
    Epi e;
    float rlo[kMaxTaps], rhi[kMaxTaps];  // taps reversed: rlo[t] = dec_lo[L - 1 - t]
};

// Geometry of the standard 1 s frame (N = 22 050, levels 0..8) at compile time: the same node lengths, pitches and LDS
// offsets plan_top computes.  The STD instance of the kernel takes them from here, its level loop is unrolled, and
// every division, pitch product and image offset is a constant: the kernel is bound by instruction issue (64 % of the
// SIMD cycles vector-ALU busy, 2 358 vector + 675 scalar instructions per wave of which 1 250 are the packed FMAs).
template <int L, int NT = kTopThreads> struct Std3 {
    static constexpr int N = 22050, Ks = kKsMax;
    static constexpr int n_at(int k) {
        int n = N;
        for (int i = 0; i < k; ++i) n = (n + L - 2 + (n & 1)) / 2;
        return n;
    }
    // Outputs per work item at level k (2^(k-2) parents in a half frame, both children of each): the smallest even run
    // with which the level is ONE round of the 1024 threads.  Every level of a half frame holds about 5 500 positions
    // whatever k, so pairs (W = 2) are 2.7 rounds -- three, the last one a quarter full -- and W = 4 is 1.36 rounds, two
    // with the second a third full (which is why four outputs per item measured 1-2 % in round 4); W = 6 is 0.90-0.97 of
    // a round for coif4 / sym5 / db8 (W = 10 at coif4's level 7, 14 at level 1).  The item's window is L + 2 (W - 1) samples: 1.5 LDS
    // reads of 16 bytes per position instead of 3.5, and the index / address / bounds arithmetic of an item is paid once
    // per W positions -- the kernel is bound by the SUM of its LDS and vector-instruction time (profiles/r05_frontend_floor.md).
    // Levels 1 .. Ks-1 run lanes along the positions of a node: neighbouring lanes' windows start 2 W floats apart, and
    // 16-byte LDS reads are conflict-free when W / 2 is odd (W = 6: 48-byte lane stride, the 16 lanes of a phase cover the
    // 64 banks once; W = 8 would put lanes 0, 4, 8, 12 on the same banks).  The last level runs lanes along the NODES
    // (pitch / 4 odd: conflict-free at any W).
    static constexpr int run_at(int k) {
        const int parents = k < 2 ? 1 : 1 << (k - 2);
        for (int w = 6; w <= kMaxRun; w += (k == Ks ? 2 : 4))
            if (parents * ((n_at(k) + w - 1) / w) <= NT) return w;
        return kMaxRun;
    }
    // slack behind a node: the window of an item of the node's last positions (run of the level that READS it) runs
    // 2 (W - 1) + 3 floats past the last sample it needs (31 floats cover the runs of the 1024-thread instance)
    static constexpr int pitch_at(int k) {
        return padded_pitch(n_at(k), L, NT == kTopThreads || k >= Ks ? 31 : 2 * (run_at(k + 1) - 1) + 5);
    }
    static constexpr long size_at(int k) { return (k == 0 ? 1L : (1L << (k - 1))) * pitch_at(k); }
    static constexpr int off_at(int k) { return (k & 1) && k < Ks ? (int)((kTopLdsFloats - size_at(k)) & ~3L) : 0; }
};

// FIN: -1 = the level-Ks image is a hand-off to the deep kernel (no epilogue), else the epilogue mode
// STD: the standard frame at 8 levels with compile-time geometry (Std3); otherwise everything comes from T3Params
template <int L, int FIN, bool STD = false, int NT = kTopThreads>
__global__ void __launch_bounds__(NT) wpt3_top_kernel(const T3Params p) {
    using SG = Std3<L, NT>;
    constexpr int kTopThreads = NT;  // (shadows the default of wpt_shared.h inside this kernel)
    const int pN = STD ? SG::N : p.N;
    const int pKs = STD ? SG::Ks : p.Ks;
    auto n_of = [&](int k) { return STD ? SG::n_at(k) : p.n[k]; };
    auto off_of = [&](int k) { return STD ? SG::off_at(k) : p.off[k]; };
    auto pitch_of = [&](int k) { return STD ? SG::pitch_at(k) : p.pitch[k]; };
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PAD = L - 2;
    const int tid = threadIdx.x;
    // The two halves of a frame read the same 88 KB: they are EIGHT workgroups apart -- same XCD under the round-robin
    // placement of workgroups over the 8 XCDs, so one L2 serves both, and close enough in dispatch order to run at the
    // same time, so the second reader hits (rounds 2-3 placed them B apart: at B = 4096 the line had long left the
    // L2 and every frame came from HBM twice).  Blocks of 16: ids 0..7 = frames 8 g .. 8 g + 7 low halves, 8..15 high.
    int b, h;
    if (p.B % 8 == 0) {
        const int grp = blockIdx.x >> 4, r = blockIdx.x & 15;
        b = 8 * grp + (r & 7);
        h = r >> 3;
    } else {
        b = blockIdx.x % p.B;
        h = blockIdx.x / p.B;
    }

    // ---- frame -> level-0 image (with both reflect pads) ----
    {
        const float* xg = p.x + (size_t)b * pN;
        float* X0 = lds + off_of(0) + PAD;
        if ((pN & 1) == 0) {
            const float2* xv = reinterpret_cast<const float2*>(xg);  // frames are 8-byte aligned
            const int n2 = pN >> 1;
            // 11 loads in flight per thread: the 22 050-sample frame arrives in ONE round of memory latency
            constexpr int UN = 11 * 1024 / NT;
            for (int base = 0; base < n2; base += kTopThreads * UN) {
                float2 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int i = base + u * kTopThreads + tid;
                    v[u] = i < n2 ? xv[i] : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int i = base + u * kTopThreads + tid;
                    if (i < n2) *reinterpret_cast<float2*>(X0 + 2 * i) = v[u];
                }
            }
        } else {
            for (int i = tid; i < pN; i += kTopThreads) X0[i] = xg[i];
        }
        if (tid >= 1 && tid <= PAD) X0[-tid] = xg[tid];
        const int j = tid - 64;
        if (j >= 1 && j <= PAD + (pN & 1)) X0[pN - 1 + j] = xg[pN - 1 - j];
    }
    __syncthreads();
#if defined(AFD_TOP_STOP)
    if (AFD_TOP_STOP == 0) { if (tid == 0) p.dst[blockIdx.x] = lds[off_of(0) + PAD + 5]; return; }
#endif

    // ---- level 1: this workgroup's child of the frame (h = 0 low-pass, 1 high-pass) ----
    if (pKs == 1) return;  // not built: the caller keeps one-level transforms on the first-generation kernel
    {
        const float* src = lds + off_of(0);
        float* node = lds + off_of(1) + PAD;
        const int n1 = n_of(1);
        const float* taps = h ? p.rhi : p.rlo;
        if constexpr (STD) {
            // one round: every thread a run of W1 outputs of the one child (14 for the standard frame: a 50-sample window)
            constexpr int W1 = SG::run_at(1);
            constexpr int items = (SG::n_at(1) + W1 - 1) / W1;
            static_assert(items <= kTopThreads, "level 1 is one round of the workgroup");
            if (tid < items) {
                const int i = W1 * tid;
                Window<L, W1> win;
                win.load(src + 2 * i);
                float c1[W1];
                static_for<W1>([&](auto u) { c1[u.value] = win.template dot<u.value>(taps); });
#pragma unroll
                for (int u = 0; u < W1; ++u)
                    if (i + u < n1) put<L>(node, i + u, n1, c1[u]);
            }
        } else {
            for (int j = tid; 2 * j < n1; j += kTopThreads) {
                Window<L> win;
                win.load(src + 4 * j);
                put<L>(node, 2 * j, n1, win.template dot<0>(taps));
                if (2 * j + 1 < n1) put<L>(node, 2 * j + 1, n1, win.template dot<1>(taps));
            }
        }
    }
    __syncthreads();
#if defined(AFD_TOP_STOP)
    if (AFD_TOP_STOP == 1) { if (tid == 0) p.dst[blockIdx.x] = lds[off_of(1) + PAD + 5]; return; }
#endif

    // ---- levels 2 .. Ks-1: both children of every node, lanes along the output index ----
    // W outputs of both children per work item (Std3::run_at for the compile-time geometry, pairs otherwise)
    auto level = [&](const int k, auto wtag) {
        constexpr int W = decltype(wtag)::value;
        const int Mp = 1 << (k - 2);  // parents (nodes of level k-1 in this half)
        const int nk = n_of(k);
        const int mk = (nk + W - 1) / W;  // items per node
        const int total = Mp * mk;
        const float* src0 = lds + off_of(k - 1);
        float* dst0 = lds + off_of(k) + PAD;
        const int pin = pitch_of(k - 1), pout = pitch_of(k);
        const unsigned magic = STD ? 0u : p.magic[k];
        const int padr = PAD + (nk & 1);
        for (int idx = tid; idx < total; idx += kTopThreads) {
            const int q = STD ? idx / mk : (int)__umulhi((unsigned)idx, magic);
            const int i = W * (idx - q * mk);
            Window<L, W> win;
            win.load(src0 + q * pin + 2 * i);
            float ca[W], cd[W];
            static_for<W>([&](auto u) {
                ca[u.value] = win.template dot<u.value>(p.rlo);
                cd[u.value] = win.template dot<u.value>(p.rhi);
            });
            // odd-frequency parents list their children (d, a)
            const int par = (k == 2) ? h : (q & 1);
            float* na = dst0 + (2 * q + par) * pout;
            float* nd = dst0 + (2 * q + 1 - par) * pout;
#pragma unroll
            for (int u = 0; u < W; u += 2) {
                if (i + u + 1 < nk) {
                    *reinterpret_cast<f2*>(na + i + u) = f2{ca[u], ca[u + 1]};
                    *reinterpret_cast<f2*>(nd + i + u) = f2{cd[u], cd[u + 1]};
                } else if (i + u < nk) {
                    na[i + u] = ca[u];
                    nd[i + u] = cd[u];
                }
            }
            // the L-2 coefficients next to a border also fill the pad slots that mirror them
            if (i <= PAD) {
#pragma unroll
                for (int u = 0; u < W; ++u) {
                    const int iu = i + u;
                    if (iu >= 1 && iu <= PAD && iu < nk) { na[-iu] = ca[u]; nd[-iu] = cd[u]; }
                }
            }
            if (nk - 2 - (i + W - 1) < padr) {
#pragma unroll
                for (int u = 0; u < W; ++u) {
                    const int iu = i + u;
                    if ((unsigned)(nk - 2 - iu) < (unsigned)padr) { na[2 * (nk - 1) - iu] = ca[u]; nd[2 * (nk - 1) - iu] = cd[u]; }
                }
            }
        }
        __syncthreads();
    };
    if constexpr (STD) {  // six inlined copies with a constant level: every length, pitch and offset folds
        level(2, std::integral_constant<int, SG::run_at(2)>{}); level(3, std::integral_constant<int, SG::run_at(3)>{});
#if defined(AFD_TOP_STOP)
        if (AFD_TOP_STOP == 3) { if (tid == 0) p.dst[blockIdx.x] = lds[off_of(3) + PAD + 5]; return; }
#endif
        level(4, std::integral_constant<int, SG::run_at(4)>{}); level(5, std::integral_constant<int, SG::run_at(5)>{});
#if defined(AFD_TOP_STOP)
        if (AFD_TOP_STOP == 5) { if (tid == 0) p.dst[blockIdx.x] = lds[off_of(5) + PAD + 5]; return; }
#endif
        level(6, std::integral_constant<int, SG::run_at(6)>{}); level(7, std::integral_constant<int, SG::run_at(7)>{});
#if defined(AFD_TOP_STOP)
        if (AFD_TOP_STOP == 7) { if (tid == 0) p.dst[blockIdx.x] = lds[off_of(7) + PAD + 5]; return; }
#endif
    } else {
        for (int k = 2; k < pKs; ++k) level(k, std::integral_constant<int, 2>{});
    }

    // ---- level Ks: lanes along the nodes, results leave the chip packet-contiguous ----
    {
        const int k = pKs;
        const int logM = k - 2;  // parents in this half: 2^(k-2)
        const int Mp = 1 << logM;
        const int nk = n_of(k);
        constexpr int WL = STD ? SG::run_at(SG::Ks) : 2;  // outputs per item (one round of the workgroup in the STD instance)
        const int total = ((nk + WL - 1) / WL) << logM;
        const float* src0 = lds + off_of(k - 1);
        const int pin = pitch_of(k - 1);
        const size_t P = (size_t)1 << k;
        const size_t chan = (size_t)nk * P;
        const int nch = (FIN >= 0 && (p.e.flags & AFD_WPT_SIGN)) ? 2 : 1;
        float* outb = p.dst + (size_t)b * nch * chan;
        for (int idx = tid; idx < total; idx += kTopThreads) {
            const int q = idx & (Mp - 1);
            const int i = WL * (idx >> logM);
            Window<L, WL> win;
            win.load(src0 + q * pin + 2 * i);
            const int par = (k == 2) ? h : (q & 1);  // odd-frequency parents list their children (d, a)
            float* o = outb + (size_t)i * P + 2 * ((size_t)(h << logM) + q);
            static_for<WL>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (i + u < nk) {
                    const float ca = win.template dot<u>(p.rlo);
                    const float cd = win.template dot<u>(p.rhi);
                    f2 v;
                    v.x = par ? cd : ca;
                    v.y = par ? ca : cd;
                    if (FIN < 0) {
                        *reinterpret_cast<f2*>(o + u * P) = v;
                    } else {
                        f2 r;
                        r.x = epi_value<(FIN < 0 ? 0 : FIN)>(v.x, p.e);
                        r.y = epi_value<(FIN < 0 ? 0 : FIN)>(v.y, p.e);
                        *reinterpret_cast<f2*>(o + u * P) = r;
                        if (p.e.flags & AFD_WPT_SIGN) {
                            f2 sg;
                            sg.x = v.x < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                            sg.y = v.y < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                            *reinterpret_cast<f2*>(o + chan + u * P) = sg;
                        }
                    }
                }
            });
        }
    }
}

int child_len3(int n, int L) { return (n + L - 2 + (n & 1)) / 2; }

// LDS plan of the top kernel; false when a level pair does not fit
bool plan_top(T3Params& p, int L) {
    long size[kKsMax + 1];
    for (int k = 0; k <= p.Ks; ++k) {
        // node = [L-2 left pad | n samples | L-2 (+1) right pad | up to 3 floats an odd node's last item reads]
        // (31 floats of slack behind a node cover the runs of the compile-time instances, L = 10 / 16 / 24; the run-time
        // instance works in pairs and 5 suffice -- which is what lets the 60-tap coif10 keep two levels in LDS)
        const int pitch = padded_pitch(p.n[k], L, (L == 10 || L == 16 || L == 24) ? 31 : 5);
        p.pitch[k] = pitch;
        const long nodes = k == 0 ? 1 : (1L << (k - 1));
        size[k] = nodes * pitch;
        p.magic[k] = (unsigned)(0x100000000ULL / (unsigned)((p.n[k] + 1) / 2)) + 1u;
    }
    // images alternate between the two ends of the carve; the last level is never stored
    for (int k = 0; k < p.Ks; ++k) {
        if (size[k] > kTopLdsFloats) return false;
        if (k + 1 < p.Ks && size[k] + size[k + 1] > kTopLdsFloats) return false;
        p.off[k] = (k & 1) ? (int)((kTopLdsFloats - size[k]) & ~3L) : 0;
    }
    p.off[p.Ks] = 0;
    return true;
}

// the runtime plan is exactly the compile-time geometry of Std3<L> (the standard frame at 8 levels)
template <int L>
bool is_std_plan(const T3Params& p) {
    using SG = Std3<L>;
    if (p.N != SG::N || p.Ks != SG::Ks || getenv("AFD_WPT_TOP_RUNTIME")) return false;
    for (int k = 0; k <= SG::Ks; ++k)
        if (p.n[k] != SG::n_at(k) || p.pitch[k] != SG::pitch_at(k) || p.off[k] != SG::off_at(k)) return false;
    return true;
}

// the compile-time LDS plan of an instance with its own thread count fits the arena (same rule as plan_top)
template <int L, int NT>
constexpr bool std_plan_fits() {
    using SG = Std3<L, NT>;
    for (int k = 0; k < SG::Ks; ++k) {
        if (SG::size_at(k) > kTopLdsFloats) return false;
        if (k + 1 < SG::Ks && SG::size_at(k) + SG::size_at(k + 1) > kTopLdsFloats) return false;
    }
    return true;
}

template <int L, int FIN, bool STD, int NT = kTopThreads>
int launch_top_as(const T3Params& p, hipStream_t stream) {
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt3_top_kernel<L, FIN, STD, NT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kTopLdsFloats * 4);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr.mark();
    }
    hipLaunchKernelGGL((wpt3_top_kernel<L, FIN, STD, NT>), dim3((unsigned)p.B * 2), dim3(NT),
                       (size_t)kTopLdsFloats * 4, stream, p);
    return afd::check_launch("wpt3_top_kernel");
}

template <int L, int FIN>
int launch_top(const T3Params& p, hipStream_t stream) {
    // the tap counts of the shipped and BASELINE configurations get the compile-time instance
    if constexpr (L == 24 || L == 10 || L == 16) {
        if (is_std_plan<L>(p)) {
            if constexpr (L == 10) {
                static const int nt = getenv("AFD_WPT_TOP_THREADS") ? atoi(getenv("AFD_WPT_TOP_THREADS")) : 1024;
                // round 6 A/B (profiles/r06_sym5_top_threads.txt): half the threads with runs of 14-22 outputs per work
                // item (one 16-byte LDS read per 1.7 positions instead of 1.5 per position) is SLOWER, 0.467 -> 0.550 ms
                // at level 8 / B = 4096; 256 threads with runs of 22-46 spill and take 1.9 ms.  Kept for the record.
                static_assert(std_plan_fits<L, 512>(), "LDS plan of the 512-thread instance");
                if (nt == 512) return launch_top_as<L, FIN, true, 512>(p, stream);
            }
            return launch_top_as<L, FIN, true>(p, stream);
        }
    }
    return launch_top_as<L, FIN, false>(p, stream);
}

template <int L>
int launch3(T3Params& p, const float* dec_lo, const float* dec_hi, float* out, void* ws, int level, int t_len,
            hipStream_t stream) {
    const int C = (p.e.flags & AFD_WPT_SIGN) ? 2 : 1;
    afd::ScopedTiming timing(AFD_K_WPT, 4.0 * p.B * ((double)p.N + (double)C * t_len * (double)(1L << level)), stream);
    const int mode = epi_mode(p.e.flags, p.e.power);
    if (level <= kKsMax) {
        p.dst = out;
        // the sign channel and the slow powers take the generic instance (flags read at run time there)
        if (mode == EPI_LOG2) return launch_top<L, EPI_LOG2>(p, stream);
        if (mode == EPI_RAW) return launch_top<L, EPI_RAW>(p, stream);
        return launch_top<L, EPI_SLOW>(p, stream);
    }
    if constexpr (HasShape3<L>::value) {
        // level 14 of the standard frame: levels 1..8 here, hand-off image [B][n8][256], levels 9..14 in lattice form
        // (wpt4.hip).  The caller has checked that the taps have a lattice (wpt4_available).
        p.dst = static_cast<float*>(ws);
        int rc = launch_top<L, -1>(p, stream);
        if (rc != AFD_OK) return rc;
        return afd::wpt4_deep(static_cast<const float*>(ws), out, p.B, dec_lo, dec_hi, L, p.e.flags, p.e.power, p.e.eps,
                              p.e.k1, p.e.k0, p.e.mean, p.e.inv_std, p.e.sgn_neg, p.e.sgn_pos, stream);
    }
    return 1;
}

}  // namespace

namespace afd {

// the level-8 hand-off image [B][n8][256] of the level-14 transform of standard frames; 0 for everything else
size_t wpt3_workspace_bytes(int B, int N, int L, int level) {
    if (level != 14 || (L != 24 && L != 10 && L != 16)) return 0;
    int n = N;
    for (int k = 1; k <= 8; ++k) {
        if (L - 2 + (n & 1) >= n) return 0;
        n = child_len3(n, L);
    }
    return (size_t)B * n * 256 * sizeof(float);
}

// returns AFD_OK, an error, or 1 = "not this generation's case" (the caller falls back to the generic kernel of wpt.hip)
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
        if (!ok || !afd::wpt4_available(dec_lo, dec_hi, L)) return 1;  // no lattice: the caller's generic kernels
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
    const double scale = norm ? 1.0 / (double)std : 1.0;
    e.k1 = (float)((flags & AFD_WPT_LOG) ? 0.6931471805599453 * scale : scale);
    e.k0 = norm ? (float)(-(double)mean / (double)std) : 0.f;
    e.sgn_neg = norm ? (-1.f - sign_mean) / sign_std : -1.f;
    e.sgn_pos = norm ? (1.f - sign_mean) / sign_std : 1.f;
    for (int m = 0; m < L; ++m) {
        p.rlo[m] = dec_lo[L - 1 - m];
        p.rhi[m] = dec_hi[L - 1 - m];
    }
    switch (L) {  // every even length up to kMaxTaps (db2..db32, sym2..sym20, coif1..coif10, dmey, bior / rbio)
        case 4: return launch3<4>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 6: return launch3<6>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 8: return launch3<8>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 10: return launch3<10>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 12: return launch3<12>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 14: return launch3<14>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 16: return launch3<16>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 18: return launch3<18>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 20: return launch3<20>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 22: return launch3<22>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 24: return launch3<24>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 26: return launch3<26>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 28: return launch3<28>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 30: return launch3<30>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 32: return launch3<32>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 34: return launch3<34>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 36: return launch3<36>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 38: return launch3<38>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 40: return launch3<40>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 42: return launch3<42>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 44: return launch3<44>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 46: return launch3<46>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 48: return launch3<48>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 50: return launch3<50>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 52: return launch3<52>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 54: return launch3<54>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 56: return launch3<56>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 58: return launch3<58>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 60: return launch3<60>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 62: return launch3<62>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        case 64: return launch3<64>(p, dec_lo, dec_hi, out, ws, level, t_len, stream);
        default: return 1;
    }
}

}  // namespace afd
