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
//       as columns.  Three phases per work item (64 level-8 nodes of one frame; persistent 8-wave workgroups):
//         8 -> 9   A [2 n9 x n8], banded row tiles, 64 columns
//         9 -> 10  A [2 n10 x n9], 128 columns
//         10 -> 14 ONE composite matrix C [16 n14 x n10] = the product of the four level matrices along
//                  each of the 16 filter paths (deep nodes are 24..44 samples long: the composite has
//                  fewer entries than the four steps it replaces, 16 896 against 21 800 products per
//                  level-10 node for coif4), 256 columns.  Its fragments stay in registers for the whole
//                  kernel (a wave owns half of the row tiles and two of the eight column tiles): inside a
//                  stream of f32 matrix instructions every operand register filled from LDS or memory costs
//                  about as much as a matrix instruction (tools/micro), so operands are loaded once and
//                  reused across as many instructions as the register file allows (2 waves per SIMD).
//       Matrices are built on the host in double precision from the taps, once per wavelet and device.
#include "wpt_shared.h"

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

using namespace afd::wptc;

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

// FIN: -1 = the level-Ks image is a hand-off to the deep kernel (no epilogue), else the epilogue mode
template <int L, int FIN>
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
            // 11 loads in flight per thread: the 22 050-sample frame arrives in ONE round of memory latency
            constexpr int UN = 11;
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
        const float* taps = h ? p.rhi : p.rlo;
        for (int j = tid; 2 * j < n1; j += kTopThreads) {
            Window<L> win;
            win.load(src + 4 * j);
            put<L>(node, 2 * j, n1, win.template dot<0>(taps));
            if (2 * j + 1 < n1) put<L>(node, 2 * j + 1, n1, win.template dot<1>(taps));
        }
    }
    __syncthreads();

    // ---- levels 2 .. Ks-1: both children of every node, lanes along the output index ----
    for (int k = 2; k < p.Ks; ++k) {
        const int Mp = 1 << (k - 2);  // parents (nodes of level k-1 in this half)
        const int nk = p.n[k];
        const int mk = (nk + 1) >> 1;  // items (output pairs) per node
        const int total = Mp * mk;
        const float* src0 = lds + p.off[k - 1];
        float* dst0 = lds + p.off[k] + PAD;
        const int pin = p.pitch[k - 1], pout = p.pitch[k];
        const unsigned magic = p.magic[k];
        const int padr = PAD + (nk & 1);
        for (int idx = tid; idx < total; idx += kTopThreads) {
            const int q = (int)__umulhi((unsigned)idx, magic);
            const int i = 2 * (idx - q * mk);
            Window<L> win;
            win.load(src0 + q * pin + 2 * i);
            const float ca0 = win.template dot<0>(p.rlo), cd0 = win.template dot<0>(p.rhi);
            const float ca1 = win.template dot<1>(p.rlo), cd1 = win.template dot<1>(p.rhi);
            // odd-frequency parents list their children (d, a)
            const int par = (k == 2) ? h : (q & 1);
            float* na = dst0 + (2 * q + par) * pout;
            float* nd = dst0 + (2 * q + 1 - par) * pout;
            const bool two = i + 1 < nk;
            if (two) {
                *reinterpret_cast<f2*>(na + i) = f2{ca0, ca1};
                *reinterpret_cast<f2*>(nd + i) = f2{cd0, cd1};
            } else {
                na[i] = ca0;
                nd[i] = cd0;
            }
            // the L-2 coefficients next to a border also fill the pad slots that mirror them
            if (i <= PAD) {
                if (i >= 1) { na[-i] = ca0; nd[-i] = cd0; }
                if (two && i + 1 <= PAD) { na[-i - 1] = ca1; nd[-i - 1] = cd1; }
            }
            if (nk - 2 - i < padr + 1) {
                if ((unsigned)(nk - 2 - i) < (unsigned)padr) { na[2 * (nk - 1) - i] = ca0; nd[2 * (nk - 1) - i] = cd0; }
                if (two && (unsigned)(nk - 3 - i) < (unsigned)padr) { na[2 * (nk - 1) - i - 1] = ca1; nd[2 * (nk - 1) - i - 1] = cd1; }
            }
        }
        __syncthreads();
    }

    // ---- level Ks: lanes along the nodes, results leave the chip packet-contiguous ----
    {
        const int k = p.Ks;
        const int logM = k - 2;  // parents in this half: 2^(k-2)
        const int Mp = 1 << logM;
        const int nk = p.n[k];
        const int total = ((nk + 1) >> 1) << logM;
        const float* src0 = lds + p.off[k - 1];
        const int pin = p.pitch[k - 1];
        const size_t P = (size_t)1 << k;
        const size_t chan = (size_t)nk * P;
        const int nch = (FIN >= 0 && (p.e.flags & AFD_WPT_SIGN)) ? 2 : 1;
        float* outb = p.dst + (size_t)b * nch * chan;
        for (int idx = tid; idx < total; idx += kTopThreads) {
            const int q = idx & (Mp - 1);
            const int i = 2 * (idx >> logM);
            Window<L> win;
            win.load(src0 + q * pin + 2 * i);
            const int par = (k == 2) ? h : (q & 1);  // odd-frequency parents list their children (d, a)
            float* o = outb + (size_t)i * P + 2 * ((size_t)(h << logM) + q);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (i + u < nk) {
                    const float ca = u ? win.template dot<1>(p.rlo) : win.template dot<0>(p.rlo);
                    const float cd = u ? win.template dot<1>(p.rhi) : win.template dot<0>(p.rhi);
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
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// deep levels on the matrix cores
// ------------------------------------------------------------------------------------------------
constexpr int kDeepWaves = 8;
constexpr int kDeepThreads = kDeepWaves * 64;
constexpr int kGroup = 64;  // level-8 nodes per work item

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
    // LDS images, position-major: X8 [R8][64] and X10 [R10][256] share region A, X9 [R9][128] is region B
    static constexpr int R8 = 2 * P1::KS, R9 = 2 * P2::KS, R10 = 2 * KS3;
    static constexpr int A_FLOATS = (R8 * 64 > R10 * 256) ? R8 * 64 : R10 * 256;
    static constexpr int B_FLOATS = R9 * 128;
    // fragment tables: four k-steps of a lane are one 16-byte element and the 64 lanes of a load are
    // consecutive (1 KB per load instruction).  Inside a stream of f32 matrix instructions a vector-memory
    // instruction costs by the cache lines it touches, not by its bytes (measured, tools/micro): the same
    // 4 KB per wave cost 1 345 cycles as lane-strided 16-byte pieces and 231 as whole lines
    static constexpr int pad4(int v) { return (v + 3) / 4 * 4; }
    static constexpr int KP1 = pad4(P1::KSB), KP2 = pad4(P2::KSB), KP3 = pad4(KS3);
    static constexpr int off1 = 0;
    static constexpr int off2 = off1 + P1::T * KP1 * 64;
    static constexpr int off3 = off2 + P2::T * KP2 * 64;
    static constexpr int tab_floats = off3 + T3 * KP3 * 64;
};

struct D3Params {
    const float* ws;
    const float* tab;
    float* out;
    int groups;  // B * 8 work items (frame, 32 level-8 nodes)
    short kst1[kDeepWaves], kst2[kDeepWaves];
    Epi e;
};

// stepwise level for one wave: row tile mt of A (fragments a[]) times the 32 columns [32 nt, 32 nt + 32) of
// src (row stride 1 << LOGS); children written position-major to dst (row stride 1 << LOGD).  The B
// fragments are all requested before the first matrix instruction (one LDS round trip per tile).
template <class LV, int KP, int LOGS, int LOGD>
__device__ __forceinline__ void step_tile(const float (&a)[KP], int mt, int kst, int nt,
                                          const float* __restrict__ src, float* __restrict__ dst, int lane) {
    constexpr int S = 1 << LOGS;
    const int half = lane >> 5, col = lane & 31;
    const float* bp = src + ((size_t)kst * 2 + half) * S + nt * 32 + col;
    float bv[LV::KSB];
#pragma unroll
    for (int st = 0; st < LV::KSB; ++st) bv[st] = bp[st * 2 * S];
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < LV::KSB; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st], bv[st], acc, 0, 0, 0);
    // D: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); rows (2i, 2i+1) = (cA[i], cD[i])
    const int n = nt * 32 + col;
    const int i0 = mt * 16 + half * 2;
    const bool odd = col & 1;  // odd-frequency parents list their children (d, a)
    float* dp = dst + ((size_t)i0 << LOGD) + 2 * n;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const int di = (r >> 2) * 4 + ((r & 3) >> 1);  // i - i0: compile-time
        float e0 = acc[r], e1 = acc[r + 1];
        asm volatile("" : "+v"(e0), "+v"(e1));  // pinned: the select must not become a dynamic vector index
        f2 v;
        v.x = odd ? e1 : e0;
        v.y = odd ? e0 : e1;
        if (i0 + di < LV::n_out) *reinterpret_cast<f2*>(dp + ((size_t)di << LOGD)) = v;
    }
}

// fragments of row tile mt: KP floats per lane, KP % 4 == 0; table [mt][KP / 4][lane][4]
// 4 x 4 transpose inside every quad of lanes: on return register i of lane j (j = lane & 3) holds what
// register j of lane i held.  Two exchange stages (lane ^ 1, lane ^ 2) over DPP quad permutes -- vector
// ALU work, which runs beside the f32 matrix instructions at no measurable cost (tools/micro).
__device__ __forceinline__ void quad_transpose4(float& v0, float& v1, float& v2, float& v3, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    auto xchg = [](float keep_lo, float keep_hi, bool hi, int ctrl_is_2, float& out_lo, float& out_hi) {
        const float send = hi ? keep_lo : keep_hi;
        const int si = __builtin_bit_cast(int, send);
        const int ri = ctrl_is_2 ? __builtin_amdgcn_mov_dpp(si, 0x4E, 0xF, 0xF, true)   // quad_perm [2,3,0,1]
                                 : __builtin_amdgcn_mov_dpp(si, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
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

template <int KP>
__device__ __forceinline__ void load_frags(float (&a)[KP], const float* tab, int mt, int lane) {
    const f4* t = reinterpret_cast<const f4*>(tab) + (size_t)mt * (KP / 4) * 64 + lane;
#pragma unroll
    for (int j = 0; j < KP / 4; ++j) {
        const f4 v = t[j * 64];
        a[4 * j] = v.x; a[4 * j + 1] = v.y; a[4 * j + 2] = v.z; a[4 * j + 3] = v.w;
    }
}

// Persistent workgroups (one per CU, 2 waves per SIMD, up to 256 registers each): a workgroup walks the
// (frame, 64 level-8 nodes) items with a stride of the grid.  The composite's fragments are loaded once per
// kernel; the next item's level-8 nodes travel from the hand-off image into registers during the current item.
template <class SH, int MODE, bool SIGN>
__global__ void __launch_bounds__(kDeepThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
wpt3_deep_kernel(const D3Params p) {
    using D = Deep3<SH>;
    using P1 = typename D::P1;
    using P2 = typename D::P2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int groups = 256 / kGroup;
    constexpr int n8 = SH::n[0];
    constexpr int NLD = (n8 * kGroup + kDeepThreads - 1) / kDeepThreads;  // hand-off floats per thread
    static_assert(NLD * kDeepThreads >= D::R8 * kGroup, "the zero row of the level-8 image comes with the load");
    static_assert(D::R9 == SH::n[1] && D::R10 == SH::n[2], "even node lengths at levels 9 and 10: no pad rows");
    float* XA = lds;
    float* XB = lds + D::A_FLOATS;

    int item = blockIdx.x;
    if (item >= p.groups) return;
    float xr[NLD];
#define AFD_FETCH_X8(it)                                                                      \
    {                                                                                         \
        const int fb = (it) / groups, fg = (it) - fb * groups;                                \
        const float* wsb = p.ws + (size_t)fb * n8 * 256 + fg * kGroup;                        \
        _Pragma("unroll") for (int u = 0; u < NLD; ++u) {                                     \
            const int e = u * kDeepThreads + tid;                                             \
            xr[u] = e < n8 * kGroup ? wsb[(size_t)(e >> 6) * 256 + (e & 63)] : 0.f;           \
        }                                                                                     \
    }
    AFD_FETCH_X8(item);

    // composite phase: wave = (half of the row tiles, two of the eight 32-column tiles).  The node image is
    // the A operand and the matrix fragment the B operand, i.e. the wave computes the TRANSPOSED tile
    // D[node][row]: a lane owns one composite row (time, packet offset f) and its 16 registers are 16 nodes.
    // Four registers (nodes 8g + {0..3}) are then transposed inside the lane quads (packet offsets 4a + {0..3}),
    // after which lane j of a quad holds four consecutive packets of node 8g + j: one 16-byte store per lane,
    // the four quads of a 16-lane group complete the node's 64-byte line -- a store instruction writes 16
    // whole lines (cost inside an f32 matrix stream goes by the lines touched, tools/micro).
    constexpr int KS3 = D::KS3, T3 = D::T3, TH = (T3 + 1) / 2;
    const int rh = wave & 1, cp = wave >> 1;
    const int half = lane >> 5, col = lane & 31;
    const int rt0 = rh * TH;
    const size_t P = 16384;
    const size_t chan = (size_t)D::n14 * P;
    // this lane's composite row inside a tile: time col >> 4, packet offset f = col & 15 of an even-frequency
    // node (an odd-frequency node's 16 descendants come out in reversed order: 15 - f)
    const int lane_t = col >> 4, lane_f = col & 15;
    float fa[TH][D::KP3];
#pragma unroll
    for (int r = 0; r < TH; ++r)
        if (rt0 + r < T3) load_frags<D::KP3>(fa[r], p.tab + D::off3, rt0 + r, lane);

    for (; item < p.groups; item += (int)gridDim.x) {
        const int b = item / groups, grp = item - b * groups;
        // the table pointer is made opaque once per item: hoisted out of the item loop, the (loop-invariant)
        // fragment loads of the stepwise phases would be kept live across it
        const float* tab = p.tab;
        asm volatile("" : "+s"(tab));
        // level-8 nodes -> LDS; rows past the node length meet zero matrix columns but must be finite
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int e = u * kDeepThreads + tid;
            if (e < D::R8 * kGroup) XA[e] = xr[u];
        }
        __syncthreads();
        if (item + (int)gridDim.x < p.groups) AFD_FETCH_X8(item + (int)gridDim.x);
        // 8 -> 9: (row tile, column tile) tasks, 64 columns -> 128
        for (int task = wave; task < 2 * P1::T; task += kDeepWaves) {
            const int mt = task % P1::T, nt = task / P1::T;
            float a1[D::KP1];
            load_frags<D::KP1>(a1, tab + D::off1, mt, lane);
            step_tile<P1, D::KP1, 6, 7>(a1, mt, p.kst1[mt], nt, XA, XB, lane);
        }
        __syncthreads();
        // 9 -> 10: 128 columns -> 256
        for (int task = wave; task < 4 * P2::T; task += kDeepWaves) {
            const int mt = task % P2::T, nt = task / P2::T;
            float a2[D::KP2];
            load_frags<D::KP2>(a2, tab + D::off2, mt, lane);
            step_tile<P2, D::KP2, 7, 8>(a2, mt, p.kst2[mt], nt, XB, XA, lane);
        }
        __syncthreads();
        // 10 -> 14: both column tiles' node fragments are requested up front
        float bf[2][KS3];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < KS3; ++s) bf[c][s] = XA[(2 * s + half) * 256 + 32 * (2 * cp + c) + col];
        __syncthreads();  // XA may be overwritten by the next item's level-8 nodes from here on
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            // registers q of this lane are nodes 8 (q >> 2) + 4 half + (q & 3) of the column tile
            // after the quad transpose: this lane writes packets 4a .. 4a + 3 (a = lane_f >> 2) of node
            // 8g + (lane & 3) + 4 half; an odd-frequency node's packets come out reversed (15 - f)
            const int qj = lane & 3, qa = lane_f >> 2;
            float* outq = p.out + (size_t)b * (SIGN ? 2 : 1) * chan +
                          (size_t)16 * (grp * 256 + 32 * (2 * cp + c) + 4 * half + qj) + (size_t)lane_t * P +
                          ((qj & 1) ? 12 - 4 * qa : 4 * qa);
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                const int rt = rt0 + r;
                if (rt < T3) {
                    f16v acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                    for (int s = 0; s < KS3; ++s)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[c][s], fa[r][s], acc, 0, 0, 0);
                    const size_t trow = (size_t)(2 * rt) * P;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float v0 = acc[4 * g], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
                        quad_transpose4(v0, v1, v2, v3, lane);
                        const bool odd = qj & 1;
                        const float w0 = odd ? v3 : v0, w1 = odd ? v2 : v1, w2 = odd ? v1 : v2, w3 = odd ? v0 : v3;
                        float* o = outq + trow + 16 * 8 * g;
                        f4 w;
                        w.x = epi_value<MODE>(w0, p.e); w.y = epi_value<MODE>(w1, p.e);
                        w.z = epi_value<MODE>(w2, p.e); w.w = epi_value<MODE>(w3, p.e);
                        *reinterpret_cast<f4*>(o) = w;
                        if (SIGN) {
                            f4 sg;
                            sg.x = w0 < 0.f ? p.e.sgn_neg : p.e.sgn_pos; sg.y = w1 < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                            sg.z = w2 < 0.f ? p.e.sgn_neg : p.e.sgn_pos; sg.w = w3 < 0.f ? p.e.sgn_neg : p.e.sgn_pos;
                            *reinterpret_cast<f4*>(o + chan) = sg;
                        }
                    }
                }
            }
        }
    }
#undef AFD_FETCH_X8
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

// fragment (row tile rt, lane l, k-step s) = A[32 rt + (l & 31)][2 (kst + s) + (l >> 5)]; kp floats per lane,
// stored [rt][kp / 4][lane][4]
void put_fragments(std::vector<float>& tab, int off, const Mat& a, int tiles, int ksb, int kp, const int* kst) {
    for (int rt = 0; rt < tiles; ++rt)
        for (int l = 0; l < 64; ++l)
            for (int s = 0; s < ksb; ++s) {
                const int row = 32 * rt + (l & 31), col = 2 * ((kst ? kst[rt] : 0) + s) + (l >> 5);
                const double v = (row < a.rows && col < a.cols) ? a.at(row, col) : 0.0;
                tab[(size_t)off + (((size_t)rt * (kp / 4) + s / 4) * 64 + l) * 4 + (s & 3)] = (float)v;
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
    put_fragments(tab, D::off1, lv[0], D::P1::T, D::P1::KSB, D::KP1, kst1i);
    put_fragments(tab, D::off2, lv[1], D::P2::T, D::P2::KSB, D::KP2, kst2i);
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
    put_fragments(tab, D::off3, comp, D::T3, D::KS3, D::KP3, nullptr);
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
    long size[kKsMax + 1];
    for (int k = 0; k <= p.Ks; ++k) {
        // node = [L-2 left pad | n samples | L-2 (+1) right pad | up to 3 floats an odd node's last item reads]
        const int pitch = padded_pitch(p.n[k], L);
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

template <int L, int FIN>
int launch_top(const T3Params& p, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt3_top_kernel<L, FIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kTopLdsFloats * 4);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = true;
    }
    hipLaunchKernelGGL((wpt3_top_kernel<L, FIN>), dim3((unsigned)p.B * 2), dim3(kTopThreads),
                       (size_t)kTopLdsFloats * 4, stream, p);
    return afd::check_launch("wpt3_top_kernel");
}

template <class SH, int MODE, bool SIGN>
int launch_deep(const D3Params& q, hipStream_t stream) {
    using D = Deep3<SH>;
    constexpr size_t lds = (size_t)(D::A_FLOATS + D::B_FLOATS) * 4;
    static int grid = 0;
    if (!grid) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt3_deep_kernel<SH, MODE, SIGN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int dev = 0, cus = 0;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: %s", hipGetErrorString(e));
        grid = cus > 0 ? cus : 256;  // one persistent workgroup per CU (8 waves, 2 per SIMD)
    }
    const int g = q.groups < grid ? q.groups : grid;
    hipLaunchKernelGGL((wpt3_deep_kernel<SH, MODE, SIGN>), dim3((unsigned)g), dim3(kDeepThreads), lds, stream, q);
    return afd::check_launch("wpt3_deep_kernel");
}

template <int L>
int launch3(T3Params& p, const float* dec_lo, const float* dec_hi, float* out, void* ws, int level, int t_len,
            hipStream_t stream) {
    const int C = (p.e.flags & AFD_WPT_SIGN) ? 2 : 1;
    afd::ScopedTiming timing(AFD_K_WPT, 4.0 * p.B * ((double)p.N + (double)C * t_len * (double)(1L << level)), stream);
    const int mode = epi_mode(p.e.flags, p.e.power);
    const bool sign = p.e.flags & AFD_WPT_SIGN;
    if (level <= kKsMax) {
        p.dst = out;
        // the sign channel and the slow powers take the generic instance (flags read at run time there)
        if (mode == EPI_LOG2) return launch_top<L, EPI_LOG2>(p, stream);
        if (mode == EPI_RAW) return launch_top<L, EPI_RAW>(p, stream);
        return launch_top<L, EPI_SLOW>(p, stream);
    }
    if constexpr (HasShape3<L>::value) {
        using SH = Shape3<L>;
        p.dst = static_cast<float*>(ws);
        int rc = launch_top<L, -1>(p, stream);
        if (rc != AFD_OK) return rc;
        // levels 9..14: the lattice kernel (wpt4.hip); AFD_WPT_DEEP_MFMA=1 keeps the matrix-core composite below
        rc = afd::wpt4_deep(static_cast<const float*>(ws), out, p.B, dec_lo, dec_hi, L, p.e.flags, p.e.power, p.e.eps,
                            p.e.k1, p.e.k0, p.e.mean, p.e.inv_std, p.e.sgn_neg, p.e.sgn_pos, stream);
        if (rc != 1) return rc;
        D3Params q{};
        rc = get_tables<SH>(dec_lo, dec_hi, stream, &q.tab, q.kst1, q.kst2);
        if (rc != AFD_OK) return rc;
        q.ws = static_cast<const float*>(ws);
        q.out = out;
        q.e = p.e;
        q.groups = p.B * (256 / kGroup);
        if (mode == EPI_LOG2) return sign ? launch_deep<SH, EPI_LOG2, true>(q, stream) : launch_deep<SH, EPI_LOG2, false>(q, stream);
        if (mode == EPI_RAW) return sign ? launch_deep<SH, EPI_RAW, true>(q, stream) : launch_deep<SH, EPI_RAW, false>(q, stream);
        return sign ? launch_deep<SH, EPI_SLOW, true>(q, stream) : launch_deep<SH, EPI_SLOW, false>(q, stream);
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
    const double scale = norm ? 1.0 / (double)std : 1.0;
    e.k1 = (float)((flags & AFD_WPT_LOG) ? 0.6931471805599453 * scale : scale);
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
