// Haar wavelet packets, level 14, of the standard 1 s @ 22 050 Hz frame -- BASELINE.json
// configs[3] ("packets-haar level-14 front-end only, batch 4096, HBM-roofline throughput run").
//
// With 2 taps the transform is an add/subtract network (no products: the 1/sqrt(2) per level is
// applied once at the end as 2^-7, exactly), so the generic filter-bank kernels are pure
// overhead here.  A 512-thread workgroup owns one level-1 half of a frame (its low-pass or its
// high-pass node, 8192 of the 16384 packets); two workgroups run per CU, and the two halves of
// a frame are given to workgroups on the same XCD so the frame leaves HBM once.  The half
// lives in one 44 KB LDS image that every level overwrites in place:
//   levels 1-2   straight from global memory: element i of the two level-2 nodes is a +/-
//                combination of x[4i .. 4i+3];
//   levels 3-8   two radix-8 passes in LDS (three levels per pass): each work item reads eight
//                consecutive samples of a node into registers, barrier, writes element i of the
//                eight great-grandchildren into the parent's own slot (split in eight).  The slots
//                past the end of an odd-length node hold the samples the reflect rule mirrors there
//                (compile-time index maps, written by one thread per node after each pass), so
//                every element is the plain butterfly: the kernel is bound by its vector
//                instructions (75 % issue occupancy with radix-4 passes and per-element tail
//                selects), three levels per pass cut the index arithmetic per output by 40 %;
//   levels 9-14  in registers: thread = level-10 node (512 per half, 22 samples each),
//                compile-time recursion down to its 16 leaf nodes of 2 samples;
//   store        a thread owns 16 consecutive packets of both time rows (one 64-byte segment
//                per row): 4 x 4 float4 transpose inside lane quads, log-power / sign /
//                normalise in registers, float4 stores into [B][C][T=2][P=16384].
// The next frame's samples are loaded, a third at a time, while the LDS passes run.
// Nodes sit in frequency (Gray) order: child 2F of parent F is the low-pass child when F is
// even and the high-pass child when F is odd.
// Reflect rule for odd node lengths: xe[n] = x[n-2] (the last pair is (x[n-1], x[n-2])).
// Algorithmic bytes per frame: 4 * (22050 + 32768) = 219 272.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

constexpr int kThreads = 512;
constexpr int kN = 22050;
// node lengths of the Haar tree for N = 22050
constexpr int kLen[15] = {22050, 11025, 5513, 2757, 1379, 690, 345, 173, 87, 44, 22, 11, 6, 3, 2};
// LDS slot per node at levels 2, 5, 8 (one eighth every three levels).  A workgroup holds one
// level-1 half of a frame: 2 * 5632 floats = 44 KB
constexpr int kCap2 = 5632;
constexpr int kLdsFloats = 2 * kCap2;
constexpr int kLoadItems = 11;  // ... and in the frame load (both halves read the whole frame)

struct HaarParams {
    const float* x;
    float* out;
    int B;
    unsigned flags;
    float eps, mean, inv_std, scale, sgn_neg, sgn_pos;
    float scale2, k1, k0;  // scale^2; ln 2 / std (or scale / std without the log); -mean / std
};

// (1/sqrt 2)^14 scale, log(v^2 + eps) and (x - mean) / std in three instructions per coefficient.  With
// s^2 = 2^-14 exactly, log((v s)^2 + eps) = ln 2 (log2(v v + eps 2^14) - 14): one FMA, v_log_f32, one FMA with
// k0 = -mean / std - 14 k1 (the kernel is bound by its vector instructions, not by HBM; the epilogue mode is a
// template parameter so that no per-element select is left)
template <bool LOG>
__device__ __forceinline__ float haar_epilogue(float v, const HaarParams& p) {
    if (LOG) return fmaf(__builtin_amdgcn_logf(fmaf(v, v, p.eps)), p.k1, p.k0);
    return fmaf(v, p.k1, p.k0);
}

// registers -> leaves.  `s` is +1 when the node's frequency index is even (first child =
// sum), -1 when odd (first child = difference); below the root it is a literal, so the
// products fold into adds / subtracts.  POS = leaf offset of the subtree inside the thread.
template <int LEN, int POS>
struct Sub {
    static __device__ __forceinline__ void run(const float (&v)[LEN], float s, float (&out)[2][16]) {
        constexpr int NOUT = (LEN + 1) / 2;
        float first[NOUT], second[NOUT];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const float x0 = v[2 * i];
            const float x1 = (2 * i + 1 < LEN) ? v[2 * i + 1] : v[2 * i - 1 < 0 ? 0 : 2 * i - 1];
            first[i] = fmaf(s, x1, x0);
            second[i] = fmaf(-s, x1, x0);
        }
        Sub<NOUT, 2 * POS>::run(first, 1.f, out);
        Sub<NOUT, 2 * POS + 1>::run(second, -1.f, out);
    }
};

template <int POS>
struct Sub<2, POS> {
    static __device__ __forceinline__ void run(const float (&v)[2], float, float (&out)[2][16]) {
        out[0][POS] = v[0];
        out[1][POS] = v[1];
    }
};

// v = four float4 per lane.  4 x 4 transpose of float4s among the lanes {l, l + 16, l + 32, l + 48} of a wave
// (l = lane & 15, k = lane >> 4): on return float4 j of lane (l, k) is what float4 k of lane (l, j) was.
// gfx950's v_permlane16_swap_b32 exchanges the odd 16-lane rows of its first operand with the even rows of its
// second -- a 2 x 2 block exchange between lanes 16 apart in ONE instruction, no select and no DPP wait states
// (the quad-permute form of round 2 took a v_mov_dpp, three v_cndmask and s_nops per exchange);
// v_permlane32_swap_b32 does the same between the wave's halves.
// Inline assembly, not __builtin_amdgcn_permlane16_swap: hipcc 7.2 loses the second result of the builtin when
// swaps are chained (it stored one register group four times -- tools/micro/pl_chain.hip reproduces it).
template <int DIST>
__device__ __forceinline__ void lane_swap(float& a, float& b) {
    if (DIST == 16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    else asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

__device__ __forceinline__ void row_transpose(float (&v)[16]) {
    // a vector-ALU result read by v_permlane*_swap needs two wait states (the compiler cannot see into the asm)
    asm volatile("s_nop 1");
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        lane_swap<16>(v[0 + c], v[4 + c]);
        lane_swap<16>(v[8 + c], v[12 + c]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        lane_swap<32>(v[0 + c], v[8 + c]);
        lane_swap<32>(v[4 + c], v[12 + c]);
    }
}

// The four samples feeding element i of a node's grandchildren are x[4i .. 4i+3]; for the last element of an
// odd-length node the reflect rule (twice: at the node and at its children) gives, for n = 4k + 1,
// (x[n-1], x[n-2], x[n-3], x[n-2]) and, for n = 4k + 3, (x[n-3], x[n-2], x[n-1], x[n-2]): the slots past the
// node end hold exactly those mirrored samples (write_tail_pads), so every element is a plain butterfly.
// element i of the grandchildren 4F .. 4F+3 of a node with frequency parity `odd`
__device__ __forceinline__ float4 butterfly4(const float4 x, bool odd) {
    const float s = odd ? -1.f : 1.f;
    const float a0 = fmaf(s, x.y, x.x), a1 = fmaf(s, x.w, x.z);    // first child (even index)
    const float d0 = fmaf(-s, x.y, x.x), d1 = fmaf(-s, x.w, x.z);  // second child (odd index)
    return make_float4(a0 + a1, a0 - a1, d0 - d1, d0 + d1);
}

// The reflect rule of an odd-length node, xe[n] = x[n-2], as an index map
constexpr int refl1(int j, int n) { return (j >= n) ? n - 2 : j; }

// Base position that slot p >= n of a node of length n must mirror so that a plain radix-8 element over
// x[8i .. 8i+7] equals three reflect-extended Haar levels: follow the element's inputs down the three levels,
// applying the rule at each (node lengths n -> (n+1)/2 -> ...).
constexpr int pad_src8(int p, int n) {
    const int n1 = (n + 1) / 2, n2 = (n1 + 1) / 2;
    const int i = p >> 3, s = p & 7;
    const int j2 = refl1(2 * i + ((s >> 2) & 1), n2);   // level + 2 element
    const int j1 = refl1(2 * j2 + ((s >> 1) & 1), n1);  // level + 1 element
    return refl1(2 * j1 + (s & 1), n);
}

// reflect pads for a radix-8 reader: slots n .. 8 ceil(n/8) - 1 of every node (one thread per node)
template <int n, int cap, int nodes>
__device__ __forceinline__ void write_pads8(float* buf, int tid) {
    constexpr int n3 = (((n + 1) / 2 + 1) / 2 + 1) / 2;  // elements three levels down
    constexpr int end = 8 * n3;
    static_assert(nodes <= kThreads && end <= cap && end - n <= 7, "pad slots");
    if (tid < nodes) {
        float* x = buf + tid * cap;
        float v[8];
#pragma unroll
        for (int k = 0; k < end - n; ++k) v[k] = x[pad_src8(n + k, n)];
#pragma unroll
        for (int k = 0; k < end - n; ++k) x[n + k] = v[k];
    }
}

// reflect pad for the radix-4 reader of the level-8 nodes (n = 87 = 4k + 3): x[n] = x[n-2]
template <int n, int cap, int nodes>
__device__ __forceinline__ void write_pad4(float* buf, int tid) {
    static_assert(n % 4 == 3 && n + 1 <= cap && nodes <= kThreads, "pad slot");
    if (tid < nodes) {
        float* x = buf + tid * cap;
        x[n] = x[n - 2];
    }
}

// element i of the great-grandchildren 8F .. 8F+7 of a node with frequency parity `odd`: two radix-4 halves
// (x[0..3] -> element 2i of the grandchildren, x[4..7] -> element 2i + 1), then one more level; grandchildren
// 4F, 4F+2 have even frequency (first child = sum), 4F+1, 4F+3 odd (first child = difference)
__device__ __forceinline__ void butterfly8(const float4 lo, const float4 hi, bool odd, float (&out)[8]) {
    const float4 p = butterfly4(lo, odd), q = butterfly4(hi, odd);
    out[0] = p.x + q.x; out[1] = p.x - q.x;
    out[2] = p.y - q.y; out[3] = p.y + q.y;
    out[4] = p.z + q.z; out[5] = p.z - q.z;
    out[6] = p.w - q.w; out[7] = p.w + q.w;
}

// one radix-8 pass: levels LEV -> LEV + 3 for the 2^(LEV-1) nodes of the half, in place (element i of the
// eight great-grandchildren goes to the parent's own slot, split in eight); then the pads its reader needs
constexpr int kItems8 = 3;
template <int LEV, int CAPIN>
__device__ __forceinline__ void lds_pass8(float* buf, int tid) {
    constexpr int M = 1 << (LEV - 1);
    constexpr int n_out = kLen[LEV + 3];
    constexpr int capOut = CAPIN / 8;
    constexpr int total = M * n_out;
    static_assert(total <= kItems8 * kThreads, "pass does not fit the register staging");
    static_assert(n_out <= capOut && 8 * n_out <= CAPIN, "slot capacity");
    float4 va[kItems8], vb[kItems8];
#pragma unroll
    for (int r = 0; r < kItems8; ++r) {
        int w = r * kThreads + tid;
        w = w < total ? w : total - 1;
        const int q = w / n_out, i = w - q * n_out;
        const float4* src = reinterpret_cast<const float4*>(buf + q * CAPIN + 8 * i);
        va[r] = src[0];
        vb[r] = src[1];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kItems8; ++r) {
        const int w = r * kThreads + tid;
        if (w < total) {
            const int q = w / n_out, i = w - q * n_out;
            float g[8];
            butterfly8(va[r], vb[r], q & 1, g);
            float* o = buf + q * CAPIN + i;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k * capOut] = g[k];
        }
    }
    __syncthreads();
    if constexpr (LEV + 3 == 8) {
        write_pad4<n_out, capOut, 8 * M>(buf, tid);
    } else {
        write_pads8<n_out, capOut, 8 * M>(buf, tid);
    }
    __syncthreads();
}

// Items [R0, R1) of the frame load: x[4i .. 4i+3] for the level-2 element i; the last element
// (i = 5512) pairs the level-1 samples (11024, 11023), i.e. x[22048], x[22049], x[22046],
// x[22047].  No branches: every load is issued before the first use.
template <int R0, int R1>
__device__ __forceinline__ void load_items(const float* xb, int tid, float2 (&lo)[kLoadItems],
                                           float2 (&hi)[kLoadItems]) {
    static_assert((kLoadItems - 1) * kThreads < kLen[2] && kLoadItems * kThreads >= kLen[2], "only the last round is ragged");
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        int i = r * kThreads + tid;
        i = i < kLen[2] ? i : 0;
        const int at = (i == kLen[2] - 1) ? kN - 4 : 4 * i;
        // frames start 8-byte aligned (22050 * 4 bytes per frame): float2 loads
        lo[r] = *reinterpret_cast<const float2*>(xb + at);
        hi[r] = *reinterpret_cast<const float2*>(xb + at + 2);
    }
}

// ... and their element i of the level-2 nodes 2h, 2h + 1 (h = level-1 half: 0 low-pass, 1 high-pass;
// s = +1 / -1 is uniform over the workgroup).  Only this half's level-1 child is formed: a = x0 + s x1,
// a' = x2 + s x3, then (a + s a', a - s a') -- the children of an odd-frequency node come (difference, sum).
template <int R0, int R1, int H>
__device__ __forceinline__ void level2_items_h(const float2 (&lo)[kLoadItems], const float2 (&hi)[kLoadItems],
                                               int tid, float2 (&g)[kLoadItems]) {
    constexpr float s = H ? -1.f : 1.f;
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        float2 p0 = lo[r], p1 = hi[r];
        if (r == kLoadItems - 1) {
            const bool tail = r * kThreads + tid == kLen[2] - 1;
            p0 = tail ? hi[r] : lo[r];
            p1 = tail ? lo[r] : hi[r];
        }
        const float a0 = fmaf(s, p0.y, p0.x), a1 = fmaf(s, p1.y, p1.x);
        g[r] = make_float2(fmaf(s, a1, a0), fmaf(-s, a1, a0));
    }
}

template <int R0, int R1>
__device__ __forceinline__ void level2_items(const float2 (&lo)[kLoadItems], const float2 (&hi)[kLoadItems],
                                             int tid, int h, float2 (&g)[kLoadItems]) {
    // uniform branch: with a literal sign the products fold into adds / subtracts
    if (h) level2_items_h<R0, R1, 1>(lo, hi, tid, g);
    else level2_items_h<R0, R1, 0>(lo, hi, tid, g);
}

__device__ __forceinline__ void write_level2(float* buf, int tid, const float2 (&g)[kLoadItems]) {
#pragma unroll
    for (int r = 0; r < kLoadItems; ++r) {
        const int i = r * kThreads + tid;
        if (r < kLoadItems - 1 || i < kLen[2]) {
            buf[i] = g[r].x;
            buf[kCap2 + i] = g[r].y;
        }
    }
}

template <bool LOG, bool SIGN>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) wpt_haar14_kernel(const HaarParams p) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    const int tid = threadIdx.x;
    constexpr bool sign_ch = SIGN;
    const size_t P = 16384;
    const size_t frame_out = (size_t)(sign_ch ? 2 : 1) * 2 * P;

    // Workgroups w and w + 8 sit on the same XCD (round-robin dispatch): they take the two
    // halves of the same frames, so the second read of a frame is an L2 hit.
    const int w = blockIdx.x;
    const int half = (w >> 3) & 1;
    const int fstride = gridDim.x >> 1;
    int b = ((w >> 4) << 3) | (w & 7);
    if (b >= p.B) return;
    {
        float2 lo[kLoadItems], hi[kLoadItems], g[kLoadItems];
        load_items<0, kLoadItems>(p.x + (size_t)b * kN, tid, lo, hi);
        level2_items<0, kLoadItems>(lo, hi, tid, half, g);
        write_level2(buf, tid, g);
    }
    __syncthreads();
    for (; b < p.B; b += fstride) {
        // opaque copy of the thread index: keeps the per-item index arithmetic of the passes
        // inside the loop (hoisted out, it would pin ~100 registers across the whole frame)
        int lt = tid;
        asm volatile("" : "+v"(lt));
        // next frame's samples: a third of them in flight during each LDS pass (all at once
        // would hold 44 registers across a pass)
        const int nb = b + fstride;
        const float* xn = p.x + (size_t)(nb < p.B ? nb : b) * kN;
        float2 lo[kLoadItems], hi[kLoadItems], nxt[kLoadItems];
        load_items<0, 6>(xn, lt, lo, hi);
        write_pads8<kLen[2], kCap2, 2>(buf, lt);  // level-2 image of this frame (written before the last barrier)
        __syncthreads();
        lds_pass8<2, kCap2>(buf, lt);  // levels 2 -> 5
        level2_items<0, 6>(lo, hi, lt, half, nxt);
        load_items<6, kLoadItems>(xn, lt, lo, hi);
        lds_pass8<5, kCap2 / 8>(buf, lt);  // levels 5 -> 8
        level2_items<6, kLoadItems>(lo, hi, lt, half, nxt);
        // ---- levels 9-10: thread = level-10 node `tid`; its level-8 grandparent is shared by
        // four threads, each forms the level-9 node it needs (44 samples) ----
        float v10[22];
        {
            const int q8 = lt >> 2;
            const int F9 = lt >> 1;
            const float sg9 = ((F9 ^ q8) & 1) ? -1.f : 1.f;
            const float sg10 = ((lt ^ F9) & 1) ? -1.f : 1.f;
            const float* nd = buf + q8 * 88;
            // two halves, so that at most 11 float4 reads are in flight beside `nxt`
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const int i = 11 * h + k;
                    // 87 samples: level-9 element 43 is the reflected pair (86, 85) -- slot 87 holds x[85]
                    const float4 xv = *reinterpret_cast<const float4*>(nd + 4 * i);
                    v10[i] = fmaf(sg10, fmaf(sg9, xv.w, xv.z), fmaf(sg9, xv.y, xv.x));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        // the LDS image is free again: level 2 of the next frame goes in now
        if (nb < p.B) write_level2(buf, lt, nxt);
        // ---- levels 11-14 in registers, then the thread's 2 x 16 outputs ----
        float out[2][16];
        Sub<22, 0>::run(v10, (lt & 1) ? -1.f : 1.f, out);
        // A thread's 16 packets of one time row are one 64-byte segment.  Transpose 4 x 4 float4 among the
        // lanes {l, l + 16, l + 32, l + 48}: lane (l, k) then holds quarter k of the segments of threads
        // 16 j + l (j = register group), so a store instruction writes 16 consecutive whole segments (1 KB).
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            row_transpose(out[t]);
        }
        float* ob = p.out + (size_t)b * frame_out + 8192 * half + 16 * ((lt & ~63) + (lt & 15)) + 4 * ((lt >> 4) & 3);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 r;
                r.x = haar_epilogue<LOG>(out[t][4 * k], p);
                r.y = haar_epilogue<LOG>(out[t][4 * k + 1], p);
                r.z = haar_epilogue<LOG>(out[t][4 * k + 2], p);
                r.w = haar_epilogue<LOG>(out[t][4 * k + 3], p);
                typedef float f4v __attribute__((ext_vector_type(4)));
                const f4v rv = {r.x, r.y, r.z, r.w};
                __builtin_nontemporal_store(rv, reinterpret_cast<f4v*>(ob + (size_t)t * P + 256 * k));
            }
        }
        if (sign_ch) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float4 sgn;
                    sgn.x = out[t][4 * k] < 0.f ? p.sgn_neg : p.sgn_pos;
                    sgn.y = out[t][4 * k + 1] < 0.f ? p.sgn_neg : p.sgn_pos;
                    sgn.z = out[t][4 * k + 2] < 0.f ? p.sgn_neg : p.sgn_pos;
                    sgn.w = out[t][4 * k + 3] < 0.f ? p.sgn_neg : p.sgn_pos;
                    *reinterpret_cast<float4*>(ob + (size_t)(2 + t) * P + 256 * k) = sgn;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

namespace afd {

// 0 = launched, 1 = not this kernel's case
int wpt_haar14_forward(const float* x, int B, int N, const float* dec_lo, int L, int level,
                       unsigned flags, float power, float eps, float mean, float std, float sign_mean,
                       float sign_std, float* out, hipStream_t stream) {
    if (L != 2 || level != 14 || N != kN || power != 2.0f) return 1;
    const float s = 0.70710678118654752f;
    if (fabsf(dec_lo[0] - s) > 1e-6f || fabsf(dec_lo[1] - s) > 1e-6f) return 1;
    static int n_cu = 0;
    if (!n_cu) {
        hipError_t e = hipSuccess;
        const void* kerns[4] = {reinterpret_cast<const void*>(&wpt_haar14_kernel<false, false>),
                                reinterpret_cast<const void*>(&wpt_haar14_kernel<false, true>),
                                reinterpret_cast<const void*>(&wpt_haar14_kernel<true, false>),
                                reinterpret_cast<const void*>(&wpt_haar14_kernel<true, true>)};
        for (const void* kf : kerns)
            if (e == hipSuccess) e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsFloats * 4);
        int dev = 0;
        if (e == hipSuccess) e = hipGetDevice(&dev);
        int cus = 0;
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt haar: %s", hipGetErrorString(e));
        n_cu = cus > 0 ? cus : 256;
    }
    HaarParams p{};
    p.x = x;
    p.out = out;
    p.B = B;
    p.flags = flags;
    p.eps = eps;
    p.mean = mean;
    p.inv_std = (float)(1.0 / (double)(std == 0.f ? 1.f : std));
    p.sgn_neg = (flags & AFD_WPT_NORM) ? (-1.f - sign_mean) / sign_std : -1.f;
    p.sgn_pos = (flags & AFD_WPT_NORM) ? (1.f - sign_mean) / sign_std : 1.f;
    p.scale = 1.0f / 128.0f;  // (1/sqrt 2)^14
    {
        const bool norm = flags & AFD_WPT_NORM;
        const double inv = norm ? 1.0 / (double)(std == 0.f ? 1.f : std) : 1.0;
        const bool lg = flags & AFD_WPT_LOG;
        p.scale2 = p.scale * p.scale;  // exact: a power of two
        p.eps = lg ? eps * 16384.0f : eps;  // eps / s^2, exact
        const double k1 = (lg ? 0.6931471805599453 : (double)p.scale) * inv;
        p.k1 = (float)k1;
        p.k0 = (float)((norm ? -(double)mean * inv : 0.0) - (lg ? 14.0 * k1 : 0.0));
    }
    afd::ScopedTiming timing(AFD_K_WPT, 4.0 * B * ((double)N + ((flags & AFD_WPT_SIGN) ? 2.0 : 1.0) * 32768.0), stream);
    // two workgroups (frame halves) per CU, 16 per XCD pair up on a frame; the grid is a
    // multiple of 16 so that every workgroup has its partner
    int pairs = B < n_cu ? B : n_cu;
    pairs = (pairs + 7) / 8 * 8;
    const int grid = 2 * pairs;
    const bool lg = flags & AFD_WPT_LOG, sg = flags & AFD_WPT_SIGN;
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), (size_t)kLdsFloats * 4, stream, p);
    };
    if (lg && sg) launch(wpt_haar14_kernel<true, true>);
    else if (lg) launch(wpt_haar14_kernel<true, false>);
    else if (sg) launch(wpt_haar14_kernel<false, true>);
    else launch(wpt_haar14_kernel<false, false>);
    return afd::check_launch("wpt_haar14_kernel");
}

}  // namespace afd
