// Haar wavelet packets, level 14, of the standard 1 s @ 22 050 Hz frame -- BASELINE.json
// configs[3] ("packets-haar level-14 front-end only, batch 4096, HBM-roofline throughput run").
//
// With 2 taps the transform is an add/subtract network (no products: the 1/sqrt(2) per level is
// applied once at the end as 2^-7, exactly), so the generic filter-bank kernels are pure
// overhead here.  This kernel does one frame per workgroup, one level-2 quarter at a time:
//   levels 1-2   straight from global memory: element i of the quarter is a +/- combination of
//                x[4i .. 4i+3] (one float4 load; the frame is re-read per quarter from L2);
//   levels 3-8   in LDS, node-major slots, lanes over positions, float2 reads;
//   levels 9-14  in registers: one thread per level-9 node (44 samples), compile-time
//                recursion down to the 32 leaf nodes of 2 samples;
//   store        leaves are transposed through LDS and written as coalesced float4 rows of the
//                [B][C][T=2][P=16384] output, log-power / sign / normalise applied on the way.
// Reflect rule for odd node lengths: xe[n] = x[n-2] (the last pair is (x[n-1], x[n-2])).
// Algorithmic bytes per frame: 4 * (22050 + 32768) = 219 272.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kN = 22050;
// node lengths of the Haar tree for N = 22050
constexpr int kLen[15] = {22050, 11025, 5513, 2757, 1379, 690, 345, 173, 87, 44, 22, 11, 6, 3, 2};
// LDS slot (even capacity >= length) per node at levels 2..8 inside one quarter
constexpr int kCap[9] = {0, 0, 5632, 2816, 1408, 704, 352, 176, 88};
constexpr int kBufFloats = 5632;  // one level of a quarter (64 * 88 = 5632 at level 8 too)

struct HaarParams {
    const float* x;
    float* out;
    int B;
    unsigned flags;
    float eps, mean, inv_std, scale;
};

__device__ __forceinline__ float haar_epilogue(float v, const HaarParams& p) {
    v *= p.scale;
    if (p.flags & AFD_WPT_LOG) v = __builtin_amdgcn_logf(fmaf(v, v, p.eps)) * 0.6931471805599453f;
    if (p.flags & AFD_WPT_NORM) v = (v - p.mean) * p.inv_std;
    return v;
}

// registers -> leaves: node of LEN samples with frequency index F (local to the quarter);
// children of an even-F node are (a, d), of an odd-F node (d, a)
template <int LEN>
struct Sub {
    static __device__ __forceinline__ void run(const float (&v)[LEN], int F, float* leaves) {
        constexpr int NOUT = (LEN + 1) / 2;
        float a[NOUT], d[NOUT];
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const float x0 = v[2 * i];
            const float x1 = (2 * i + 1 < LEN) ? v[2 * i + 1] : v[2 * i - 1 < 0 ? 0 : 2 * i - 1];
            a[i] = x0 + x1;
            d[i] = x0 - x1;
        }
        const int par = F & 1;
        Sub<NOUT>::run(a, 2 * F + par, leaves);
        Sub<NOUT>::run(d, 2 * F + 1 - par, leaves);
    }
};

template <>
struct Sub<2> {
    static __device__ __forceinline__ void run(const float (&v)[2], int F, float* leaves) {
        // leaves[t * 4096 + packet]
        leaves[F] = v[0];
        leaves[4096 + F] = v[1];
    }
};

__global__ void __launch_bounds__(kThreads) wpt_haar14_kernel(const HaarParams p) {
    // 54 KB: two workgroups per CU.  Level 8 ends in bufA (six swaps), so the leaf transpose
    // buffer can share storage with bufB.
    __shared__ __attribute__((aligned(16))) float bufA[kBufFloats];
    __shared__ __attribute__((aligned(16))) float bufB[2 * 4096];
    float* leaves = bufB;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float* xb = p.x + (size_t)b * kN;
    const bool sign_ch = p.flags & AFD_WPT_SIGN;
    const size_t P = 16384;
    float* outb = p.out + (size_t)b * (sign_ch ? 2 : 1) * 2 * P;

    for (int quarter = 0; quarter < 4; ++quarter) {
        // frequency index `quarter` at level 2 -> filters: Gray code, MSB = level 1, 1 = detail
        const int g = quarter ^ (quarter >> 1);
        const float s1 = (g & 2) ? -1.f : 1.f;  // level-1 filter sign
        const float s2 = (g & 1) ? -1.f : 1.f;  // level-2 filter sign
        // ---- levels 1-2 from global: element i <- x[4i..4i+3]; last element by reflect ----
        // 8 float4-equivalents in flight per thread before the first LDS store (a load -> store
        // chain would pay one L2 latency per element)
        for (int base = 0; base < kLen[2]; base += kThreads * 8) {
            float2 lo[8], hi[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = base + r * kThreads + tid;
                if (i < kLen[2] - 1) {
                    // frames start 8-byte aligned (22050 * 4 bytes per frame): two float2 loads
                    lo[r] = *reinterpret_cast<const float2*>(xb + 4 * i);
                    hi[r] = *reinterpret_cast<const float2*>(xb + 4 * i + 2);
                } else {
                    // level-1 node has 11025 samples (odd): pair (l1[11024], l1[11023])
                    lo[r] = *reinterpret_cast<const float2*>(xb + 22048);
                    hi[r] = *reinterpret_cast<const float2*>(xb + 22046);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = base + r * kThreads + tid;
                if (i < kLen[2]) bufA[i] = (lo[r].x + s1 * lo[r].y) + s2 * (hi[r].x + s1 * hi[r].y);
            }
        }
        __syncthreads();
        // ---- levels 3-8 in LDS: M parents of n_in samples in slots of capIn ----
        float* src = bufA;
        float* dst = bufB;
        int Fb = quarter;
#pragma unroll 1
        for (int lev = 3; lev <= 8; ++lev) {
            const int M = 1 << (lev - 3);
            const int n_in = kLen[lev - 1], n_out = kLen[lev];
            const int capIn = kCap[lev - 1], capOut = kCap[lev];
            const int total = M * n_out;
            const float inv_nout = 1.0f / (float)n_out;
            for (int w = tid; w < total; w += kThreads) {
                int q = (int)((float)w * inv_nout);  // w / n_out through the reciprocal + fix-up
                int i = w - q * n_out;
                if (i < 0) { i += n_out; --q; } else if (i >= n_out) { i -= n_out; ++q; }
                const float* nd = src + q * capIn;
                float x0, x1;
                if (2 * i + 1 < n_in) {
                    const float2 v = *reinterpret_cast<const float2*>(nd + 2 * i);
                    x0 = v.x;
                    x1 = v.y;
                } else {
                    x0 = nd[2 * i];
                    x1 = nd[2 * i - 1];
                }
                const int par = (Fb + q) & 1;
                dst[(2 * q + par) * capOut + i] = x0 + x1;
                dst[(2 * q + 1 - par) * capOut + i] = x0 - x1;
            }
            __syncthreads();
            float* t = src;
            src = dst;
            dst = t;
            Fb *= 2;
        }
        // src: 64 level-8 nodes (87 samples, slot 88), frequency order; Fb = quarter * 64
        // ---- levels 9-14 in registers: thread = level-10 node (256 per quarter).  The two
        // threads of a level-9 node both form its 44 samples from the level-8 parent (cheaper
        // than leaving half the workgroup idle), then take one level-10 child each.
        {
            const int q8 = tid >> 2;
            const int c9 = (tid >> 1) & 1;
            const int c10 = tid & 1;
            const int F8 = Fb + q8;
            const float sg9 = ((c9 ^ (F8 & 1)) != 0) ? -1.f : 1.f;
            const float* nd = src + q8 * kCap[8];
            float v9[44];
#pragma unroll
            for (int i = 0; i < 43; ++i) {
                const float2 xv = *reinterpret_cast<const float2*>(nd + 2 * i);
                v9[i] = xv.x + sg9 * xv.y;
            }
            v9[43] = nd[86] + sg9 * nd[85];  // 87 samples: the last pair reflects
            const int F9 = 2 * q8 + c9;  // local to the quarter; parity = global parity
            const float sg10 = ((c10 ^ (F9 & 1)) != 0) ? -1.f : 1.f;
            float v10[22];
#pragma unroll
            for (int i = 0; i < 22; ++i) v10[i] = v9[2 * i] + sg10 * v9[2 * i + 1];
            Sub<22>::run(v10, 2 * F9 + c10, leaves);
        }
        __syncthreads();
        // ---- coalesced store of the quarter: rows t = 0, 1, packets [quarter*4096, +4096) ----
        for (int e = tid; e < 2 * 1024; e += kThreads) {
            const int t = e >> 10;
            const int c4 = e & 1023;
            const float4 v = *reinterpret_cast<const float4*>(leaves + t * 4096 + 4 * c4);
            float4 r;
            r.x = haar_epilogue(v.x, p);
            r.y = haar_epilogue(v.y, p);
            r.z = haar_epilogue(v.z, p);
            r.w = haar_epilogue(v.w, p);
            float* o = outb + (size_t)t * P + quarter * 4096 + 4 * c4;
            *reinterpret_cast<float4*>(o) = r;
            if (sign_ch) {
                float4 sgn;
                sgn.x = v.x < 0.f ? -1.f : 1.f;
                sgn.y = v.y < 0.f ? -1.f : 1.f;
                sgn.z = v.z < 0.f ? -1.f : 1.f;
                sgn.w = v.w < 0.f ? -1.f : 1.f;
                if (p.flags & AFD_WPT_NORM) {
                    sgn.x = (sgn.x - p.mean) * p.inv_std;
                    sgn.y = (sgn.y - p.mean) * p.inv_std;
                    sgn.z = (sgn.z - p.mean) * p.inv_std;
                    sgn.w = (sgn.w - p.mean) * p.inv_std;
                }
                *reinterpret_cast<float4*>(o + 2 * P) = sgn;
            }
        }
        __syncthreads();
    }
}

}  // namespace

namespace afd {

// 0 = launched, 1 = not this kernel's case
int wpt_haar14_forward(const float* x, int B, int N, const float* dec_lo, int L, int level,
                       unsigned flags, float power, float eps, float mean, float std, float* out,
                       hipStream_t stream) {
    if (L != 2 || level != 14 || N != kN || power != 2.0f) return 1;
    const float s = 0.70710678118654752f;
    if (fabsf(dec_lo[0] - s) > 1e-6f || fabsf(dec_lo[1] - s) > 1e-6f) return 1;
    HaarParams p{};
    p.x = x;
    p.out = out;
    p.B = B;
    p.flags = flags;
    p.eps = eps;
    p.mean = mean;
    p.inv_std = (float)(1.0 / (double)(std == 0.f ? 1.f : std));
    p.scale = 1.0f / 128.0f;  // (1/sqrt 2)^14
    afd::ScopedTiming timing(AFD_K_WPT, 4.0 * B * ((double)N + ((flags & AFD_WPT_SIGN) ? 2.0 : 1.0) * 32768.0), stream);
    hipLaunchKernelGGL(wpt_haar14_kernel, dim3(B), dim3(kThreads), 0, stream, p);
    return afd::check_launch("wpt_haar14_kernel");
}

}  // namespace afd
