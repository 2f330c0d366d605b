// Per-packet (per tree node) statistics and the reference's "block norm" epilogue.
//
// Reference: src/audiofakedetect/wavelet_math.py:194-203 -- inside the loop over the level's nodes
// every node (a [B, T] slab of coefficients) updates one WelfordEstimator and, with
// block_norm, is divided by max|node| over the whole batch before the stack / log / sign steps
// (:206-218).  The maximum is a batch-global reduction, so the work is two passes over the raw
// coefficients [B][T][P] (P fastest, the layout afd_wpt_forward writes with flags == 0):
//   afd_packet_stats      one read : per packet sum v, sum v^2 (double) and max |v|
//   afd_packet_block_norm one read + one write: v / max|v|, log / sign / normalise epilogue
// Both are HBM-bound; 16-byte accesses per lane, a lane owns 4 neighbouring packets.
#include "../../include/afd_hip.h"
#include "afd_common.h"

namespace {

constexpr int kT = 256;

template <int VEC> struct Vec;
template <> struct Vec<1> { using type = float; };
template <> struct Vec<4> { using type = float4; };

template <int VEC> __device__ __forceinline__ void unpack(const typename Vec<VEC>::type& v, float (&f)[VEC]);
template <> __device__ __forceinline__ void unpack<1>(const float& v, float (&f)[1]) { f[0] = v; }
template <> __device__ __forceinline__ void unpack<4>(const float4& v, float (&f)[4]) {
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}

// block = cols x lanes threads: `cols` column groups of VEC packets, `lanes` rows in flight
template <int VEC>
__global__ __launch_bounds__(kT) void packet_stats_kernel(const float* __restrict__ x, long rows, int P,
                                                         int cols, double* __restrict__ sums,
                                                         unsigned* __restrict__ absmax) {
    using V = typename Vec<VEC>::type;
    const int lanes = kT / cols;
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    const int cg = blockIdx.x * cols + tx;  // column group
    const int groups = P / VEC;
    double s[VEC], q[VEC];
    float m[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = 0.0, q[k] = 0.0, m[k] = 0.f;
    if (cg < groups && ty < lanes) {
        const long step = (long)lanes * gridDim.y;
        const V* base = reinterpret_cast<const V*>(x) + cg;
        long r = (long)blockIdx.y * lanes + ty;
        // 4 rows in flight per lane
        for (; r + 3 * step < rows; r += 4 * step) {
            V v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = base[(r + u * step) * groups];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[VEC];
                unpack<VEC>(v[u], f);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    s[k] += (double)f[k];
                    q[k] += (double)f[k] * (double)f[k];
                    m[k] = fmaxf(m[k], fabsf(f[k]));
                }
            }
        }
        for (; r < rows; r += step) {
            float f[VEC];
            unpack<VEC>(base[r * groups], f);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                s[k] += (double)f[k];
                q[k] += (double)f[k] * (double)f[k];
                m[k] = fmaxf(m[k], fabsf(f[k]));
            }
        }
    }
    // fold the row lanes of the block, then one atomic per packet and block
    __shared__ double red_s[kT * VEC], red_q[kT * VEC];
    __shared__ float red_m[kT * VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        red_s[threadIdx.x * VEC + k] = s[k];
        red_q[threadIdx.x * VEC + k] = q[k];
        red_m[threadIdx.x * VEC + k] = m[k];
    }
    __syncthreads();
    if (ty == 0 && cg < groups) {
        for (int l = 1; l < lanes; ++l) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                s[k] += red_s[(l * cols + tx) * VEC + k];
                q[k] += red_q[(l * cols + tx) * VEC + k];
                m[k] = fmaxf(m[k], red_m[(l * cols + tx) * VEC + k]);
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int p = cg * VEC + k;
            atomicAdd(sums + p, s[k]);
            atomicAdd(sums + P + p, q[k]);
            // |v| >= 0: the IEEE bit patterns order like unsigned integers; a NaN input (fmaxf drops
            // it) cannot reach here
            atomicMax(absmax + p, __float_as_uint(m[k]));
        }
    }
}

__device__ __noinline__ float pow_log_precise(float v, float power, float eps) {
    return logf(powf(fabsf(v), power) + eps);
}

struct BnParams {
    unsigned flags;
    float power, eps, mean, std, sgn_neg, sgn_pos;
    long TP;       // T * P
    long chan;     // output channel stride = T * P
    int P, C;
};

__device__ __forceinline__ float bn_value(float v, const BnParams& p) {
    if (p.flags & AFD_WPT_LOG) {
        // same arithmetic as the fused transform epilogues (wpt3.hip / wpt4.hip / wpt_haar.hip)
        if (p.power == 2.0f) v = __builtin_amdgcn_logf(fmaf(v, v, p.eps)) * 0.6931471805599453f;
        else v = pow_log_precise(v, p.power, p.eps);
    }
    if (p.flags & AFD_WPT_NORM) v = (v - p.mean) / p.std;
    return v;
}

template <int VEC>
__global__ __launch_bounds__(kT) void packet_block_norm_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ absmax,
                                                              float* __restrict__ out, long n_groups,
                                                              BnParams p) {
    using V = typename Vec<VEC>::type;
    const long stride = (long)gridDim.x * kT;
    const int groups = p.P / VEC;
    for (long g = (long)blockIdx.x * kT + threadIdx.x; g < n_groups; g += stride) {
        const long e = g * VEC;           // element index in [B][T][P]
        const int pk = (int)(g % groups) * VEC;
        const long b = e / p.TP, within = e - b * p.TP;
        float f[VEC], d[VEC], r[VEC], sg[VEC];
        unpack<VEC>(reinterpret_cast<const V*>(x)[g], f);
        if (absmax) unpack<VEC>(reinterpret_cast<const V*>(absmax + pk)[0], d);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float v = absmax ? f[k] / d[k] : f[k];
            r[k] = bn_value(v, p);
            sg[k] = v < 0.f ? p.sgn_neg : p.sgn_pos;
        }
        float* o = out + b * p.C * p.chan + within;
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
            if (p.flags & AFD_WPT_SIGN) *reinterpret_cast<float4*>(o + p.chan) = make_float4(sg[0], sg[1], sg[2], sg[3]);
        } else {
            o[0] = r[0];
            if (p.flags & AFD_WPT_SIGN) o[p.chan] = sg[0];
        }
    }
}

}  // namespace

extern "C" int afd_packet_stats(const float* x, long rows, int P, double* sums, float* absmax,
                                afd_stream_t stream) {
    if (!x || !sums || !absmax) return afd::fail(AFD_ERR_ARG, "packet stats: null pointer");
    if (rows < 0 || P < 1) return afd::fail(AFD_ERR_ARG, "packet stats: bad shape");
    if (rows == 0) return AFD_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int groups = vec ? P / 4 : P;
    int cols = 1;
    while (cols < kT && cols < groups) cols *= 2;  // power of two <= 256
    const int lanes = kT / cols;
    const unsigned gx = (unsigned)((groups + cols - 1) / cols);
    long gy = (rows + (long)lanes * 8 - 1) / ((long)lanes * 8);
    const long cap = 2048 / gx > 0 ? 2048 / gx : 1;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    if (vec)
        hipLaunchKernelGGL(packet_stats_kernel<4>, dim3(gx, (unsigned)gy), dim3(kT), 0, s, x, rows, P, cols, sums,
                           reinterpret_cast<unsigned*>(absmax));
    else
        hipLaunchKernelGGL(packet_stats_kernel<1>, dim3(gx, (unsigned)gy), dim3(kT), 0, s, x, rows, P, cols, sums,
                           reinterpret_cast<unsigned*>(absmax));
    return afd::check_launch("packet_stats_kernel");
}

extern "C" int afd_packet_block_norm(const float* x, int B, int T, int P, const float* absmax,
                                     unsigned flags, float power, float eps, float mean, float std,
                                     float sign_mean, float sign_std, float* out, afd_stream_t stream) {
    if (!x || !out) return afd::fail(AFD_ERR_ARG, "packet block norm: null pointer");
    if (B < 0 || T < 1 || P < 1) return afd::fail(AFD_ERR_ARG, "packet block norm: bad shape");
    if ((flags & AFD_WPT_SIGN) && !(flags & AFD_WPT_LOG))
        return afd::fail(AFD_ERR_ARG, "packet block norm: AFD_WPT_SIGN needs AFD_WPT_LOG");
    if ((flags & AFD_WPT_NORM) && (std == 0.f || ((flags & AFD_WPT_SIGN) && sign_std == 0.f)))
        return afd::fail(AFD_ERR_ARG, "packet block norm: std == 0");
    if (B == 0) return AFD_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    BnParams p;
    p.flags = flags;
    p.power = power;
    p.eps = eps;
    p.mean = mean;
    p.std = std;
    p.sgn_neg = (flags & AFD_WPT_NORM) ? (-1.f - sign_mean) / sign_std : -1.f;
    p.sgn_pos = (flags & AFD_WPT_NORM) ? (1.f - sign_mean) / sign_std : 1.f;
    p.TP = (long)T * P;
    p.chan = p.TP;
    p.P = P;
    p.C = (flags & AFD_WPT_SIGN) ? 2 : 1;
    const long n = (long)B * T * P;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) |
                         reinterpret_cast<uintptr_t>(absmax);
    const bool vec = (P % 4 == 0) && ((al & 15) == 0);
    const long n_groups = vec ? n / 4 : n;
    long blocks = (n_groups + kT * 4 - 1) / (kT * 4);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (vec)
        hipLaunchKernelGGL(packet_block_norm_kernel<4>, dim3((unsigned)blocks), dim3(kT), 0, s, x, absmax, out,
                           n_groups, p);
    else
        hipLaunchKernelGGL(packet_block_norm_kernel<1>, dim3((unsigned)blocks), dim3(kT), 0, s, x, absmax, out,
                           n_groups, p);
    return afd::check_launch("packet_block_norm_kernel");
}
