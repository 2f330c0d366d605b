// HBM-bound layers of the DCNN / LCNN step on gfx950: PReLU, MaxPool2d(2,2), BatchNorm
// (batch statistics + apply, forward and backward, PReLU fused on the input side),
// dropout (+ the cnn -> dil_conv permute), Linear + mean over time, cross entropy, Adam,
// scalar Normalize and the [R][C] -> [C][R] transpose used for STFT features.
//
// Replaces the torch.nn launches behind reference src/audiofakedetect/models.py:254-298,
// 301-313 (PReLU / MaxPool2d / SyncBatchNorm / Dropout / permute().contiguous() /
// Flatten+Linear / .mean(1)), train_classifier.py:970 (CrossEntropyLoss), :986 and
// :1215-1219 (Adam with coupled L2) and wavelet_math.py:380-382 (Normalize).
//
// All kernels are one pass over their operands with coalesced 4-byte-per-lane accesses,
// planes (n, c) mapped to blockIdx.y so that per-channel parameters are wave-uniform.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

constexpr int kT = 256;

__device__ __forceinline__ float prelu(float z, float a) { return z > 0.f ? z : a * z; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum of up to 3 values; result valid in thread 0
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c) {
    __shared__ float red[3][kT / 64];
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = a;
        red[1][wave] = b;
        red[2][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = b = c = 0.f;
        for (int w = 0; w < kT / 64; ++w) {
            a += red[0][w];
            b += red[1][w];
            c += red[2][w];
        }
    }
    __syncthreads();
}

// counter-based uniform in [0,1): splitmix64 of (seed, index)
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void normalize_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                 float mean, float std) {
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT)
        y[i] = (x[i] - mean) / std;
}

// per-channel form over [B][C][plane]: blockIdx.y = (frame, channel) plane, statistics by value (C <= 8)
struct ChanStats { float mean[8], std[8]; };
__global__ void normalize_channels_kernel(const float* __restrict__ x, float* __restrict__ y, size_t plane, int C,
                                          const ChanStats st) {
    const int ch = blockIdx.y % C;
    const float mean = st.mean[ch], std = st.std[ch];
    const size_t base = (size_t)blockIdx.y * plane;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < plane; i += (size_t)gridDim.x * kT)
        y[base + i] = (x[base + i] - mean) / std;
}

// y[p][c][r] = x[p][r][c]
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C) {
    __shared__ float tile[32][33];
    const size_t plane = (size_t)blockIdx.z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8)
        if (r0 + j < R && c0 + tx < C) tile[j][tx] = x[plane + (size_t)(r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) y[plane + (size_t)(c0 + j) * R + r0 + tx] = tile[tx][j];
}

// ------------------------------- PReLU (+ dropout) -------------------------------------
__global__ void prelu_dropout_fwd_kernel(const float* __restrict__ z, const float* __restrict__ slope,
                                         float* __restrict__ y, size_t n, float p,
                                         unsigned long long seed) {
    const float a = slope[0];
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) {
        float v = prelu(z[i], a);
        if (p > 0.f) v = uniform01(seed, i) >= p ? v * scale : 0.f;
        y[i] = v;
    }
}

__global__ void prelu_dropout_bwd_kernel(const float* __restrict__ z, const float* __restrict__ slope,
                                         const float* __restrict__ dy, float* __restrict__ dz,
                                         float* __restrict__ dslope, size_t n, float p,
                                         unsigned long long seed) {
    const float a = slope[0];
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float ds = 0.f, u0 = 0.f, u1 = 0.f;
    auto one = [&](float g, float zz, size_t i) {
        if (p > 0.f) g = uniform01(seed, i) >= p ? g * scale : 0.f;
        if (zz <= 0.f) ds += g * zz;
        return zz > 0.f ? g : a * g;
    };
    // 16-byte accesses where the three tensors allow (the element form ran at 2.5 TB/s)
    const bool vec = ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dz)) & 15) == 0;
    const size_t n4 = vec ? n >> 2 : 0;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n4; i += (size_t)gridDim.x * kT) {
        const float4 g = reinterpret_cast<const float4*>(dy)[i];
        const float4 zz = reinterpret_cast<const float4*>(z)[i];
        reinterpret_cast<float4*>(dz)[i] =
            make_float4(one(g.x, zz.x, 4 * i), one(g.y, zz.y, 4 * i + 1), one(g.z, zz.z, 4 * i + 2), one(g.w, zz.w, 4 * i + 3));
    }
    for (size_t i = 4 * n4 + (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT)
        dz[i] = one(dy[i], z[i], i);
    block_sum3(ds, u0, u1);
    if (threadIdx.x == 0 && ds != 0.f) atomicAdd(dslope, ds);
}

// ------------------------------- PReLU + MaxPool 2x2 -----------------------------------
// q = e / d, r = e % d for 0 <= e < 2^23 through a float reciprocal (exact after one fix-up)
__device__ __forceinline__ void divmod_small(int e, int d, float inv, int& q, int& r) {
    q = (int)((float)e * inv);
    r = e - q * d;
    if (r < 0) {
        r += d;
        --q;
    } else if (r >= d) {
        r -= d;
        ++q;
    }
}

// z [NC][H][W] -> u [NC][Hp][Wp] = max of prelu over the 2x2 window (first max wins, like
// torch); idx bits 0-1 = argmax position dy*2+dx, bit 2 = the winning z was <= 0 (so the
// backward pass needs neither z nor a second look at the window).  VEC: float2 row loads.
// two floats at any 4-byte address (rows of odd-width images): still one dwordx2 access
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

template <bool VEC>
__global__ void __launch_bounds__(kT)
prelu_pool_fwd_kernel(const float* __restrict__ z, const float* __restrict__ slope,
                      float* __restrict__ u, unsigned char* __restrict__ idx, int H, int W, int Hp,
                      int Wp, float invWp) {
    constexpr int UN = 4;  // pooled pixels per thread per round: 8 loads in flight, then the stores
    const float a = slope ? slope[0] : 1.f;
    const size_t plane = blockIdx.y;
    const float* zp = z + plane * (size_t)H * W;
    const size_t obase = plane * (size_t)Hp * Wp;
    const int total = Hp * Wp;
    for (int base = blockIdx.x * kT * UN; base < total; base += gridDim.x * kT * UN) {
        float zv[UN][4];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = base + r * kT + threadIdx.x;
            int py, px;
            divmod_small(i < total ? i : 0, Wp, invWp, py, px);
            const float* r0 = zp + (size_t)(2 * py) * W + 2 * px;
            if (VEC) {
                const f32x2u t = *reinterpret_cast<const f32x2u*>(r0);
                const f32x2u b = *reinterpret_cast<const f32x2u*>(r0 + W);
                zv[r][0] = t[0]; zv[r][1] = t[1]; zv[r][2] = b[0]; zv[r][3] = b[1];
            } else {
                zv[r][0] = r0[0]; zv[r][1] = r0[1]; zv[r][2] = r0[W]; zv[r][3] = r0[W + 1];
            }
        }
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = base + r * kT + threadIdx.x;
            if (i >= total) continue;
            const float z0 = zv[r][0], z1 = zv[r][1], z2 = zv[r][2], z3 = zv[r][3];
            float best = slope ? prelu(z0, a) : z0, zb = z0;
            int bi = 0;
            float v = slope ? prelu(z1, a) : z1;
            if (v > best) { best = v; bi = 1; zb = z1; }
            v = slope ? prelu(z2, a) : z2;
            if (v > best) { best = v; bi = 2; zb = z2; }
            v = slope ? prelu(z3, a) : z3;
            if (v > best) { best = v; bi = 3; zb = z3; }
            u[obase + i] = best;
            idx[obase + i] = (unsigned char)(bi | ((slope && zb <= 0.f) ? 4 : 0));
        }
    }
}

// Work items are (plane, chunk of kT * UN pooled pixels) pairs walked by a bounded grid: the
// slope gradient is one float atomic per WORKGROUP, and 200 000 workgroups adding to one address
// serialise in L2 (measured: 2.8 ms for an 8 GB pass that takes 1.3 ms at the HBM rate).
template <bool VEC>
__global__ void __launch_bounds__(kT)
prelu_pool_bwd_kernel(const float* __restrict__ u, const float* __restrict__ slope,
                      const unsigned char* __restrict__ idx, const float* __restrict__ du,
                      float* __restrict__ dz, float* __restrict__ dslope, int H, int W, int Hp, int Wp,
                      float invWp, int chunks, long items, const float* __restrict__ coef, int C) {
    constexpr int UN = 4;  // pooled pixels per thread per round: 12 loads in flight, then stores
    const float a = slope ? slope[0] : 1.f;
    const float inva = (slope && a != 0.f) ? 1.f / a : 0.f;
    const int total = Hp * Wp;
    float ds = 0.f, u0 = 0.f, u1 = 0.f;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const size_t plane = (size_t)(item / chunks);
        const int base = (int)(item - (long)plane * chunks) * kT * UN;
        // AFFINE (coef): the gradient of the pooled tensor is A[c] du + B[c] u + K[c] -- the backward of a
        // BatchNorm(affine=False) that follows the pool, applied where du and u are read anyway
        float kA = 1.f, kB = 0.f, kK = 0.f;
        if (coef) {
            const int c = (int)(plane % (size_t)C);
            kA = coef[4 * c]; kB = coef[4 * c + 1]; kK = coef[4 * c + 2];
        }
        float* dzp = dz + plane * (size_t)H * W;
        const size_t pbase = plane * (size_t)Hp * Wp;
        int code[UN];
        float g[UN], uu[UN];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = base + r * kT + threadIdx.x;
            const bool ok = i < total;
            code[r] = ok ? idx[pbase + i] : 0;
            g[r] = ok ? du[pbase + i] : 0.f;
            uu[r] = (ok && (slope || coef)) ? u[pbase + i] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = base + r * kT + threadIdx.x;
            if (i >= total) continue;
            int py, px;
            divmod_small(i, Wp, invWp, py, px);
            float gg = fmaf(kA, g[r], fmaf(kB, uu[r], kK));
            if (code[r] & 4) {
                ds += gg * uu[r] * inva;
                gg *= a;
            }
            const int pos = code[r] & 3;
            float* r0 = dzp + (size_t)(2 * py) * W + 2 * px;
            const float g0 = pos == 0 ? gg : 0.f, g1 = pos == 1 ? gg : 0.f;
            const float g2 = pos == 2 ? gg : 0.f, g3 = pos == 3 ? gg : 0.f;
            if (VEC) {
                f32x2u t01 = {g0, g1}, t23 = {g2, g3};
                *reinterpret_cast<f32x2u*>(r0) = t01;
                *reinterpret_cast<f32x2u*>(r0 + W) = t23;
            } else {
                r0[0] = g0; r0[1] = g1; r0[W] = g2; r0[W + 1] = g3;
            }
            if ((W & 1) && px == Wp - 1) {  // odd width: last column belongs to no window
                dzp[(size_t)(2 * py) * W + W - 1] = 0.f;
                dzp[(size_t)(2 * py + 1) * W + W - 1] = 0.f;
            }
            if ((H & 1) && py == Hp - 1) {  // odd height: last row
                dzp[(size_t)(H - 1) * W + 2 * px] = 0.f;
                dzp[(size_t)(H - 1) * W + 2 * px + 1] = 0.f;
                if ((W & 1) && px == Wp - 1) dzp[(size_t)(H - 1) * W + W - 1] = 0.f;
            }
        }
    }
    if (slope) {
        block_sum3(ds, u0, u1);
        if (threadIdx.x == 0 && ds != 0.f) atomicAdd(dslope, ds);
    }
}

// The pool's backward WITHOUT the scatter: gg[i] = the value prelu_pool_bwd_kernel would route to position (code & 3) of
// window i (BatchNorm-backward affine and PReLU slope applied), at pooled resolution.  The consumers of the
// gradient (the F(4x4) backward-data and backward-weight kernels, wino44*.hip) expand it from gg and the codes while
// they load: the dense gradient -- three quarters zeros, written once and read twice -- never exists.
__global__ void __launch_bounds__(kT)
prelu_pool_bwd_compact_kernel(const float* __restrict__ u, const float* __restrict__ slope,
                              const unsigned char* __restrict__ idx, const float* __restrict__ du,
                              float* __restrict__ gg, float* __restrict__ dslope, int HWp, int chunks, long items,
                              const float* __restrict__ coef, int C) {
    constexpr int UN = 4;
    const float a = slope ? slope[0] : 1.f;
    const float inva = (slope && a != 0.f) ? 1.f / a : 0.f;
    float ds = 0.f, u0 = 0.f, u1 = 0.f;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const size_t plane = (size_t)(item / chunks);
        const int base = (int)(item - (long)plane * chunks) * kT * UN;
        float kA = 1.f, kB = 0.f, kK = 0.f;
        if (coef) {
            const int c = (int)(plane % (size_t)C);
            kA = coef[4 * c]; kB = coef[4 * c + 1]; kK = coef[4 * c + 2];
        }
        const size_t pbase = plane * (size_t)HWp;
        int code[UN];
        float g[UN], uu[UN];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = base + r * kT + threadIdx.x;
            const bool ok = i < HWp;
            code[r] = ok ? idx[pbase + i] : 0;
            g[r] = ok ? du[pbase + i] : 0.f;
            uu[r] = (ok && (slope || coef)) ? u[pbase + i] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = base + r * kT + threadIdx.x;
            if (i >= HWp) continue;
            float v = fmaf(kA, g[r], fmaf(kB, uu[r], kK));
            if (code[r] & 4) {
                ds += v * uu[r] * inva;
                v *= a;
            }
            gg[pbase + i] = v;
        }
    }
    if (slope) {
        block_sum3(ds, u0, u1);
        if (threadIdx.x == 0 && ds != 0.f) atomicAdd(dslope, ds);
    }
}

// the same with four pooled pixels per lane and access (16-byte loads of du and u, the four codes as one dword, a
// 16-byte store): the 4-byte-per-lane form above moved 2.8 TB/s where the BatchNorm passes reach 5.4 on the same tensors
__global__ void __launch_bounds__(kT)
prelu_pool_bwd_compact4_kernel(const float* __restrict__ u, const float* __restrict__ slope,
                               const unsigned char* __restrict__ idx, const float* __restrict__ du,
                               float* __restrict__ gg, float* __restrict__ dslope, int Q /* quads per plane */, int chunks,
                               long items, const float* __restrict__ coef, int C) {
    constexpr int UN = 2;
    const float a = slope ? slope[0] : 1.f;
    const float inva = (slope && a != 0.f) ? 1.f / a : 0.f;
    const bool need_u = slope || coef;
    float ds = 0.f, u0 = 0.f, u1 = 0.f;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const size_t plane = (size_t)(item / chunks);
        const int base = (int)(item - (long)plane * chunks) * kT * UN;
        float kA = 1.f, kB = 0.f, kK = 0.f;
        if (coef) {
            const int c = (int)(plane % (size_t)C);
            kA = coef[4 * c]; kB = coef[4 * c + 1]; kK = coef[4 * c + 2];
        }
        const size_t qbase = plane * (size_t)Q;
        const float4* du4 = reinterpret_cast<const float4*>(du) + qbase;
        const float4* u4 = reinterpret_cast<const float4*>(u) + qbase;
        const unsigned* c4 = reinterpret_cast<const unsigned*>(idx) + qbase;
        float4* g4 = reinterpret_cast<float4*>(gg) + qbase;
        unsigned code[UN];
        float4 g[UN], uu[UN];
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = base + r * kT + threadIdx.x;
            const bool ok = i < Q;
            code[r] = ok ? c4[i] : 0u;
            g[r] = ok ? du4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            uu[r] = (ok && need_u) ? u4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < UN; ++r) {
            const int i = base + r * kT + threadIdx.x;
            if (i >= Q) continue;
            const float gv[4] = {g[r].x, g[r].y, g[r].z, g[r].w}, uv[4] = {uu[r].x, uu[r].y, uu[r].z, uu[r].w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = fmaf(kA, gv[j], fmaf(kB, uv[j], kK));
                if ((code[r] >> (8 * j)) & 4u) {
                    ds += v * uv[j] * inva;
                    v *= a;
                }
                o[j] = v;
            }
            g4[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (slope) {
        block_sum3(ds, u0, u1);
        if (threadIdx.x == 0 && ds != 0.f) atomicAdd(dslope, ds);
    }
}

// ------------------------------- BatchNorm ---------------------------------------------
// Visits the HW elements of one (n, c) plane with 16-byte accesses: a scalar head up to the
// first 16-byte boundary of the tensor, float4 body, scalar tail.  f1(i) / f4(i) receive the
// element index inside the plane; blocks along x share the plane.
template <typename F1, typename F4>
__device__ __forceinline__ void plane_loop(size_t plane_base, int HW, int bx, int nbx, F1 f1, F4 f4) {
    int head = (int)((4 - (plane_base & 3)) & 3);
    if (head > HW) head = HW;
    const int nvec = (HW - head) >> 2;
    // four independent float4 visits per trip: with __restrict__ operands the compiler hoists
    // the four loads above the first store (one memory latency per trip instead of four)
    for (int v0 = bx * kT * 4 + threadIdx.x; v0 < nvec; v0 += nbx * kT * 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = v0 + r * kT;
            if (v < nvec) f4(head + 4 * v);
        }
    }
    if (bx == 0) {
        if ((int)threadIdx.x < head) f1((int)threadIdx.x);
        const int t = head + 4 * nvec + (int)threadIdx.x;
        if (threadIdx.x < 4 && t < HW) f1(t);
    }
}

// per-channel sum and sum of squares of (optionally PReLU'd) x [N][C][HW]; double atomics
__global__ void bn_stats_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                double* __restrict__ sums, int N, int C, int HW) {
    const int c = blockIdx.x;
    const bool act = slope != nullptr;
    const float a = act ? slope[0] : 1.f;
    float s = 0.f, q = 0.f, un = 0.f;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const size_t base = ((size_t)n * C + c) * HW;
        const float* p = x + base;
        float ls = 0.f, lq = 0.f;
        plane_loop(base, HW, blockIdx.z, gridDim.z,
                   [&](int i) {
                       float v = p[i];
                       if (act) v = prelu(v, a);
                       ls += v;
                       lq += v * v;
                   },
                   [&](int i) {
                       float4 v = *reinterpret_cast<const float4*>(p + i);
                       if (act) {
                           v.x = prelu(v.x, a); v.y = prelu(v.y, a);
                           v.z = prelu(v.z, a); v.w = prelu(v.w, a);
                       }
                       ls += (v.x + v.y) + (v.z + v.w);
                       lq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                   });
        s += ls;
        q += lq;
    }
    block_sum3(s, q, un);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], (double)s);
        atomicAdd(&sums[C + c], (double)q);
    }
}

// y = (prelu?(x) - mean) * invstd * gamma + beta
__global__ void bn_apply_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    float* __restrict__ y, int C, int HW) {
    const size_t plane = blockIdx.y;
    const int c = (int)(plane % C);
    const bool act = slope != nullptr;
    const float a = act ? slope[0] : 1.f;
    const float m = mean[c];
    const float sc = invstd[c] * (gamma ? gamma[c] : 1.f);
    const float sh = beta ? beta[c] : 0.f;
    const size_t base = plane * (size_t)HW;
    const float* xp = x + base;
    float* yp = y + base;
    auto one = [&](float v) { return ((act ? prelu(v, a) : v) - m) * sc + sh; };
    plane_loop(base, HW, blockIdx.x, gridDim.x, [&](int i) { yp[i] = one(xp[i]); },
               [&](int i) {
                   const float4 v = *reinterpret_cast<const float4*>(xp + i);
                   *reinterpret_cast<float4*>(yp + i) = make_float4(one(v.x), one(v.y), one(v.z), one(v.w));
               });
}

// per-channel sum(dy), sum(dy * xhat)
__global__ void bn_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                    const float* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, double* __restrict__ sums,
                                    int N, int C, int HW) {
    const int c = blockIdx.x;
    const bool act = slope != nullptr;
    const float a = act ? slope[0] : 1.f;
    const float m = mean[c], is = invstd[c];
    float s = 0.f, q = 0.f, un = 0.f;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const size_t base = ((size_t)n * C + c) * HW;
        const float* xp = x + base;
        const float* gp = dy + base;
        float ls = 0.f, lq = 0.f;
        auto one = [&](float v, float g) {
            if (act) v = prelu(v, a);
            ls += g;
            lq += g * (v - m) * is;
        };
        plane_loop(base, HW, blockIdx.z, gridDim.z, [&](int i) { one(xp[i], gp[i]); },
                   [&](int i) {
                       const float4 v = *reinterpret_cast<const float4*>(xp + i);
                       const float4 g = *reinterpret_cast<const float4*>(gp + i);
                       one(v.x, g.x); one(v.y, g.y); one(v.z, g.z); one(v.w, g.w);
                   });
        s += ls;
        q += lq;
    }
    block_sum3(s, q, un);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], (double)s);
        atomicAdd(&sums[C + c], (double)q);
    }
}

// dx = gamma * invstd * (dy - mdy - xhat * mdyx); through the fused PReLU: dz, dslope
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                    const float* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ mdy, const float* __restrict__ mdyx,
                                    float* __restrict__ dx, float* __restrict__ dslope, int C,
                                    int HW, double* __restrict__ dxsum, int N, int G) {
    // workgroup (x, y): channel c = y % C of the images G (y / C) .. + G - 1 (G = 1: one plane per workgroup; small planes
    // -- the level-8 dilated stack -- share a workgroup, whose two atomics then close 16 k elements instead of 2 k)
    const int c = (int)(blockIdx.y % C);
    const int n_begin = (int)(blockIdx.y / C) * G;
    const int n_end = min(n_begin + G, N);
    const bool act = slope != nullptr;
    const float a = act ? slope[0] : 1.f;
    const float m = mean[c], is = invstd[c];
    const float gs = is * (gamma ? gamma[c] : 1.f);
    const float k0 = mdy[c], k1 = mdyx[c];
    float ds = 0.f, u0 = 0.f, u1 = 0.f;
    for (int n = n_begin; n < n_end; ++n) {
    const size_t base = ((size_t)n * C + c) * (size_t)HW;
    const float* xp = x + base;
    const float* gp = dy + base;
    float* op = dx + base;
    auto one = [&](float zz, float gy) {
        const float v = act ? prelu(zz, a) : zz;
        const float xh = (v - m) * is;
        float g = gs * (gy - k0 - xh * k1);
        if (act && zz <= 0.f) {
            ds += g * zz;
            g *= a;
        }
        u0 += g;  // per-channel sum of the result: the bias gradient of the convolution that produced x
        return g;
    };
    plane_loop(base, HW, blockIdx.x, gridDim.x, [&](int i) { op[i] = one(xp[i], gp[i]); },
               [&](int i) {
                   const float4 v = *reinterpret_cast<const float4*>(xp + i);
                   const float4 g = *reinterpret_cast<const float4*>(gp + i);
                   *reinterpret_cast<float4*>(op + i) =
                       make_float4(one(v.x, g.x), one(v.y, g.y), one(v.z, g.z), one(v.w, g.w));
               });
    }
    if (act || dxsum) {
        block_sum3(ds, u0, u1);
        if (threadIdx.x == 0 && act && ds != 0.f) atomicAdd(dslope, ds);
        if (threadIdx.x == 0 && dxsum) atomicAdd(dxsum + c, (double)u0);
    }
}

// ------------------------------- dropout + permute(0,2,1,3) ----------------------------
// x [B][C][H][W] -> y [B][H][C][W]; mask index = flat index of x
__global__ void dropout_permute_kernel(const float* __restrict__ x, float* __restrict__ y, int C,
                                       int H, int W, float p, unsigned long long seed,
                                       int inverse) {
    // forward: reads x[b][c][h][w], writes y[b][h][c][w]
    // inverse: reads dy[b][h][c][w], writes dx[b][c][h][w] (same mask, same scale)
    const size_t b = blockIdx.y;
    const size_t per = (size_t)C * H * W;
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < per; i += (size_t)gridDim.x * kT) {
        const int w = (int)(i % W);
        const int h = (int)((i / W) % H);
        const int c = (int)(i / ((size_t)W * H));
        const size_t xi = b * per + i;
        const size_t yi = b * per + ((size_t)h * C + c) * W + w;
        const bool keep = p > 0.f ? (uniform01(seed, xi) >= p) : true;
        if (!inverse) y[yi] = keep ? x[xi] * scale : 0.f;
        else y[xi] = keep ? x[yi] * scale : 0.f;
    }
}

// the same with four consecutive w per thread (W % 4 == 0, C * H * W < 2^31): 16-byte accesses, 32-bit index arithmetic
// (the element form spends ~60 instructions per element on 64-bit divisions); the mask is still indexed per element
__global__ void dropout_permute4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W4,
                                        float p, unsigned long long seed, int inverse) {
    const size_t b = blockIdx.y;
    const unsigned per4 = (unsigned)C * (unsigned)H * (unsigned)W4;
    const size_t base = b * (size_t)per4 * 4;
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const float4* src = reinterpret_cast<const float4*>((inverse ? x : x) + base);
    float4* dst = reinterpret_cast<float4*>(y + base);
    for (unsigned i = blockIdx.x * kT + threadIdx.x; i < per4; i += gridDim.x * kT) {
        const unsigned w4 = i % (unsigned)W4;
        const unsigned r = i / (unsigned)W4;
        const unsigned h = r % (unsigned)H, c = r / (unsigned)H;
        const unsigned xi4 = i;                                    // [c][h][w4]
        const unsigned yi4 = (h * (unsigned)C + c) * (unsigned)W4 + w4;  // [h][c][w4]
        float4 v = src[inverse ? yi4 : xi4];
        if (p > 0.f) {
            const size_t e = base + (size_t)xi4 * 4;
            v.x = uniform01(seed, e) >= p ? v.x * scale : 0.f;
            v.y = uniform01(seed, e + 1) >= p ? v.y * scale : 0.f;
            v.z = uniform01(seed, e + 2) >= p ? v.z * scale : 0.f;
            v.w = uniform01(seed, e + 3) >= p ? v.w * scale : 0.f;
        }
        dst[inverse ? xi4 : yi4] = v;
    }
}

// ------------------------------- Linear + mean(1) --------------------------------------
// x [B][TD][F], w [O][F], bias [O] -> y [B][O] = bias + (1/TD) sum_t x[b,t,:] . w[o,:]
// One workgroup of 1024 threads per frame (the batch is the only parallel dimension of the reduction that keeps a
// fixed summation order; 16 waves with 16-byte loads cover the 243k elements of a level-14 frame in 20 rounds)
constexpr int kLinT = 1024;
template <int O>
__global__ void __launch_bounds__(kLinT)
linear_mean_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                       const float* __restrict__ bias, float* __restrict__ y, int TD, int F) {
    const int b = blockIdx.x;
    const float* xb = x + (size_t)b * TD * F;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    if ((F & 3) == 0) {
        const int F4 = F >> 2;
        for (int f = threadIdx.x; f < F4; f += kLinT) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < TD; ++t) {
                const float4 v = reinterpret_cast<const float4*>(xb + (size_t)t * F)[f];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
#pragma unroll
            for (int o = 0; o < O; ++o) {
                const float4 wv = reinterpret_cast<const float4*>(w + (size_t)o * F)[f];
                acc[o] = fmaf(s.x, wv.x, fmaf(s.y, wv.y, fmaf(s.z, wv.z, fmaf(s.w, wv.w, acc[o]))));
            }
        }
    } else {
        for (int f = threadIdx.x; f < F; f += kLinT) {
            float s = 0.f;
            for (int t = 0; t < TD; ++t) s += xb[(size_t)t * F + f];
#pragma unroll
            for (int o = 0; o < O; ++o) acc[o] = fmaf(s, w[(size_t)o * F + f], acc[o]);
        }
    }
    __shared__ float red[O][kLinT / 64];
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float v = wave_sum(acc[o]);
        if ((threadIdx.x & 63) == 0) red[o][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < O) {
        float s = 0.f;
        for (int k = 0; k < kLinT / 64; ++k) s += red[threadIdx.x][k];
        y[(size_t)b * O + threadIdx.x] = s / (float)TD + bias[threadIdx.x];
    }
}

// dx[b][t][f] = (1/TD) sum_o dy[b][o] w[o][f]
template <int O>
__global__ void linear_mean_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                         float* __restrict__ dx, int TD, int F) {
    const int b = blockIdx.y;
    float g[O];
#pragma unroll
    for (int o = 0; o < O; ++o) g[o] = dy[(size_t)b * O + o] / (float)TD;
    for (int f = blockIdx.x * kT + threadIdx.x; f < F; f += gridDim.x * kT) {
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < O; ++o) v = fmaf(g[o], w[(size_t)o * F + f], v);
        for (int t = 0; t < TD; ++t) dx[((size_t)b * TD + t) * F + f] = v;
    }
}

// dw[o][f] = (1/TD) sum_b dy[b][o] sum_t x[b][t][f];  db[o] = sum_b dy[b][o]
// block = 64 features x 16 batch groups (F is 320 in the shipped configs: five workgroups; with 4 groups a thread
// walked 32 frames x 13 time steps of dependent 4-byte loads -- 74 us at B = 128 for 2 x 320 results; now 8 frames per
// thread with the time steps of a frame requested together)
constexpr int kLinGroups = 16;
template <int O>
__global__ void __launch_bounds__(64 * kLinGroups)
linear_mean_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                         float* __restrict__ dw, float* __restrict__ db, int B, int TD, int F) {
    __shared__ float red[O][kLinGroups][65];
    const int fl = threadIdx.x & 63;
    const int g = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + fl;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    if (f < F) {
        for (int b = g; b < B; b += kLinGroups) {
            const float* xb = x + (size_t)b * TD * F + f;
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            int t = 0;
            for (; t + 3 < TD; t += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) s[u] += xb[(size_t)(t + u) * F];
            }
            for (; t < TD; ++t) s[0] += xb[(size_t)t * F];
            const float sx = (s[0] + s[1]) + (s[2] + s[3]);
#pragma unroll
            for (int o = 0; o < O; ++o) acc[o] = fmaf(sx, dy[(size_t)b * O + o], acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < O; ++o) red[o][g][fl] = acc[o];
    __syncthreads();
    if (g == 0 && f < F) {
#pragma unroll
        for (int o = 0; o < O; ++o) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < kLinGroups; ++k) v += red[o][k][fl];  // fixed order: deterministic
            dw[(size_t)o * F + f] = v / (float)TD;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < O) {
        float sb = 0.f;
        for (int b = 0; b < B; ++b) sb += dy[(size_t)b * O + threadIdx.x];
        db[threadIdx.x] = sb;
    }
}

// ------------------------------- cross entropy -----------------------------------------
// logits [B][O], labels int64 [B] -> loss (mean), dlogits = (softmax - onehot) / B,
// correct = #(argmax == label)
__global__ void ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                          float* __restrict__ loss, float* __restrict__ dlogits,
                          float* __restrict__ correct, int B, int O) {
    float ls = 0.f, cs = 0.f, u = 0.f;
    for (int b = threadIdx.x; b < B; b += kT) {
        const float* l = logits + (size_t)b * O;
        float mx = l[0];
        int am = 0;
        for (int o = 1; o < O; ++o)
            if (l[o] > mx) { mx = l[o]; am = o; }
        float se = 0.f;
        for (int o = 0; o < O; ++o) se += expf(l[o] - mx);
        const float lse = logf(se) + mx;
        const int lab = (int)labels[b];
        ls += lse - l[lab];
        cs += (am == lab) ? 1.f : 0.f;
        if (dlogits)
            for (int o = 0; o < O; ++o)
                dlogits[(size_t)b * O + o] = (expf(l[o] - lse) - (o == lab ? 1.f : 0.f)) / (float)B;
    }
    block_sum3(ls, cs, u);
    if (threadIdx.x == 0) {
        loss[0] = ls / (float)B;
        if (correct) correct[0] = cs;
    }
}

// ------------------------------- Adam (coupled L2) -------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                            float wd, float bc1, float bc2_sqrt, float gscale) {
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) {
        const float pi = p[i];
        const float gi = fmaf(wd, pi, g[i] * gscale);
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

inline unsigned grid1d(size_t n, int cap = 4096) {
    size_t b = (n + kT - 1) / kT;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

#define AFD_STREAM static_cast<hipStream_t>(stream)

// grid-stride double-precision sum / sum of squares; one double atomic pair per workgroup
__global__ void __launch_bounds__(256)
moments_kernel(const float* __restrict__ x, size_t n, double* __restrict__ acc) {
    __shared__ double red[2][4];
    double s = 0.0, q = 0.0;
    const size_t n4 = n >> 2;
    // head to a 16-byte boundary is not needed: feature tensors come from the allocator
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const bool aligned = (reinterpret_cast<size_t>(x) & 15) == 0;
    if (aligned) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            const float4 v = x4[i];
            const float a = v.x + v.y, b = v.z + v.w;
            const float qa = fmaf(v.x, v.x, v.y * v.y), qb = fmaf(v.z, v.z, v.w * v.w);
            s += (double)a + (double)b;
            q += (double)qa + (double)qb;
        }
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            s += x[i];
            q += (double)x[i] * x[i];
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
            s += x[i];
            q += (double)x[i] * x[i];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        q += __shfl_down(q, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][wave] = s;
        red[1][wave] = q;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(acc + 1, (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        atomicAdd(acc + 2, (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
        if (blockIdx.x == 0) atomicAdd(acc, (double)n);
    }
}

extern "C" int afd_moments_accumulate(const float* x, size_t n, double* acc, afd_stream_t stream) {
    if (!x || !acc) return afd::fail(AFD_ERR_ARG, "moments: null pointer");
    if (n == 0) return AFD_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 4.0 * (double)n, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(moments_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, acc);
    return afd::check_launch("moments_kernel");
}

extern "C" int afd_normalize_forward(const float* x, float* y, size_t n, float mean, float std,
                                     afd_stream_t stream) {
    if (!x || !y || std == 0.f) return afd::fail(AFD_ERR_ARG, "normalize: bad argument");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 8.0 * (double)n, AFD_STREAM);
    hipLaunchKernelGGL(normalize_kernel, dim3(grid1d(n)), dim3(kT), 0, AFD_STREAM, x, y, n, mean, std);
    return afd::check_launch("normalize_kernel");
}

extern "C" int afd_normalize_channels_forward(const float* x, float* y, int B, int C, size_t plane, const float* means,
                                              const float* stds, afd_stream_t stream) {
    if (!x || !y || !means || !stds || B < 1 || C < 1 || plane < 1) return afd::fail(AFD_ERR_ARG, "normalize channels: bad argument");
    if (C > 8 || (long)B * C > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "normalize channels: %d channels x %d frames", C, B);
    ChanStats st{};
    for (int c = 0; c < C; ++c) {
        if (stds[c] == 0.f) return afd::fail(AFD_ERR_ARG, "normalize channels: zero std");
        st.mean[c] = means[c];
        st.std[c] = stds[c];
    }
    size_t gx = (plane + kT - 1) / kT;
    if (gx > 1024) gx = 1024;
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 8.0 * (double)B * C * (double)plane, AFD_STREAM);
    hipLaunchKernelGGL(normalize_channels_kernel, dim3((unsigned)gx, (unsigned)(B * C)), dim3(kT), 0, AFD_STREAM, x, y,
                       plane, C, st);
    return afd::check_launch("normalize_channels_kernel");
}

extern "C" int afd_transpose_last2(const float* x, float* y, int planes, int R, int C,
                                   afd_stream_t stream) {
    if (!x || !y || planes < 1 || R < 1 || C < 1) return afd::fail(AFD_ERR_ARG, "transpose: bad argument");
    if (planes > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "transpose: too many planes");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 8.0 * (double)planes * R * C, AFD_STREAM);
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, planes), dim3(kT), 0,
                       AFD_STREAM, x, y, R, C);
    return afd::check_launch("transpose_kernel");
}

extern "C" int afd_prelu_dropout_forward(const float* z, const float* slope, float* y, size_t n,
                                         float p, uint64_t seed, afd_stream_t stream) {
    if (!z || !slope || !y || p < 0.f || p >= 1.f) return afd::fail(AFD_ERR_ARG, "prelu fwd: bad argument");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 8.0 * (double)n, AFD_STREAM);
    hipLaunchKernelGGL(prelu_dropout_fwd_kernel, dim3(grid1d(n)), dim3(kT), 0, AFD_STREAM, z, slope,
                       y, n, p, (unsigned long long)seed);
    return afd::check_launch("prelu_dropout_fwd_kernel");
}

extern "C" int afd_prelu_dropout_backward(const float* z, const float* slope, const float* dy,
                                          float* dz, float* dslope, size_t n, float p,
                                          uint64_t seed, afd_stream_t stream) {
    if (!z || !slope || !dy || !dz || !dslope) return afd::fail(AFD_ERR_ARG, "prelu bwd: null pointer");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 12.0 * (double)n, AFD_STREAM);
    hipLaunchKernelGGL(prelu_dropout_bwd_kernel, dim3(grid1d(n, 1024)), dim3(kT), 0, AFD_STREAM, z,
                       slope, dy, dz, dslope, n, p, (unsigned long long)seed);
    return afd::check_launch("prelu_dropout_bwd_kernel");
}

extern "C" int afd_prelu_pool_forward(const float* z, const float* slope, float* u, uint8_t* idx,
                                      int NC, int H, int W, afd_stream_t stream) {
    if (!z || !u || !idx || NC < 1 || H < 2 || W < 2) return afd::fail(AFD_ERR_ARG, "pool fwd: bad argument");
    const int Hp = H / 2, Wp = W / 2;
    if ((long)Hp * Wp >= (1L << 23)) return afd::fail(AFD_ERR_UNSUPPORTED, "pool: plane too large");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, (double)NC * (4.0 * H * W + 5.0 * Hp * Wp), AFD_STREAM);
    const unsigned gx = grid1d((size_t)Hp * Wp, 64);
    // float2 rows need 8-byte alignment: even W, even plane size, 8-byte aligned base
    const bool vec = true;  // dwordx2 loads need 4-byte alignment only
    for (int p0 = 0; p0 < NC; p0 += 65535) {
        const int np = NC - p0 < 65535 ? NC - p0 : 65535;
        const float* zp = z + (size_t)p0 * H * W;
        float* up = u + (size_t)p0 * Hp * Wp;
        uint8_t* ip = idx + (size_t)p0 * Hp * Wp;
        if (vec)
            hipLaunchKernelGGL(prelu_pool_fwd_kernel<true>, dim3(gx, np), dim3(kT), 0, AFD_STREAM, zp, slope, up, ip, H, W, Hp, Wp, 1.0f / Wp);
        else
            hipLaunchKernelGGL(prelu_pool_fwd_kernel<false>, dim3(gx, np), dim3(kT), 0, AFD_STREAM, zp, slope, up, ip, H, W, Hp, Wp, 1.0f / Wp);
    }
    return afd::check_launch("prelu_pool_fwd_kernel");
}

extern "C" int afd_prelu_pool_backward(const float* u, const float* slope, const uint8_t* idx,
                                       const float* du, float* dz, float* dslope, int NC, int H,
                                       int W, afd_stream_t stream) {
    return afd_prelu_pool_backward_affine(u, slope, idx, du, nullptr, 1, dz, dslope, NC, H, W, stream);
}

extern "C" int afd_prelu_pool_backward_compact(const float* u, const float* slope, const uint8_t* idx,
                                               const float* du, const float* coef, int C, float* gg,
                                               float* dslope, int NC, int Hp, int Wp, afd_stream_t stream) {
    if (!u || !idx || !du || !gg || (slope && !dslope)) return afd::fail(AFD_ERR_ARG, "pool bwd (compact): null pointer");
    if (coef && (C < 1 || NC % C != 0)) return afd::fail(AFD_ERR_ARG, "pool bwd (compact): planes are not a multiple of the channels");
    if (NC < 1 || Hp < 1 || Wp < 1 || (long)Hp * Wp >= (1L << 30)) return afd::fail(AFD_ERR_ARG, "pool bwd (compact): bad geometry");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 13.0 * (double)NC * Hp * Wp, AFD_STREAM);
    const int HWp = Hp * Wp;
    const int chunks = (HWp + kT * 4 - 1) / (kT * 4);
    const long items = (long)NC * chunks;
    const unsigned blocks = (unsigned)(items < 8192 ? items : 8192);
    // planes of a multiple of four pooled pixels behind 16-byte aligned tensors (the level-14 and level-8 models): the
    // 16-byte-per-lane form -- one dwordx4 of du and u and one dword of codes in, one dwordx4 out
    const bool vec = HWp % 4 == 0 && (((uintptr_t)u | (uintptr_t)du | (uintptr_t)gg) & 15) == 0 && ((uintptr_t)idx & 3) == 0;
    if (vec) {
        const int chunks4 = (HWp / 4 + kT * 2 - 1) / (kT * 2);
        const long items4 = (long)NC * chunks4;
        const unsigned blocks4 = (unsigned)(items4 < 16384 ? items4 : 16384);
        hipLaunchKernelGGL(prelu_pool_bwd_compact4_kernel, dim3(blocks4), dim3(kT), 0, AFD_STREAM, u, slope, idx, du, gg, dslope,
                           HWp / 4, chunks4, items4, coef, C);
        return afd::check_launch("prelu_pool_bwd_compact_kernel");
    }
    hipLaunchKernelGGL(prelu_pool_bwd_compact_kernel, dim3(blocks), dim3(kT), 0, AFD_STREAM, u, slope, idx, du, gg, dslope,
                       HWp, chunks, items, coef, C);
    return afd::check_launch("prelu_pool_bwd_compact_kernel");
}

extern "C" int afd_prelu_pool_backward_affine(const float* u, const float* slope, const uint8_t* idx,
                                              const float* du, const float* coef, int C, float* dz,
                                              float* dslope, int NC, int H, int W, afd_stream_t stream) {
    if (!u || !idx || !du || !dz || (slope && !dslope)) return afd::fail(AFD_ERR_ARG, "pool bwd: null pointer");
    if (coef && (C < 1 || NC % C != 0)) return afd::fail(AFD_ERR_ARG, "pool bwd: planes are not a multiple of the channels");
    const int Hp = H / 2, Wp = W / 2;
    if ((long)Hp * Wp >= (1L << 23)) return afd::fail(AFD_ERR_UNSUPPORTED, "pool: plane too large");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, (double)NC * (9.0 * Hp * Wp + 4.0 * H * W), AFD_STREAM);
    const unsigned gx = grid1d(((size_t)Hp * Wp + 3) / 4, 16);
    const bool vec = true;  // dwordx2 stores need 4-byte alignment only
    {
        const int chunks = (Hp * Wp + kT * 4 - 1) / (kT * 4);
        const long items = (long)NC * chunks;
        const unsigned blocks = (unsigned)(items < 8192 ? items : 8192);
        (void)gx;
        if (vec)
            hipLaunchKernelGGL(prelu_pool_bwd_kernel<true>, dim3(blocks), dim3(kT), 0, AFD_STREAM, u, slope, idx, du, dz, dslope, H, W, Hp, Wp, 1.0f / Wp, chunks, items, coef, C);
        else
            hipLaunchKernelGGL(prelu_pool_bwd_kernel<false>, dim3(blocks), dim3(kT), 0, AFD_STREAM, u, slope, idx, du, dz, dslope, H, W, Hp, Wp, 1.0f / Wp, chunks, items, coef, C);
    }
    return afd::check_launch("prelu_pool_bwd_kernel");
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, int C, double count, float eps,
                                   float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ nbt, double* __restrict__ count_out,
                                   float* __restrict__ fold_tab) {
    const double cnt = count < 0.0 ? sums[2 * C] : count;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const double m = sums[c] / cnt;
        double var = sums[C + c] / cnt - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (fold_tab) {  // [C][2] = (mean, invstd): the table the folded convolution launches read
            fold_tab[2 * c] = mean[c];
            fold_tab[2 * c + 1] = invstd[c];
        }
        if (running_mean) {
            const double unbiased = var * (cnt / (cnt > 1.0 ? cnt - 1.0 : 1.0));
            running_mean[c] = running_mean[c] * (1.f - momentum) + (float)m * momentum;
            running_var[c] = running_var[c] * (1.f - momentum) + (float)unbiased * momentum;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (nbt) *nbt += 1;
        if (count_out) *count_out = cnt;
    }
}

__global__ void bn_bwd_means_kernel(const double* __restrict__ sums, int C, double count,
                                    const double* __restrict__ count_dev, float* __restrict__ mdy,
                                    float* __restrict__ mdyx, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, float* __restrict__ tab4) {
    const double cnt = count < 0.0 ? *count_dev : count;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        const float a = (float)(sums[c] / cnt), b = (float)(sums[C + c] / cnt);
        mdy[c] = a;
        mdyx[c] = b;
        if (tab4) {  // [C][4] = (mean, invstd, mean of g, mean of g xhat): afd_conv3x3_backward_data_bnapply's table
            tab4[4 * c] = mean[c];
            tab4[4 * c + 1] = invstd[c];
            tab4[4 * c + 2] = a;
            tab4[4 * c + 3] = b;
        }
    }
}

extern "C" int afd_bn_finalize(const double* sums, int C, double count, float eps, float momentum,
                               float* mean, float* invstd, float* running_mean, float* running_var,
                               long long* nbt, double* count_out, float* fold_tab, afd_stream_t stream) {
    if (!sums || !mean || !invstd || C < 1) return afd::fail(AFD_ERR_ARG, "bn finalize: bad argument");
    if ((running_mean == nullptr) != (running_var == nullptr)) return afd::fail(AFD_ERR_ARG, "bn finalize: running stats");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 40.0 * C, AFD_STREAM);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, AFD_STREAM, sums, C, count, eps,
                       momentum, mean, invstd, running_mean, running_var, nbt, count_out, fold_tab);
    return afd::check_launch("bn_finalize_kernel");
}

extern "C" int afd_bn_backward_means(const double* sums, int C, double count, const double* count_dev,
                                     float* mdy, float* mdyx, const float* mean, const float* invstd, float* tab4,
                                     afd_stream_t stream) {
    if (!sums || !mdy || !mdyx || C < 1 || (count < 0.0 && !count_dev) || (tab4 && (!mean || !invstd)))
        return afd::fail(AFD_ERR_ARG, "bn backward means: bad argument");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 24.0 * C, AFD_STREAM);
    hipLaunchKernelGGL(bn_bwd_means_kernel, dim3((C + 255) / 256), dim3(256), 0, AFD_STREAM, sums, C, count,
                       count_dev, mdy, mdyx, mean, invstd, tab4);
    return afd::check_launch("bn_bwd_means_kernel");
}

// ---- the small matrices of a BatchNorm folded into the 1x1 convolution after it (ops._BNConv1x1*) ----
namespace {

// wf[co][ci] = w[co][ci] * invstd[ci];  bf[co] = b[co] - sum_ci wf[co][ci] * mean[ci]; one wave per output channel
__global__ void __launch_bounds__(64)
bn_fold_forward_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean,
                       const float* __restrict__ invstd, float* __restrict__ wf, float* __restrict__ bf, int Cin) {
    const int co = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int ci = lane; ci < Cin; ci += 64) {
        const float v = w[(size_t)co * Cin + ci] * invstd[ci];
        wf[(size_t)co * Cin + ci] = v;
        acc = fmaf(v, mean[ci], acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) bf[co] = (b ? b[co] : 0.f) - acc;
}

// dw[co][ci] = (G[co][ci] - db[co] mean[ci]) invstd[ci]  (gradient against the normalised input from the one
// against the un-normalised input); sums[ci] = sum_co w[co][ci] db[co] = sum_px dxhat[ci],
// sums[Cin + ci] = sum_co w[co][ci] dw[co][ci] = sum_px dxhat[ci] xhat[ci]; thread = input channel
__global__ void __launch_bounds__(128)
bn_fold_backward_weights_kernel(const float* __restrict__ G, const float* __restrict__ db,
                                const float* __restrict__ w, const float* __restrict__ mean,
                                const float* __restrict__ invstd, float* __restrict__ dw,
                                double* __restrict__ sums, int C, int Cin) {
    const int ci = blockIdx.x * 128 + threadIdx.x;
    if (ci >= Cin) return;
    const float m = mean[ci], is = invstd[ci];
    double s1 = 0.0, s2 = 0.0;
    for (int co = 0; co < C; ++co) {
        const float d = db[co], wv = w[(size_t)co * Cin + ci];
        const float g = (G[(size_t)co * Cin + ci] - d * m) * is;
        dw[(size_t)co * Cin + ci] = g;
        s1 += (double)wv * (double)d;
        s2 += (double)wv * (double)g;
    }
    sums[ci] = s1;
    sums[Cin + ci] = s2;
}

// alpha = -invstd^2 E[dxhat xhat], beta = invstd^2 E[dxhat xhat] mean - invstd E[dxhat]
__global__ void __launch_bounds__(128)
bn_fold_backward_affine_kernel(const double* __restrict__ sums, double count, const double* __restrict__ count_dev,
                               const float* __restrict__ mean, const float* __restrict__ invstd,
                               float* __restrict__ alpha, float* __restrict__ beta, int Cin) {
    const int ci = blockIdx.x * 128 + threadIdx.x;
    if (ci >= Cin) return;
    const double cnt = count < 0.0 ? count_dev[0] : count;
    const float m1 = (float)(sums[ci] / cnt), m2 = (float)(sums[Cin + ci] / cnt);
    const float is = invstd[ci];
    const float s2m2 = is * is * m2;
    alpha[ci] = -s2m2;
    beta[ci] = s2m2 * mean[ci] - is * m1;
}

// coef[c] = (A, B, K, 0): dP = A g + B P + K is the BatchNorm backward with the batch means folded in
__global__ void __launch_bounds__(128)
bn_backward_coef_kernel(const float* __restrict__ mean, const float* __restrict__ invstd,
                        const float* __restrict__ mdy, const float* __restrict__ mdyx, float* __restrict__ coef,
                        int C) {
    const int c = blockIdx.x * 128 + threadIdx.x;
    if (c >= C) return;
    const float is = invstd[c];
    const float s2m = is * is * mdyx[c];
    coef[4 * c] = is;
    coef[4 * c + 1] = -s2m;
    coef[4 * c + 2] = s2m * mean[c] - is * mdy[c];
    coef[4 * c + 3] = 0.f;
}

}  // namespace

extern "C" int afd_bn_fold_forward(const float* w, const float* b, const float* mean, const float* invstd,
                                   float* wf, float* bf, int C, int Cin, afd_stream_t stream) {
    if (!w || !mean || !invstd || !wf || !bf || C < 1 || Cin < 1) return afd::fail(AFD_ERR_ARG, "bn fold: bad argument");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 8.0 * C * Cin, AFD_STREAM);
    hipLaunchKernelGGL(bn_fold_forward_kernel, dim3(C), dim3(64), 0, AFD_STREAM, w, b, mean, invstd, wf, bf, Cin);
    return afd::check_launch("bn_fold_forward_kernel");
}

extern "C" int afd_bn_fold_backward_weights(const float* G, const float* db, const float* w, const float* mean,
                                            const float* invstd, float* dw, double* sums, int C, int Cin,
                                            afd_stream_t stream) {
    if (!G || !db || !w || !mean || !invstd || !dw || !sums || C < 1 || Cin < 1)
        return afd::fail(AFD_ERR_ARG, "bn fold backward: bad argument");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 12.0 * C * Cin, AFD_STREAM);
    hipLaunchKernelGGL(bn_fold_backward_weights_kernel, dim3((Cin + 127) / 128), dim3(128), 0, AFD_STREAM, G, db, w,
                       mean, invstd, dw, sums, C, Cin);
    return afd::check_launch("bn_fold_backward_weights_kernel");
}

extern "C" int afd_bn_fold_backward_affine(const double* sums, double count, const double* count_dev,
                                           const float* mean, const float* invstd, float* alpha, float* beta,
                                           int Cin, afd_stream_t stream) {
    if (!sums || !mean || !invstd || !alpha || !beta || Cin < 1 || (count < 0.0 && !count_dev))
        return afd::fail(AFD_ERR_ARG, "bn fold backward affine: bad argument");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 32.0 * Cin, AFD_STREAM);
    hipLaunchKernelGGL(bn_fold_backward_affine_kernel, dim3((Cin + 127) / 128), dim3(128), 0, AFD_STREAM, sums, count,
                       count_dev, mean, invstd, alpha, beta, Cin);
    return afd::check_launch("bn_fold_backward_affine_kernel");
}

extern "C" int afd_bn_backward_coef(const float* mean, const float* invstd, const float* mdy, const float* mdyx,
                                    float* coef, int C, afd_stream_t stream) {
    if (!mean || !invstd || !mdy || !mdyx || !coef || C < 1) return afd::fail(AFD_ERR_ARG, "bn backward coef: bad argument");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 32.0 * C, AFD_STREAM);
    hipLaunchKernelGGL(bn_backward_coef_kernel, dim3((C + 127) / 128), dim3(128), 0, AFD_STREAM, mean, invstd, mdy,
                       mdyx, coef, C);
    return afd::check_launch("bn_backward_coef_kernel");
}

// few channels (the dilated stack's BatchNorms have 3): a plane is shared by workgroups of at least 16 k elements until
// about six of them sit on a CU (384 workgroups on 256 CUs ran at 2.4 TB/s: 84 -> 64 us); more than that and the
// double atomics on the 2 C addresses take over (3 072 workgroups: 95 us)
// small planes (the level-8 dilated stack: 13 channels of 64 x 32): a workgroup takes the planes of several images so that
// it sums at least 16 k elements -- one workgroup per 2 048-element plane was 1 664 workgroups ending in 3 328 double atomics
// on 26 addresses (26 us for 1.7 MB; round 6)
static int small_plane_rows(int gy, int N, int HW) {
    while (gy > 1 && (long)HW * ((N + gy - 1) / gy) < 16384) gy = (gy + 1) / 2;
    return gy;
}

static int plane_splits(int C, int gy, int HW) {
    int gz = 1;
    while ((long)C * gy * gz < 1024 && gz < 16 && HW / (gz * 2) >= 16384) gz *= 2;
    return gz;
}

extern "C" int afd_bn_stats(const float* x, const float* slope, double* sums, int N, int C, int HW,
                            afd_stream_t stream) {
    if (!x || !sums || N < 1 || C < 1 || HW < 1) return afd::fail(AFD_ERR_ARG, "bn stats: bad argument");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 4.0 * (double)N * C * HW, AFD_STREAM);
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, AFD_STREAM);
    if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "bn stats: memset: %s", hipGetErrorString(e));
    int gy = N;
    while ((long)gy * C > 4096 && gy > 1) gy = (gy + 1) / 2;
    gy = small_plane_rows(gy, N, HW);
    const int gz = plane_splits(C, gy, HW);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, gy, gz), dim3(kT), 0, AFD_STREAM, x, slope, sums, N, C, HW);
    return afd::check_launch("bn_stats_kernel");
}

extern "C" int afd_bn_apply_forward(const float* x, const float* slope, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta,
                                    float* y, int N, int C, int HW, afd_stream_t stream) {
    if (!x || !mean || !invstd || !y) return afd::fail(AFD_ERR_ARG, "bn apply: null pointer");
    if ((long)N * C > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "bn apply: N*C > 65535");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 8.0 * (double)N * C * HW, AFD_STREAM);
    unsigned bx = (unsigned)(HW / 16384);  // at least 16k elements per workgroup
    if (bx < 1) bx = 1;
    if (bx > 16) bx = 16;
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(bx, N * C), dim3(kT), 0, AFD_STREAM, x,
                       slope, mean, invstd, gamma, beta, y, C, HW);
    return afd::check_launch("bn_apply_fwd_kernel");
}

extern "C" int afd_bn_backward_stats(const float* x, const float* slope, const float* dy,
                                     const float* mean, const float* invstd, double* sums, int N,
                                     int C, int HW, afd_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !sums) return afd::fail(AFD_ERR_ARG, "bn bwd stats: null pointer");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 8.0 * (double)N * C * HW, AFD_STREAM);
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, AFD_STREAM);
    if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "bn bwd stats: memset: %s", hipGetErrorString(e));
    int gy = N;
    while ((long)gy * C > 4096 && gy > 1) gy = (gy + 1) / 2;
    gy = small_plane_rows(gy, N, HW);
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(C, gy, plane_splits(C, gy, HW)), dim3(kT), 0, AFD_STREAM, x, slope, dy, mean,
                       invstd, sums, N, C, HW);
    return afd::check_launch("bn_bwd_stats_kernel");
}

extern "C" int afd_bn_backward_apply(const float* x, const float* slope, const float* dy,
                                     const float* mean, const float* invstd, const float* gamma,
                                     const float* mean_dy, const float* mean_dy_xhat, float* dx,
                                     float* dslope, int N, int C, int HW, afd_stream_t stream) {
    return afd_bn_backward_apply_sums(x, slope, dy, mean, invstd, gamma, mean_dy, mean_dy_xhat, dx, dslope, nullptr, N,
                                      C, HW, stream);
}

extern "C" int afd_bn_backward_apply_sums(const float* x, const float* slope, const float* dy,
                                          const float* mean, const float* invstd, const float* gamma,
                                          const float* mean_dy, const float* mean_dy_xhat, float* dx,
                                          float* dslope, double* dx_sums, int N, int C, int HW,
                                          afd_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !mean_dy || !mean_dy_xhat || !dx || (slope && !dslope))
        return afd::fail(AFD_ERR_ARG, "bn bwd apply: null pointer");
    if ((long)N * C > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "bn bwd apply: N*C > 65535");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 12.0 * (double)N * C * HW, AFD_STREAM);
    // at least 16k elements per workgroup (each ends in one float atomic on dslope; short
    // workgroups also pay the block reduction per few hundred elements)
    unsigned bx = (unsigned)(HW / 16384);
    if (bx < 1) bx = 1;
    if (bx > 16) bx = 16;
    int G = 1;  // images per workgroup: small planes share one (at least 16 k elements, at most 16 images)
    while (G < 16 && G < N && (long)HW * G < 16384) G *= 2;
    const int ngroups = (N + G - 1) / G;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bx, ngroups * C), dim3(kT), 0, AFD_STREAM, x,
                       slope, dy, mean, invstd, gamma, mean_dy, mean_dy_xhat, dx, dslope, C, HW, dx_sums, N, G);
    return afd::check_launch("bn_bwd_apply_kernel");
}

extern "C" int afd_dropout_permute(const float* x, float* y, int B, int C, int H, int W, float p,
                                   uint64_t seed, int inverse, afd_stream_t stream) {
    if (!x || !y || B < 1 || p < 0.f || p >= 1.f) return afd::fail(AFD_ERR_ARG, "dropout_permute: bad argument");
    if (B > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "dropout_permute: batch > 65535");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 8.0 * (double)B * C * H * W, AFD_STREAM);
    if ((W & 3) == 0 && (size_t)C * H * W < 0x7fffffffULL && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0) {
        hipLaunchKernelGGL(dropout_permute4_kernel, dim3(grid1d((size_t)C * H * (W / 4), 64), B), dim3(kT), 0, AFD_STREAM, x, y,
                           C, H, W / 4, p, (unsigned long long)seed, inverse);
        return afd::check_launch("dropout_permute4_kernel");
    }
    hipLaunchKernelGGL(dropout_permute_kernel, dim3(grid1d((size_t)C * H * W, 256), B), dim3(kT), 0,
                       AFD_STREAM, x, y, C, H, W, p, (unsigned long long)seed, inverse);
    return afd::check_launch("dropout_permute_kernel");
}

extern "C" int afd_linear_mean_forward(const float* x, const float* w, const float* bias, float* y,
                                       int B, int TD, int F, int O, afd_stream_t stream) {
    if (!x || !w || !bias || !y) return afd::fail(AFD_ERR_ARG, "linear fwd: null pointer");
    if (O != 2) return afd::fail(AFD_ERR_UNSUPPORTED, "linear: only 2 classes (reference models.py:297)");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 4.0 * ((double)B * TD * F + (double)F * O), AFD_STREAM);
    hipLaunchKernelGGL(linear_mean_fwd_kernel<2>, dim3(B), dim3(kLinT), 0, AFD_STREAM, x, w, bias, y, TD, F);
    return afd::check_launch("linear_mean_fwd_kernel");
}

extern "C" int afd_linear_mean_backward(const float* x, const float* w, const float* dy, float* dx,
                                        float* dw, float* db, int B, int TD, int F, int O,
                                        afd_stream_t stream) {
    if (!x || !w || !dy || !dx || !dw || !db) return afd::fail(AFD_ERR_ARG, "linear bwd: null pointer");
    if (O != 2) return afd::fail(AFD_ERR_UNSUPPORTED, "linear: only 2 classes");
    if (B > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "linear: batch > 65535");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 8.0 * (double)B * TD * F, AFD_STREAM);
    hipLaunchKernelGGL(linear_mean_bwd_x_kernel<2>, dim3(grid1d(F, 64), B), dim3(kT), 0, AFD_STREAM,
                       dy, w, dx, TD, F);
    hipLaunchKernelGGL(linear_mean_bwd_w_kernel<2>, dim3((F + 63) / 64), dim3(64 * kLinGroups), 0, AFD_STREAM, x,
                       dy, dw, db, B, TD, F);
    return afd::check_launch("linear_mean_bwd kernels");
}

extern "C" int afd_cross_entropy(const float* logits, const int64_t* labels, float* loss,
                                 float* dlogits, float* correct, int B, int O, afd_stream_t stream) {
    if (!logits || !labels || !loss || B < 1 || O < 2) return afd::fail(AFD_ERR_ARG, "cross entropy: bad argument");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 16.0 * B * O, AFD_STREAM);
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(kT), 0, AFD_STREAM, logits,
                       reinterpret_cast<const long long*>(labels), loss, dlogits, correct, B, O);
    return afd::check_launch("ce_kernel");
}

// Many small tensors -> slices of one arena in ONE launch (the optimizer's gradient arena: autograd hands every
// parameter its own gradient tensor; adding each into the arena was ~50 four-microsecond launches per step).  The table
// travels as a kernel argument; a null source zero-fills its slice (a parameter that took no gradient).
constexpr int kGatherMax = 96;
struct GatherTable {
    const float* src[kGatherMax];
    long off[kGatherMax];
    int count[kGatherMax];
};

__global__ void __launch_bounds__(kT)
multi_gather_kernel(const GatherTable t, float* __restrict__ dst) {
    const int k = blockIdx.y;
    const float* __restrict__ s = t.src[k];
    float* __restrict__ d = dst + t.off[k];
    const int n = t.count[k];
    for (int i = blockIdx.x * kT + threadIdx.x; i < n; i += gridDim.x * kT) d[i] = s ? s[i] : 0.f;
}

extern "C" int afd_multi_gather(const float* const* srcs, const long* offsets, const long* counts, int n, float* dst,
                                afd_stream_t stream) {
    if (n < 0 || (n > 0 && (!srcs || !offsets || !counts || !dst))) return afd::fail(AFD_ERR_ARG, "multi gather: bad argument");
    double total = 0.0;
    for (int k = 0; k < n; ++k) total += counts[k] > 0 ? (double)counts[k] : 0.0;
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 8.0 * total, AFD_STREAM);
    for (int base = 0; base < n; base += kGatherMax) {
        GatherTable t{};
        const int m = n - base < kGatherMax ? n - base : kGatherMax;
        long big = 1;
        for (int k = 0; k < m; ++k) {
            if (counts[base + k] < 0 || counts[base + k] > 0x7fffffffL || offsets[base + k] < 0)
                return afd::fail(AFD_ERR_ARG, "multi gather: tensor %d has a bad size or offset", base + k);
            t.src[k] = srcs[base + k];
            t.off[k] = offsets[base + k];
            t.count[k] = (int)counts[base + k];
            if (counts[base + k] > big) big = counts[base + k];
        }
        long bx = (big + kT * 4 - 1) / (kT * 4);
        if (bx > 64) bx = 64;
        hipLaunchKernelGGL(multi_gather_kernel, dim3((unsigned)bx, (unsigned)m), dim3(kT), 0, AFD_STREAM, t, dst);
    }
    return afd::check_launch("multi_gather_kernel");
}

extern "C" int afd_adam_step(float* params, const float* grads, float* m, float* v, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step,
                             float grad_scale, afd_stream_t stream) {
    if (!params || !grads || !m || !v || step < 1) return afd::fail(AFD_ERR_ARG, "adam: bad argument");
    afd::ScopedBytes timing(AFD_K_ELEMENTWISE, 28.0 * (double)n, AFD_STREAM);
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n)), dim3(kT), 0, AFD_STREAM, params, grads, m, v, n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
    return afd::check_launch("adam_kernel");
}
