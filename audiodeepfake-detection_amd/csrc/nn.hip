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
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < n; i += (size_t)gridDim.x * kT) {
        float g = dy[i];
        if (p > 0.f) g = uniform01(seed, i) >= p ? g * scale : 0.f;
        const float zz = z[i];
        dz[i] = zz > 0.f ? g : a * g;
        if (zz <= 0.f) ds += g * zz;
    }
    block_sum3(ds, u0, u1);
    if (threadIdx.x == 0 && ds != 0.f) atomicAdd(dslope, ds);
}

// ------------------------------- PReLU + MaxPool 2x2 -----------------------------------
// z [NC][H][W] -> u [NC][Hp][Wp] (= max of prelu over the window, first max wins like
// torch), idx = argmax position 0..3 (dy*2+dx)
__global__ void prelu_pool_fwd_kernel(const float* __restrict__ z, const float* __restrict__ slope,
                                      float* __restrict__ u, unsigned char* __restrict__ idx, int H,
                                      int W, int Hp, int Wp) {
    const float a = slope ? slope[0] : 1.f;
    const size_t plane = blockIdx.y;
    const float* zp = z + plane * (size_t)H * W;
    const size_t obase = plane * (size_t)Hp * Wp;
    const int total = Hp * Wp;
    for (int i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
        const int py = i / Wp, px = i - py * Wp;
        const float* r0 = zp + (size_t)(2 * py) * W + 2 * px;
        const float2 t = *reinterpret_cast<const float2*>(r0);
        const float2 b = *reinterpret_cast<const float2*>(r0 + W);
        float best = slope ? prelu(t.x, a) : t.x;
        int bi = 0;
        float v = slope ? prelu(t.y, a) : t.y;
        if (v > best) { best = v; bi = 1; }
        v = slope ? prelu(b.x, a) : b.x;
        if (v > best) { best = v; bi = 2; }
        v = slope ? prelu(b.y, a) : b.y;
        if (v > best) { best = v; bi = 3; }
        u[obase + i] = best;
        idx[obase + i] = (unsigned char)bi;
    }
}

// unaligned-safe variant (odd W or odd plane offsets): scalar loads
__global__ void prelu_pool_fwd_kernel_s(const float* __restrict__ z, const float* __restrict__ slope,
                                        float* __restrict__ u, unsigned char* __restrict__ idx,
                                        int H, int W, int Hp, int Wp) {
    const float a = slope ? slope[0] : 1.f;
    const size_t plane = blockIdx.y;
    const float* zp = z + plane * (size_t)H * W;
    const size_t obase = plane * (size_t)Hp * Wp;
    const int total = Hp * Wp;
    for (int i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
        const int py = i / Wp, px = i - py * Wp;
        const float* r0 = zp + (size_t)(2 * py) * W + 2 * px;
        float best = slope ? prelu(r0[0], a) : r0[0];
        int bi = 0;
        float v = slope ? prelu(r0[1], a) : r0[1];
        if (v > best) { best = v; bi = 1; }
        v = slope ? prelu(r0[W], a) : r0[W];
        if (v > best) { best = v; bi = 2; }
        v = slope ? prelu(r0[W + 1], a) : r0[W + 1];
        if (v > best) { best = v; bi = 3; }
        u[obase + i] = best;
        idx[obase + i] = (unsigned char)bi;
    }
}

// dz [NC][H][W] fully written (zeros outside the argmax, incl. the odd last row/col)
__global__ void prelu_pool_bwd_kernel(const float* __restrict__ z, const float* __restrict__ slope,
                                      const unsigned char* __restrict__ idx,
                                      const float* __restrict__ du, float* __restrict__ dz,
                                      float* __restrict__ dslope, int H, int W, int Hp, int Wp) {
    const float a = slope ? slope[0] : 1.f;
    const size_t plane = blockIdx.y;
    const float* zp = z + plane * (size_t)H * W;
    float* dzp = dz + plane * (size_t)H * W;
    const size_t pbase = plane * (size_t)Hp * Wp;
    const int total = H * W;
    float ds = 0.f, u0 = 0.f, u1 = 0.f;
    for (int i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
        const int y = i / W, x = i - y * W;
        const int py = y >> 1, px = x >> 1;
        float g = 0.f;
        if (py < Hp && px < Wp) {
            const int pos = ((y & 1) << 1) | (x & 1);
            if (idx[pbase + (size_t)py * Wp + px] == pos) {
                g = du[pbase + (size_t)py * Wp + px];
                if (slope) {
                    const float zz = zp[i];
                    if (zz <= 0.f) {
                        ds += g * zz;
                        g *= a;
                    }
                }
            }
        }
        dzp[i] = g;
    }
    if (slope) {
        block_sum3(ds, u0, u1);
        if (threadIdx.x == 0 && ds != 0.f) atomicAdd(dslope, ds);
    }
}

// ------------------------------- BatchNorm ---------------------------------------------
// per-channel sum and sum of squares of (optionally PReLU'd) x [N][C][HW]; double atomics
__global__ void bn_stats_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                double* __restrict__ sums, int N, int C, int HW) {
    const int c = blockIdx.x;
    const float a = slope ? slope[0] : 1.f;
    float s = 0.f, q = 0.f, u = 0.f;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const float* p = x + ((size_t)n * C + c) * HW;
        float ls = 0.f, lq = 0.f;
        for (int i = threadIdx.x; i < HW; i += kT) {
            float v = p[i];
            if (slope) v = prelu(v, a);
            ls += v;
            lq += v * v;
        }
        s += ls;
        q += lq;
    }
    block_sum3(s, q, u);
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
    const float a = slope ? slope[0] : 1.f;
    const float m = mean[c];
    const float sc = invstd[c] * (gamma ? gamma[c] : 1.f);
    const float sh = beta ? beta[c] : 0.f;
    const float* xp = x + plane * (size_t)HW;
    float* yp = y + plane * (size_t)HW;
    for (int i = blockIdx.x * kT + threadIdx.x; i < HW; i += gridDim.x * kT) {
        float v = xp[i];
        if (slope) v = prelu(v, a);
        yp[i] = (v - m) * sc + sh;
    }
}

// per-channel sum(dy), sum(dy * xhat)
__global__ void bn_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ slope,
                                    const float* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, double* __restrict__ sums,
                                    int N, int C, int HW) {
    const int c = blockIdx.x;
    const float a = slope ? slope[0] : 1.f;
    const float m = mean[c], is = invstd[c];
    float s = 0.f, q = 0.f, u = 0.f;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const size_t off = ((size_t)n * C + c) * HW;
        float ls = 0.f, lq = 0.f;
        for (int i = threadIdx.x; i < HW; i += kT) {
            float v = x[off + i];
            if (slope) v = prelu(v, a);
            const float g = dy[off + i];
            ls += g;
            lq += g * (v - m) * is;
        }
        s += ls;
        q += lq;
    }
    block_sum3(s, q, u);
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
                                    int HW) {
    const size_t plane = blockIdx.y;
    const int c = (int)(plane % C);
    const float a = slope ? slope[0] : 1.f;
    const float m = mean[c], is = invstd[c];
    const float gs = is * (gamma ? gamma[c] : 1.f);
    const float k0 = mdy[c], k1 = mdyx[c];
    const size_t off = plane * (size_t)HW;
    float ds = 0.f, u0 = 0.f, u1 = 0.f;
    for (int i = blockIdx.x * kT + threadIdx.x; i < HW; i += gridDim.x * kT) {
        const float zz = x[off + i];
        const float v = slope ? prelu(zz, a) : zz;
        const float xh = (v - m) * is;
        float g = gs * (dy[off + i] - k0 - xh * k1);
        if (slope && zz <= 0.f) {
            ds += g * zz;
            g *= a;
        }
        dx[off + i] = g;
    }
    if (slope) {
        block_sum3(ds, u0, u1);
        if (threadIdx.x == 0 && ds != 0.f) atomicAdd(dslope, ds);
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

// ------------------------------- Linear + mean(1) --------------------------------------
// x [B][TD][F], w [O][F], bias [O] -> y [B][O] = bias + (1/TD) sum_t x[b,t,:] . w[o,:]
template <int O>
__global__ void linear_mean_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                       const float* __restrict__ bias, float* __restrict__ y, int TD,
                                       int F) {
    const int b = blockIdx.x;
    const float* xb = x + (size_t)b * TD * F;
    float acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = 0.f;
    for (int f = threadIdx.x; f < F; f += kT) {
        float s = 0.f;
        for (int t = 0; t < TD; ++t) s += xb[(size_t)t * F + f];
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = fmaf(s, w[(size_t)o * F + f], acc[o]);
    }
    __shared__ float red[O][kT / 64];
#pragma unroll
    for (int o = 0; o < O; ++o) {
        const float v = wave_sum(acc[o]);
        if ((threadIdx.x & 63) == 0) red[o][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < O) {
        float s = 0.f;
        for (int k = 0; k < kT / 64; ++k) s += red[threadIdx.x][k];
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
template <int O>
__global__ void linear_mean_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                         float* __restrict__ dw, float* __restrict__ db, int B,
                                         int TD, int F) {
    for (int f = blockIdx.x * kT + threadIdx.x; f < F; f += gridDim.x * kT) {
        float acc[O];
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] = 0.f;
        for (int b = 0; b < B; ++b) {
            float s = 0.f;
            for (int t = 0; t < TD; ++t) s += x[((size_t)b * TD + t) * F + f];
#pragma unroll
            for (int o = 0; o < O; ++o) acc[o] = fmaf(s, dy[(size_t)b * O + o], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < O; ++o) dw[(size_t)o * F + f] = acc[o] / (float)TD;
    }
    if (blockIdx.x == 0 && threadIdx.x < O) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += dy[(size_t)b * O + threadIdx.x];
        db[threadIdx.x] = s;
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

extern "C" int afd_normalize_forward(const float* x, float* y, size_t n, float mean, float std,
                                     afd_stream_t stream) {
    if (!x || !y || std == 0.f) return afd::fail(AFD_ERR_ARG, "normalize: bad argument");
    hipLaunchKernelGGL(normalize_kernel, dim3(grid1d(n)), dim3(kT), 0, AFD_STREAM, x, y, n, mean, std);
    return afd::check_launch("normalize_kernel");
}

extern "C" int afd_transpose_last2(const float* x, float* y, int planes, int R, int C,
                                   afd_stream_t stream) {
    if (!x || !y || planes < 1 || R < 1 || C < 1) return afd::fail(AFD_ERR_ARG, "transpose: bad argument");
    if (planes > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "transpose: too many planes");
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, planes), dim3(kT), 0,
                       AFD_STREAM, x, y, R, C);
    return afd::check_launch("transpose_kernel");
}

extern "C" int afd_prelu_dropout_forward(const float* z, const float* slope, float* y, size_t n,
                                         float p, uint64_t seed, afd_stream_t stream) {
    if (!z || !slope || !y || p < 0.f || p >= 1.f) return afd::fail(AFD_ERR_ARG, "prelu fwd: bad argument");
    hipLaunchKernelGGL(prelu_dropout_fwd_kernel, dim3(grid1d(n)), dim3(kT), 0, AFD_STREAM, z, slope,
                       y, n, p, (unsigned long long)seed);
    return afd::check_launch("prelu_dropout_fwd_kernel");
}

extern "C" int afd_prelu_dropout_backward(const float* z, const float* slope, const float* dy,
                                          float* dz, float* dslope, size_t n, float p,
                                          uint64_t seed, afd_stream_t stream) {
    if (!z || !slope || !dy || !dz || !dslope) return afd::fail(AFD_ERR_ARG, "prelu bwd: null pointer");
    hipLaunchKernelGGL(prelu_dropout_bwd_kernel, dim3(grid1d(n, 1024)), dim3(kT), 0, AFD_STREAM, z,
                       slope, dy, dz, dslope, n, p, (unsigned long long)seed);
    return afd::check_launch("prelu_dropout_bwd_kernel");
}

extern "C" int afd_prelu_pool_forward(const float* z, const float* slope, float* u, uint8_t* idx,
                                      int NC, int H, int W, afd_stream_t stream) {
    if (!z || !u || !idx || NC < 1 || H < 2 || W < 2) return afd::fail(AFD_ERR_ARG, "pool fwd: bad argument");
    if (NC > 65535 * 16) return afd::fail(AFD_ERR_UNSUPPORTED, "pool: too many planes");
    const int Hp = H / 2, Wp = W / 2;
    const unsigned gx = grid1d((size_t)Hp * Wp, 256);
    // float2 loads need 8-byte aligned rows: even W and an even plane size
    const bool aligned = (W % 2 == 0) && (((size_t)H * W) % 2 == 0) && (((uintptr_t)z & 7) == 0);
    for (int p0 = 0; p0 < NC; p0 += 65535) {
        const int np = NC - p0 < 65535 ? NC - p0 : 65535;
        const float* zp = z + (size_t)p0 * H * W;
        float* up = u + (size_t)p0 * Hp * Wp;
        uint8_t* ip = idx + (size_t)p0 * Hp * Wp;
        if (aligned)
            hipLaunchKernelGGL(prelu_pool_fwd_kernel, dim3(gx, np), dim3(kT), 0, AFD_STREAM, zp, slope, up, ip, H, W, Hp, Wp);
        else
            hipLaunchKernelGGL(prelu_pool_fwd_kernel_s, dim3(gx, np), dim3(kT), 0, AFD_STREAM, zp, slope, up, ip, H, W, Hp, Wp);
    }
    return afd::check_launch("prelu_pool_fwd_kernel");
}

extern "C" int afd_prelu_pool_backward(const float* z, const float* slope, const uint8_t* idx,
                                       const float* du, float* dz, float* dslope, int NC, int H,
                                       int W, afd_stream_t stream) {
    if (!z || !idx || !du || !dz || (slope && !dslope)) return afd::fail(AFD_ERR_ARG, "pool bwd: null pointer");
    const int Hp = H / 2, Wp = W / 2;
    const unsigned gx = grid1d((size_t)H * W, 256);
    for (int p0 = 0; p0 < NC; p0 += 65535) {
        const int np = NC - p0 < 65535 ? NC - p0 : 65535;
        hipLaunchKernelGGL(prelu_pool_bwd_kernel, dim3(gx, np), dim3(kT), 0, AFD_STREAM,
                           z + (size_t)p0 * H * W, slope, idx + (size_t)p0 * Hp * Wp,
                           du + (size_t)p0 * Hp * Wp, dz + (size_t)p0 * H * W, dslope, H, W, Hp, Wp);
    }
    return afd::check_launch("prelu_pool_bwd_kernel");
}

extern "C" int afd_bn_stats(const float* x, const float* slope, double* sums, int N, int C, int HW,
                            afd_stream_t stream) {
    if (!x || !sums || N < 1 || C < 1 || HW < 1) return afd::fail(AFD_ERR_ARG, "bn stats: bad argument");
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, AFD_STREAM);
    if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "bn stats: memset: %s", hipGetErrorString(e));
    int gy = N;
    while ((long)gy * C > 4096 && gy > 1) gy = (gy + 1) / 2;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C, gy), dim3(kT), 0, AFD_STREAM, x, slope, sums, N, C, HW);
    return afd::check_launch("bn_stats_kernel");
}

extern "C" int afd_bn_apply_forward(const float* x, const float* slope, const float* mean,
                                    const float* invstd, const float* gamma, const float* beta,
                                    float* y, int N, int C, int HW, afd_stream_t stream) {
    if (!x || !mean || !invstd || !y) return afd::fail(AFD_ERR_ARG, "bn apply: null pointer");
    if ((long)N * C > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "bn apply: N*C > 65535");
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(grid1d(HW, 64), N * C), dim3(kT), 0, AFD_STREAM, x,
                       slope, mean, invstd, gamma, beta, y, C, HW);
    return afd::check_launch("bn_apply_fwd_kernel");
}

extern "C" int afd_bn_backward_stats(const float* x, const float* slope, const float* dy,
                                     const float* mean, const float* invstd, double* sums, int N,
                                     int C, int HW, afd_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !sums) return afd::fail(AFD_ERR_ARG, "bn bwd stats: null pointer");
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, AFD_STREAM);
    if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "bn bwd stats: memset: %s", hipGetErrorString(e));
    int gy = N;
    while ((long)gy * C > 4096 && gy > 1) gy = (gy + 1) / 2;
    hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(C, gy), dim3(kT), 0, AFD_STREAM, x, slope, dy, mean,
                       invstd, sums, N, C, HW);
    return afd::check_launch("bn_bwd_stats_kernel");
}

extern "C" int afd_bn_backward_apply(const float* x, const float* slope, const float* dy,
                                     const float* mean, const float* invstd, const float* gamma,
                                     const float* mean_dy, const float* mean_dy_xhat, float* dx,
                                     float* dslope, int N, int C, int HW, afd_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !mean_dy || !mean_dy_xhat || !dx || (slope && !dslope))
        return afd::fail(AFD_ERR_ARG, "bn bwd apply: null pointer");
    if ((long)N * C > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "bn bwd apply: N*C > 65535");
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid1d(HW, 64), N * C), dim3(kT), 0, AFD_STREAM, x,
                       slope, dy, mean, invstd, gamma, mean_dy, mean_dy_xhat, dx, dslope, C, HW);
    return afd::check_launch("bn_bwd_apply_kernel");
}

extern "C" int afd_dropout_permute(const float* x, float* y, int B, int C, int H, int W, float p,
                                   uint64_t seed, int inverse, afd_stream_t stream) {
    if (!x || !y || B < 1 || p < 0.f || p >= 1.f) return afd::fail(AFD_ERR_ARG, "dropout_permute: bad argument");
    if (B > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "dropout_permute: batch > 65535");
    hipLaunchKernelGGL(dropout_permute_kernel, dim3(grid1d((size_t)C * H * W, 256), B), dim3(kT), 0,
                       AFD_STREAM, x, y, C, H, W, p, (unsigned long long)seed, inverse);
    return afd::check_launch("dropout_permute_kernel");
}

extern "C" int afd_linear_mean_forward(const float* x, const float* w, const float* bias, float* y,
                                       int B, int TD, int F, int O, afd_stream_t stream) {
    if (!x || !w || !bias || !y) return afd::fail(AFD_ERR_ARG, "linear fwd: null pointer");
    if (O != 2) return afd::fail(AFD_ERR_UNSUPPORTED, "linear: only 2 classes (reference models.py:297)");
    hipLaunchKernelGGL(linear_mean_fwd_kernel<2>, dim3(B), dim3(kT), 0, AFD_STREAM, x, w, bias, y, TD, F);
    return afd::check_launch("linear_mean_fwd_kernel");
}

extern "C" int afd_linear_mean_backward(const float* x, const float* w, const float* dy, float* dx,
                                        float* dw, float* db, int B, int TD, int F, int O,
                                        afd_stream_t stream) {
    if (!x || !w || !dy || !dx || !dw || !db) return afd::fail(AFD_ERR_ARG, "linear bwd: null pointer");
    if (O != 2) return afd::fail(AFD_ERR_UNSUPPORTED, "linear: only 2 classes");
    if (B > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "linear: batch > 65535");
    hipLaunchKernelGGL(linear_mean_bwd_x_kernel<2>, dim3(grid1d(F, 64), B), dim3(kT), 0, AFD_STREAM,
                       dy, w, dx, TD, F);
    hipLaunchKernelGGL(linear_mean_bwd_w_kernel<2>, dim3(grid1d(F, 1024)), dim3(kT), 0, AFD_STREAM, x,
                       dy, dw, db, B, TD, F);
    return afd::check_launch("linear_mean_bwd kernels");
}

extern "C" int afd_cross_entropy(const float* logits, const int64_t* labels, float* loss,
                                 float* dlogits, float* correct, int B, int O, afd_stream_t stream) {
    if (!logits || !labels || !loss || B < 1 || O < 2) return afd::fail(AFD_ERR_ARG, "cross entropy: bad argument");
    hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(kT), 0, AFD_STREAM, logits,
                       reinterpret_cast<const long long*>(labels), loss, dlogits, correct, B, O);
    return afd::check_launch("ce_kernel");
}

extern "C" int afd_adam_step(float* params, const float* grads, float* m, float* v, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step,
                             float grad_scale, afd_stream_t stream) {
    if (!params || !grads || !m || !v || step < 1) return afd::fail(AFD_ERR_ARG, "adam: bad argument");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n)), dim3(kT), 0, AFD_STREAM, params, grads, m, v, n,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale);
    return afd::check_launch("adam_kernel");
}
