// Direct (VALU) convolutions for the DCNN's dilated stack when it has very few channels.
//
// Reference models.py:286-300: three Conv2d(time_dim, time_dim, k, padding, dilation) with
// (k, pad, dil) = (3,1,1), (5,2,2), (7,2,4) on [B, time_dim, 64, P/8].  For level-14 packets
// time_dim = 24 // 8 = 3: on the MFMA implicit-GEMM path 3 of 32 output-channel rows are live
// and the three layers cost 21 ms of a 178 ms step.  Here every lane owns one image column and
// ROWS output rows of one residue class modulo the dilation (rows y, y+d, .. share all but one
// of their input rows), keeps ROWS x C accumulators, reads each input sample once from an LDS
// tile (zero-filled halo, no bounds checks in the loop) and feeds it to up to ROWS x C FMAs
// whose weight operand is a scalar register.
//   forward        out[o] = sum in[i] (*) w[o][i]                         (pad)
//   backward-data  the same kernel, weights transposed + flipped           (pad' = (K-1)d - pad)
//   backward-weight  one wave per kernel row ky, lanes over columns, C*C*K accumulators per
//                  lane over a strided tile list, one partial slab per workgroup, summed by a
//                  second kernel (deterministic, no float atomics).
// Bound: fp32 VALU (441 FMA per 3 x 4-byte pixel at C = 3, K = 7); algorithmic flops as for
// the implicit-GEMM path: 2 N C C K K Hout Wout.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

struct DirGeom {
    int N, Hin, Win, Hout, Wout, pad;  // out = in + 2 pad - (K-1) dil
    int tilesX, groups;                // groups = row groups per residue class
};

// uniform-index weight read: the compiler turns these into scalar loads
template <int C, int K, bool FLIP>
__device__ __forceinline__ float wread(const float* __restrict__ w, int o, int i, int ky, int kx) {
    constexpr int KK = K * K;
    return FLIP ? w[(i * C + o) * KK + (K - 1 - ky) * K + (K - 1 - kx)]
                : w[(o * C + i) * KK + ky * K + kx];
}

// Stage in[ci][iy0 + r*DIL][ix0 .. ix0+PC) for r < R into tile[ci][r][PC], zero outside the
// image.  Lanes run along a patch row (coalesced, no per-element index arithmetic: these are
// vector-ALU kernels, every staging instruction is an FMA slot lost); rows are either all taken
// by every thread (PC >= NT: the (channel, row) of a load is a compile-time constant), or, for
// tiles narrower than the workgroup, addressed through a flat element index.  Loads go out in batches of 8 before their LDS stores.
template <int C, int R, int PC, int DIL, int NT>
__device__ __forceinline__ void stage_rows(const float* __restrict__ img, int H, int W, int iy0,
                                           int ix0, float* tile, int tid) {
    constexpr int ROWSALL = C * R;
    const size_t plane = (size_t)H * W;
    if constexpr (PC >= NT) {
        constexpr int CP = (PC + NT - 1) / NT;
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
            const int col = tid + cp * NT;
            const int ix = ix0 + col;
            const bool inx = col < PC && ix >= 0 && ix < W;
            const float* src = img + ix;
#pragma unroll
            for (int r0 = 0; r0 < ROWSALL; r0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = r0 + u;
                    const int ci = row / R, r = row - ci * R;  // compile-time
                    const int iy = iy0 + r * DIL;             // wave-uniform
                    v[u] = (row < ROWSALL && inx && iy >= 0 && iy < H) ? src[(size_t)ci * plane + (size_t)iy * W] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = r0 + u;
                    if (row < ROWSALL && col < PC) tile[row * PC + col] = v[u];
                }
            }
        }
    } else {
        // narrow tiles (the backward-weight kernel): flat element index over the whole tile
        constexpr int total = C * R * PC;
        constexpr int per = (total + NT - 1) / NT;
#pragma unroll
        for (int u0 = 0; u0 < per; u0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + (u0 + u) * NT;
                const int row = e / PC, c = e - row * PC;
                const int ci = row / R, r = row - ci * R;
                const int iy = iy0 + r * DIL, ix = ix0 + c;
                const bool ok = (u0 + u < per) && (e < total) && iy >= 0 && iy < H && ix >= 0 && ix < W;
                v[u] = ok ? img[(size_t)ci * plane + (size_t)iy * W + ix] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = tid + (u0 + u) * NT;
                if (u0 + u < per && e < total) tile[e] = v[u];
            }
        }
    }
}

template <int C, int K, int DIL, int ROWS, bool FLIP>
__global__ void __launch_bounds__(256)
dilconv_direct_kernel(const DirGeom g, const float* __restrict__ in, const float* __restrict__ w,
                      const float* __restrict__ bias, float* __restrict__ out) {
    constexpr int TX = 256;
    constexpr int R = ROWS + K - 1;
    constexpr int PC = TX + (K - 1) * DIL;
    __shared__ float tile[C * R * PC];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TX;
    const int cls = blockIdx.y % DIL, grp = blockIdx.y / DIL;
    const int y0 = cls + DIL * ROWS * grp;
    const int n = blockIdx.z;
    stage_rows<C, R, PC, DIL, TX>(in + (size_t)n * C * g.Hin * g.Win, g.Hin, g.Win, y0 - g.pad,
                                  x0 - g.pad, tile, tid);
    __syncthreads();
    float acc[ROWS][C];
#pragma unroll
    for (int j = 0; j < ROWS; ++j)
#pragma unroll
        for (int o = 0; o < C; ++o) acc[j][o] = bias ? bias[o] : 0.f;
#pragma unroll
    for (int i = 0; i < C; ++i) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float v = tile[(i * R + r) * PC + tid + kx * DIL];
#pragma unroll
                for (int j = 0; j < ROWS; ++j) {
                    const int ky = r - j;
                    if (ky < 0 || ky >= K) continue;
#pragma unroll
                    for (int o = 0; o < C; ++o)
                        acc[j][o] = fmaf(v, wread<C, K, FLIP>(w, o, i, ky, kx), acc[j][o]);
                }
            }
        }
    }
    const int x = x0 + tid;
    if (x >= g.Wout) return;
    const size_t plane = (size_t)g.Hout * g.Wout;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int y = y0 + j * DIL;
        if (y < g.Hout) {
#pragma unroll
            for (int o = 0; o < C; ++o) out[((size_t)n * C + o) * plane + (size_t)y * g.Wout + x] = acc[j][o];
        }
    }
}

// partial[block][C*C*K*K + C]: weight gradient sums of this block's tiles, then the bias sums
template <int C, int K, int DIL, int ROWS>
__global__ void __launch_bounds__(64 * K)
dilconv_wgrad_kernel(const DirGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                     float* __restrict__ partial, int ntiles) {
    constexpr int TX = 64;
    constexpr int NT = 64 * K;
    constexpr int R = ROWS + K - 1;
    constexpr int PC = TX + (K - 1) * DIL;
    __shared__ float xt[C * R * PC];
    __shared__ float dyt[C * ROWS * TX];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ky = tid >> 6;
    float acc[C][C][K];
    float accb[C];
#pragma unroll
    for (int o = 0; o < C; ++o) {
        accb[o] = 0.f;
#pragma unroll
        for (int i = 0; i < C; ++i)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[o][i][kx] = 0.f;
    }
    const int rgs = DIL * g.groups;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int xtile = t % g.tilesX;
        const int rest = t / g.tilesX;
        const int rg = rest % rgs, n = rest / rgs;
        const int cls = rg % DIL, grp = rg / DIL;
        const int y0 = cls + DIL * ROWS * grp;
        const int x0 = xtile * TX;
        __syncthreads();  // the previous tile has been consumed
        stage_rows<C, R, PC, DIL, NT>(x + (size_t)n * C * g.Hin * g.Win, g.Hin, g.Win, y0 - g.pad,
                                      x0 - g.pad, xt, tid);
        stage_rows<C, ROWS, TX, DIL, NT>(dy + (size_t)n * C * g.Hout * g.Wout, g.Hout, g.Wout, y0,
                                         x0, dyt, tid);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            float d[C];
#pragma unroll
            for (int o = 0; o < C; ++o) d[o] = dyt[(o * ROWS + j) * TX + lane];
            if (ky == 0) {
#pragma unroll
                for (int o = 0; o < C; ++o) accb[o] += d[o];
            }
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const float* row = xt + (i * R + j + ky) * PC + lane;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float v = row[kx * DIL];
#pragma unroll
                    for (int o = 0; o < C; ++o) acc[o][i][kx] = fmaf(d[o], v, acc[o][i][kx]);
                }
            }
        }
    }
    // wave sums -> lane 0 -> this block's slab
    constexpr int KK = K * K;
    float* slab = partial + (size_t)blockIdx.x * (C * C * KK + C);
#pragma unroll
    for (int o = 0; o < C; ++o) {
#pragma unroll
        for (int i = 0; i < C; ++i) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                float v = acc[o][i][kx];
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                if (lane == 0) slab[(o * C + i) * KK + ky * K + kx] = v;
            }
        }
    }
    if (ky == 0) {
#pragma unroll
        for (int o = 0; o < C; ++o) {
            float v = accb[o];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) slab[C * C * KK + o] = v;
        }
    }
}

// dw[e] = sum over slabs (fixed order); e >= n_w are the bias sums
__global__ void __launch_bounds__(256)
dilconv_reduce_kernel(const float* __restrict__ partial, int nslabs, int n_w, int stride,
                      float* __restrict__ dw, float* __restrict__ dbias) {
    __shared__ float red[256];
    const int e = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nslabs; b += 256) s += partial[(size_t)b * stride + e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (e < n_w) dw[e] = red[0];
        else if (dbias) dbias[e - n_w] = red[0];
    }
}

// rows per thread: 5 when that wastes no more of the last row group than 4
int pick_rows(int Hout, int dil) {
    const int per = (Hout + dil - 1) / dil;
    const int waste4 = (per + 3) / 4 * 4 - per, waste5 = (per + 4) / 5 * 5 - per;
    return waste5 <= waste4 ? 5 : 4;
}

template <int C, int K, int DIL, int ROWS, bool FLIP>
void launch_direct(const DirGeom& g, const float* in, const float* w, const float* bias, float* out,
                   hipStream_t s) {
    hipLaunchKernelGGL((dilconv_direct_kernel<C, K, DIL, ROWS, FLIP>),
                       dim3(g.tilesX, DIL * g.groups, g.N), dim3(256), 0, s, g, in, w, bias, out);
}

template <int C, int K, int DIL, bool FLIP>
int run_direct(DirGeom g, const float* in, const float* w, const float* bias, float* out,
               hipStream_t s) {
    const int rows = pick_rows(g.Hout, DIL);
    const int per = (g.Hout + DIL - 1) / DIL;
    g.tilesX = (g.Wout + 255) / 256;
    g.groups = (per + rows - 1) / rows;
    if ((long)DIL * g.groups > 65535 || g.N > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: grid too large");
    if (rows == 5) launch_direct<C, K, DIL, 5, FLIP>(g, in, w, bias, out, s);
    else launch_direct<C, K, DIL, 4, FLIP>(g, in, w, bias, out, s);
    return afd::check_launch("dilconv_direct_kernel");
}

constexpr int kWgradBlocks = 1021;  // prime: a count sharing a factor with the tiles per row walks tile columns in lock step

template <int C, int K, int DIL>
int run_wgrad(DirGeom g, const float* x, const float* dy, float* dw, float* dbias, float* partial,
              hipStream_t s) {
    const int rows = pick_rows(g.Hout, DIL);
    const int per = (g.Hout + DIL - 1) / DIL;
    g.tilesX = (g.Wout + 63) / 64;
    g.groups = (per + rows - 1) / rows;
    const long ntiles = (long)g.N * DIL * g.groups * g.tilesX;
    if (ntiles > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv wgrad: too many tiles");
    const int blocks = ntiles < kWgradBlocks ? (int)ntiles : kWgradBlocks;
    if (rows == 5)
        hipLaunchKernelGGL((dilconv_wgrad_kernel<C, K, DIL, 5>), dim3(blocks), dim3(64 * K), 0, s, g, x,
                           dy, partial, (int)ntiles);
    else
        hipLaunchKernelGGL((dilconv_wgrad_kernel<C, K, DIL, 4>), dim3(blocks), dim3(64 * K), 0, s, g, x,
                           dy, partial, (int)ntiles);
    int rc = afd::check_launch("dilconv_wgrad_kernel");
    if (rc) return rc;
    const int n_w = C * C * K * K;
    hipLaunchKernelGGL(dilconv_reduce_kernel, dim3(n_w + C), dim3(256), 0, s, partial, blocks, n_w,
                       n_w + C, dw, dbias);
    return afd::check_launch("dilconv_reduce_kernel");
}

// dispatch over the (K, dil) pairs of the reference's dilated stack and C = 1..4
template <int C>
int dispatch(int mode, int K, int dil, const DirGeom& g, const float* a, const float* b,
             const float* bias, float* out, float* dbias, float* partial, hipStream_t s) {
#define AFD_DIL_CASE(KK, DD)                                                      \
    if (K == KK && dil == DD) {                                                   \
        if (mode == 0) return run_direct<C, KK, DD, false>(g, a, b, bias, out, s); \
        if (mode == 1) return run_direct<C, KK, DD, true>(g, a, b, nullptr, out, s); \
        return run_wgrad<C, KK, DD>(g, a, b, out, dbias, partial, s);             \
    }
    AFD_DIL_CASE(3, 1)
    AFD_DIL_CASE(5, 2)
    AFD_DIL_CASE(7, 4)
#undef AFD_DIL_CASE
    return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: K=%d dil=%d not built", K, dil);
}

int dispatch_c(int C, int mode, int K, int dil, const DirGeom& g, const float* a, const float* b,
               const float* bias, float* out, float* dbias, float* partial, hipStream_t s) {
    switch (C) {
        case 1: return dispatch<1>(mode, K, dil, g, a, b, bias, out, dbias, partial, s);
        case 2: return dispatch<2>(mode, K, dil, g, a, b, bias, out, dbias, partial, s);
        case 3: return dispatch<3>(mode, K, dil, g, a, b, bias, out, dbias, partial, s);
        case 4: return dispatch<4>(mode, K, dil, g, a, b, bias, out, dbias, partial, s);
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: C=%d not built", C);
}

}  // namespace

namespace afd {

bool dilconv_applicable(int Cin, int Cout, int K, int dil) {
    if (getenv("AFD_NO_DIRECT_CONV")) return false;
    if (Cin != Cout || Cin < 1 || Cin > 4) return false;
    return (K == 3 && dil == 1) || (K == 5 && dil == 2) || (K == 7 && dil == 4);
}

size_t dilconv_workspace_bytes(int C, int K) {
    return (size_t)kWgradBlocks * (C * C * K * K + C) * sizeof(float);
}

int dilconv_forward(const float* x, const float* w, const float* bias, float* y, int N, int C, int H,
                    int W, int K, int pad, int dil, hipStream_t s) {
    DirGeom g{};
    g.N = N; g.Hin = H; g.Win = W; g.pad = pad;
    g.Hout = H + 2 * pad - dil * (K - 1);
    g.Wout = W + 2 * pad - dil * (K - 1);
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, 2.0 * N * C * C * K * K * (double)g.Hout * g.Wout, s);
    return dispatch_c(C, 0, K, dil, g, x, w, bias, y, nullptr, nullptr, s);
}

int dilconv_backward_data(const float* dy, const float* w, float* dx, int N, int C, int H, int W,
                          int K, int pad, int dil, hipStream_t s) {
    DirGeom g{};
    g.N = N;
    g.Hin = H + 2 * pad - dil * (K - 1);
    g.Win = W + 2 * pad - dil * (K - 1);
    g.pad = dil * (K - 1) - pad;
    g.Hout = H;
    g.Wout = W;
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, 2.0 * N * C * C * K * K * (double)g.Hin * g.Win, s);
    return dispatch_c(C, 1, K, dil, g, dy, w, nullptr, dx, nullptr, nullptr, s);
}

int dilconv_backward_weight(const float* x, const float* dy, float* dw, float* dbias, int N, int C,
                            int H, int W, int K, int pad, int dil, void* ws, size_t ws_bytes,
                            hipStream_t s) {
    if (!ws || ws_bytes < dilconv_workspace_bytes(C, K))
        return afd::fail(AFD_ERR_WORKSPACE, "dilconv wgrad: workspace too small");
    DirGeom g{};
    g.N = N; g.Hin = H; g.Win = W; g.pad = pad;
    g.Hout = H + 2 * pad - dil * (K - 1);
    g.Wout = W + 2 * pad - dil * (K - 1);
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, 2.0 * N * C * C * K * K * (double)g.Hout * g.Wout, s);
    return dispatch_c(C, 2, K, dil, g, x, dy, nullptr, dw, dbias, static_cast<float*>(ws), s);
}

}  // namespace afd
