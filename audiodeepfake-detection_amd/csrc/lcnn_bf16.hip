// bf16 matrix-core path of the LCNN evaluation forward (BASELINE.json configs[4]: "STFT + LCNN, bf16").
//
// Replaces, for the reference's LCNN (src/audiofakedetect/models.py:68-131) under bf16 autocast semantics for
// the matrix products: the nine convolutions (5x5, 3x3, 1x1) and the LSTM / Linear projections.  Operands are
// rounded to bf16 (round-to-nearest-even), products accumulate in fp32 on v_mfma_f32_32x32x16_bf16, tensors
// stay fp32 in HBM; max-feature-map, pooling, BatchNorm, the LSTM cell and the final Linear + mean stay fp32.
//
//   afd_conv2d_forward_bf16   implicit GEMM, M = Cout (32-row tiles), N = 128 output pixels of one image,
//                             K = Cin*k*k in chunks of 64.  The im2col tile is gathered straight into LDS as
//                             bf16: a thread owns one pixel and eight consecutive k of a chunk (its eight
//                             taps are coalesced loads across the 64 pixel lanes) and writes them as one
//                             16-byte fragment row; weights are converted once per call into a [Cout][Kpad]
//                             bf16 image that the waves read as 16-byte A fragments from L2.  With `mfm` the
//                             epilogue applies MaxFeatureMap2D (max over the two channel halves) and writes
//                             Cout / 2 channels: the full-width convolution output never reaches HBM.
//   afd_gemm_nt_bf16          C[M][N] = A[M][K] B[N][K]^T + bias (+ C): 64x64 tile per workgroup, both
//                             operands converted while they are staged.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kPix = 128;    // output pixels per workgroup (32 per wave)
constexpr int kChunk = 64;   // k per LDS stage
constexpr int kPitch = 72;   // bf16 per LDS row: 144 B keeps the 16-byte fragments aligned and spreads the banks

struct CG {
    int N, Cin, H, W, Cout, pad, Hout, Wout, Ktot, Kpad, tiles;
    int half, HP;  // MFM: Cout / 2 and its padding to whole 32-row tiles
};

// w [Cout][Cin][K][K] f32 -> wb [rows][Kpad] bf16, k = (ci * K + ky) * K + kx, zero padded.  With the
// max-feature-map fused the two channel halves each start on a tile boundary (rows [0, HP) and [HP, 2 HP)),
// so that the partners co and co + Cout/2 sit at the same row of two tiles of one wave.
__global__ void convert_weights_kernel(const float* __restrict__ w, __bf16* __restrict__ wb, int Cout, int Ktot,
                                       int Kpad, int total, int half, int HP) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int row = e / Kpad, k = e - row * Kpad;
    int co = row;
    bool ok = row < Cout;
    if (HP) {
        const int r = row < HP ? row : row - HP;
        ok = r < half;
        co = row < HP ? r : half + r;
    }
    wb[e] = (ok && k < Ktot) ? (__bf16)w[(size_t)co * Ktot + k] : (__bf16)0.f;
}

// workgroup = 128 output pixels of one image x all output channels; wave = 32 pixels x all MT row tiles
// (the pixel fragment is read once per k-step and reused by every row tile).  MFM: the epilogue writes
// max(y[co], y[co + Cout/2]) -- MaxFeatureMap2D (reference models.py:203-209) -- and only Cout/2 channels.
template <int K, int MT, bool MFM>
__global__ void __launch_bounds__(kThreads) conv_bf16_kernel(const CG g, const float* __restrict__ x,
                                                             const __bf16* __restrict__ wb,
                                                             const float* __restrict__ bias, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) __bf16 Bs[kPix][kPitch];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / g.tiles;
    const int p0 = (blockIdx.x - n * g.tiles) * kPix;
    const int HW = g.Hout * g.Wout;
    // staging role: pixel `sp`, k-groups sg, sg + 2, sg + 4, sg + 6 of a chunk
    const int sp = tid & (kPix - 1), sg = tid / kPix;
    const int p = p0 + sp;
    const bool pvalid = p < HW;
    const int oy = pvalid ? p / g.Wout : 0, ox = pvalid ? p - oy * g.Wout : 0;
    const float* xn = x + (size_t)n * g.Cin * g.H * g.W;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    for (int k0 = 0; k0 < g.Kpad; k0 += kChunk) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (kChunk / 8) / (kThreads / kPix); ++j) {
            const int kg = sg + (kThreads / kPix) * j;  // group of eight k
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + 8 * kg + e;
                const int ci = k / (K * K), rem = k - ci * (K * K);
                const int ky = rem / K, kx = rem - ky * K;
                const int iy = oy + ky - g.pad, ix = ox + kx - g.pad;
                float f = 0.f;
                if (pvalid && k < g.Ktot && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
                    f = xn[((size_t)ci * g.H + iy) * g.W + ix];
                v[e] = (__bf16)f;
            }
            *reinterpret_cast<bf16x8*>(&Bs[sp][8 * kg]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kChunk / 16; ++ks) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[32 * wave + r][16 * ks + 8 * h]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(
                    wb + (size_t)(32 * mt + r) * g.Kpad + k0 + 16 * ks + 8 * h);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mt], 0, 0, 0);
            }
        }
    }
    // D: column = lane & 31 (pixel), row = (q & 3) + 8 (q >> 2) + 4 (lane >> 5) (output channel inside the tile)
    const int pp = p0 + 32 * wave + r;
    if (pp < HW) {
        if (MFM) {
            float* yn = y + (size_t)n * g.half * HW + pp;
#pragma unroll
            for (int i = 0; i < MT / 2; ++i) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (co < g.half) {
                        const float a = acc[i][q] + (bias ? bias[co] : 0.f);
                        const float b = acc[i + MT / 2][q] + (bias ? bias[g.half + co] : 0.f);
                        yn[(size_t)co * HW] = b > a ? b : a;
                    }
                }
            }
        } else {
            float* yn = y + (size_t)n * g.Cout * HW + pp;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (co < g.Cout) yn[(size_t)co * HW] = acc[mt][q] + (bias ? bias[co] : 0.f);
                }
            }
        }
    }
}

template <int K, bool MFM>
int launch_conv(const CG& g, int rows, const float* x, const __bf16* wb, const float* bias, float* y, hipStream_t s) {
    const unsigned grid = (unsigned)(g.N * g.tiles);
    switch (rows / 32) {
        case 1:
            if (MFM) return afd::fail(AFD_ERR_UNSUPPORTED, "conv bf16: max-feature-map needs two row tiles");
            hipLaunchKernelGGL((conv_bf16_kernel<K, 1, false>), dim3(grid), dim3(kThreads), 0, s, g, x, wb, bias, y);
            break;
        case 2: hipLaunchKernelGGL((conv_bf16_kernel<K, 2, MFM>), dim3(grid), dim3(kThreads), 0, s, g, x, wb, bias, y); break;
        case 3:
            if (MFM) return afd::fail(AFD_ERR_UNSUPPORTED, "conv bf16: odd tile count with max-feature-map");
            hipLaunchKernelGGL((conv_bf16_kernel<K, 3, false>), dim3(grid), dim3(kThreads), 0, s, g, x, wb, bias, y);
            break;
        case 4: hipLaunchKernelGGL((conv_bf16_kernel<K, 4, MFM>), dim3(grid), dim3(kThreads), 0, s, g, x, wb, bias, y); break;
        default: return afd::fail(AFD_ERR_UNSUPPORTED, "conv bf16: Cout %d > 128", g.Cout);
    }
    return afd::check_launch("conv_bf16_kernel");
}

// C[M][N] = A[M][K] . B[N][K]^T + bias[N] (+ C): 64x64 tile, four waves of one 32x32 accumulator each
__global__ void __launch_bounds__(kThreads)
gemm_nt_bf16_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                    float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int accumulate) {
    __shared__ __attribute__((aligned(16))) __bf16 As[64][kPitch];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[64][kPitch];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int k0 = 0; k0 < K; k0 += kChunk) {
        __syncthreads();
        // thread -> (row = tid / 8 (+32), group of eight k = tid % 8): 8 lanes read 256 contiguous bytes of a row
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (tid >> 3) + 32 * j, kg = tid & 7;
            bf16x8 va, vb;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + 8 * kg + e;
                va[e] = (__bf16)((m0 + row < M && k < K) ? A[(size_t)(m0 + row) * lda + k] : 0.f);
                vb[e] = (__bf16)((n0 + row < N && k < K) ? B[(size_t)(n0 + row) * ldb + k] : 0.f);
            }
            *reinterpret_cast<bf16x8*>(&As[row][8 * kg]) = va;
            *reinterpret_cast<bf16x8*>(&Bs[row][8 * kg]) = vb;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kChunk / 16; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[wm + r][16 * ks + 8 * h]);
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(&Bs[wn + r][16 * ks + 8 * h]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn + r;
    if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = m0 + wm + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (m < M) {
                float v = acc[q] + bv;
                if (accumulate) v += C[(size_t)m * ldc + n];
                C[(size_t)m * ldc + n] = v;
            }
        }
    }
}

// The same product on 128 x 128 tiles (round 5): four waves of 2 x 2 accumulator tiles each, 16-byte loads of the fp32
// operands a chunk ahead (registers), rounded to bf16 on their way into a double-buffered LDS image, one barrier per
// 32-deep chunk.  The 64 x 64 kernel above gathers its operands element by element behind two barriers per chunk and
// ran the LSTM input projections of the bf16 evaluation forward ([6 B x 512] . [1024 x 512]^T) at 47 TFLOP/s.
// Needs K % 32 == 0 and 16-byte aligned rows (lda, ldb multiples of 4, aligned bases): the host falls back otherwise.
constexpr int kGK = 32, kGPitch = 40;  // bf16 per LDS row: 80-byte rows, 16-byte aligned, banks spread
__global__ void __launch_bounds__(kThreads)
gemm_nt_bf16_128_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                        float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int accumulate) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][128][kGPitch];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][128][kGPitch];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    // thread -> (row = tid / 8 + 32 j, float4 tid % 8 of the row's 32 k): rows past the matrix read its last row and are
    // never stored
    const int lrow = tid >> 3, c4 = tid & 7;
    const float* ap[4];
    const float* bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ra = m0 + lrow + 32 * j, rb = n0 + lrow + 32 * j;
        ap[j] = A + (size_t)(ra < M ? ra : M - 1) * lda + 4 * c4;
        bp[j] = B + (size_t)(rb < N ? rb : N - 1) * ldb + 4 * c4;
    }
    float4 ra[4], rb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const float4*>(ap[j] + k0);
            rb[j] = *reinterpret_cast<const float4*>(bp[j] + k0);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 va, vb;
            va[0] = (__bf16)ra[j].x; va[1] = (__bf16)ra[j].y; va[2] = (__bf16)ra[j].z; va[3] = (__bf16)ra[j].w;
            vb[0] = (__bf16)rb[j].x; vb[1] = (__bf16)rb[j].y; vb[2] = (__bf16)rb[j].z; vb[3] = (__bf16)rb[j].w;
            *reinterpret_cast<bf16x4*>(&As[buf][lrow + 32 * j][4 * c4]) = va;
            *reinterpret_cast<bf16x4*>(&Bs[buf][lrow + 32 * j][4 * c4]) = vb;
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
    const int nchunks = K / kGK;
    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunks) fetch((c + 1) * kGK);  // in flight during this chunk's matrix instructions
#pragma unroll
        for (int ks = 0; ks < kGK / 16; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(&As[buf][wm + 32 * i + r][16 * ks + 8 * h]);
                b[i] = *reinterpret_cast<const bf16x8*>(&Bs[buf][wn + 32 * i + r][16 * ks + 8 * h]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < nchunks) stash(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + r;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (m < M) {
                    float v = acc[i][j][q] + bv;
                    if (accumulate) v += C[(size_t)m * ldc + n];
                    C[(size_t)m * ldc + n] = v;
                }
            }
    }
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

#define AFD_STREAM static_cast<hipStream_t>(stream)

extern "C" size_t afd_conv2d_bf16_workspace_bytes(int Cin, int Cout, int K) {
    if (Cin < 1 || Cout < 1 || K < 1) return 0;
    // with the max-feature-map fused each channel half is padded to whole tiles: at most Cout + 62 rows
    return (size_t)(round_up(Cout, 32) + 64) * round_up(Cin * K * K, kChunk) * sizeof(__bf16);
}

extern "C" int afd_conv2d_forward_bf16(const float* x, const float* w, const float* bias, float* y, int N,
                                       int Cin, int H, int W, int Cout, int K, int pad, int mfm, void* ws,
                                       size_t ws_bytes, afd_stream_t stream) {
    if (!x || !w || !y) return afd::fail(AFD_ERR_ARG, "conv bf16: null pointer");
    if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return afd::fail(AFD_ERR_ARG, "conv bf16: bad shape");
    if (K != 1 && K != 3 && K != 5) return afd::fail(AFD_ERR_UNSUPPORTED, "conv bf16: kernel size %d", K);
    if (pad < 0 || pad >= K) return afd::fail(AFD_ERR_ARG, "conv bf16: padding %d", pad);
    if (mfm && (Cout & 1)) return afd::fail(AFD_ERR_ARG, "conv bf16: max-feature-map over an odd channel count");
    if (!ws || ws_bytes < afd_conv2d_bf16_workspace_bytes(Cin, Cout, K))
        return afd::fail(AFD_ERR_WORKSPACE, "conv bf16: workspace too small");
    CG g{};
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.pad = pad;
    g.Hout = H + 2 * pad - (K - 1);
    g.Wout = W + 2 * pad - (K - 1);
    if (g.Hout < 1 || g.Wout < 1) return afd::fail(AFD_ERR_ARG, "conv bf16: empty output");
    g.Ktot = Cin * K * K;
    g.Kpad = round_up(g.Ktot, kChunk);
    g.tiles = (g.Hout * g.Wout + kPix - 1) / kPix;
    g.half = mfm ? Cout / 2 : 0;
    g.HP = mfm ? round_up(Cout / 2, 32) : 0;
    const int rows = mfm ? 2 * g.HP : round_up(Cout, 32);
    if (rows > 128) return afd::fail(AFD_ERR_UNSUPPORTED, "conv bf16: Cout %d", Cout);
    if ((long)N * g.tiles > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv bf16: grid too large");
    __bf16* wb = static_cast<__bf16*>(ws);
    const int total = rows * g.Kpad;
    hipLaunchKernelGGL(convert_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, AFD_STREAM, w, wb, Cout,
                       g.Ktot, g.Kpad, total, g.half, g.HP);
    int rc = afd::check_launch("convert_weights_kernel");
    if (rc) return rc;
    if (mfm) {
        if (K == 1) return launch_conv<1, true>(g, rows, x, wb, bias, y, AFD_STREAM);
        if (K == 3) return launch_conv<3, true>(g, rows, x, wb, bias, y, AFD_STREAM);
        return launch_conv<5, true>(g, rows, x, wb, bias, y, AFD_STREAM);
    }
    if (K == 1) return launch_conv<1, false>(g, rows, x, wb, bias, y, AFD_STREAM);
    if (K == 3) return launch_conv<3, false>(g, rows, x, wb, bias, y, AFD_STREAM);
    return launch_conv<5, false>(g, rows, x, wb, bias, y, AFD_STREAM);
}

extern "C" int afd_gemm_nt_bf16(const float* A, const float* B, const float* bias, float* C, int M, int N, int K,
                                int lda, int ldb, int ldc, int accumulate, afd_stream_t stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || lda < K || ldb < K || ldc < N)
        return afd::fail(AFD_ERR_ARG, "gemm_nt bf16: bad argument");
    afd::ScopedTiming timing(AFD_K_LCNN_BF16, 2.0 * M * (double)N * K, AFD_STREAM);
    timing.bytes(4.0 * ((double)M * K + (double)N * K + (double)M * N));
    const bool aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && lda % 4 == 0 && ldb % 4 == 0;
    if (M >= 128 && N >= 128 && K % kGK == 0 && aligned && !getenv("AFD_GEMM_BF16_SMALL")) {
        timing.issued(2.0 * ((M + 127) / 128 * 128) * (double)((N + 127) / 128 * 128) * K);
        hipLaunchKernelGGL(gemm_nt_bf16_128_kernel, dim3((N + 127) / 128, (M + 127) / 128), dim3(kThreads), 0, AFD_STREAM, A, B,
                           bias, C, M, N, K, lda, ldb, ldc, accumulate);
        return afd::check_launch("gemm_nt_bf16_128_kernel");
    }
    timing.issued(2.0 * ((M + 63) / 64 * 64) * (double)((N + 63) / 64 * 64) * ((K + kChunk - 1) / kChunk * kChunk));
    hipLaunchKernelGGL(gemm_nt_bf16_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(kThreads), 0, AFD_STREAM, A, B,
                       bias, C, M, N, K, lda, ldb, ldc, accumulate);
    return afd::check_launch("gemm_nt_bf16_kernel");
}
