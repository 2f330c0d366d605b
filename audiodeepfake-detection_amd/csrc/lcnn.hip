// LCNN-specific layers for gfx950: max-feature-map, dense f32-MFMA GEMM (C = A B^T + bias)
// and the LSTM cell update used by the two bidirectional LSTM layers.
//
// Replaces, for the reference's LCNN (src/audiofakedetect/models.py:68-131):
//   MaxFeatureMap2D (:161-209)          -> afd_mfm_forward / afd_mfm_backward
//   nn.LSTM inside BLSTMLayer (:212-237)-> afd_gemm_nt (input and recurrent projections) +
//                                          afd_lstm_cell (gate non-linearities, c/h update)
//   nn.Linear(512, classes) + mean(1)   -> afd_linear_mean_forward (nn.hip)
// The convolutions / pooling / batch norm of LCNN reuse conv.hip and nn.hip.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kT = 256;

// y[n][c][hw] = max(x[n][c][hw], x[n][c + C/2][hw]); sel = 1 where the second half won
__global__ void mfm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                               unsigned char* __restrict__ sel, int Ch, size_t HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < total; i += (size_t)gridDim.x * kT) {
        const size_t n = i / (Ch * HW);
        const size_t r = i - n * Ch * HW;
        const float a = x[n * 2 * Ch * HW + r];
        const float b = x[n * 2 * Ch * HW + Ch * HW + r];
        const bool s = b > a;
        y[i] = s ? b : a;
        if (sel) sel[i] = s ? 1 : 0;
    }
}

__global__ void mfm_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ sel,
                               float* __restrict__ dx, int Ch, size_t HW, size_t total) {
    for (size_t i = (size_t)blockIdx.x * kT + threadIdx.x; i < total; i += (size_t)gridDim.x * kT) {
        const size_t n = i / (Ch * HW);
        const size_t r = i - n * Ch * HW;
        const float g = dy[i];
        const bool s = sel[i] != 0;
        dx[n * 2 * Ch * HW + r] = s ? 0.f : g;
        dx[n * 2 * Ch * HW + Ch * HW + r] = s ? g : 0.f;
    }
}

// C[M][N] = A[M][K] . B[N][K]^T + bias[N] (+ C if accumulate).  64x64 tile per workgroup
// (4 waves, each 32x32 = one MFMA accumulator), K staged through LDS in slabs of 32.
__global__ void __launch_bounds__(256)
gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
               float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, int accumulate) {
    __shared__ float As[32][65];  // [k][m]
    __shared__ float Bs[32][65];  // [k][n]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 32) {
        __syncthreads();
        // 64 rows x 32 k of A and B: thread -> (row = tid / 4, k = (tid % 4) * 8 .. +8)
        {
            const int row = tid >> 2, kk = (tid & 3) * 8;
            float va[8], vb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + kk + j;
                va[j] = (m0 + row < M && k < K) ? A[(size_t)(m0 + row) * lda + k] : 0.f;
                vb[j] = (n0 + row < N && k < K) ? B[(size_t)(n0 + row) * ldb + k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                As[kk + j][row] = va[j];
                Bs[kk + j][row] = vb[j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int k = 2 * ks + half;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k][wm + l31], Bs[k][wn + l31], acc, 0, 0, 0);
        }
    }
    const int n = n0 + wn + l31;
    if (n < N) {
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < M) {
                float v = acc[r] + bv;
                if (accumulate) v += C[(size_t)m * ldc + n];
                C[(size_t)m * ldc + n] = v;
            }
        }
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gates [B][4H] (i | f | g | o, torch order) = pre-activation sums; updates c [B][H] in place
// and writes h into hout (row stride ldh, e.g. a slice of the [T][B][2H] layer output)
__global__ void lstm_cell_kernel(const float* __restrict__ gates, float* __restrict__ c,
                                 float* __restrict__ hout, float* __restrict__ hstate, int Bn, int H,
                                 int ldh) {
    const int total = Bn * H;
    for (int i = blockIdx.x * kT + threadIdx.x; i < total; i += gridDim.x * kT) {
        const int b = i / H, j = i - b * H;
        const float* g = gates + (size_t)b * 4 * H;
        const float ig = sigmoidf_(g[j]);
        const float fg = sigmoidf_(g[H + j]);
        const float gg = tanhf(g[2 * H + j]);
        const float og = sigmoidf_(g[3 * H + j]);
        const float cn = fg * c[i] + ig * gg;
        const float hn = og * tanhf(cn);
        c[i] = cn;
        hstate[i] = hn;
        hout[(size_t)b * ldh + j] = hn;
    }
}

// One backward step of the cell.  gates = the saved pre-activation sums of the step, c / cprev
// the cell state after / before it (cprev may be NULL: zero), dh = gradient reaching h_t
// (layer output gradient + recurrent part), dc = running cell-state gradient (in: from step
// t+1, out: for step t-1), dpre = gradient of the pre-activations [B][4H].
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c,
                                     const float* __restrict__ cprev, const float* __restrict__ dh,
                                     int lddh, float* __restrict__ dc, float* __restrict__ dpre, int Bn,
                                     int H) {
    const int total = Bn * H;
    for (int idx = blockIdx.x * kT + threadIdx.x; idx < total; idx += gridDim.x * kT) {
        const int b = idx / H, j = idx - b * H;
        const float* g = gates + (size_t)b * 4 * H;
        const float ig = sigmoidf_(g[j]);
        const float fg = sigmoidf_(g[H + j]);
        const float gg = tanhf(g[2 * H + j]);
        const float og = sigmoidf_(g[3 * H + j]);
        const float tc = tanhf(c[idx]);
        const float dhv = dh[(size_t)b * lddh + j];
        const float dcv = fmaf(dhv * og, 1.f - tc * tc, dc[idx]);
        const float cp = cprev ? cprev[idx] : 0.f;
        float* d = dpre + (size_t)b * 4 * H;
        d[j] = dcv * gg * ig * (1.f - ig);
        d[H + j] = dcv * cp * fg * (1.f - fg);
        d[2 * H + j] = dcv * ig * (1.f - gg * gg);
        d[3 * H + j] = dhv * tc * og * (1.f - og);
        dc[idx] = dcv * fg;
    }
}

inline unsigned grid1(size_t n) {
    size_t b = (n + kT - 1) / kT;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

#define AFD_STREAM static_cast<hipStream_t>(stream)

extern "C" int afd_mfm_forward(const float* x, float* y, uint8_t* sel, int N, int C, int HW,
                               afd_stream_t stream) {
    if (!x || !y || N < 1 || C < 2 || (C & 1) || HW < 1) return afd::fail(AFD_ERR_ARG, "mfm fwd: bad argument");
    const size_t total = (size_t)N * (C / 2) * HW;
    hipLaunchKernelGGL(mfm_fwd_kernel, dim3(grid1(total)), dim3(kT), 0, AFD_STREAM, x, y, sel, C / 2,
                       (size_t)HW, total);
    return afd::check_launch("mfm_fwd_kernel");
}

extern "C" int afd_mfm_backward(const float* dy, const uint8_t* sel, float* dx, int N, int C, int HW,
                                afd_stream_t stream) {
    if (!dy || !sel || !dx || (C & 1)) return afd::fail(AFD_ERR_ARG, "mfm bwd: bad argument");
    const size_t total = (size_t)N * (C / 2) * HW;
    hipLaunchKernelGGL(mfm_bwd_kernel, dim3(grid1(total)), dim3(kT), 0, AFD_STREAM, dy, sel, dx, C / 2,
                       (size_t)HW, total);
    return afd::check_launch("mfm_bwd_kernel");
}

extern "C" int afd_gemm_nt(const float* A, const float* B, const float* bias, float* C, int M, int N,
                           int K, int lda, int ldb, int ldc, int accumulate, afd_stream_t stream) {
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || lda < K || ldb < K || ldc < N)
        return afd::fail(AFD_ERR_ARG, "gemm_nt: bad argument");
    hipLaunchKernelGGL(gemm_nt_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, AFD_STREAM, A,
                       B, bias, C, M, N, K, lda, ldb, ldc, accumulate);
    return afd::check_launch("gemm_nt_kernel");
}

extern "C" int afd_lstm_cell(const float* gates, float* c, float* hout, float* hstate, int B, int H,
                             int ldh, afd_stream_t stream) {
    if (!gates || !c || !hout || !hstate || B < 1 || H < 1 || ldh < H)
        return afd::fail(AFD_ERR_ARG, "lstm cell: bad argument");
    hipLaunchKernelGGL(lstm_cell_kernel, dim3(grid1((size_t)B * H)), dim3(kT), 0, AFD_STREAM, gates, c,
                       hout, hstate, B, H, ldh);
    return afd::check_launch("lstm_cell_kernel");
}

extern "C" int afd_lstm_cell_backward(const float* gates, const float* c, const float* cprev,
                                      const float* dh, int lddh, float* dc, float* dpre, int B, int H,
                                      afd_stream_t stream) {
    if (!gates || !c || !dh || !dc || !dpre || B < 1 || H < 1 || lddh < H)
        return afd::fail(AFD_ERR_ARG, "lstm cell bwd: bad argument");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid1((size_t)B * H)), dim3(kT), 0, AFD_STREAM, gates, c,
                       cprev, dh, lddh, dc, dpre, B, H);
    return afd::check_launch("lstm_cell_bwd_kernel");
}
