// Power spectrogram (STFT) front end for gfx950 as a matrix DFT on the f32 MFMA.
//
// Replaces torchaudio.transforms.Spectrogram(n_fft, hop, power) + log of the reference's
// STFTLayer (src/audiofakedetect/wavelet_math.py:25-68; torch.stft defaults: periodic Hann
// window of n_fft, centre, reflect pad n_fft/2, one-sided, not normalised) and the scalar
// Normalize (:380-382).  The reference calls it with n_fft = 2*num_of_scales-1 = 511 = 7*73
// (no radix-2 FFT): here X = Basis . frames is an exact-f32 MFMA contraction
//   A = basisT [K][2F]  (k-major; column f: w[k] cos(2 pi f k/n), column F'+f: -w[k] sin)
//   B[k][t] = xe[t*hop + k]   read straight from the reflect-padded frame in LDS (no im2col)
// one wave = 32 frequency bins (a cos tile and a sin tile) x 32 frames, so re^2 + im^2 is a
// per-register epilogue; output [B][1][F][T], T fastest (torch.stft's layout).
//
// Algorithmic bytes per frame = 4 (N + F T); flops = 2 * 2F * n_fft * T (52.9 MFLOP at
// n_fft 511, hop 220).
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cmath>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct StftParams {
    const float* x;
    const float* basis;  // [KP][MP]
    float* out;
    int B, N, n_fft, hop, F, T, FP, MP, KP, TG, seg;
    unsigned flags;
    float power, eps, mean, std;
};

__global__ void __launch_bounds__(256) stft_mfma_kernel(const StftParams p) {
    extern __shared__ __attribute__((aligned(16))) float seg[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int b = blockIdx.z;
    const int fblk = blockIdx.x;       // 32 bins
    const int t0 = blockIdx.y * p.TG;  // first frame of this group
    // reflect-padded segment of the frame: padded position P = t0*hop + s, sample j = P - n_fft/2
    const float* xb = p.x + (size_t)b * p.N;
    const int pad = p.n_fft / 2;
    for (int s = tid; s < p.seg; s += 256) {
        int j = t0 * p.hop + s - pad;
        j = j < 0 ? -j : j;
        j = j >= p.N ? 2 * (p.N - 1) - j : j;
        seg[s] = (j >= 0 && j < p.N) ? xb[j] : 0.f;
    }
    __syncthreads();

    const int tl = wave * 32 + l31;  // frame inside the group handled by this lane's column
    const bool tok = (tl < p.TG) && (t0 + tl < p.T);
    const float* bcol = seg + (tok ? tl * p.hop : 0);
    const float* acos = p.basis + fblk * 32 + l31;
    const float* asin = acos + p.FP;

    f32x16 ac, as;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ac[r] = 0.f;
        as[r] = 0.f;
    }
    const int ksteps = p.KP >> 1;
#pragma unroll 8
    for (int ks = 0; ks < ksteps; ++ks) {
        const int k = 2 * ks + half;
        const float a0 = acos[(size_t)k * p.MP];
        const float a1 = asin[(size_t)k * p.MP];
        const float bv = (k < p.n_fft) ? bcol[k] : 0.f;
        ac = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, ac, 0, 0, 0);
        as = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, as, 0, 0, 0);
    }
    if (!tok) return;
    const int t = t0 + tl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = fblk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (f < p.F) {
            float v = ac[r] * ac[r] + as[r] * as[r];
            if (p.power != 2.0f) v = powf(sqrtf(v), p.power);
            if (p.flags & AFD_STFT_LOG) v = logf(v + p.eps);
            if (p.flags & AFD_STFT_NORM) v = (v - p.mean) / p.std;
            p.out[((size_t)b * p.F + f) * p.T + t] = v;
        }
    }
}

}  // namespace

extern "C" int afd_stft_dims(int N, int n_fft, int hop, int* F, int* T, int* basis_rows,
                             int* basis_cols) {
    if (N < 2 || n_fft < 2 || hop < 1 || n_fft / 2 >= N) return afd::fail(AFD_ERR_ARG, "stft: bad geometry");
    const int f = n_fft / 2 + 1;
    if (F) *F = f;
    if (T) *T = 1 + N / hop;
    const int fp = ((f + 31) / 32) * 32;
    if (basis_rows) *basis_rows = (n_fft + 1) & ~1;
    if (basis_cols) *basis_cols = 2 * fp;
    return AFD_OK;
}

// host: basis[k][m], m < FP: w[k] cos(2 pi m k / n), m >= FP: -w[k] sin(2 pi (m-FP) k / n),
// w = periodic Hann; zero for padded rows / columns
extern "C" int afd_stft_basis(int n_fft, float* basis) {
    if (!basis || n_fft < 2) return afd::fail(AFD_ERR_ARG, "stft basis: bad argument");
    const int f = n_fft / 2 + 1;
    const int fp = ((f + 31) / 32) * 32;
    const int kp = (n_fft + 1) & ~1;
    const int mp = 2 * fp;
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < kp; ++k) {
        const double w = k < n_fft ? 0.5 - 0.5 * std::cos(2.0 * pi * k / n_fft) : 0.0;
        for (int m = 0; m < mp; ++m) {
            const int fi = m < fp ? m : m - fp;
            double v = 0.0;
            if (fi < f && k < n_fft) {
                const long ph = ((long)fi * k) % n_fft;
                const double ang = 2.0 * pi * (double)ph / n_fft;
                v = m < fp ? w * std::cos(ang) : -w * std::sin(ang);
            }
            basis[(size_t)k * mp + m] = (float)v;
        }
    }
    return AFD_OK;
}

extern "C" int afd_stft_forward(const float* x, int B, int N, int n_fft, int hop, const float* basis,
                                unsigned flags, float power, float eps, float mean, float std,
                                float* out, afd_stream_t stream) {
    if (!x || !basis || !out || B < 1) return afd::fail(AFD_ERR_ARG, "stft: null pointer");
    StftParams p{};
    int rc = afd_stft_dims(N, n_fft, hop, &p.F, &p.T, &p.KP, &p.MP);
    if (rc) return rc;
    if ((flags & AFD_STFT_NORM) && std == 0.f) return afd::fail(AFD_ERR_ARG, "stft: std == 0");
    if (B > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "stft: batch > 65535");
    p.x = x; p.basis = basis; p.out = out; p.B = B; p.N = N; p.n_fft = n_fft; p.hop = hop;
    p.FP = p.MP / 2;
    p.flags = flags; p.power = power; p.eps = eps; p.mean = mean; p.std = std;
    // frames per workgroup: 4 waves x 32, shrunk until the padded segment fits LDS
    int tg = 128;
    while (tg > 32 && ((long)(tg - 1) * hop + n_fft) > 36000) tg >>= 1;
    if ((long)(tg - 1) * hop + n_fft > 36000) return afd::fail(AFD_ERR_UNSUPPORTED, "stft: hop %d too large", hop);
    if (p.T < tg) tg = ((p.T + 31) / 32) * 32;
    p.TG = tg;
    p.seg = (tg - 1) * hop + n_fft;
    const size_t lds = (size_t)p.seg * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int waves = tg / 32;
    afd::ScopedTiming timing(AFD_K_STFT, 4.0 * B * ((double)N + (double)p.F * p.T), s);
    hipLaunchKernelGGL(stft_mfma_kernel, dim3(p.FP / 32, (p.T + tg - 1) / tg, B), dim3(256), lds, s, p);
    (void)waves;
    return afd::check_launch("stft_mfma_kernel");
}
