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
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kKC = 64;  // basis rows per LDS chunk

struct StftParams {
    const float* x;
    const float* basis;  // [KP][MP]
    float* out;
    int B, N, n_fft, hop, F, T, FP, MP, KP, TG, seg, segpad, FB;
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
    const int t0 = blockIdx.y * p.TG;  // first frame of this group
    // reflect-padded segment of the frame: padded position P = t0*hop + s, sample j = P - n_fft/2
    const float* xb = p.x + (size_t)b * p.N;
    const int pad = p.n_fft / 2;
    // groups of four samples, eight groups in flight per thread; a group that touches the
    // reflected edges (or the end of the segment) goes element by element
    const int ngroups = (p.seg + 3) >> 2;
    for (int g0 = 0; g0 < ngroups; g0 += 256 * 8) {
        f32x4u v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = 4 * (g0 + u * 256 + tid);
            const int j0 = t0 * p.hop + s - pad;
            f32x4u q = {0.f, 0.f, 0.f, 0.f};
            if (s + 3 < p.seg && j0 >= 0 && j0 + 3 < p.N) {
                q = *reinterpret_cast<const f32x4u*>(xb + j0);
            } else if (s < p.seg) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int j = j0 + e;
                    j = j < 0 ? -j : j;
                    j = j >= p.N ? 2 * (p.N - 1) - j : j;
                    if (s + e < p.seg && j >= 0 && j < p.N) q[e] = xb[j];
                }
            }
            v[u] = q;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int s = 4 * (g0 + u * 256 + tid);
            if (s < p.segpad) *reinterpret_cast<float4*>(seg + s) = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
        }
    }
    __syncthreads();

    const int tl = wave * 32 + l31;  // frame inside the group handled by this lane's column
    const bool tok = (tl < p.TG) && (t0 + tl < p.T);
    const float* bcol = seg + (tok ? tl * p.hop : 0);
    // A (basis) tile of this bin block, staged through LDS in chunks of kKC rows (two buffers:
    // the next chunk's loads are issued before this chunk's MFMAs and stored after them).  A
    // fragment reads from global memory paid an L2 round trip per k-step in every wave.
    float* As = seg + p.segpad;  // [2][kKC][64]: columns 0-31 cos, 32-63 sin of the block
    // the staged segment serves p.FB consecutive bin blocks (staging it once per block cost more
    // than the block's matrix work)
    for (int fb = 0; fb < p.FB; ++fb) {
    const int fblk = blockIdx.x * p.FB + fb;
    if (fblk * 32 >= p.FP) break;
    const float* gcos = p.basis + fblk * 32;
    const int nchunks = p.KP / kKC;
    constexpr int kPer = kKC * 64 / 4 / 256;  // float4 per thread per chunk
    float4 stage[kPer];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = tid + u * 256;        // float4 index inside the chunk: row = i / 16
            const int row = i >> 4, c4 = i & 15;  // 16 float4 per row: 8 cos, 8 sin
            const float* src = gcos + (size_t)(c * kKC + row) * p.MP + (c4 < 8 ? 4 * c4 : p.FP + 4 * (c4 - 8));
            stage[u] = *reinterpret_cast<const float4*>(src);
        }
    };
    auto store_chunk = [&](int buf) {
        float4* dst = reinterpret_cast<float4*>(As + buf * kKC * 64);
#pragma unroll
        for (int u = 0; u < kPer; ++u) dst[tid + u * 256] = stage[u];
    };
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    f32x16 ac, as;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        ac[r] = 0.f;
        as[r] = 0.f;
    }
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) load_chunk(c + 1);
        const float* arow = As + (c & 1) * kKC * 64 + l31 + half * 64;
        const float* brow = bcol + c * kKC + half;
        const bool kfull = (c + 1) * kKC <= p.n_fft;  // only the last chunk runs past n_fft
        // groups of 4 k-steps with ping-pong fragment registers: the 12 LDS reads of group g+1 are
        // spread between the 8 MFMAs of group g (one wave per SIMD here: nothing else hides them)
        constexpr int G = 4;
        float a0[2][G], a1[2][G], bv[2][G];
        auto read_group = [&](int g, int pp) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int kl2 = 2 * (g * G + j);  // even row of the k-step (+ half folded in the bases)
                a0[pp][j] = arow[kl2 * 64];
                a1[pp][j] = arow[kl2 * 64 + 32];
                bv[pp][j] = (kfull || c * kKC + kl2 + half < p.n_fft) ? brow[kl2] : 0.f;
            }
        };
        read_group(0, 0);
#pragma unroll
        for (int g = 0; g < kKC / 2 / G; ++g) {
            const int pp = g & 1;
            if (g + 1 < kKC / 2 / G) read_group(g + 1, pp ^ 1);
#pragma unroll
            for (int j = 0; j < G; ++j) {
                ac = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[pp][j], bv[pp][j], ac, 0, 0, 0);
                as = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[pp][j], bv[pp][j], as, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2 * G; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < nchunks) store_chunk((c + 1) & 1);
        __syncthreads();
    }
    if (tok) {
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
    __syncthreads();  // the A buffers are re-used by the next bin block
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
    if (basis_rows) *basis_rows = (n_fft + 63) & ~63;  // whole LDS chunks of 64 rows (zero rows past n_fft)
    if (basis_cols) *basis_cols = 2 * fp;
    return AFD_OK;
}

// host: basis[k][m], m < FP: w[k] cos(2 pi m k / n), m >= FP: -w[k] sin(2 pi (m-FP) k / n),
// w = periodic Hann; zero for padded rows / columns
extern "C" int afd_stft_basis(int n_fft, float* basis) {
    if (!basis || n_fft < 2) return afd::fail(AFD_ERR_ARG, "stft basis: bad argument");
    const int f = n_fft / 2 + 1;
    const int fp = ((f + 31) / 32) * 32;
    const int kp = (n_fft + 63) & ~63;
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
    while (tg > 32 && ((long)(tg - 1) * hop + n_fft) > 30000) tg >>= 1;
    if ((long)(tg - 1) * hop + n_fft > 30000) return afd::fail(AFD_ERR_UNSUPPORTED, "stft: hop %d too large", hop);
    if (p.T < tg) tg = ((p.T + 31) / 32) * 32;
    p.TG = tg;
    p.seg = (tg - 1) * hop + n_fft;
    p.segpad = (p.seg + 3) & ~3;
    const size_t lds = ((size_t)p.segpad + 2 * kKC * 64) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int waves = tg / 32;
    afd::ScopedTiming timing(AFD_K_STFT, 4.0 * B * ((double)N + (double)p.F * p.T), s);
    // bin blocks per workgroup
    const int nblk = p.FP / 32, tgroups = (p.T + tg - 1) / tg;
    // cost model in units of one block's matrix work: rounds over the CUs x (staging ~0.7 + blocks)
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu < 1) ncu = 256;
    int fb = 1;
    double best = 1e30;
    for (int cand = 1; cand <= nblk; ++cand) {
        const long wgs = (long)((nblk + cand - 1) / cand) * tgroups * B;
        const double cost = (double)((wgs + ncu - 1) / ncu) * (0.7 + cand);
        if (cost < best) { best = cost; fb = cand; }
    }
    p.FB = fb;
    hipLaunchKernelGGL(stft_mfma_kernel, dim3((nblk + fb - 1) / fb, tgroups, B), dim3(256), lds, s, p);
    (void)waves;
    return afd::check_launch("stft_mfma_kernel");
}
