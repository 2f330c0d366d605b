// Power spectrogram (STFT) front end for gfx950 as a matrix DFT on the f32 MFMA.
//
// Replaces torchaudio.transforms.Spectrogram(n_fft, hop, power) + log of the reference's
// STFTLayer (src/audiofakedetect/wavelet_math.py:25-68; torch.stft defaults: periodic Hann
// window of n_fft, centre, reflect pad n_fft/2, one-sided, not normalised) and the scalar
// Normalize (:380-382).  The reference calls it with n_fft = 2*num_of_scales-1 = 511 = 7*73
// (no radix-2 FFT): here X = Basis . frames is an exact-f32 MFMA contraction, FOLDED on the window's
// symmetry (round 6).  The periodic Hann window has w[0] = 0 and w[k] = w[n-k], so with
//   e_t[k] = xe[t*hop + k] + xe[t*hop + n - k],  o_t[k] = xe[t*hop + k] - xe[t*hop + n - k],  k = 1 .. floor(n/2)
//   Re X[f] =  sum_k w[k] cos(2 pi f k/n) e_t[k],   Im X[f] = -sum_k w[k] sin(2 pi f k/n) o_t[k]
// (an even n counts its Nyquist sample k = n/2 once: that basis row carries w/2) -- K = 255 per tile instead of
// 511: half the matrix instructions, exact in real arithmetic.
//   A = basisT [K][2F]  (k-major, row r = sample k = r + 1; column f: w[k] cos, column F'+f: -w[k] sin)
//   B: e_t / o_t formed from two LDS reads of the reflect-padded frame (no im2col) and one add / subtract
// one wave = 32 frequency bins (a cos tile and a sin tile) x 32 frames, so re^2 + im^2 is a
// per-register epilogue; output [B][1][F][T], T fastest (torch.stft's layout).
//
// Algorithmic bytes per frame = 4 (N + F T); flops issued = 2 * 2F * floor(n_fft/2) * T (26.4 MFLOP at
// n_fft 511, hop 220; 52.9 in unfolded form).
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cmath>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4a __attribute__((ext_vector_type(4)));

constexpr int kKC = 64;        // folded basis rows per LDS chunk = 32 k-steps of the 32x32x2 instruction
constexpr int kAPitch = 36;    // floats per (tile, k-half, bin) row of a chunk image: 32 k-steps + 4 (16-byte reads of
                               // 16 neighbouring lanes then cover the 64 banks once)
constexpr int kABuf = 4 * 32 * kAPitch;  // one chunk image: [cos | sin][k-half][bin][k-step]
constexpr int kMargin = 64;    // zero floats on both sides of the staged segment: the padded rows of the last chunk
                               // (zero basis) read at most 63 samples outside a frame's window, and must read finite values

struct StftParams {
    const float* x;
    const float* basis;  // [bin block][chunk][cos | sin][k-half][bin][k-step]: the LDS chunk images, unpadded
    float* out;
    int B, N, n_fft, hop, F, T, FP, MP, KP, TG, seg, segpad, FB;
    unsigned flags;
    float power, eps, mean, std;
};

__global__ void __launch_bounds__(256) stft_mfma_kernel(const StftParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* seg = lds + kMargin;
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
    if (tid < kMargin) {
        lds[tid] = 0.f;
        seg[p.segpad + tid] = 0.f;
    }
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

    const int tl = wave * 32 + l31;  // frame inside the group handled by this lane's column
    const bool tok = (tl < p.TG) && (t0 + tl < p.T);
    const float* bcol = seg + (tok ? tl * p.hop : 0);
    // The basis tiles of this workgroup's bin blocks, staged through LDS in chunks of kKC rows (two images: the next
    // chunk's loads are issued before this chunk's matrix instructions and stored after them; the chunk sequence runs
    // across bin blocks, so a block's first chunk arrives under the previous block's last).  The staged segment
    // serves p.FB consecutive bin blocks (staging it once per block cost more than the block's matrix work).
    float* As = seg + p.segpad + kMargin;
    const int nchunks = p.KP / kKC;
    const int fblk0 = blockIdx.x * p.FB;
    const int nblk_here = min(p.FB, p.FP / 32 - fblk0);
    const int total = nblk_here * nchunks;
    const float4* gsrc = reinterpret_cast<const float4*>(p.basis + (size_t)fblk0 * nchunks * (kKC * 64)) + tid;
    // float4 i of a chunk image (i = tid + 256 u) lands at row (4 i) / 32, column (4 i) % 32 of the padded image
    const int arow0 = (tid >> 3) * kAPitch + 4 * (tid & 7);
    float4 st0, st1, st2, st3;
    auto load_chunk = [&](int it) {
        const float4* src = gsrc + (size_t)it * (kKC * 64 / 4);
        st0 = src[0]; st1 = src[256]; st2 = src[512]; st3 = src[768];
    };
    auto store_chunk = [&](int buf) {
        float* dst = As + buf * kABuf + arow0;
        *reinterpret_cast<float4*>(dst) = st0;
        *reinterpret_cast<float4*>(dst + 32 * kAPitch) = st1;
        *reinterpret_cast<float4*>(dst + 64 * kAPitch) = st2;
        *reinterpret_cast<float4*>(dst + 96 * kAPitch) = st3;
    };
    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    f32x16 ac, as;
    const int alane = (half * 32 + l31) * kAPitch;
    for (int it = 0; it < total; ++it) {
        const int c = it % nchunks;
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ac[r] = 0.f;
                as[r] = 0.f;
            }
        }
        if (it + 1 < total) load_chunk(it + 1);
        const float* acos = As + (it & 1) * kABuf + alane;
        const float* asin = acos + 64 * kAPitch;
        // k-step j of the chunk, k-half h: basis row r = c kKC + 32 h + j, sample k = r + 1 and its mirror n_fft - k
        const float* bf = bcol + 1 + c * kKC + half * 32;
        const float* br = bcol + p.n_fft - 1 - c * kKC - half * 32;
        // groups of 4 k-steps with ping-pong fragment registers: the 6 LDS reads and 8 add / subtracts of group g+1
        // are spread between the 8 matrix instructions of group g (one wave per SIMD here: nothing else hides them)
        constexpr int G = 4;
        f32x4a ca[2], sa[2];
        float be[2][G], bo[2][G];
        auto read_group = [&](int g, int pp) {
            ca[pp] = *reinterpret_cast<const f32x4a*>(acos + G * g);
            sa[pp] = *reinterpret_cast<const f32x4a*>(asin + G * g);
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const float xa = bf[G * g + j];
                const float xr = br[-(G * g + j)];
                be[pp][j] = xa + xr;
                bo[pp][j] = xa - xr;
            }
        };
        read_group(0, 0);
#pragma unroll
        for (int g = 0; g < kKC / 2 / G; ++g) {
            const int pp = g & 1;
            if (g + 1 < kKC / 2 / G) read_group(g + 1, pp ^ 1);
#pragma unroll
            for (int j = 0; j < G; ++j) {
                ac = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[pp][j], be[pp][j], ac, 0, 0, 0);
                as = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[pp][j], bo[pp][j], as, 0, 0, 0);
            }
            if (g + 1 < kKC / 2 / G) {
#pragma unroll
                for (int j = 0; j < 2 * G; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (j < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (it + 1 < total) store_chunk((it + 1) & 1);
        if (c == nchunks - 1 && tok) {
            const int fblk = fblk0 + it / nchunks;
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
        __syncthreads();
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
    if (basis_rows) *basis_rows = (n_fft / 2 + 63) & ~63;  // folded: samples 1 .. n_fft/2, whole LDS chunks of 64 rows
    if (basis_cols) *basis_cols = 2 * fp;
    return AFD_OK;
}

// host: the folded basis, row r = sample k = r + 1 (k = 1 .. n/2; w[0] = 0 drops sample 0): w[k] cos(2 pi f k / n) and
// -w[k] sin(2 pi f k / n), w = periodic Hann (halved on an even n's Nyquist row, whose two folded samples are the same
// one); zero for padded rows / bins.  basis_rows x basis_cols of afd_stft_dims is its size, the order is the kernel's
extern "C" int afd_stft_basis(int n_fft, float* basis) {
    if (!basis || n_fft < 2) return afd::fail(AFD_ERR_ARG, "stft basis: bad argument");
    const int f = n_fft / 2 + 1;
    const int fp = ((f + 31) / 32) * 32;
    const int kfold = n_fft / 2;
    const int kp = (kfold + 63) & ~63;
    const int nchunks = kp / 64;
    const double pi = 3.14159265358979323846;
    // chunk images in the kernel's LDS order: [bin block][chunk][cos | sin][k-half h][bin][k-step j], row r = 64 c + 32 h + j
    for (int fb = 0; fb < fp / 32; ++fb)
        for (int c = 0; c < nchunks; ++c)
            for (int cs = 0; cs < 2; ++cs)
                for (int h = 0; h < 2; ++h)
                    for (int bin = 0; bin < 32; ++bin)
                        for (int j = 0; j < 32; ++j) {
                            const int k = 64 * c + 32 * h + j + 1;
                            const int fi = 32 * fb + bin;
                            double v = 0.0;
                            if (fi < f && k <= kfold) {
                                double w = 0.5 - 0.5 * std::cos(2.0 * pi * k / n_fft);
                                if (2 * k == n_fft) w *= 0.5;
                                const long ph = ((long)fi * k) % n_fft;
                                const double ang = 2.0 * pi * (double)ph / n_fft;
                                v = cs == 0 ? w * std::cos(ang) : -w * std::sin(ang);
                            }
                            basis[((((size_t)(fb * nchunks + c) * 2 + cs) * 2 + h) * 32 + bin) * 32 + j] = (float)v;
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
    // (any number of frames per workgroup works -- lanes past it idle -- so a large hop shrinks the group as far as it must:
    // hop 1024 leaves 28 frames per workgroup)
    constexpr long kSegMax = 29000;
    if (n_fft > kSegMax) return afd::fail(AFD_ERR_UNSUPPORTED, "stft: n_fft %d does not fit LDS", n_fft);
    int tg = 128;
    if ((long)(tg - 1) * hop + n_fft > kSegMax) tg = (int)((kSegMax - n_fft) / hop) + 1;
    const int t32 = ((p.T + 31) / 32) * 32;
    if (t32 < tg) tg = t32;
    p.TG = tg;
    p.seg = (tg - 1) * hop + n_fft;
    p.segpad = (p.seg + 3) & ~3;
    const size_t lds = ((size_t)p.segpad + 2 * kMargin + 2 * kABuf) * sizeof(float);
    static afd::PerDeviceOnce attr_set;
    if (!attr_set.done()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int waves = tg / 32;
    afd::ScopedTiming timing(AFD_K_STFT, 4.0 * B * ((double)N + (double)p.F * p.T), s);
    // bin blocks per workgroup
    const int nblk = p.FP / 32, tgroups = (p.T + tg - 1) / tg;
    // cost model in units of one block's (folded) matrix work: rounds over the CUs x (staging ~1.4 + blocks)
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (ncu < 1) ncu = 256;
    int fb = 1;
    double best = 1e30;
    for (int cand = 1; cand <= nblk; ++cand) {
        const long wgs = (long)((nblk + cand - 1) / cand) * tgroups * B;
        const double cost = (double)((wgs + ncu - 1) / ncu) * (1.4 + cand);
        if (cost < best) { best = cost; fb = cand; }
    }
    p.FB = fb;
    // matrix flops the launch issues: (cos + sin) columns x folded rows x frames, padded tiles included
    timing.issued(2.0 * (2.0 * p.FP) * (double)p.KP * ((double)tgroups * tg) * B);
    hipLaunchKernelGGL(stft_mfma_kernel, dim3((nblk + fb - 1) / fb, tgroups, B), dim3(256), lds, s, p);
    (void)waves;
    return afd::check_launch("stft_mfma_kernel");
}
