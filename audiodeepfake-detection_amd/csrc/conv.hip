// 2-D convolution (stride 1, square kernel, dilation) for the DCNN / LCNN blocks on gfx950.
//
// Replaces the cuDNN forward / backward-data / backward-weight launches behind
// nn.Conv2d in the reference (src/audiofakedetect/models.py:255-291) with implicit-GEMM
// kernels on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (fp32 in, fp32
// accumulate: logits must stay within 1e-4 of the fp32 CPU reference, so no bf16 here).
//
//   forward / backward-data (same kernel, weights repacked flipped+transposed for dgrad):
//     M = output channels (32 per wave), N = output pixels (NT x 32 per wave),
//     K = (input channel, ky, kx) in chunks of CI_T channels.  Per chunk the workgroup
//     stages the repacked weights [k][cout] and the input patch [ci][rows][cols] in LDS;
//     B fragments are read straight from the patch at koff[k] + pixel offset (no im2col
//     buffer), A fragments from the weight slab.  NCHW fp32 everywhere.
//   backward-weight:
//     M = output channels, N = (ci, ky, kx) columns, K = output pixels; each workgroup
//     walks a strided list of 64-pixel tiles, accumulates in registers, writes one partial
//     slab; a second kernel sums the slabs (deterministic, no float atomics) and also
//     produces the bias gradient.
//
// Pixel tiles come in two shapes chosen on the host: RECT (TH x TW rectangles, x-windowed
// patch; for the 16 386-wide level-14 images) and FLAT (128 consecutive pixels of the
// flattened image, full-width patch rows; no tail waste on 129-wide images).
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kLdsBudget = 52 * 1024;  // bytes per workgroup -> 3 workgroups per CU

struct ConvGeom {
    int N, Cin, H, W, Cout, K, pad, dil, Hout, Wout;
    int mode;        // 0 RECT, 1 FLAT
    int TWlog2, TH;  // RECT tile
    int PIX;         // pixels per tile
    int tilesX, tilesY;
    int PR, PC;      // patch rows / cols
    int CI_T, nchunks, CKP, CO_PAD;
    int patchFloats;
    int MW, NW, WNB;  // wave tile (MW x NW 32x32 tiles), waves along the pixel axis
    int fast_stage;   // 1: a chunk's staging loads fit one register round per thread
    int two_level;
};

__device__ __forceinline__ void tile_origin(const ConvGeom& g, int t, int& oy0, int& ox0,
                                            int& p0) {
    if (g.mode == 0) {
        const int tyi = t / g.tilesX;
        const int txi = t - tyi * g.tilesX;
        oy0 = tyi * g.TH;
        ox0 = txi << g.TWlog2;
        p0 = 0;
    } else {
        p0 = t * g.PIX;
        oy0 = p0 / g.Wout;
        ox0 = 0;
    }
}

__device__ __forceinline__ void tile_pixel(const ConvGeom& g, int pj, int oy0, int ox0, int p0,
                                           int& oy, int& ox) {
    if (g.mode == 0) {
        oy = oy0 + (pj >> g.TWlog2);
        ox = ox0 + (pj & ((1 << g.TWlog2) - 1));
    } else {
        const int p = p0 + pj;
        oy = p / g.Wout;
        ox = p - oy * g.Wout;
    }
}

// Stage the input patch [CI_T][PR][PC]: columns go to lanes (coalesced along x), rows to
// waves in groups of R.  All R loads of a group are issued before the first LDS store, so a
// wave keeps R global loads in flight (a load-store-load-store chain is latency bound).
template <int R>
__device__ __forceinline__ void stage_patch(const ConvGeom& g, const float* __restrict__ x, int n,
                                            int chunk, int iy0, int ix0, float* patch,
                                            int nthreads) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = nthreads >> 6;
    const int rows = g.CI_T * g.PR;
    const size_t plane = (size_t)g.H * g.W;
    const float* xn = x + (size_t)n * g.Cin * plane;
    for (int pc0 = 0; pc0 < g.PC; pc0 += 64) {
        const int pc = pc0 + lane;
        const int ix = ix0 + pc;
        const bool colok = pc < g.PC;
        const bool inx = colok && ix >= 0 && ix < g.W;
        for (int row0 = wave * R; row0 < rows; row0 += nwaves * R) {
            float v[R];
            int ci_l = row0 / g.PR;
            int pr = row0 - ci_l * g.PR;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int ci = chunk * g.CI_T + ci_l;
                const int iy = iy0 + pr;
                const bool ok = inx && (row0 + r < rows) && (ci < g.Cin) && (iy >= 0) && (iy < g.H);
                v[r] = ok ? xn[(size_t)ci * plane + (size_t)iy * g.W + ix] : 0.f;
                if (++pr == g.PR) {
                    pr = 0;
                    ++ci_l;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (colok && row0 + r < rows) patch[(row0 + r) * g.PC + pc] = v[r];
        }
    }
}

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

// Register prefetch of one channel chunk (weight slab + input patch): the loads are issued
// right after the barrier that publishes the PREVIOUS chunk and stay in flight behind its
// MFMAs; they are written to LDS one phase later (issue-early / write-late).
constexpr int kZV = 16;  // wgrad: dz floats per thread (rows per wave)
constexpr int kPW = 18;  // wgrad: patch floats per thread

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

constexpr int kWV = 6;   // float4 of weights per thread
constexpr int kPV = 12;  // patch floats per thread

__device__ __forceinline__ void prefetch_chunk(const ConvGeom& g, const float* __restrict__ xn,
                                               const float* __restrict__ wp, int chunk, int iy0,
                                               int ix0, float invPC, float invPR, int tid,
                                               int nthreads, float4 (&w)[kWV], float (&p)[kPV]) {
    const int wslab = g.CKP * g.CO_PAD;
    const int nvec = wslab >> 2;
    const float4* src = reinterpret_cast<const float4*>(wp + (size_t)chunk * wslab);
#pragma unroll
    for (int u = 0; u < kWV; ++u) {
        const int i = tid + u * nthreads;
        w[u] = i < nvec ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int total = g.CI_T * g.PR * g.PC;
    const size_t plane = (size_t)g.H * g.W;
#pragma unroll
    for (int u = 0; u < kPV; ++u) {
        const int e = tid + u * nthreads;
        int row, pc, ci_l, pr;
        divmod_small(e, g.PC, invPC, row, pc);
        divmod_small(row, g.PR, invPR, ci_l, pr);
        const int ci = chunk * g.CI_T + ci_l;
        const int iy = iy0 + pr;
        const int ix = ix0 + pc;
        const bool ok = (e < total) && (ci < g.Cin) && (iy >= 0) && (iy < g.H) && (ix >= 0) && (ix < g.W);
        p[u] = ok ? xn[(size_t)ci * plane + (size_t)iy * g.W + ix] : 0.f;
    }
}

__device__ __forceinline__ void store_chunk(const ConvGeom& g, float* wlds, float* patch, int tid,
                                            int nthreads, const float4 (&w)[kWV],
                                            const float (&p)[kPV]) {
    const int nvec = (g.CKP * g.CO_PAD) >> 2;
    float4* dst = reinterpret_cast<float4*>(wlds);
#pragma unroll
    for (int u = 0; u < kWV; ++u) {
        const int i = tid + u * nthreads;
        if (i < nvec) dst[i] = w[u];
    }
    const int total = g.CI_T * g.PR * g.PC;
#pragma unroll
    for (int u = 0; u < kPV; ++u) {
        const int e = tid + u * nthreads;
        if (e < total) patch[e] = p[u];
    }
}

// ---------------------------------------------------------------------------------------
// forward / backward-data
// ---------------------------------------------------------------------------------------
constexpr int igemm_min_waves(int mw, int nw, bool two) {
    const int acc = mw * nw * 16 * (two ? 2 : 1);
    return acc <= 64 ? 4 : (acc <= 100 ? 3 : 2);
}

template <int MW, int NW, bool TWO>
__global__ void __launch_bounds__(512)
conv_igemm_kernel(const ConvGeom g, const float* __restrict__ x, const float* __restrict__ wp,
                  const float* __restrict__ bias, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wlds = smem;
    float* patch = wlds + g.CKP * g.CO_PAD;
    int* koff = reinterpret_cast<int*>(patch + g.patchFloats);

    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int WMB = (g.CO_PAD >> 5) / MW;  // waves along the channel axis
    const int wmb = wave % WMB;
    const int wnb = wave / WMB;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int tpi = g.tilesX * g.tilesY;
    const int n = blockIdx.x / tpi;
    const int t = blockIdx.x - n * tpi;
    int oy0, ox0, p0;
    tile_origin(g, t, oy0, ox0, p0);
    const int iy0 = oy0 - g.pad;
    const int ix0 = ox0 - g.pad;

    int pbase[NW];
    int oyv[NW], oxv[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int pj = (wnb * NW + i) * 32 + l31;
        int oy, ox;
        tile_pixel(g, pj, oy0, ox0, p0, oy, ox);
        const bool ok = (oy < g.Hout) && (ox < g.Wout);
        pbase[i] = ok ? (oy - oy0) * g.PC + (ox - ox0) : 0;
        oyv[i] = ok ? oy : -1;
        oxv[i] = ox;
    }

    const int KK = g.K * g.K;
    for (int kl = tid; kl < g.CKP + 6; kl += nthreads) {
        int o = 0;
        if (kl < g.CI_T * KK) {
            const int ci_l = kl / KK;
            const int r = kl - ci_l * KK;
            const int ky = r / g.K;
            const int kx = r - ky * g.K;
            o = ci_l * g.PR * g.PC + ky * g.dil * g.PC + kx * g.dil;
        }
        koff[kl] = o;
    }

    f32x16 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;

    const float invPC = 1.0f / (float)g.PC;
    const float invPR = 1.0f / (float)g.PR;
    const float* xn = x + (size_t)n * g.Cin * g.H * g.W;
    const int wslab = g.CKP * g.CO_PAD;
    for (int chunk = 0; chunk < g.nchunks; ++chunk) {
        __syncthreads();  // the previous chunk's fragments have been read
        if (g.fast_stage) {
            // every staging load of this thread is in flight before the first LDS store:
            // one global-latency round per chunk instead of one per row group
            float4 pfw[kWV];
            float pfp[kPV];
            prefetch_chunk(g, xn, wp, chunk, iy0, ix0, invPC, invPR, tid, nthreads, pfw, pfp);
            store_chunk(g, wlds, patch, tid, nthreads, pfw, pfp);
        } else {
            const float4* src = reinterpret_cast<const float4*>(wp + (size_t)chunk * wslab);
            float4* dst = reinterpret_cast<float4*>(wlds);
            for (int i = tid; i < (wslab >> 2); i += nthreads) dst[i] = src[i];
            stage_patch<6>(g, x, n, chunk, iy0, ix0, patch, nthreads);
        }
        __syncthreads();
        const float* arow = wlds + wmb * MW * 32 + l31;
        const int ksteps = g.CKP >> 1;
        if (TWO) {
            // two-level accumulation: a fresh fp32 chain per channel chunk, then one add --
            // keeps the rounding error of K = Cin*K*K <= 1152 products near sqrt(chunk) * eps
            f32x16 part[MW][NW];
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int i = 0; i < NW; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[m][i][r] = 0.f;
#pragma unroll 4
            for (int ks = 0; ks < ksteps; ++ks) {
                const int k = 2 * ks + half;
                const int off = koff[k];
                float a[MW], b[NW];
#pragma unroll
                for (int m = 0; m < MW; ++m) a[m] = arow[k * g.CO_PAD + m * 32];
#pragma unroll
                for (int i = 0; i < NW; ++i) b[i] = patch[off + pbase[i]];
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int i = 0; i < NW; ++i)
                        part[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[i], part[m][i], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int i = 0; i < NW; ++i) acc[m][i] += part[m][i];
        } else {
#pragma unroll 4
            for (int ks = 0; ks < ksteps; ++ks) {
                const int k = 2 * ks + half;
                const int off = koff[k];
                float a[MW], b[NW];
#pragma unroll
                for (int m = 0; m < MW; ++m) a[m] = arow[k * g.CO_PAD + m * 32];
#pragma unroll
                for (int i = 0; i < NW; ++i) b[i] = patch[off + pbase[i]];
#pragma unroll
                for (int m = 0; m < MW; ++m)
#pragma unroll
                    for (int i = 0; i < NW; ++i)
                        acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[i], acc[m][i], 0, 0, 0);
            }
        }
    }

    // D layout: column = lane & 31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const size_t plane = (size_t)g.Hout * g.Wout;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        if (oyv[i] < 0) continue;
        const size_t pix = (size_t)oyv[i] * g.Wout + oxv[i];
#pragma unroll
        for (int m = 0; m < MW; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (wmb * MW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < g.Cout) {
                    const float bv = bias ? bias[co] : 0.f;
                    y[((size_t)n * g.Cout + co) * plane + pix] = acc[m][i][r] + bv;
                }
            }
        }
    }
}

// weights [Cout][Cin][K][K] -> slabs [chunk][CKP][CO_PAD] (k-major, cout fastest)
__global__ void repack_fwd_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin,
                                  int Cout, int KK, int CI_T, int nchunks, int CKP, int CO_PAD) {
    const int total = nchunks * CKP * CO_PAD;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % CO_PAD;
        const int kl = (i / CO_PAD) % CKP;
        const int c = i / (CO_PAD * CKP);
        const int ci = c * CI_T + kl / KK;
        const int r = kl % KK;
        float v = 0.f;
        if (kl < CI_T * KK && ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * KK + r];
        wp[i] = v;
    }
}

// backward-data: the "input channels" are the forward's Cout, the "output channels" its
// Cin, taps flipped: wp[c][kl][ci] = w[co][ci][KK-1-r], co = c*CI_T + kl/KK
__global__ void repack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin,
                                    int Cout, int KK, int CI_T, int nchunks, int CKP,
                                    int CO_PAD) {
    const int total = nchunks * CKP * CO_PAD;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i % CO_PAD;
        const int kl = (i / CO_PAD) % CKP;
        const int c = i / (CO_PAD * CKP);
        const int co = c * CI_T + kl / KK;
        const int r = kl % KK;
        float v = 0.f;
        if (kl < CI_T * KK && co < Cout && ci < Cin)
            v = w[((size_t)co * Cin + ci) * KK + (KK - 1 - r)];
        wp[i] = v;
    }
}

// RECT (1 x pix rectangles) for wide images, FLAT (pix consecutive pixels) otherwise
void set_tiling(ConvGeom& g, int pix, bool rect) {
    const int halo = (g.K - 1) * g.dil;
    g.PIX = pix;
    if (rect) {
        g.mode = 0;
        g.TWlog2 = 0;
        while ((1 << (g.TWlog2 + 1)) <= pix) ++g.TWlog2;
        g.TH = 1;
        g.tilesX = (g.Wout + pix - 1) / pix;
        g.tilesY = g.Hout;
        g.PR = 1 + halo;
        g.PC = pix + halo;
    } else {
        g.mode = 1;
        g.TWlog2 = 0;
        g.TH = 0;
        const long total = (long)g.Hout * g.Wout;
        g.tilesX = (int)((total + pix - 1) / pix);
        g.tilesY = 1;
        const int rows = (pix + g.Wout - 2) / g.Wout + 1;
        g.PR = (rows < g.Hout ? rows : g.Hout) + halo;
        g.PC = g.Wout + halo;
    }
}

int ilog2_floor(int v) {
    int l = 0;
    while ((1 << (l + 1)) <= v) ++l;
    return l;
}

// wave tiling per padded channel count: (MW x NW) 32x32 tiles per wave, WNB waves along pixels
struct IgemmCfg {
    int MW, NW, WNB, two;
};

IgemmCfg pick_cfg(int co_pad) {
    IgemmCfg c;
    switch (co_pad / 32) {
        case 1: c = {1, 2, 4, 1}; break;   // 32 ch  x 256 px, 4 waves
        case 2: c = {1, 2, 2, 1}; break;   // 64 ch  x 128 px, 2 x 2 waves
        case 3: c = {3, 1, 4, 1}; break;   // 96 ch  x 128 px, 4 waves
        default: c = {2, 2, 2, 1}; break;  // 128 ch x 128 px, 2 x 2 waves
    }
    return c;
}

// geometry for a conv whose input is [N][Cin][H][W] and output [N][Cout][Hout][Wout]
int plan_igemm(ConvGeom& g, int N, int Cin, int H, int W, int Cout, int K, int pad, int dil,
               int Hout, int Wout) {
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.K = K; g.pad = pad; g.dil = dil;
    g.Hout = Hout; g.Wout = Wout;
    g.CO_PAD = ((Cout + 31) / 32) * 32;
    const IgemmCfg cfg = pick_cfg(g.CO_PAD);
    g.MW = cfg.MW; g.NW = cfg.NW; g.WNB = cfg.WNB; g.two_level = cfg.two;
    const int pix = cfg.NW * cfg.WNB * 32;
    g.PIX = pix;
    const int nthreads = (g.CO_PAD / 32 / cfg.MW) * cfg.WNB * 64;
    const int KK = K * K;
    int best = 0, best_fast = 0;
    for (int attempt = 0; attempt < 2 && best == 0; ++attempt) {
        const bool rect = (Wout >= 4 * pix) != (attempt == 1);
        if (rect && Wout < pix) continue;
        set_tiling(g, pix, rect);
        // largest channel chunk whose weights + patch fit the LDS budget
        for (int ct = 1; ct <= Cin && ct <= 32; ++ct) {
            const int ckp = (ct * KK + 3) & ~3;
            const long bytes = 4L * ((long)ckp * g.CO_PAD + (long)ct * g.PR * g.PC + 12 + ckp + 6L * g.CO_PAD);
            const bool regs_ok = ((long)ckp * g.CO_PAD / 4 <= (long)kWV * nthreads) &&
                                 ((long)ct * g.PR * g.PC <= (long)kPV * nthreads);
            if (bytes <= kLdsBudget) {
                best = ct;
                if (regs_ok) best_fast = ct;
            }
        }
    }
    if (best == 0) return afd::fail(AFD_ERR_UNSUPPORTED, "conv: tile does not fit LDS (K=%d dil=%d W=%d)", K, dil, W);
    g.fast_stage = 0;
    if (best_fast * 2 >= best || best_fast >= 8) {  // one-round staging unless it halves the chunk
        best = best_fast;
        g.fast_stage = 1;
    }
    // prefer a chunk size that divides Cin (no zero-padded k rows)
    int ct = best;
    for (int c = best; c >= 1 && c * 2 > best; --c)
        if (Cin % c == 0) { ct = c; break; }
    g.CI_T = ct;
    g.nchunks = (Cin + ct - 1) / ct;
    g.CKP = (ct * KK + 3) & ~3;
    g.patchFloats = ((ct * g.PR * g.PC + 3) / 4) * 4;
    return AFD_OK;
}

size_t igemm_lds_bytes(const ConvGeom& g) {
    return 4 * ((size_t)g.CKP * g.CO_PAD + g.patchFloats + g.CKP + 8 + 6 * (size_t)g.CO_PAD);
}

size_t repack_floats(const ConvGeom& g) { return (size_t)g.nchunks * g.CKP * g.CO_PAD; }

template <int MW, int NW>
void launch_igemm_t(const ConvGeom& g, const float* x, const float* wp, const float* bias, float* y,
                    unsigned blocks, int threads, hipStream_t s) {
    if (g.two_level)
        hipLaunchKernelGGL((conv_igemm_kernel<MW, NW, true>), dim3(blocks), dim3(threads),
                           igemm_lds_bytes(g), s, g, x, wp, bias, y);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<MW, NW, false>), dim3(blocks), dim3(threads),
                           igemm_lds_bytes(g), s, g, x, wp, bias, y);
}

int launch_igemm(const ConvGeom& g, const float* x, const float* wp, const float* bias, float* y,
                 hipStream_t s) {
    const int threads = (g.CO_PAD / 32 / g.MW) * g.WNB * 64;
    if (threads > 512 || threads < 64) return afd::fail(AFD_ERR_UNSUPPORTED, "conv: Cout %d too large", g.Cout);
    const long blocks = (long)g.N * g.tilesX * g.tilesY;
    if (blocks > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv: grid too large");
    afd::ScopedTiming timing(AFD_K_CONV_IGEMM,
                             2.0 * g.N * g.Cout * (double)g.Hout * g.Wout * g.Cin * g.K * g.K, s);
    // lower bound of the issued matrix flops: channel-tile padding counted, pixel- and k-tile padding not
    timing.issued(2.0 * g.N * g.CO_PAD * (double)g.Hout * g.Wout * g.Cin * g.K * g.K);
    timing.bytes(4.0 * g.N * ((double)g.Cin * g.H * g.W + (double)g.Cout * g.Hout * g.Wout));
    const int key = g.MW * 10 + g.NW;
    switch (key) {
        case 12: launch_igemm_t<1, 2>(g, x, wp, bias, y, (unsigned)blocks, threads, s); break;
        case 14: launch_igemm_t<1, 4>(g, x, wp, bias, y, (unsigned)blocks, threads, s); break;
        case 21: launch_igemm_t<2, 1>(g, x, wp, bias, y, (unsigned)blocks, threads, s); break;
        case 22: launch_igemm_t<2, 2>(g, x, wp, bias, y, (unsigned)blocks, threads, s); break;
        case 31: launch_igemm_t<3, 1>(g, x, wp, bias, y, (unsigned)blocks, threads, s); break;
        case 41: launch_igemm_t<4, 1>(g, x, wp, bias, y, (unsigned)blocks, threads, s); break;
        default: return afd::fail(AFD_ERR_UNSUPPORTED, "conv: wave tile %dx%d not built", g.MW, g.NW);
    }
    return afd::check_launch("conv_igemm_kernel");
}

// ---------------------------------------------------------------------------------------
// backward-weight
// ---------------------------------------------------------------------------------------
struct WgradGeom {
    ConvGeom c;    // tile geometry over the forward OUTPUT pixels (PIX = 64), patch over input
    int MT, NG;    // waves = MT * NG
    int NTILES;    // column tiles = ceil(CI_T*KK / 32)
    int NCOL;      // NTILES * 32
    int S;         // splits
    int PIXP;      // padded dz row
    int fast_stage;
    long totalTiles;
};

template <int NTW>
__global__ void __launch_bounds__(512)
conv_wgrad_kernel(const WgradGeom wg, const float* __restrict__ x, const float* __restrict__ dz,
                  float* __restrict__ part, float* __restrict__ partb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = wg.c;
    float* dzl = smem;                                  // [CO_PAD][PIXP]
    float* patch = dzl + g.CO_PAD * wg.PIXP;            // [CI_T][PR][PC]
    int* pixoff = reinterpret_cast<int*>(patch + g.patchFloats);  // [PIX]
    float* bsum_lds = reinterpret_cast<float*>(pixoff + g.PIX + 8);  // [CO_PAD]; pixoff padded

    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = nthreads >> 6;
    const float invPC = 1.0f / (float)g.PC;
    const float invPR = 1.0f / (float)g.PR;
    for (int i = tid; i < g.CO_PAD; i += nthreads) bsum_lds[i] = 0.f;
    if (tid < 8) pixoff[g.PIX + tid] = 0;
    const int m = wave % wg.MT;
    const int grp = wave / wg.MT;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int chunk = blockIdx.y;
    const int split = blockIdx.x;
    const int KK = g.K * g.K;

    int joff[NTW];
    bool nvalid[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int nt = grp + wg.NG * i;
        nvalid[i] = nt < wg.NTILES;
        const int c = nt * 32 + l31;
        int o = 0;
        if (c < g.CI_T * KK) {
            const int ci_l = c / KK;
            const int r = c - ci_l * KK;
            const int ky = r / g.K;
            const int kx = r - ky * g.K;
            o = ci_l * g.PR * g.PC + ky * g.dil * g.PC + kx * g.dil;
        }
        joff[i] = o;
    }

    f32x16 acc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int tpi = g.tilesX * g.tilesY;
    for (long tile = split; tile < wg.totalTiles; tile += wg.S) {
        const int n = (int)(tile / tpi);
        const int t = (int)(tile - (long)n * tpi);
        int oy0, ox0, p0;
        tile_origin(g, t, oy0, ox0, p0);
        __syncthreads();
        {
        int oy, ox;
        tile_pixel(g, lane, oy0, ox0, p0, oy, ox);
        const bool pok = (oy < g.Hout) && (ox < g.Wout);
        const size_t oplane = (size_t)g.Hout * g.Wout;
        const float* dzn = dz + (size_t)n * g.Cout * oplane + (pok ? (size_t)oy * g.Wout + ox : 0);
        if (wave == 0) pixoff[lane] = pok ? (oy - oy0) * g.PC + (ox - ox0) : 0;
        // rows wave, wave + nwaves, ...: up to 16 loads in flight per wave, then the stores
        for (int j0 = 0; wave + j0 * nwaves < g.CO_PAD; j0 += 16) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = wave + (j0 + r) * nwaves;
                v[r] = (pok && co < g.Cout) ? dzn[(size_t)co * oplane] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = wave + (j0 + r) * nwaves;
                if (co < g.CO_PAD) dzl[co * wg.PIXP + lane] = v[r];
            }
        }
        stage_patch<8>(g, x, n, chunk, oy0 - g.pad, ox0 - g.pad, patch, nthreads);
        __syncthreads();
        if (chunk == 0 && tid < g.CO_PAD) {
            const float* row = dzl + tid * wg.PIXP;
            float sm = 0.f;
            for (int pj = 0; pj < g.PIX; ++pj) sm += row[pj];
            bsum_lds[tid] += sm;
        }
        }
        __syncthreads();
        const float* arow = dzl + (m * 32 + l31) * wg.PIXP;
        const int ksteps = g.PIX >> 1;  // 32
        // two-level accumulation (per 64-pixel tile, then across tiles) while the second
        // accumulator set fits (NTW <= 3); see the forward kernel
        constexpr bool kTwo = NTW <= 3;
        f32x16 part[kTwo ? NTW : 1];
        if (kTwo) {
#pragma unroll
            for (int i = 0; i < NTW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[i][r] = 0.f;
        }
        // Two-deep software pipeline with ping-pong fragment registers: fragments of pixel
        // step s+1 and the patch offset of step s+2 are requested before the MFMAs of step s
        // (hipcc otherwise emits read -> wait -> MFMA chains; with 1.5-3 waves per SIMD the
        // other waves do not cover that).  pixoff is padded, dz rows have slack: the tail
        // prefetches in-bounds garbage, no branches.
        float a0, a1, b0[NTW], b1[NTW];
        {
            const int po0 = pixoff[half];
            a0 = arow[half];
#pragma unroll
            for (int i = 0; i < NTW; ++i) b0[i] = patch[joff[i] + po0];
        }
        int pon = pixoff[2 + half];
        for (int ks = 0; ks < ksteps; ks += 2) {
            const int k1 = 2 * ks + 2 + half;
            a1 = arow[k1];
#pragma unroll
            for (int i = 0; i < NTW; ++i) b1[i] = patch[joff[i] + pon];
            const int ponn = pixoff[k1 + 2];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                if (kTwo) part[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[i], part[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0[i], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int k2 = k1 + 2;
            a0 = arow[k2];
#pragma unroll
            for (int i = 0; i < NTW; ++i) b0[i] = patch[joff[i] + ponn];
            pon = pixoff[k2 + 2];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                if (kTwo) part[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[i], part[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[i], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kTwo) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i] += part[i];
        }
    }

    float* slab = part + ((size_t)split * gridDim.y + chunk) * g.CO_PAD * wg.NCOL;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if (!nvalid[i]) continue;
        const int nt = grp + wg.NG * i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[(size_t)co * wg.NCOL + nt * 32 + l31] = acc[i][r];
        }
    }
    __syncthreads();
    if (chunk == 0 && tid < g.CO_PAD) partb[(size_t)split * g.CO_PAD + tid] = bsum_lds[tid];
}

// Sums the S partial slabs.  Block = 32 gradient elements x 8 slab groups: lane group g sums
// slabs g, g+8, ... with four independent chains (loads in flight), LDS combines the groups --
// a one-thread-per-element loop over S strided loads is a pure latency chain.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ partb,
                    float* __restrict__ dw, float* __restrict__ db, int Cin, int Cout, int KK,
                    int CI_T, int nchunks, int CO_PAD, int NCOL, int S, int swapped = 0, int SB = 0) {
    __shared__ float red[8][33];
    const int total = Cout * Cin * KK;
    const int e = threadIdx.x & 31;
    const int g = threadIdx.x >> 5;
    const int nblk_w = (total + 31) / 32;
    const size_t stride = (size_t)nchunks * CO_PAD * NCOL;
    float sum = 0.f;
    bool valid = false;
    const float* p = nullptr;
    size_t step = 0;
    int out_i = -1;
    if ((int)blockIdx.x < nblk_w) {
        const int i = blockIdx.x * 32 + e;
        if (i < total) {
            const int r = i % KK;
            const int ci = (i / KK) % Cin;
            const int co = i / (KK * Cin);
            if (!swapped) {
                const int chunk = ci / CI_T;
                const int col = (ci - chunk * CI_T) * KK + r;
                p = part + ((size_t)chunk * CO_PAD + co) * NCOL + col;
            } else {
                // slabs of the role-swapped product D[ci][co][t'] = sum x[ci][q] dy[co][q + t']: dw[co][ci][t] = D[ci][co][8 - t]
                const int chunk = co / CI_T;
                const int col = (co - chunk * CI_T) * KK + (KK - 1 - r);
                p = part + ((size_t)chunk * CO_PAD + ci) * NCOL + col;
            }
            step = stride;
            valid = true;
            out_i = i;
        }
    } else if (db) {  // trailing blocks: bias gradient
        const int i = (blockIdx.x - nblk_w) * 32 + e;
        if (i < Cout) {
            p = partb + i;
            step = swapped ? (size_t)Cout : (size_t)CO_PAD;
            valid = true;
            out_i = i;
            if (swapped) S = SB;  // partial sums of bias_partial_kernel
        }
    }
    if (valid) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = g;
        for (; k + 24 < S; k += 32) {
            s0 += p[(size_t)k * step];
            s1 += p[(size_t)(k + 8) * step];
            s2 += p[(size_t)(k + 16) * step];
            s3 += p[(size_t)(k + 24) * step];
        }
        for (; k < S; k += 8) s0 += p[(size_t)k * step];
        sum = (s0 + s1) + (s2 + s3);
    }
    red[g][e] = sum;
    __syncthreads();
    if (g == 0 && valid) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j][e];
        if ((int)blockIdx.x < nblk_w) dw[out_i] = t;
        else db[out_i] = t;
    }
}

// Bias gradient of the role-swapped backward-weight product (below): block (s, c) sums channel c of the images
// s, s + SB, ... into partb[s][c]; wgrad_reduce_kernel's trailing blocks add the SB partial sums in a fixed order.
__global__ void __launch_bounds__(256)
bias_partial_kernel(const float* __restrict__ dy, float* __restrict__ partb, int N, int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.y, SB = gridDim.x;
    const int nv = HW / 4;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = blockIdx.x; n < N; n += SB) {
        const float4* pl = reinterpret_cast<const float4*>(dy + ((size_t)n * C + c) * HW);
        int i = threadIdx.x;
        for (; i + 768 < nv; i += 1024) {  // four 16-byte loads in flight per thread
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = pl[i + 256 * q];
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] += (v[q].x + v[q].y) + (v[q].z + v[q].w);
        }
        for (; i < nv; i += 256) {
            const float4 u = pl[i];
            a[0] += (u.x + u.y) + (u.z + u.w);
        }
    }
    float v = (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partb[(size_t)blockIdx.x * C + c] = (red[0] + red[1]) + (red[2] + red[3]);
}

constexpr int kBiasSplits = 32;

// the bias gradient handed over by the producer of dy (afd_bn_backward_apply_sums): double sums -> db
__global__ void bias_from_sums_kernel(const double* __restrict__ sums, float* __restrict__ db, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) db[i] = (float)sums[i];
}

// One 32-channel tile of Cout against 64..128 input channels: the product is computed with the roles swapped --
// M = Cin (x rows as the staged operand), N = (Cout, tap) columns built from dy with its zero halo -- because a
// single M tile reads one LDS operand pair per matrix instruction (measured 128 -> 32 channels at level 14:
// 2.83 ms as is, 2.08 ms swapped; 64 -> 32: 1.42 / 1.11 ms).
bool wgrad_swap_applicable(int Cin, int H, int W, int Cout, int K, int pad, int dil, int dy_rows, int dy_cols) {
    if (getenv("AFD_NO_WGRAD_SWAP")) return false;
    if (K != 3 || pad != 1 || dil != 1 || dy_rows < H || dy_cols < W) return false;
    if (Cin % 32 != 0 || Cin > 128 || Cout % 32 != 0 || ((size_t)H * W) % 4 != 0) return false;
    // measured: one channel tile of Cout, and 96 -> 128 channels, whose swapped form runs on the 64-channel workgroups
    const bool pays = (Cout == 32 && Cin >= 64) || (Cin == 96 && Cout == 128);
    if (!pays) return false;
    return afd::wgrad3x3_applicable(Cout, H, W, Cin, K, pad, dil);
}

// The Winograd-domain backward-weight (wino44_wgrad.hip) wherever it applies: measured faster than the direct
// kernels on the level-14 images (block 3: 10.9 -> 6.8 ms) and on the level-8 / STFT ones (backward-weight class
// 2.0-2.2 -> 1.5-1.6 ms per step); AFD_NO_WINO44_WGRAD=1 puts the direct kernels back.
bool wino_wgrad_pays(int Cin, int H, int W, int Cout, int K, int pad, int dil) {
    return afd::wino44_wgrad_applicable(Cin, H, W, Cout, K, pad, dil);
}

int plan_wgrad(WgradGeom& wg, int N, int Cin, int H, int W, int Cout, int K, int pad, int dil) {
    ConvGeom& g = wg.c;
    const int Hout = H + 2 * pad - dil * (K - 1);
    const int Wout = W + 2 * pad - dil * (K - 1);
    const int pix = 64;
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.K = K; g.pad = pad; g.dil = dil;
    g.Hout = Hout; g.Wout = Wout; g.PIX = pix;
    g.CO_PAD = ((Cout + 31) / 32) * 32;
    const int KK = K * K;
    wg.PIXP = pix + 1;
    wg.MT = g.CO_PAD / 32;
    const int ngmax = 8 / wg.MT > 1 ? 8 / wg.MT : 1;
    const long fixed = 4L * ((long)g.CO_PAD * wg.PIXP + pix + 16 + g.CO_PAD);
    int best = 0;
    for (int attempt = 0; attempt < 2 && best == 0; ++attempt) {
        const bool rect = (Wout >= 4 * pix) != (attempt == 1);
        if (rect && Wout < pix) continue;
        set_tiling(g, pix, rect);
        // LDS pitch of the patch: the 32 lanes of a B read are 32 different (ci,ky,kx)
        // columns at offsets ci*PR*PC + ky*PC + kx; PC = 3 (mod 32) for 3x3 (and an odd
        // channel stride for 1x1) spreads them over 32 banks.  The extra columns are staged
        // like any other input column and never read.
        if (K == 3 && dil == 1) g.PC += (3 - g.PC % 32 + 32) % 32;
        if (K == 1 && ((g.PR * g.PC) & 1) == 0) g.PC += 1;
        for (int ct = 1; ct <= Cin && ct <= 32; ++ct) {
            const long bytes = fixed + 4L * ct * g.PR * g.PC;
            if (bytes <= 72 * 1024 && (ct * KK + 31) / 32 <= 9 * ngmax) best = ct;
        }
    }
    if (best == 0) return afd::fail(AFD_ERR_UNSUPPORTED, "wgrad: tile does not fit LDS");
    int ct = best;
    for (int c = best; c >= 1 && c * 2 > best; --c)
        if (Cin % c == 0) { ct = c; break; }
    g.CI_T = ct;
    g.nchunks = (Cin + ct - 1) / ct;
    g.CKP = 0;
    g.patchFloats = ((ct * g.PR * g.PC + 3) / 4) * 4;
    wg.NTILES = (ct * KK + 31) / 32;
    wg.NCOL = wg.NTILES * 32;
    // waves = MT * NG (at most 8); every wave computes NTW = ceil(NTILES / NG) column tiles,
    // tiles past NTILES are wasted MFMAs: take the NG with the fewest wasted slots (ties: more
    // waves), e.g. 9 tiles -> NG 3 x 3 tiles rather than NG 4 x 3 (3 idle slots).
    int NG = 1;
    double best_eff = -1.0;
    for (int cand = 1; cand <= ngmax && cand <= wg.NTILES; ++cand) {
        const int ntw = (wg.NTILES + cand - 1) / cand;
        if (ntw > 9) continue;
        const int bucket = ntw <= 6 ? ntw : 9;
        const double eff = (double)wg.NTILES / (bucket * cand) + 1e-3 * cand;
        if (eff > best_eff) {
            best_eff = eff;
            NG = cand;
        }
    }
    wg.NG = NG;
    {
        const int nthreads = wg.MT * wg.NG * 64;
        (void)nthreads;
        wg.fast_stage = false;  // the register-staged variant measured slower (round 1) and is never selected
    }
    wg.totalTiles = (long)N * g.tilesX * g.tilesY;
    long S = 1024 / g.nchunks;
    if (S < 1) S = 1;
    if (S > wg.totalTiles) S = wg.totalTiles;
    wg.S = (int)S;
    return AFD_OK;
}

size_t wgrad_lds_bytes(const WgradGeom& wg) {
    return 4 * ((size_t)wg.c.CO_PAD * wg.PIXP + wg.c.patchFloats + wg.c.PIX + 8 + wg.c.CO_PAD);
}

size_t wgrad_ws_floats(const WgradGeom& wg) {
    return (size_t)wg.S * wg.c.nchunks * wg.c.CO_PAD * wg.NCOL + (size_t)wg.S * wg.c.CO_PAD;
}

template <int NTW>
int launch_wgrad_t(const WgradGeom& wg, const float* x, const float* dz, float* part, float* partb,
                   hipStream_t s) {
    static afd::PerDeviceOnce attr_set;
    if (!attr_set.done()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<NTW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set.mark();
    }
    afd::ScopedTiming timing(AFD_K_CONV_WGRAD, 2.0 * wg.c.N * wg.c.Cout * (double)wg.c.Hout *
                                                   wg.c.Wout * wg.c.Cin * wg.c.K * wg.c.K, s);
    timing.issued(2.0 * wg.c.N * wg.c.CO_PAD * (double)wg.c.Hout * wg.c.Wout * wg.c.Cin * wg.c.K * wg.c.K);
    timing.bytes(4.0 * wg.c.N * ((double)wg.c.Cin * wg.c.H * wg.c.W + (double)wg.c.Cout * wg.c.Hout * wg.c.Wout));
    hipLaunchKernelGGL(conv_wgrad_kernel<NTW>, dim3(wg.S, wg.c.nchunks), dim3(wg.MT * wg.NG * 64),
                       wgrad_lds_bytes(wg), s, wg, x, dz, part, partb);
    return afd::check_launch("conv_wgrad_kernel");
}

// ---------------------------------------------------------------------------------------
// backward-weight, second generation: producer / consumer waves and a double-buffered LDS
// image.  Waves 0-3 (one per SIMD) only issue MFMAs: the (channel tile, column tile) pairs of
// the chunk are dealt to them, TPW per wave.  Waves 4-7 only move data: while tile t is in
// the MFMAs they stage tile t+1 (dz rows, input patch with zero halo, pixel offsets) into the
// other buffer and keep the bias sums.  One barrier per tile; the loaders' address arithmetic
// and memory latency run in the issue slots the 64-cycle MFMAs leave free.
// ---------------------------------------------------------------------------------------
constexpr int kW2Threads = 512;
constexpr int kW2Loaders = 256;
constexpr int kW2MaxTpw = 7;
constexpr int kW2MaxDz = 8;      // dz groups of 4 pixels per loader thread (CO_PAD * 16 / 256)
constexpr int kW2MaxPatch = 14;  // patch groups of 4 columns per loader thread

struct Wgrad2Geom {
    WgradGeom w;   // c, MT, NTILES, NCOL, S, PIXP, totalTiles as in the first generation
    int PAIRS;     // MT * NTILES
    int TPW;       // pairs per MFMA wave
    int bufFloats; // one LDS buffer: dz [CO_PAD][PIXP] | patch | pixoff [PIX + 8]
    int nDz;       // dz rows staged per loader thread = CO_PAD / 4
    int nPatch;    // patch elements staged per loader thread
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// loader side (RECT tiles: 64 consecutive pixels of one output row): tile -> LDS buffer.
// Work items are groups of four consecutive floats (one dwordx4 load; rows start at any
// 4-byte address); a group that crosses the image border falls back to four predicated
// loads.  Every global load of the tile is issued before the first LDS store: the loaders
// have one MFMA loop (a few microseconds) per tile, two HBM round trips do not fit in it.
// bsl [CO_PAD][16] (LDS, one cell per group) accumulates dz for the bias gradient.
__device__ __forceinline__ void w2_stage_tile(const Wgrad2Geom& w2, const float* __restrict__ x,
                                              const float* __restrict__ dz, long tile, int chunk,
                                              int ltid, const unsigned (&dzo)[kW2MaxDz],
                                              const unsigned (&xo)[kW2MaxPatch], float* buf,
                                              float* bsl) {
    // opaque thread index: keeps the per-item (channel, row, column) arithmetic inside the
    // tile loop (hoisted out of it, it would pin several registers per staged item)
    asm volatile("" : "+v"(ltid));
    const WgradGeom& wg = w2.w;
    const ConvGeom& g = wg.c;
    const int tpi = g.tilesX * g.tilesY;
    const size_t iplane = (size_t)g.H * g.W;
    const size_t oplane = (size_t)g.Hout * g.Wout;
    const int n = (int)(tile / tpi);
    const int t = (int)(tile - (long)n * tpi);
    int oy0, ox0, p0;
    tile_origin(g, t, oy0, ox0, p0);
    const int dzFloats = g.CO_PAD * wg.PIXP;

    const float* dzn = dz + (size_t)n * g.Cout * oplane + (size_t)oy0 * g.Wout + ox0;
    const float* xn = x + ((size_t)n * g.Cin + (size_t)chunk * g.CI_T) * iplane;
    const int iy0 = oy0 - g.pad, ix0 = ox0 - g.pad;
    const int cin_left = g.Cin - chunk * g.CI_T;
    const int G = g.PC >> 2;  // groups per patch row (PC is a multiple of 4 here)
    const int pitems = g.CI_T * g.PR * G;
    const float invG = 1.0f / (float)G;
    const float invPR = 1.0f / (float)g.PR;
    float* patch = buf + dzFloats;

    f32x4u dv[kW2MaxDz], pv[kW2MaxPatch];
    // Interior tile (the common case; the test is scalar): every item is one unpredicated
    // dwordx4 load at a wave-uniform base plus this lane's constant offset -- two vector adds
    // per load and nothing else.  The f32 MFMA runs on the vector ALU (MI355X_MICROARCH.md,
    // "Matrix cores"): every VALU instruction of a loader wave is a cycle the MFMA wave of the
    // same SIMD does not get, so the per-item index arithmetic below would not hide behind
    // the matrix work, it would add to it.
    const bool interior = ox0 + g.PIX <= g.Wout && ix0 >= 0 && ix0 + g.PC <= g.W && iy0 >= 0 &&
                          iy0 + g.PR <= g.H && cin_left >= g.CI_T;
    if (interior) {
        const float* xb = xn + (size_t)iy0 * g.W + ix0;
#pragma unroll
        for (int u = 0; u < kW2MaxDz; ++u)
            if (u < w2.nDz) dv[u] = *reinterpret_cast<const f32x4u*>(dzn + dzo[u]);
#pragma unroll
        for (int u = 0; u < kW2MaxPatch; ++u)
            if (u < w2.nPatch) pv[u] = *reinterpret_cast<const f32x4u*>(xb + xo[u]);
    } else {
#pragma unroll
    for (int u = 0; u < kW2MaxDz; ++u) {
        f32x4u v = {0.f, 0.f, 0.f, 0.f};
        if (u < w2.nDz) {
            const int item = ltid + u * kW2Loaders;
            const int co = item >> 4, px = (item & 15) << 2;
            if (co < g.Cout) {
                const float* src = dzn + (size_t)co * oplane + px;
                if (ox0 + px + 3 < g.Wout) {
                    v = *reinterpret_cast<const f32x4u*>(src);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ox0 + px + j < g.Wout) v[j] = src[j];
                }
            }
        }
        dv[u] = v;
        __builtin_amdgcn_sched_barrier(0);  // border tiles are rare: one item at a time, few live registers
    }
#pragma unroll
    for (int u = 0; u < kW2MaxPatch; ++u) {
        f32x4u v = {0.f, 0.f, 0.f, 0.f};
        if (u < w2.nPatch) {
            const int item = ltid + u * kW2Loaders;
            int row, g4, ci_l, pr;
            divmod_small(item, G, invG, row, g4);
            divmod_small(row, g.PR, invPR, ci_l, pr);
            const int iy = iy0 + pr, ix = ix0 + 4 * g4;
            if (item < pitems && ci_l < cin_left && iy >= 0 && iy < g.H) {
                const float* src = xn + (size_t)ci_l * iplane + (size_t)iy * g.W + ix;
                if (ix >= 0 && ix + 3 < g.W) {
                    v = *reinterpret_cast<const f32x4u*>(src);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ix + j >= 0 && ix + j < g.W) v[j] = src[j];
                }
            }
        }
        pv[u] = v;
        __builtin_amdgcn_sched_barrier(0);
    }
    }
#pragma unroll
    for (int u = 0; u < kW2MaxDz; ++u) {
        if (u < w2.nDz) {
            const int item = ltid + u * kW2Loaders;
            const int co = item >> 4, px = (item & 15) << 2;
            float* d = buf + co * wg.PIXP + px;
            const bool live = co < g.Cout;  // padded channel rows stay zero (the interior path loads anything there)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = live ? dv[u][j] : 0.f;
            if (bsl && live) bsl[item] += (dv[u][0] + dv[u][1]) + (dv[u][2] + dv[u][3]);
        }
    }
#pragma unroll
    for (int u = 0; u < kW2MaxPatch; ++u) {
        const int item = ltid + u * kW2Loaders;
        if (u < w2.nPatch && item < pitems) *reinterpret_cast<float4*>(patch + 4 * item) = make_float4(pv[u][0], pv[u][1], pv[u][2], pv[u][3]);
    }
}

template <int TPW>
__global__ void __launch_bounds__(kW2Threads)
conv_wgrad2_kernel(const Wgrad2Geom w2, const float* __restrict__ x, const float* __restrict__ dz,
                   float* __restrict__ part, float* __restrict__ partb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const WgradGeom& wg = w2.w;
    const ConvGeom& g = wg.c;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool loader = wave >= 4;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int chunk = blockIdx.y;
    const int split = blockIdx.x;
    const int KK = g.K * g.K;
    const int dzFloats = g.CO_PAD * wg.PIXP;

    float* bsl = chunk == 0 ? smem + 2 * w2.bufFloats : nullptr;
    if (bsl)
        for (int i = tid; i < g.CO_PAD * 16; i += kW2Threads) bsl[i] = 0.f;
    __syncthreads();

    // The two roles run separate tile loops with matching barrier counts (s_barrier counts
    // waves, not program locations): neither role carries the other's registers.
    const long ntiles = wg.totalTiles;
    if (loader) {
        const int ltid = tid - 256;
        // this lane's element offsets inside an interior tile (constant over the tile loop)
        unsigned dzo[kW2MaxDz], xo[kW2MaxPatch];
        {
            const int G = g.PC >> 2;
            const int pitems = g.CI_T * g.PR * G;
#pragma unroll
            for (int u = 0; u < kW2MaxDz; ++u) {
                const int item = ltid + u * kW2Loaders;
                const int co = item >> 4, px = (item & 15) << 2;
                dzo[u] = co < g.Cout ? (unsigned)co * (unsigned)(g.Hout * g.Wout) + px : 0u;
            }
#pragma unroll
            for (int u = 0; u < kW2MaxPatch; ++u) {
                const int item = ltid + u * kW2Loaders;
                const int row = item / G, g4 = item - row * G;
                const int ci_l = row / g.PR, pr = row - ci_l * g.PR;
                xo[u] = item < pitems ? (unsigned)ci_l * (unsigned)(g.H * g.W) + (unsigned)(pr * g.W + 4 * g4) : 0u;
            }
        }
        long tile = split;
        int cur = 0;
        if (tile < ntiles) w2_stage_tile(w2, x, dz, tile, chunk, ltid, dzo, xo, smem, bsl);
        __syncthreads();
        for (; tile < ntiles; tile += wg.S) {
            const long tnext = tile + wg.S;
            if (tnext < ntiles)
                w2_stage_tile(w2, x, dz, tnext, chunk, ltid, dzo, xo, smem + (cur ^ 1) * w2.bufFloats, bsl);
            __syncthreads();
            cur ^= 1;
        }
        if (chunk == 0) {
            // bias gradient: 16 group cells per row, summed in a fixed order
            const int co = ltid;
            if (co < g.CO_PAD) {
                float v = 0.f;
                for (int j = 0; j < 16; ++j) v += bsl[co * 16 + j];
                partb[(size_t)split * g.CO_PAD + co] = v;
            }
        }
        return;
    }

    int aoff[TPW], joff[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int p = wave * TPW + q;
        const int pp = p < w2.PAIRS ? p : 0;
        const int m = pp / wg.NTILES, nt = pp - m * wg.NTILES;
        aoff[q] = (m * 32 + l31) * wg.PIXP;
        const int c = nt * 32 + l31;
        int o = 0;
        if (c < g.CI_T * KK) {
            const int ci_l = c / KK;
            const int r = c - ci_l * KK;
            const int ky = r / g.K;
            const int kx = r - ky * g.K;
            o = ci_l * g.PR * g.PC + ky * g.dil * g.PC + kx * g.dil;
        }
        joff[q] = dzFloats + o;  // the patch follows the dz rows inside a buffer
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    int cur = 0;
    __syncthreads();  // the first tile is staged
    for (long tile = split; tile < ntiles; tile += wg.S) {
        {
            const float* buf = smem + cur * w2.bufFloats;
            // RECT tile: pixel k of the tile sits k columns into the patch row, so every
            // fragment address is a per-lane base plus a compile-time offset (pixels past
            // the image edge read in-bounds patch columns against dz = 0).  Ping-pong
            // fragment registers; the reads of step s+1 are spread between the MFMAs of
            // step s (one MFMA, two LDS reads, ...) so that the matrix pipe neither waits
            // for a block of reads to issue nor for their latency at the step boundary.
            const float* ap[TPW];
            const float* bp[TPW];
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                ap[q] = buf + aoff[q] + half;
                bp[q] = buf + joff[q] + half;
            }
            float a0[TPW], a1[TPW], b0[TPW], b1[TPW];
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                a0[q] = ap[q][0];
                b0[q] = bp[q][0];
            }
#pragma unroll
            for (int ks = 0; ks < 32; ks += 2) {
#pragma unroll
                for (int q = 0; q < TPW; ++q) {
                    a1[q] = ap[q][2 * ks + 2];
                    b1[q] = bp[q][2 * ks + 2];
                }
#pragma unroll
                for (int q = 0; q < TPW; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[q], b0[q], acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < TPW; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < 32) {
#pragma unroll
                    for (int q = 0; q < TPW; ++q) {
                        a0[q] = ap[q][2 * ks + 4];
                        b0[q] = bp[q][2 * ks + 4];
                    }
                }
#pragma unroll
                for (int q = 0; q < TPW; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[q], b1[q], acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < TPW; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    float* slab = part + ((size_t)split * gridDim.y + chunk) * g.CO_PAD * wg.NCOL;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int p = wave * TPW + q;
        if (p >= w2.PAIRS) continue;
        const int m = p / wg.NTILES, nt = p - m * wg.NTILES;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[(size_t)co * wg.NCOL + nt * 32 + l31] = acc[q][r];
        }
    }
}

int plan_wgrad2(Wgrad2Geom& w2, int N, int Cin, int H, int W, int Cout, int K, int pad, int dil) {
    WgradGeom& wg = w2.w;
    ConvGeom& g = wg.c;
    const int Hout = H + 2 * pad - dil * (K - 1);
    const int Wout = W + 2 * pad - dil * (K - 1);
    const int pix = 64;
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.K = K; g.pad = pad; g.dil = dil;
    g.Hout = Hout; g.Wout = Wout; g.PIX = pix;
    g.CO_PAD = ((Cout + 31) / 32) * 32;
    if (g.CO_PAD > 128) return AFD_ERR_UNSUPPORTED;
    const int KK = K * K;
    wg.PIXP = pix + 1;
    wg.MT = g.CO_PAD / 32;
    wg.NG = 0;
    wg.fast_stage = 0;
    w2.nDz = g.CO_PAD / 16;  // groups of 4 pixels per loader thread
    int best_ct = 0;
    double best_eff = 0.0;
    // RECT tiles only (one output row, 64 consecutive pixels): the vectorised loader relies on
    // it; narrow images stay on the first-generation kernel
    if (Wout < 4 * pix) return AFD_ERR_UNSUPPORTED;
    set_tiling(g, pix, true);
    g.PC = (g.PC + 3) & ~3;  // whole groups of four columns per patch row
    for (int ct = 1; ct <= Cin && ct <= 128; ++ct) {
        const int ntiles = (ct * KK + 31) / 32;
        const int pairs = wg.MT * ntiles;
        const int tpw = (pairs + 3) / 4;
        if (tpw > kW2MaxTpw) break;
        const long patch = (long)ct * g.PR * g.PC;
        const long buf = (long)g.CO_PAD * wg.PIXP + patch + pix + 8;
        if ((2 * buf + (long)g.CO_PAD * 16) * 4 > 156 * 1024) break;
        if ((patch / 4 + kW2Loaders - 1) / kW2Loaders > kW2MaxPatch) break;
        const int nchunks = (Cin + ct - 1) / ct;
        // useful MFMA slots / issued slots, over all chunks (padded channels and columns)
        const double eff = ((double)Cin * KK / 32.0 * wg.MT) / ((double)nchunks * 4 * tpw);
        if (eff > best_eff + 1e-9 || (eff > best_eff - 1e-9 && ct > best_ct)) {
            best_eff = eff;
            best_ct = ct;
        }
    }
    if (best_ct == 0 || best_eff < 0.6) return AFD_ERR_UNSUPPORTED;
    const int ct = best_ct;
    g.CI_T = ct;
    g.nchunks = (Cin + ct - 1) / ct;
    g.CKP = 0;
    g.patchFloats = ((ct * g.PR * g.PC + 3) / 4) * 4;
    wg.NTILES = (ct * KK + 31) / 32;
    wg.NCOL = wg.NTILES * 32;
    w2.PAIRS = wg.MT * wg.NTILES;
    w2.TPW = (w2.PAIRS + 3) / 4;
    w2.bufFloats = g.CO_PAD * wg.PIXP + g.patchFloats + pix + 8;
    w2.nPatch = (g.patchFloats / 4 + kW2Loaders - 1) / kW2Loaders;
    wg.totalTiles = (long)N * g.tilesX * g.tilesY;
    long S = 1024 / g.nchunks;
    if (S < 1) S = 1;
    if (S > wg.totalTiles) S = wg.totalTiles;
    wg.S = (int)S;
    if ((size_t)g.CO_PAD * Hout * Wout >= 0x7fffffffULL || (size_t)ct * H * W >= 0x7fffffffULL)
        return AFD_ERR_UNSUPPORTED;  // 32-bit element offsets inside a tile
    return AFD_OK;
}

template <int TPW>
int launch_wgrad2_t(const Wgrad2Geom& w2, const float* x, const float* dz, float* part, float* partb,
                    hipStream_t s) {
    static afd::PerDeviceOnce attr_set;
    if (!attr_set.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad2_kernel<TPW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set.mark();
    }
    const ConvGeom& g = w2.w.c;
    afd::ScopedTiming timing(AFD_K_CONV_WGRAD, 2.0 * g.N * g.Cout * (double)g.Hout * g.Wout * g.Cin * g.K * g.K, s);
    timing.issued(2.0 * g.N * g.CO_PAD * (double)g.Hout * g.Wout * g.Cin * g.K * g.K);
    timing.bytes(4.0 * g.N * ((double)g.Cin * g.H * g.W + (double)g.Cout * g.Hout * g.Wout));
    hipLaunchKernelGGL(conv_wgrad2_kernel<TPW>, dim3(w2.w.S, g.nchunks), dim3(kW2Threads),
                       ((size_t)2 * w2.bufFloats + (size_t)g.CO_PAD * 16) * 4, s, w2, x, dz, part, partb);
    return afd::check_launch("conv_wgrad2_kernel");
}

int launch_wgrad2(const Wgrad2Geom& w2, const float* x, const float* dz, float* part, float* partb,
                  hipStream_t s) {
    switch (w2.TPW) {
        case 1: return launch_wgrad2_t<1>(w2, x, dz, part, partb, s);
        case 2: return launch_wgrad2_t<2>(w2, x, dz, part, partb, s);
        case 3: return launch_wgrad2_t<3>(w2, x, dz, part, partb, s);
        case 4: return launch_wgrad2_t<4>(w2, x, dz, part, partb, s);
        case 5: return launch_wgrad2_t<5>(w2, x, dz, part, partb, s);
        case 6: return launch_wgrad2_t<6>(w2, x, dz, part, partb, s);
        case 7: return launch_wgrad2_t<7>(w2, x, dz, part, partb, s);
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "wgrad: %d pairs per wave", w2.TPW);
}

bool use_wgrad2() { return true; }

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t afd_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int K, int pad,
                                             int dil) {
    const int Hout = H + 2 * pad - dil * (K - 1);
    const int Wout = W + 2 * pad - dil * (K - 1);
    if (Hout < 1 || Wout < 1) return 0;
    size_t need = 0;
    ConvGeom g;
    if (plan_igemm(g, N, Cin, H, W, Cout, K, pad, dil, Hout, Wout) == AFD_OK)
        need = align_up(repack_floats(g) * 4);
    ConvGeom gd;
    const int padd = dil * (K - 1) - pad;
    if (plan_igemm(gd, N, Cout, Hout, Wout, Cin, K, padd, dil, H, W) == AFD_OK) {
        const size_t b = align_up(repack_floats(gd) * 4);
        if (b > need) need = b;
    }
    WgradGeom wg;
    if (plan_wgrad(wg, N, Cin, H, W, Cout, K, pad, dil) == AFD_OK) {
        const size_t b = align_up(wgrad_ws_floats(wg) * 4);
        if (b > need) need = b;
    }
    Wgrad2Geom w2;
    if (plan_wgrad2(w2, N, Cin, H, W, Cout, K, pad, dil) == AFD_OK) {
        const size_t b = align_up(wgrad_ws_floats(w2.w) * 4);
        if (b > need) need = b;
    }
    if (afd::wgrad3x3_applicable(Cin, H, W, Cout, K, pad, dil)) {
        int S3, nch3, ct3, cop3, ncol3;
        afd::wgrad3x3_geometry(N, Cin, H, W, Cout, Hout, Wout, &S3, &nch3, &ct3, &cop3, &ncol3);
        const size_t b = align_up(((size_t)S3 * nch3 * cop3 * ncol3 + (size_t)S3 * cop3) * 4);
        if (b > need) need = b;
    }
    if (wino_wgrad_pays(Cin, H, W, Cout, K, pad, dil)) {
        // (the split count does not grow when the caller crops dy)
        const size_t b = align_up(afd::wino44_wgrad_workspace_floats(N, Cin, H, W, Cout, Hout, Wout) * 4);
        if (b > need) need = b;
    }
    if (wgrad_swap_applicable(Cin, H, W, Cout, K, pad, dil, Hout, Wout)) {
        int S3, nch3, ct3, cop3, ncol3;
        afd::wgrad3x3_geometry(N, Cout, H, W, Cin, H, W, &S3, &nch3, &ct3, &cop3, &ncol3);
        const size_t b = align_up(((size_t)S3 * nch3 * cop3 * ncol3 + (size_t)S3 * cop3 + (size_t)kBiasSplits * Cout) * 4);
        if (b > need) need = b;
    }
    if (afd::dilconv_applicable(Cin, Cout, K, dil)) {
        const size_t b = align_up(afd::dilconv_workspace_bytes(Cin, K));
        if (b > need) need = b;
    }
    if (afd::dilmfma_applicable(Cin, Cout, H, W, K, pad, dil)) {
        const size_t b = align_up(afd::dilmfma_workspace_bytes(N, Cin, H, W, K, pad, dil));
        if (b > need) need = b;
    }
    if (afd::conv3x3_applicable(Cin, H, W, Cout, K, pad, dil)) {
        const size_t b = align_up(afd::conv3x3_workspace_bytes(Cin, Cout));
        if (b > need) need = b;
    }
    if (afd::conv3x3_applicable(Cout, Hout, Wout, Cin, K, dil * (K - 1) - pad, dil)) {
        const size_t b = align_up(afd::conv3x3_workspace_bytes(Cout, Cin));
        if (b > need) need = b;
    }
    if (afd::conv1x1_applicable(Cin, Cout, K, pad, dil) && afd::conv1x1_wgrad_applicable(Cin, Cout)) {
        const size_t b = align_up(afd::conv1x1_workspace_bytes(Cin, Cout));
        if (b > need) need = b;
    }
    return need;
}

static int check_conv_args(const void* a, const void* b, const void* c, int N, int Cin, int H,
                           int W, int Cout, int K, int pad, int dil) {
    if (!a || !b || !c) return afd::fail(AFD_ERR_ARG, "conv: null pointer");
    if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || K < 1 || dil < 1 || pad < 0)
        return afd::fail(AFD_ERR_ARG, "conv: bad geometry");
    if (Cout > 128 || Cin > 128 * 32) return afd::fail(AFD_ERR_UNSUPPORTED, "conv: Cout %d > 128", Cout);
    if (H + 2 * pad - dil * (K - 1) < 1 || W + 2 * pad - dil * (K - 1) < 1)
        return afd::fail(AFD_ERR_ARG, "conv: empty output");
    return AFD_OK;
}

extern "C" int afd_conv2d_forward(const float* x, const float* w, const float* bias, float* y,
                                  int N, int Cin, int H, int W, int Cout, int K, int pad, int dil,
                                  void* ws, size_t ws_bytes, afd_stream_t stream) {
    return afd_conv2d_forward_cropped(x, w, bias, y, N, Cin, H, W, Cout, K, pad, dil, 0x7fffffff,
                                      0x7fffffff, ws, ws_bytes, stream);
}

// only the 3x3 wide-image kernel makes use of the crop; the other paths compute all of y
extern "C" int afd_conv2d_forward_cropped(const float* x, const float* w, const float* bias, float* y,
                                          int N, int Cin, int H, int W, int Cout, int K, int pad,
                                          int dil, int out_rows, int out_cols, void* ws,
                                          size_t ws_bytes, afd_stream_t stream) {
    int rc = check_conv_args(x, w, y, N, Cin, H, W, Cout, K, pad, dil);
    if (rc) return rc;
    if (out_rows < 1 || out_cols < 1) return afd::fail(AFD_ERR_ARG, "conv fwd: empty crop");
    if (afd::dilconv_applicable(Cin, Cout, K, dil))
        return afd::dilconv_forward(x, w, bias, y, N, Cin, H, W, K, pad, dil, static_cast<hipStream_t>(stream));
    if (afd::dilmfma_applicable(Cin, Cout, H, W, K, pad, dil))
        return afd::dilmfma_forward(x, w, bias, y, N, Cin, H, W, K, pad, dil, ws, ws_bytes, static_cast<hipStream_t>(stream));
    if (afd::conv1x1_applicable(Cin, Cout, K, pad, dil))
        return afd::conv1x1_forward(x, w, bias, y, N, Cin, Cout, (long)H * W, static_cast<hipStream_t>(stream));
    if (afd::conv3x3_applicable(Cin, H, W, Cout, K, pad, dil))
        return afd::conv3x3_run(x, w, bias, y, N, Cin, H, W, Cout, 0, out_rows, out_cols, ws, ws_bytes,
                                static_cast<hipStream_t>(stream));
    const int Hout = H + 2 * pad - dil * (K - 1);
    const int Wout = W + 2 * pad - dil * (K - 1);
    ConvGeom g;
    rc = plan_igemm(g, N, Cin, H, W, Cout, K, pad, dil, Hout, Wout);
    if (rc) return rc;
    if (!ws || ws_bytes < repack_floats(g) * 4) return afd::fail(AFD_ERR_WORKSPACE, "conv fwd: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* wp = static_cast<float*>(ws);
    const int total = (int)repack_floats(g);
    hipLaunchKernelGGL(repack_fwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, wp, Cin,
                       Cout, K * K, g.CI_T, g.nchunks, g.CKP, g.CO_PAD);
    rc = afd::check_launch("repack_fwd_kernel");
    if (rc) return rc;
    return launch_igemm(g, x, wp, bias, y, s);
}

// 3x3 / pad 1 convolution + PReLU + MaxPool2d(2, 2) in one launch: the Winograd output tile is the
// pooling window.  Returns AFD_ERR_UNSUPPORTED when the layer is not one the Winograd kernel of
// wino.hip takes (the caller then runs convolution and pool separately).
extern "C" int afd_conv3x3_prelu_pool_applicable(int Cin, int H, int W, int Cout) {
    // (either Winograd kernel: the F(4x4) form takes images from 48 columns up, the F(2x2) form from 64)
    return (afd::wino44_pool_applicable(Cin, H, W, Cout) || afd::wino_applicable(Cin, H, W, Cout)) && H >= 2 && W >= 2 ? 1 : 0;
}

extern "C" int afd_conv3x3_prelu_pool_forward(const float* x, const float* w, const float* bias,
                                              const float* slope, float* u, uint8_t* idx, int N, int Cin,
                                              int H, int W, int Cout, void* ws, size_t ws_bytes,
                                              afd_stream_t stream) {
    if (!x || !w || !slope || !u || !idx) return afd::fail(AFD_ERR_ARG, "conv3x3+pool: null pointer");
    if (N < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2) return afd::fail(AFD_ERR_ARG, "conv3x3+pool: bad shape");
    if (!afd_conv3x3_prelu_pool_applicable(Cin, H, W, Cout))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3+pool: layer not on the Winograd kernel");
    if (afd::wino44_pool_applicable(Cin, H, W, Cout))
        return afd::wino44_run(x, w, bias, nullptr, N, Cin, H, W, Cout, 0, 2 * (H / 2), 2 * (W / 2), ws, ws_bytes,
                               static_cast<hipStream_t>(stream), nullptr, nullptr, slope, u, idx);
    return afd::wino_run(x, w, bias, nullptr, N, Cin, H, W, Cout, 0, 2 * (H / 2), 2 * (W / 2), ws, ws_bytes,
                         static_cast<hipStream_t>(stream), slope, u, idx);
}

extern "C" size_t afd_conv1x1_forward_stats_workspace_bytes(int Cout) {
    return afd::conv1x1_stats_workspace_bytes(Cout);
}

extern "C" int afd_conv1x1_forward_stats(const float* x, const float* w, const float* bias, const float* slope,
                                         float* y, double* sums, int N, int Cin, int Cout, long HW, void* ws,
                                         size_t ws_bytes, afd_stream_t stream) {
    if (!x || !w || !slope || !y || !sums || !ws) return afd::fail(AFD_ERR_ARG, "conv1x1 stats: null pointer");
    if (N < 1 || Cin < 1 || Cout < 1 || HW < 1 || HW > 0x7fffffffL)
        return afd::fail(AFD_ERR_ARG, "conv1x1 stats: bad shape");
    if (Cin > 128 || Cout > 128) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1 stats: more than 128 channels");
    return afd::conv1x1_forward_stats(x, w, bias, slope, y, sums, N, Cin, Cout, HW, ws, ws_bytes,
                                      static_cast<hipStream_t>(stream));
}

extern "C" int afd_conv1x1_bn_backward_data(const float* dz, const float* wf, const float* u,
                                            const float* alpha, const float* beta, float* du, int N,
                                            int Cin, int Cout, long HW, afd_stream_t stream) {
    if (!dz || !wf || !u || !alpha || !beta || !du) return afd::fail(AFD_ERR_ARG, "conv1x1 bn dgrad: null pointer");
    if (N < 1 || Cin < 1 || Cout < 1 || HW < 1 || HW > 0x7fffffffL)
        return afd::fail(AFD_ERR_ARG, "conv1x1 bn dgrad: bad shape");
    if (Cin > 128 || Cout > 128) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1 bn dgrad: more than 128 channels");
    return afd::conv1x1_backward_data_affine(dz, wf, u, alpha, beta, du, N, Cin, Cout, HW,
                                             static_cast<hipStream_t>(stream));
}

extern "C" int afd_conv2d_backward_data(const float* dy, const float* w, float* dx, int N, int Cin,
                                        int H, int W, int Cout, int K, int pad, int dil, void* ws,
                                        size_t ws_bytes, afd_stream_t stream) {
    int rc = check_conv_args(dy, w, dx, N, Cin, H, W, Cout, K, pad, dil);
    if (rc) return rc;
    if (Cin > 128) return afd::fail(AFD_ERR_UNSUPPORTED, "conv dgrad: Cin %d > 128", Cin);
    if (afd::dilconv_applicable(Cin, Cout, K, dil))
        return afd::dilconv_backward_data(dy, w, dx, N, Cin, H, W, K, pad, dil, static_cast<hipStream_t>(stream));
    if (afd::dilmfma_applicable(Cin, Cout, H, W, K, pad, dil))
        return afd::dilmfma_backward_data(dy, w, dx, N, Cin, H, W, K, pad, dil, ws, ws_bytes, static_cast<hipStream_t>(stream));
    if (afd::conv1x1_applicable(Cin, Cout, K, pad, dil))
        return afd::conv1x1_backward_data(dy, w, dx, N, Cin, Cout, (long)H * W, static_cast<hipStream_t>(stream));
    if (afd::conv3x3_applicable(Cout, H, W, Cin, K, dil * (K - 1) - pad, dil))
        return afd::conv3x3_run(dy, w, nullptr, dx, N, Cout, H, W, Cin, 1, H, W, ws, ws_bytes,
                                static_cast<hipStream_t>(stream));
    const int Hout = H + 2 * pad - dil * (K - 1);
    const int Wout = W + 2 * pad - dil * (K - 1);
    const int padd = dil * (K - 1) - pad;
    ConvGeom g;
    rc = plan_igemm(g, N, Cout, Hout, Wout, Cin, K, padd, dil, H, W);
    if (rc) return rc;
    if (!ws || ws_bytes < repack_floats(g) * 4) return afd::fail(AFD_ERR_WORKSPACE, "conv dgrad: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* wp = static_cast<float*>(ws);
    const int total = (int)repack_floats(g);
    // in the dgrad geometry: "Cin" = forward Cout, "Cout" = forward Cin
    hipLaunchKernelGGL(repack_dgrad_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, wp, Cin,
                       Cout, K * K, g.CI_T, g.nchunks, g.CKP, g.CO_PAD);
    rc = afd::check_launch("repack_dgrad_kernel");
    if (rc) return rc;
    return launch_igemm(g, dy, wp, nullptr, dx, s);
}

extern "C" int afd_conv2d_backward_weight(const float* x, const float* dy, float* dw, float* dbias,
                                          int N, int Cin, int H, int W, int Cout, int K, int pad,
                                          int dil, void* ws, size_t ws_bytes, afd_stream_t stream) {
    return afd_conv2d_backward_weight_cropped(x, dy, dw, dbias, N, Cin, H, W, Cout, K, pad, dil,
                                              0x7fffffff, 0x7fffffff, ws, ws_bytes, stream);
}

extern "C" int afd_conv2d_backward_weight_cropped(const float* x, const float* dy, float* dw,
                                                  float* dbias, int N, int Cin, int H, int W,
                                                  int Cout, int K, int pad, int dil, int dy_rows,
                                                  int dy_cols, void* ws, size_t ws_bytes,
                                                  afd_stream_t stream) {
    return afd_conv2d_backward_weight_sums(x, dy, dw, dbias, nullptr, N, Cin, H, W, Cout, K, pad, dil, dy_rows, dy_cols,
                                           ws, ws_bytes, stream);
}

extern "C" int afd_conv2d_backward_weight_sums(const float* x, const float* dy, float* dw, float* dbias,
                                               const double* dy_sums, int N, int Cin, int H, int W, int Cout,
                                               int K, int pad, int dil, int dy_rows, int dy_cols, void* ws,
                                               size_t ws_bytes, afd_stream_t stream) {
    int rc = check_conv_args(x, dy, dw, N, Cin, H, W, Cout, K, pad, dil);
    if (rc) return rc;
    if (dy_rows < 1 || dy_cols < 1) return afd::fail(AFD_ERR_ARG, "conv wgrad: empty crop");
    if (afd::dilconv_applicable(Cin, Cout, K, dil))
        return afd::dilconv_backward_weight(x, dy, dw, dbias, N, Cin, H, W, K, pad, dil, ws, ws_bytes,
                                            static_cast<hipStream_t>(stream));
    if (afd::dilmfma_applicable(Cin, Cout, H, W, K, pad, dil) && dy_rows >= H + 2 * pad - dil * (K - 1)
        && dy_cols >= W + 2 * pad - dil * (K - 1))
        return afd::dilmfma_backward_weight(x, dy, dw, dbias, N, Cin, H, W, K, pad, dil, ws, ws_bytes,
                                            static_cast<hipStream_t>(stream));
    if (afd::conv1x1_applicable(Cin, Cout, K, pad, dil) && afd::conv1x1_wgrad_applicable(Cin, Cout))
        return afd::conv1x1_backward_weight(x, dy, dw, dbias, N, Cin, Cout, (long)H * W, ws, ws_bytes,
                                            static_cast<hipStream_t>(stream));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (wino_wgrad_pays(Cin, H, W, Cout, K, pad, dil) && afd::wino44_wgrad_crop_ok(H, W, dy_rows, dy_cols)) {
        // the bias gradient: from the producer of dy when the caller has its sums, else from the dy-transform lanes
        rc = afd::wino44_wgrad_run(x, dy, dw, (dbias && !dy_sums) ? dbias : nullptr, N, Cin, H, W, Cout, dy_rows, dy_cols,
                                   ws, ws_bytes, s);
        if (rc) return rc;
        if (dbias && dy_sums) {
            hipLaunchKernelGGL(bias_from_sums_kernel, dim3((Cout + 63) / 64), dim3(64), 0, s, dy_sums, dbias, Cout);
            return afd::check_launch("bias_from_sums_kernel");
        }
        return AFD_OK;
    }
    if (wgrad_swap_applicable(Cin, H, W, Cout, K, pad, dil, dy_rows, dy_cols)) {
        int S3, nch3, ct3, cop3, ncol3;
        afd::wgrad3x3_geometry(N, Cout, H, W, Cin, H, W, &S3, &nch3, &ct3, &cop3, &ncol3);
        const size_t slabs = (size_t)S3 * nch3 * cop3 * ncol3;
        if (!ws || ws_bytes < (slabs + (size_t)S3 * cop3 + (size_t)kBiasSplits * Cout) * 4)
            return afd::fail(AFD_ERR_WORKSPACE, "conv wgrad: workspace too small");
        float* part3 = static_cast<float*>(ws);
        float* partx = part3 + slabs;                  // the kernel's per-row sums (of x here): not used
        float* partb3 = partx + (size_t)S3 * cop3;
        rc = afd::wgrad3x3_launch(dy, x, part3, partx, N, Cout, H, W, Cin, H, W, s);
        if (rc) return rc;
        // the bias gradient: per-channel sums of dy from its producer when the caller has them, else a pass over dy
        float* db_red = dbias;
        if (dbias && dy_sums) {
            hipLaunchKernelGGL(bias_from_sums_kernel, dim3((Cout + 63) / 64), dim3(64), 0, s, dy_sums, dbias, Cout);
            db_red = nullptr;
        } else if (dbias) {
            hipLaunchKernelGGL(bias_partial_kernel, dim3(kBiasSplits, Cout), dim3(256), 0, s, dy, partb3, N, Cout, H * W);
            rc = afd::check_launch("bias_partial_kernel");
            if (rc) return rc;
        }
        const int total3 = Cout * Cin * K * K;
        const int nblk3 = (total3 + 31) / 32 + (db_red ? (Cout + 31) / 32 : 0);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk3), dim3(256), 0, s, part3, partb3, dw, db_red,
                           Cin, Cout, K * K, ct3, nch3, cop3, ncol3, S3, 1, kBiasSplits);
        return afd::check_launch("wgrad_reduce_kernel");
    }
    if (afd::wgrad3x3_applicable(Cin, H, W, Cout, K, pad, dil)) {
        int S3, nch3, ct3, cop3, ncol3;
        afd::wgrad3x3_geometry(N, Cin, H, W, Cout, dy_rows, dy_cols, &S3, &nch3, &ct3, &cop3, &ncol3);
        const size_t slabs = (size_t)S3 * nch3 * cop3 * ncol3;
        if (!ws || ws_bytes < (slabs + (size_t)S3 * cop3) * 4) return afd::fail(AFD_ERR_WORKSPACE, "conv wgrad: workspace too small");
        float* part3 = static_cast<float*>(ws);
        float* partb3 = part3 + slabs;
        rc = afd::wgrad3x3_launch(x, dy, part3, partb3, N, Cin, H, W, Cout, dy_rows, dy_cols, s);
        if (rc) return rc;
        const int total3 = Cout * Cin * K * K;
        const int nblk3 = (total3 + 31) / 32 + (dbias ? (Cout + 31) / 32 : 0);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk3), dim3(256), 0, s, part3, partb3, dw, dbias,
                           Cin, Cout, K * K, ct3, nch3, cop3, ncol3, S3);
        return afd::check_launch("wgrad_reduce_kernel");
    }
    Wgrad2Geom w2;
    if (use_wgrad2() && plan_wgrad2(w2, N, Cin, H, W, Cout, K, pad, dil) == AFD_OK) {
        const WgradGeom& g2 = w2.w;
        if (!ws || ws_bytes < wgrad_ws_floats(g2) * 4) return afd::fail(AFD_ERR_WORKSPACE, "conv wgrad: workspace too small");
        float* part2 = static_cast<float*>(ws);
        float* partb2 = part2 + (size_t)g2.S * g2.c.nchunks * g2.c.CO_PAD * g2.NCOL;
        rc = launch_wgrad2(w2, x, dy, part2, partb2, s);
        if (rc) return rc;
        const int total2 = Cout * Cin * K * K;
        const int nblk2 = (total2 + 31) / 32 + (dbias ? (Cout + 31) / 32 : 0);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk2), dim3(256), 0, s, part2, partb2, dw, dbias,
                           Cin, Cout, K * K, g2.c.CI_T, g2.c.nchunks, g2.c.CO_PAD, g2.NCOL, g2.S);
        return afd::check_launch("wgrad_reduce_kernel");
    }
    WgradGeom wg;
    rc = plan_wgrad(wg, N, Cin, H, W, Cout, K, pad, dil);
    if (rc) return rc;
    if (!ws || ws_bytes < wgrad_ws_floats(wg) * 4) return afd::fail(AFD_ERR_WORKSPACE, "conv wgrad: workspace too small");
    float* part = static_cast<float*>(ws);
    float* partb = part + (size_t)wg.S * wg.c.nchunks * wg.c.CO_PAD * wg.NCOL;
    const int ntw = (wg.NTILES + wg.NG - 1) / wg.NG;
    if (ntw <= 1) rc = launch_wgrad_t<1>(wg, x, dy, part, partb, s);
    else if (ntw == 2) rc = launch_wgrad_t<2>(wg, x, dy, part, partb, s);
    else if (ntw == 3) rc = launch_wgrad_t<3>(wg, x, dy, part, partb, s);
    else if (ntw == 4) rc = launch_wgrad_t<4>(wg, x, dy, part, partb, s);
    else if (ntw == 5) rc = launch_wgrad_t<5>(wg, x, dy, part, partb, s);
    else if (ntw == 6) rc = launch_wgrad_t<6>(wg, x, dy, part, partb, s);
    else if (ntw <= 9) rc = launch_wgrad_t<9>(wg, x, dy, part, partb, s);
    else return afd::fail(AFD_ERR_UNSUPPORTED, "wgrad: %d column tiles per wave", ntw);
    if (rc) return rc;
    const int total = Cout * Cin * K * K;
    const int nblk = (total + 31) / 32 + (dbias ? (Cout + 31) / 32 : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblk), dim3(256), 0, s, part, partb, dw, dbias, Cin,
                       Cout, K * K, wg.c.CI_T, wg.c.nchunks, wg.c.CO_PAD, wg.NCOL, wg.S);
    return afd::check_launch("wgrad_reduce_kernel");
}
