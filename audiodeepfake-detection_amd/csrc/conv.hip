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

// stage input patch rows: rows go to waves, columns to lanes (coalesced along x)
__device__ __forceinline__ void stage_patch(const ConvGeom& g, const float* __restrict__ x, int n,
                                            int cin_total, int Hin, int Win, int chunk, int iy0,
                                            int ix0, float* patch, int nthreads) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwaves = nthreads >> 6;
    const int rows = g.CI_T * g.PR;
    for (int row = wave; row < rows; row += nwaves) {
        const int ci_l = row / g.PR;
        const int pr = row - ci_l * g.PR;
        const int ci = chunk * g.CI_T + ci_l;
        const int iy = iy0 + pr;
        const bool rowok = (ci < cin_total) && (iy >= 0) && (iy < Hin);
        const float* src = x + ((size_t)(n * cin_total + (rowok ? ci : 0)) * Hin + (rowok ? iy : 0)) * Win;
        float* dst = patch + row * g.PC;
        for (int pc = lane; pc < g.PC; pc += 64) {
            const int ix = ix0 + pc;
            float v = 0.f;
            if (rowok && ix >= 0 && ix < Win) v = src[ix];
            dst[pc] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// forward / backward-data
// ---------------------------------------------------------------------------------------
template <int NT>
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
    const int WM = g.CO_PAD >> 5;
    const int wm = wave % WM;
    const int wn = wave / WM;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int tpi = g.tilesX * g.tilesY;
    const int n = blockIdx.x / tpi;
    const int t = blockIdx.x - n * tpi;
    int oy0, ox0, p0;
    tile_origin(g, t, oy0, ox0, p0);
    const int iy0 = oy0 - g.pad;
    const int ix0 = ox0 - g.pad;

    int pbase[NT];
    int oyv[NT], oxv[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int pj = (wn * NT + i) * 32 + l31;
        int oy, ox;
        tile_pixel(g, pj, oy0, ox0, p0, oy, ox);
        const bool ok = (oy < g.Hout) && (ox < g.Wout);
        pbase[i] = ok ? (oy - oy0) * g.PC + (ox - ox0) : 0;
        oyv[i] = ok ? oy : -1;
        oxv[i] = ox;
    }

    const int KK = g.K * g.K;
    for (int kl = tid; kl < g.CKP; kl += nthreads) {
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

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int wslab = g.CKP * g.CO_PAD;
    for (int chunk = 0; chunk < g.nchunks; ++chunk) {
        __syncthreads();
        {
            const float4* src = reinterpret_cast<const float4*>(wp + (size_t)chunk * wslab);
            float4* dst = reinterpret_cast<float4*>(wlds);
            for (int i = tid; i < (wslab >> 2); i += nthreads) dst[i] = src[i];
        }
        stage_patch(g, x, n, g.Cin, g.H, g.W, chunk, iy0, ix0, patch, nthreads);
        __syncthreads();
        const float* arow = wlds + wm * 32 + l31;
        const int ksteps = g.CKP >> 1;
        // two-level accumulation: a fresh fp32 chain per channel chunk, then one add --
        // keeps the rounding error of K = Cin*K*K <= 1152 products near sqrt(chunk) * eps
        f32x16 part[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[i][r] = 0.f;
#pragma unroll 4
        for (int ks = 0; ks < ksteps; ++ks) {
            const int k = 2 * ks + half;
            const float a = arow[k * g.CO_PAD];
            const int off = koff[k];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const float b = patch[off + pbase[i]];
                part[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, part[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] += part[i];
    }

    // D layout: column = lane & 31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const size_t plane = (size_t)g.Hout * g.Wout;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        if (oyv[i] < 0) continue;
        const size_t pix = (size_t)oyv[i] * g.Wout + oxv[i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < g.Cout) {
                const float bv = bias ? bias[co] : 0.f;
                y[((size_t)n * g.Cout + co) * plane + pix] = acc[i][r] + bv;
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

// geometry for a conv whose input is [N][Cin][H][W] and output [N][Cout][Hout][Wout]
int plan_igemm(ConvGeom& g, int N, int Cin, int H, int W, int Cout, int K, int pad, int dil,
               int Hout, int Wout, int pix) {
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.K = K; g.pad = pad; g.dil = dil;
    g.Hout = Hout; g.Wout = Wout; g.PIX = pix;
    g.CO_PAD = ((Cout + 31) / 32) * 32;
    const int KK = K * K;
    int best = 0;
    for (int attempt = 0; attempt < 2 && best == 0; ++attempt) {
        const bool rect = (Wout >= 4 * pix) != (attempt == 1);
        if (rect && Wout < pix) continue;
        set_tiling(g, pix, rect);
        // largest channel chunk whose weights + patch fit the LDS budget
        for (int ct = 1; ct <= Cin && ct <= 32; ++ct) {
            const int ckp = (ct * KK + 1) & ~1;
            const long bytes = 4L * ((long)ckp * g.CO_PAD + (long)ct * g.PR * g.PC + 4 + ckp);
            if (bytes <= kLdsBudget) best = ct;
        }
    }
    if (best == 0) return afd::fail(AFD_ERR_UNSUPPORTED, "conv: tile does not fit LDS (K=%d dil=%d W=%d)", K, dil, W);
    // prefer a chunk size that divides Cin (no zero-padded k rows)
    int ct = best;
    for (int c = best; c >= 1 && c * 2 > best; --c)
        if (Cin % c == 0) { ct = c; break; }
    g.CI_T = ct;
    g.nchunks = (Cin + ct - 1) / ct;
    g.CKP = (ct * KK + 1) & ~1;
    g.patchFloats = ((ct * g.PR * g.PC + 3) / 4) * 4;
    return AFD_OK;
}

size_t igemm_lds_bytes(const ConvGeom& g) {
    return 4 * ((size_t)g.CKP * g.CO_PAD + g.patchFloats + g.CKP);
}

size_t repack_floats(const ConvGeom& g) { return (size_t)g.nchunks * g.CKP * g.CO_PAD; }

int launch_igemm(const ConvGeom& g, const float* x, const float* wp, const float* bias, float* y,
                 hipStream_t s) {
    const int WM = g.CO_PAD / 32;
    const int NT = 2;
    int WN = g.PIX / (32 * NT);
    const int threads = WM * WN * 64;
    if (threads > 512 || threads < 64) return afd::fail(AFD_ERR_UNSUPPORTED, "conv: Cout %d too large", g.Cout);
    const long blocks = (long)g.N * g.tilesX * g.tilesY;
    if (blocks > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv: grid too large");
    afd::ScopedTiming timing(AFD_K_CONV_IGEMM,
                             2.0 * g.N * g.Cout * (double)g.Hout * g.Wout * g.Cin * g.K * g.K, s);
    hipLaunchKernelGGL(conv_igemm_kernel<2>, dim3((unsigned)blocks), dim3(threads),
                       igemm_lds_bytes(g), s, g, x, wp, bias, y);
    return afd::check_launch("conv_igemm_kernel");
}

// pixel tile width so that a workgroup has 4..8 waves
int pick_pix(int co_pad) {
    const int WM = co_pad / 32;
    if (WM == 1) return 256;  // 1 x 4 waves
    if (WM == 2) return 128;  // 2 x 2
    if (WM == 3) return 128;  // 3 x 2
    return 128;               // 4 x 2
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

    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = nthreads >> 6;
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
    float bsum = 0.f;

    const int tpi = g.tilesX * g.tilesY;
    for (long tile = split; tile < wg.totalTiles; tile += wg.S) {
        const int n = (int)(tile / tpi);
        const int t = (int)(tile - (long)n * tpi);
        int oy0, ox0, p0;
        tile_origin(g, t, oy0, ox0, p0);
        __syncthreads();
        // dz tile: rows (channels) to waves, pixels to lanes
        for (int co = wave; co < g.CO_PAD; co += nwaves) {
            for (int pj = lane; pj < g.PIX; pj += 64) {
                int oy, ox;
                tile_pixel(g, pj, oy0, ox0, p0, oy, ox);
                float v = 0.f;
                if (co < g.Cout && oy < g.Hout && ox < g.Wout)
                    v = dz[(((size_t)n * g.Cout + co) * g.Hout + oy) * g.Wout + ox];
                dzl[co * wg.PIXP + pj] = v;
            }
        }
        for (int pj = tid; pj < g.PIX; pj += nthreads) {
            int oy, ox;
            tile_pixel(g, pj, oy0, ox0, p0, oy, ox);
            const bool ok = (oy < g.Hout) && (ox < g.Wout);
            pixoff[pj] = ok ? (oy - oy0) * g.PC + (ox - ox0) : 0;
        }
        stage_patch(g, x, n, g.Cin, g.H, g.W, chunk, oy0 - g.pad, ox0 - g.pad, patch, nthreads);
        __syncthreads();
        if (chunk == 0 && tid < g.CO_PAD) {
            const float* row = dzl + tid * wg.PIXP;
            float s = 0.f;
            for (int pj = 0; pj < g.PIX; ++pj) s += row[pj];
            bsum += s;
        }
        const float* arow = dzl + (m * 32 + l31) * wg.PIXP;
        const int ksteps = g.PIX >> 1;
        // two-level accumulation (per 64-pixel tile, then across tiles): see forward kernel
        f32x16 part[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[i][r] = 0.f;
#pragma unroll 2
        for (int ks = 0; ks < ksteps; ++ks) {
            const int k = 2 * ks + half;
            const float a = arow[k];
            const int po = pixoff[k];
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                if (nvalid[i]) {
                    const float b = patch[joff[i] + po];
                    part[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, part[i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NTW; ++i) acc[i] += part[i];
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
    if (chunk == 0 && tid < g.CO_PAD) partb[(size_t)split * g.CO_PAD + tid] = bsum;
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ partb,
                                    float* __restrict__ dw, float* __restrict__ db, int Cin,
                                    int Cout, int KK, int CI_T, int nchunks, int CO_PAD, int NCOL,
                                    int S) {
    const int total = Cout * Cin * KK;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int r = i % KK;
        const int ci = (i / KK) % Cin;
        const int co = i / (KK * Cin);
        const int chunk = ci / CI_T;
        const int col = (ci - chunk * CI_T) * KK + r;
        const size_t stride = (size_t)nchunks * CO_PAD * NCOL;
        const float* p = part + ((size_t)chunk * CO_PAD + co) * NCOL + col;
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += p[(size_t)k * stride];
        dw[i] = s;
    }
    if (db && i < Cout) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += partb[(size_t)k * CO_PAD + i];
        db[i] = s;
    }
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
    const long fixed = 4L * ((long)g.CO_PAD * wg.PIXP + pix + 8);
    int best = 0;
    for (int attempt = 0; attempt < 2 && best == 0; ++attempt) {
        const bool rect = (Wout >= 4 * pix) != (attempt == 1);
        if (rect && Wout < pix) continue;
        set_tiling(g, pix, rect);
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
    // waves = MT * NG, at most 8; per-wave column tiles NTW in {1,3,5,9}
    int NG = ngmax;
    if (NG > wg.NTILES) NG = wg.NTILES;
    wg.NG = NG;
    wg.totalTiles = (long)N * g.tilesX * g.tilesY;
    long S = 1024 / g.nchunks;
    if (S < 1) S = 1;
    if (S > wg.totalTiles) S = wg.totalTiles;
    wg.S = (int)S;
    return AFD_OK;
}

size_t wgrad_lds_bytes(const WgradGeom& wg) {
    return 4 * ((size_t)wg.c.CO_PAD * wg.PIXP + wg.c.patchFloats + wg.c.PIX);
}

size_t wgrad_ws_floats(const WgradGeom& wg) {
    return (size_t)wg.S * wg.c.nchunks * wg.c.CO_PAD * wg.NCOL + (size_t)wg.S * wg.c.CO_PAD;
}

template <int NTW>
int launch_wgrad_t(const WgradGeom& wg, const float* x, const float* dz, float* part, float* partb,
                   hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<NTW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
    afd::ScopedTiming timing(AFD_K_CONV_WGRAD, 2.0 * wg.c.N * wg.c.Cout * (double)wg.c.Hout *
                                                   wg.c.Wout * wg.c.Cin * wg.c.K * wg.c.K, s);
    hipLaunchKernelGGL(conv_wgrad_kernel<NTW>, dim3(wg.S, wg.c.nchunks), dim3(wg.MT * wg.NG * 64),
                       wgrad_lds_bytes(wg), s, wg, x, dz, part, partb);
    return afd::check_launch("conv_wgrad_kernel");
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t afd_conv2d_workspace_bytes(int N, int Cin, int H, int W, int Cout, int K, int pad,
                                             int dil) {
    const int Hout = H + 2 * pad - dil * (K - 1);
    const int Wout = W + 2 * pad - dil * (K - 1);
    if (Hout < 1 || Wout < 1) return 0;
    size_t need = 0;
    ConvGeom g;
    if (plan_igemm(g, N, Cin, H, W, Cout, K, pad, dil, Hout, Wout,
                   pick_pix(((Cout + 31) / 32) * 32)) == AFD_OK)
        need = align_up(repack_floats(g) * 4);
    ConvGeom gd;
    const int padd = dil * (K - 1) - pad;
    if (plan_igemm(gd, N, Cout, Hout, Wout, Cin, K, padd, dil, H, W,
                   pick_pix(((Cin + 31) / 32) * 32)) == AFD_OK) {
        const size_t b = align_up(repack_floats(gd) * 4);
        if (b > need) need = b;
    }
    WgradGeom wg;
    if (plan_wgrad(wg, N, Cin, H, W, Cout, K, pad, dil) == AFD_OK) {
        const size_t b = align_up(wgrad_ws_floats(wg) * 4);
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
    int rc = check_conv_args(x, w, y, N, Cin, H, W, Cout, K, pad, dil);
    if (rc) return rc;
    const int Hout = H + 2 * pad - dil * (K - 1);
    const int Wout = W + 2 * pad - dil * (K - 1);
    ConvGeom g;
    rc = plan_igemm(g, N, Cin, H, W, Cout, K, pad, dil, Hout, Wout, pick_pix(((Cout + 31) / 32) * 32));
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

extern "C" int afd_conv2d_backward_data(const float* dy, const float* w, float* dx, int N, int Cin,
                                        int H, int W, int Cout, int K, int pad, int dil, void* ws,
                                        size_t ws_bytes, afd_stream_t stream) {
    int rc = check_conv_args(dy, w, dx, N, Cin, H, W, Cout, K, pad, dil);
    if (rc) return rc;
    if (Cin > 128) return afd::fail(AFD_ERR_UNSUPPORTED, "conv dgrad: Cin %d > 128", Cin);
    const int Hout = H + 2 * pad - dil * (K - 1);
    const int Wout = W + 2 * pad - dil * (K - 1);
    const int padd = dil * (K - 1) - pad;
    ConvGeom g;
    rc = plan_igemm(g, N, Cout, Hout, Wout, Cin, K, padd, dil, H, W, pick_pix(((Cin + 31) / 32) * 32));
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
    int rc = check_conv_args(x, dy, dw, N, Cin, H, W, Cout, K, pad, dil);
    if (rc) return rc;
    WgradGeom wg;
    rc = plan_wgrad(wg, N, Cin, H, W, Cout, K, pad, dil);
    if (rc) return rc;
    if (!ws || ws_bytes < wgrad_ws_floats(wg) * 4) return afd::fail(AFD_ERR_WORKSPACE, "conv wgrad: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(ws);
    float* partb = part + (size_t)wg.S * wg.c.nchunks * wg.c.CO_PAD * wg.NCOL;
    const int ntw = (wg.NTILES + wg.NG - 1) / wg.NG;
    if (ntw <= 1) rc = launch_wgrad_t<1>(wg, x, dy, part, partb, s);
    else if (ntw <= 3) rc = launch_wgrad_t<3>(wg, x, dy, part, partb, s);
    else if (ntw <= 5) rc = launch_wgrad_t<5>(wg, x, dy, part, partb, s);
    else if (ntw <= 9) rc = launch_wgrad_t<9>(wg, x, dy, part, partb, s);
    else return afd::fail(AFD_ERR_UNSUPPORTED, "wgrad: %d column tiles per wave", ntw);
    if (rc) return rc;
    const int total = Cout * Cin * K * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, part, partb,
                       dw, dbias, Cin, Cout, K * K, wg.c.CI_T, wg.c.nchunks, wg.c.CO_PAD, wg.NCOL,
                       wg.S);
    return afd::check_launch("wgrad_reduce_kernel");
}
