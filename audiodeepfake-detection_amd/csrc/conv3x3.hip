// 3x3 (dilation 1) forward / backward-data on wide images: implicit GEMM with a vector-ALU-free
// inner loop.
//
// The f32 MFMA runs at the vector-ALU rate (MI355X_MICROARCH.md, "Matrix cores"), so address
// arithmetic between MFMAs is not hidden behind them, it competes with them: the
// general kernel (conv.hip, conv_igemm_kernel) spends ~8 VALU instructions per MFMA on its
// k -> (channel, ky, kx) table and fragment addresses and tops out near 50 % of the MFMA peak.
// This kernel fixes the tile geometry at compile time -- one output row, PIX pixels, channel
// chunks of 8, patch [8][3][PIX + 4] -- and orders K as (channel pair, ky, kx, channel parity):
// the two halves of a wave (k parity of v_mfma_f32_32x32x2_f32) then differ by ONE channel
// plane, a per-lane constant, and every A / B fragment of a chunk is a ds_read at
// lane base + compile-time immediate.  The chunk loop is unrolled (36 k-steps) with the reads
// of step s+1 placed between the MFMAs of step s.
// Staging: interior tiles load at a wave-uniform base + a per-thread constant offset (dwordx4,
// no index arithmetic); border tiles take a predicated per-element path.
// Used for the level-14 DCNN blocks 3-6 (models.py:262-270) in both directions; everything
// else stays on conv_igemm_kernel.  Same two-level fp32 accumulation as there.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kCT = 8;        // channels per chunk
constexpr int kPairs = kCT / 2;
constexpr int kKsteps = kPairs * 9;  // 36 k-steps of two channels
constexpr int kPR = 3;        // patch rows

struct G3 {
    int N, Cin, H, W, Cout;  // of THIS convolution (backward-data: roles already swapped)
    int pad;                 // 1 (forward) or 1 (backward-data of pad 1): out = in + 2 pad - 2
    int Hout, Wout;
    int Hc, Wc;  // computed extent (<= Hout, Wout): rows / columns beyond are not produced
    int nchunks, tilesX, tilesY;
};

// weights w[Cout][Cin][3][3] -> wp[chunk][kstep = (pair, ky, kx)][parity][CO_PAD]
// dgrad: the output channels are the forward's input channels, taps flipped
__global__ void repack3_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                               int CO_PAD, int nchunks, int dgrad) {
    const int total = nchunks * kKsteps * 2 * CO_PAD;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % CO_PAD;
        const int par = (i / CO_PAD) & 1;
        const int ks = (i / (2 * CO_PAD)) % kKsteps;
        const int chunk = i / (2 * CO_PAD * kKsteps);
        const int pair = ks / 9, r = ks - pair * 9;
        const int ci = chunk * kCT + 2 * pair + par;
        float v = 0.f;
        if (ci < Cin && co < Cout)
            v = dgrad ? w[((size_t)ci * Cout + co) * 9 + (8 - r)] : w[((size_t)co * Cin + ci) * 9 + r];
        wp[i] = v;
    }
}

// MW x NW 32x32 tiles per wave, WNB waves along the pixel axis; PIX = NW * WNB * 32 pixels per
// workgroup, arranged as TH rows of TW columns: TW = PIX (one row; wide level-14 images) or
// TW = 32 (PIX / 32 rows; the narrow level-8 / STFT / LCNN images).  The LDS patch pitch depends
// on the tile shape only, never on the image width, so the fragment offsets stay immediates.
template <int MW, int NW, int WMB, int WNB, bool TWO, int TW>
__global__ void __launch_bounds__(WMB * WNB * 64) __attribute__((amdgpu_waves_per_eu(TWO ? (MW * NW <= 3 ? 3 : 2) : (MW * NW <= 3 ? 4 : 3))))
conv3x3_kernel(const G3 g, const float* __restrict__ x, const float* __restrict__ wp,
               const float* __restrict__ bias, float* __restrict__ y) {
    constexpr int NT = WMB * WNB * 64;
    constexpr int CO_PAD = MW * WMB * 32;
    constexpr int PIX = NW * WNB * 32;
    static_assert(TW == 32 || TW == PIX, "tile width");
    constexpr int TH = PIX / TW;
    constexpr int kPR = TH + 2;  // patch rows (shadows the one-row constant of the file)
    constexpr int PC = TW + 4;   // 2 halo columns + 2 of padding: whole groups of four per row
    constexpr int STEP = TW == 32 ? PC : 32;  // patch offset between consecutive 32-pixel tiles
    constexpr int G4 = PC / 4;
    constexpr int WFLOATS = kKsteps * 2 * CO_PAD;
    constexpr int PFLOATS = kCT * kPR * PC;
    constexpr int WV = (WFLOATS / 4 + NT - 1) / NT;     // weight float4 per thread
    constexpr int PV = (kCT * kPR * G4 + NT - 1) / NT;  // patch groups per thread
    __shared__ __attribute__((aligned(16))) float wl[WFLOATS];
    __shared__ __attribute__((aligned(16))) float patch[PFLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wmb = wave % WMB, wnb = wave / WMB;
    const int tpi = g.tilesX * g.tilesY;
    const int n = blockIdx.x / tpi;
    const int t = blockIdx.x - n * tpi;
    const int ty = t / g.tilesX;
    const int oy0 = ty * TH;
    const int ox0 = (t - ty * g.tilesX) * TW;
    const int iy0 = oy0 - g.pad, ix0 = ox0 - g.pad;
    const size_t iplane = (size_t)g.H * g.W;

    // per-thread staging offsets (elements), constant over the chunks
    unsigned xo[PV];
#pragma unroll
    for (int u = 0; u < PV; ++u) {
        const int item = tid + u * NT;
        const int row = item / G4, g4 = item - row * G4;
        const int ci_l = row / kPR, pr = row - ci_l * kPR;
        xo[u] = item < kCT * kPR * G4 ? (unsigned)ci_l * (unsigned)(g.H * g.W) + (unsigned)(pr * g.W + 4 * g4) : 0u;
    }
    const bool interior = iy0 >= 0 && iy0 + kPR <= g.H && ix0 >= 0 && ix0 + PC <= g.W;

    // fragment bases (floats): A = wl + parity row + channel column, B = patch + parity plane + pixel
    const float* abase = wl + half * CO_PAD + wmb * MW * 32 + l31;
    const float* bbase = patch + half * kPR * PC + wnb * NW * STEP + l31;

    f32x16 acc[MW][NW];
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;

    const float* xn = x + (size_t)n * g.Cin * iplane;
    // staging registers: the loads of chunk c+1 are issued before the k-loop of chunk c and
    // written to LDS after it (interior tiles: no address arithmetic, nothing for the MFMAs to
    // wait on)
    f32x4u wv[WV], pv[PV];
    auto load_chunk = [&](int chunk) {
        const f32x4u* src = reinterpret_cast<const f32x4u*>(wp + (size_t)chunk * WFLOATS);
#pragma unroll
        for (int u = 0; u < WV; ++u) {
            const int i = tid + u * NT;
            if (i < WFLOATS / 4) wv[u] = src[i];
        }
        const float* xc = xn + (size_t)chunk * kCT * iplane;
        if (interior) {
            const float* xb = xc + (size_t)iy0 * g.W + ix0;
#pragma unroll
            for (int u = 0; u < PV; ++u) pv[u] = *reinterpret_cast<const f32x4u*>(xb + xo[u]);
        } else {
#pragma unroll
            for (int u = 0; u < PV; ++u) {
                const int item = tid + u * NT;
                const int row = item / G4, g4 = item - row * G4;
                const int ci_l = row / kPR, pr = row - ci_l * kPR;
                const int iy = iy0 + pr, ix = ix0 + 4 * g4;
                f32x4u v = {0.f, 0.f, 0.f, 0.f};
                if (item < kCT * kPR * G4 && iy >= 0 && iy < g.H) {
                    const float* src4 = xc + (size_t)ci_l * iplane + (size_t)iy * g.W + ix;
                    if (ix >= 0 && ix + 3 < g.W) {
                        v = *reinterpret_cast<const f32x4u*>(src4);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ix + j >= 0 && ix + j < g.W) v[j] = src4[j];
                    }
                }
                pv[u] = v;
            }
        }
    };
    load_chunk(0);
    for (int chunk = 0; chunk < g.nchunks; ++chunk) {
        __syncthreads();  // the previous chunk's fragments have been read
        {
            float4* wd = reinterpret_cast<float4*>(wl);
#pragma unroll
            for (int u = 0; u < WV; ++u) {
                const int i = tid + u * NT;
                if (i < WFLOATS / 4) wd[i] = make_float4(wv[u][0], wv[u][1], wv[u][2], wv[u][3]);
            }
            float4* pd = reinterpret_cast<float4*>(patch);
#pragma unroll
            for (int u = 0; u < PV; ++u) {
                const int item = tid + u * NT;
                if (item < kCT * kPR * G4) pd[item] = make_float4(pv[u][0], pv[u][1], pv[u][2], pv[u][3]);
            }
        }
        __syncthreads();
        if (chunk + 1 < g.nchunks) load_chunk(chunk + 1);
        // ---- 36 k-steps, fragment addresses = lane base + immediate ----
        // TWO: one fp32 chain per channel chunk, added to the running sum afterwards (conv
        // outputs within ~5e-7 of fp64 instead of ~2e-6, for the price of MW*NW*16 adds and
        // registers per chunk)
        f32x16 part[TWO ? MW : 1][TWO ? NW : 1];
        if (TWO) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int i = 0; i < NW; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[m][i][r] = 0.f;
        }
        float a0[MW], b0[NW], a1[MW], b1[NW];
#pragma unroll
        for (int m = 0; m < MW; ++m) a0[m] = abase[m * 32];
#pragma unroll
        for (int i = 0; i < NW; ++i) b0[i] = bbase[i * STEP];
#pragma unroll
        for (int ks = 0; ks < kKsteps; ks += 2) {
            {
                constexpr int dummy = 0;
                (void)dummy;
                const int k1 = ks + 1;
                const int pair = k1 / 9, r = k1 - pair * 9, ky = r / 3, kx = r - ky * 3;
#pragma unroll
                for (int m = 0; m < MW; ++m) a1[m] = abase[k1 * 2 * CO_PAD + m * 32];
#pragma unroll
                for (int i = 0; i < NW; ++i) b1[i] = bbase[(2 * pair * kPR + ky) * PC + kx + i * STEP];
            }
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    if (TWO) part[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[i], part[m][i], 0, 0, 0);
                    else acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m], b0[i], acc[m][i], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < MW * NW; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 2 < kKsteps) {
                const int k2 = ks + 2;
                const int pair = k2 / 9, r = k2 - pair * 9, ky = r / 3, kx = r - ky * 3;
#pragma unroll
                for (int m = 0; m < MW; ++m) a0[m] = abase[k2 * 2 * CO_PAD + m * 32];
#pragma unroll
                for (int i = 0; i < NW; ++i) b0[i] = bbase[(2 * pair * kPR + ky) * PC + kx + i * STEP];
            }
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    if (TWO) part[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[i], part[m][i], 0, 0, 0);
                    else acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m], b1[i], acc[m][i], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < MW * NW; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (TWO) {
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int i = 0; i < NW; ++i) acc[m][i] += part[m][i];
        }
    }

    // D layout: column = lane & 31 (pixel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const size_t oplane = (size_t)g.Hout * g.Wout;
    float* yn = y + (size_t)n * g.Cout * oplane;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int p = (wnb * NW + i) * 32 + l31;  // pixel inside the tile
        const int oy = oy0 + p / TW, ox = ox0 + p % TW;
        if (oy >= g.Hc || ox >= g.Wc) continue;
        float* yp = yn + (size_t)oy * g.Wout + ox;
#pragma unroll
        for (int m = 0; m < MW; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (wmb * MW + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < g.Cout) yp[(size_t)co * oplane] = acc[m][i][r] + (bias ? bias[co] : 0.f);
            }
        }
    }
}

template <int MW, int NW, int WMB, int WNB, int TW>
int launch3(G3 g, const float* x, const float* wp, const float* bias, float* y, hipStream_t s) {
    constexpr int PIX = NW * WNB * 32;
    constexpr int TH = PIX / TW;
    g.tilesX = (g.Wc + TW - 1) / TW;
    g.tilesY = (g.Hc + TH - 1) / TH;
    const long blocks = (long)g.N * g.tilesY * g.tilesX;
    if (blocks > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3: grid too large");
    hipLaunchKernelGGL((conv3x3_kernel<MW, NW, WMB, WNB, false, TW>), dim3((unsigned)blocks),
                       dim3(WMB * WNB * 64), 0, s, g, x, wp, bias, y);
    return afd::check_launch("conv3x3_kernel");
}

// one-row tiles for wide images, 32-column tiles otherwise
template <int MW, int NW, int WMB, int WNB>
int launch3_shape(const G3& g, const float* x, const float* wp, const float* bias, float* y, hipStream_t s) {
    if (g.W >= 1024) return launch3<MW, NW, WMB, WNB, NW * WNB * 32>(g, x, wp, bias, y, s);
    return launch3<MW, NW, WMB, WNB, 32>(g, x, wp, bias, y, s);
}

int run3(const G3& g, const float* x, const float* wp, const float* bias, float* y, hipStream_t s) {
    switch ((g.Cout + 31) / 32) {
        case 1: return launch3_shape<1, 2, 1, 4>(g, x, wp, bias, y, s);  // 32 ch x 256 px
        case 2: return launch3_shape<1, 2, 2, 2>(g, x, wp, bias, y, s);  // 64 ch x 128 px
        case 3: return launch3_shape<3, 1, 1, 4>(g, x, wp, bias, y, s);  // 96 ch x 128 px (x 256 px: 97 vs 115 TF/s)
        case 4: return launch3_shape<2, 2, 2, 2>(g, x, wp, bias, y, s);  // 128 ch x 128 px
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3: Cout %d > 128", g.Cout);
}

// ---------------------------------------------------------------------------------------
// backward-weight for the same layers: dw[co][ci][ky][kx] = sum_p dz[co][p] x[ci][p + tap].
// GEMM with M = output channels, N = (ci, ky, kx) columns of a chunk of CT channels,
// K = pixels in tiles of 64 (one output row of 64, or two rows of 32 on narrow images).  The (channel tile, column tile) pairs of a
// chunk are dealt to four waves, TPW each, and stay in accumulators over the whole tile list
// of the workgroup; per tile: loads -> barrier -> LDS stores -> barrier -> 32 k-steps whose
// fragment reads are lane base + immediate.  dz rows have pitch 65 and the patch rows pitch 68
// (whole float4 groups; the 32 columns of a B read land on >= 24 banks).
// One partial slab per (split, chunk), summed by conv.hip's wgrad_reduce_kernel.
// ---------------------------------------------------------------------------------------
constexpr int kWPix = 64;
constexpr int kWPitch = 65;

struct GW {
    int N, Cin, H, W, Cout;  // forward geometry; dz is [N][Cout][H][W]
    int Hc, Wc;              // dz is zero outside [:Hc, :Wc]: tiles cover that extent only
    int tilesX, tilesY, S, nchunks;
    int ntiles;
    // tile-column subset of this launch: tx = ncl ? cl[k] : col0 + k, k < ncols; slabs start at split0
    int ncols, col0, ncl, cl[4], split0;
};

template <int MT, int CT, int TPW, int TW>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad3x3_kernel(const GW g, const float* __restrict__ x, const float* __restrict__ dz,
                float* __restrict__ part, float* __restrict__ partb) {
    constexpr int CO_PAD = MT * 32;
    static_assert(TW == 64 || TW == 32, "tile width");
    constexpr int TH = kWPix / TW;
    constexpr int kPR = TH + 2;   // patch rows (shadows the one-row constant of the file)
    constexpr int kWPC = TW + 4;  // patch pitch: halo + padding to whole float4 groups
    constexpr int NTILES = CT * 9 / 32;
    static_assert(CT * 9 % 32 == 0, "whole column tiles");
    constexpr int PAIRS = MT * NTILES;
    static_assert(4 * TPW >= PAIRS, "pairs per wave");
    constexpr int NCOL = NTILES * 32;
    constexpr int DZF = CO_PAD * kWPitch;
    constexpr int PF = CT * kPR * kWPC;
    constexpr int G4 = kWPC / 4;
    constexpr int DV = CO_PAD * 16 / 256;             // dz groups (4 pixels) per thread
    constexpr int PV = (CT * kPR * G4 + 255) / 256;   // patch groups per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];  // dz | patch | bias cells
    float* patch = smem + DZF;
    float* bsl = smem + DZF + PF;  // [CO_PAD][16]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int chunk = blockIdx.y, split = blockIdx.x;
    const size_t plane = (size_t)g.H * g.W;

    for (int i = tid; i < CO_PAD * 16; i += 256) bsl[i] = 0.f;

    // staging offsets (elements) inside an interior tile
    unsigned dzo[DV], xo[PV];
#pragma unroll
    for (int u = 0; u < DV; ++u) {
        const int item = tid + u * 256;
        const int co = item >> 4, px = (item & 15) << 2;
        dzo[u] = co < g.Cout ? (unsigned)co * (unsigned)(g.H * g.W) + (unsigned)((px / TW) * g.W + px % TW) : 0u;
    }
#pragma unroll
    for (int u = 0; u < PV; ++u) {
        const int item = tid + u * 256;
        const int row = item / G4, g4 = item - row * G4;
        const int ci_l = row / kPR, pr = row - ci_l * kPR;
        xo[u] = item < CT * kPR * G4 ? (unsigned)ci_l * (unsigned)(g.H * g.W) + (unsigned)(pr * g.W + 4 * g4) : 0u;
    }
    // fragment bases
    const float* ap[TPW];
    const float* bp[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int p = wave * TPW + q;
        const int pp = p < PAIRS ? p : 0;
        const int m = pp / NTILES, nt = pp - m * NTILES;
        const int c = nt * 32 + l31;
        const int ci_l = c / 9, r = c - ci_l * 9, ky = r / 3, kx = r - ky * 3;
        ap[q] = smem + (m * 32 + l31) * kWPitch + half;
        bp[q] = patch + (ci_l * kPR + ky) * kWPC + kx + half;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int tpi = g.ncols * g.tilesY;
    for (int tile = split; tile < g.ntiles; tile += g.S) {
        const int n = tile / tpi;
        const int t = tile - n * tpi;
        const int ty = t / g.ncols;
        const int oy = ty * TH;
        const int kx_ = t - ty * g.ncols;
        const int ox0 = (g.ncl ? g.cl[kx_] : g.col0 + kx_) * TW;
        const int iy0 = oy - 1, ix0 = ox0 - 1;
        const float* dzn = dz + (size_t)n * g.Cout * plane + (size_t)oy * g.W + ox0;
        const float* xc = x + ((size_t)n * g.Cin + (size_t)chunk * CT) * plane;
        {
            f32x4u dv[DV], pv[PV];
            const bool interior = ox0 + TW <= g.W && oy + TH <= g.H && ix0 >= 0 && ix0 + kWPC <= g.W &&
                                  iy0 >= 0 && iy0 + kPR <= g.H;
            if (interior) {
                const float* xb = xc + (size_t)iy0 * g.W + ix0;
#pragma unroll
                for (int u = 0; u < DV; ++u) dv[u] = *reinterpret_cast<const f32x4u*>(dzn + dzo[u]);
#pragma unroll
                for (int u = 0; u < PV; ++u) pv[u] = *reinterpret_cast<const f32x4u*>(xb + xo[u]);
            } else {
#pragma unroll
                for (int u = 0; u < DV; ++u) {
                    const int item = tid + u * 256;
                    const int co = item >> 4, px = (item & 15) << 2;
                    f32x4u v = {0.f, 0.f, 0.f, 0.f};
                    const int r = px / TW, c = px % TW;
                    if (co < g.Cout && oy + r < g.H) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ox0 + c + j < g.W) v[j] = dzn[(size_t)co * plane + (size_t)r * g.W + c + j];
                    }
                    dv[u] = v;
                }
#pragma unroll
                for (int u = 0; u < PV; ++u) {
                    const int item = tid + u * 256;
                    const int row = item / G4, g4 = item - row * G4;
                    const int ci_l = row / kPR, pr = row - ci_l * kPR;
                    const int iy = iy0 + pr, ix = ix0 + 4 * g4;
                    f32x4u v = {0.f, 0.f, 0.f, 0.f};
                    if (item < CT * kPR * G4 && iy >= 0 && iy < g.H) {
                        const float* src = xc + (size_t)ci_l * plane + (size_t)iy * g.W + ix;
                        if (ix >= 0 && ix + 3 < g.W) {
                            v = *reinterpret_cast<const f32x4u*>(src);
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (ix + j >= 0 && ix + j < g.W) v[j] = src[j];
                        }
                    }
                    pv[u] = v;
                }
            }
            __syncthreads();  // the previous tile's fragments have been read (first pass: bias cells are zero)
#pragma unroll
            for (int u = 0; u < DV; ++u) {
                const int item = tid + u * 256;
                const int co = item >> 4, px = (item & 15) << 2;
                float* d = smem + co * kWPitch + px;
                const bool live = co < g.Cout;  // padded channel rows stay zero
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = live ? dv[u][j] : 0.f;
                if (chunk == 0 && live) bsl[item] += (dv[u][0] + dv[u][1]) + (dv[u][2] + dv[u][3]);
            }
#pragma unroll
            for (int u = 0; u < PV; ++u) {
                const int item = tid + u * 256;
                if (item < CT * kPR * G4)
                    *reinterpret_cast<float4*>(patch + 4 * item) = make_float4(pv[u][0], pv[u][1], pv[u][2], pv[u][3]);
            }
            __syncthreads();
        }
        // pixel k of the tile sits (k / TW) patch rows down, k % TW columns in
        auto boff = [](int k) { return (k / TW) * kWPC + k % TW; };
        float a0[TPW], a1[TPW], b0[TPW], b1[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            a0[q] = ap[q][0];
            b0[q] = bp[q][0];
        }
#pragma unroll
        for (int ks = 0; ks < kWPix / 2; ks += 2) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                a1[q] = ap[q][2 * ks + 2];
                b1[q] = bp[q][boff(2 * ks + 2)];
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
            if (ks + 2 < kWPix / 2) {
#pragma unroll
                for (int q = 0; q < TPW; ++q) {
                    a0[q] = ap[q][2 * ks + 4];
                    b0[q] = bp[q][boff(2 * ks + 4)];
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

    float* slab = part + ((size_t)(g.split0 + split) * g.nchunks + chunk) * CO_PAD * NCOL;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int p = wave * TPW + q;
        if (p >= PAIRS) continue;
        const int m = p / NTILES, nt = p - m * NTILES;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[(size_t)co * NCOL + nt * 32 + l31] = acc[q][r];
        }
    }
    __syncthreads();  // the last tile's bias cells are written
    if (chunk == 0 && tid < CO_PAD) {
        float v = 0.f;
        for (int j = 0; j < 16; ++j) v += bsl[tid * 16 + j];  // fixed order
        partb[(size_t)(g.split0 + split) * CO_PAD + tid] = v;
    }
}

// Second layout of the same GEMM for the interior tile columns of wide images (one row of 64 pixels per
// tile): 8 waves with the (channel tile, column tile) pairs dealt round-robin (<= 5 per wave instead of
// 7-9: no spills, 96 % balance per SIMD), and the NEXT tile's global loads issued before the k-loop
// into registers (the single workgroup per CU covers its own load latency).  The load path has no
// branch -- a join makes the compiler wait for the loads where they are issued: rows above / below
// the image are read from the clamped row and zeroed when stored to LDS, and the first / last tile
// columns (patches that cross the left / right edge) go to wgrad3x3_kernel as a second launch.
// CT = input channels per workgroup: 32, or 64 (two 32-channel chunks of the slab layout at once: dz is staged
// once for both, twice the matrix instructions per barrier)
template <int MT, int CT = 32>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((MT <= 2 && CT == 32) ? 4 : 2, (MT <= 2 && CT == 32) ? 4 : 2)))
wgrad3x3p_kernel(const GW g, const float* __restrict__ x, const float* __restrict__ dz,
                 float* __restrict__ part, float* __restrict__ partb) {
    constexpr int TW = 64, NTH = 512, NW = 8;
    constexpr int CO_PAD = MT * 32;
    constexpr int kPR = 3;
    constexpr int kWPC = TW + 4;
    constexpr int NTILES = CT * 9 / 32;  // 9
    constexpr int PAIRS = MT * NTILES;
    constexpr int TPW = (PAIRS + NW - 1) / NW;
    constexpr int NCOL = NTILES * 32;
    constexpr int DZF = CO_PAD * kWPitch;
    constexpr int PF = CT * kPR * kWPC;
    constexpr int G4 = kWPC / 4;                       // 17 groups of 4 per patch row
    constexpr int DV = CO_PAD * 16 / NTH;              // = MT
    constexpr int PGROUPS = CT * kPR * G4;             // 1632
    constexpr int PV = (PGROUPS + NTH - 1) / NTH;      // 4
    extern __shared__ __attribute__((aligned(16))) float smem[];  // dz | patch | bias cells
    float* patch = smem + DZF;
    float* bsl = smem + 2 * (DZF + PF);  // [CO_PAD][16], after the two tile images

    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int chunk = blockIdx.y, split = blockIdx.x;
    const size_t plane = (size_t)g.H * g.W;

    for (int i = tid; i < CO_PAD * 16; i += NTH) bsl[i] = 0.f;

    unsigned dzo[DV], xo[PV];
    int xpr[PV];
#pragma unroll
    for (int u = 0; u < DV; ++u) {
        const int item = tid + u * NTH;
        const int co = item >> 4, px = (item & 15) << 2;
        dzo[u] = co < g.Cout ? (unsigned)co * (unsigned)(g.H * g.W) + (unsigned)px : 0u;
    }
#pragma unroll
    for (int u = 0; u < PV; ++u) {
        const int item = tid + u * NTH;
        const bool live = item < PGROUPS;
        const int row = live ? item / G4 : 0, g4 = live ? item - row * G4 : 0;
        const int ci_l = row / kPR, pr = row - ci_l * kPR;
        xo[u] = (unsigned)ci_l * (unsigned)(g.H * g.W) + (unsigned)(4 * g4);
        xpr[u] = pr;
    }
    const float* ap[TPW];
    const float* bp[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int p = wave + NW * q;  // round-robin
        const int pp = p < PAIRS ? p : 0;
        const int m = pp / NTILES, nt = pp - m * NTILES;
        const int c = nt * 32 + l31;
        const int ci_l = c / 9, r = c - ci_l * 9, ky = r / 3, kx = r - ky * 3;
        ap[q] = smem + (m * 32 + l31) * kWPitch + half;
        bp[q] = patch + (ci_l * kPR + ky) * kWPC + kx + half;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int tpi = g.ncols * g.tilesY;
    f32x4u dv[DV], pv[PV];
    unsigned rowbad = 0;
    auto load_tile = [&](int tile) {
        const int n = tile / tpi;
        const int t = tile - n * tpi;
        const int ty = t / g.ncols;
        const int ox0 = (g.col0 + (t - ty * g.ncols)) * TW;
        const float* dzn = dz + (size_t)n * g.Cout * plane + (size_t)ty * g.W + ox0;
        const float* xb = x + ((size_t)n * g.Cin + (size_t)chunk * CT) * plane + (ox0 - 1);
        rowbad = 0;
#pragma unroll
        for (int u = 0; u < DV; ++u) dv[u] = *reinterpret_cast<const f32x4u*>(dzn + dzo[u]);
#pragma unroll
        for (int u = 0; u < PV; ++u) {
            const int iy = ty - 1 + xpr[u];
            const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
            rowbad |= (iy != iyc) ? (1u << u) : 0u;
            pv[u] = *reinterpret_cast<const f32x4u*>(xb + xo[u] + (unsigned)(iyc * g.W));
        }
    };

    // LDS image of a tile: dz | patch, two of them used alternately; bias cells after both
    constexpr int IMG = DZF + PF;
    auto store_tile = [&](int buf) {
        float* base = smem + buf * IMG;
#pragma unroll
        for (int u = 0; u < DV; ++u) {
            const int item = tid + u * NTH;
            const int co = item >> 4, px = (item & 15) << 2;
            float* d = base + co * kWPitch + px;
            const bool live = co < g.Cout;  // padded channel rows stay zero
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = live ? dv[u][j] : 0.f;
            if (chunk == 0 && live) bsl[item] += (dv[u][0] + dv[u][1]) + (dv[u][2] + dv[u][3]);
        }
#pragma unroll
        for (int u = 0; u < PV; ++u) {
            const int item = tid + u * NTH;
            const bool z = (rowbad >> u) & 1u;
            if (item < PGROUPS)
                *reinterpret_cast<float4*>(base + DZF + 4 * item) =
                    make_float4(z ? 0.f : pv[u][0], z ? 0.f : pv[u][1], z ? 0.f : pv[u][2], z ? 0.f : pv[u][3]);
        }
    };
    auto k_loop = [&](int buf) {
        const int bo = buf * IMG;
        float a0[TPW], a1[TPW], b0[TPW], b1[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            a0[q] = ap[q][bo];
            b0[q] = bp[q][bo];
        }
#pragma unroll
        for (int ks = 0; ks < kWPix / 2; ks += 2) {
#pragma unroll
            for (int q = 0; q < TPW; ++q) {
                a1[q] = ap[q][bo + 2 * ks + 2];
                b1[q] = bp[q][bo + 2 * ks + 2];
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
            if (ks + 2 < kWPix / 2) {
#pragma unroll
                for (int q = 0; q < TPW; ++q) {
                    a0[q] = ap[q][bo + 2 * ks + 4];
                    b0[q] = bp[q][bo + 2 * ks + 4];
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
    };

    // Iteration i works on tile t_i out of image i & 1 while tile t_{i+1} (in registers since the
    // previous iteration) is stored into the other image and t_{i+2} is requested: one barrier per
    // tile.  Waves 0-3 store first and then run their k-loop, waves 4-7 the other way round, so on
    // every SIMD one wave is in its matrix phase while the other one stores.
    const bool early = wave < 4;
    __syncthreads();  // bias cells are zero
    int tile = split;
    if (tile < g.ntiles) {
        load_tile(tile);
        store_tile(0);
        if (tile + g.S < g.ntiles) load_tile(tile + g.S);
    }
    __syncthreads();
    for (int it = 0; tile < g.ntiles; tile += g.S, ++it) {
        const bool more = tile + g.S < g.ntiles;
        if (early && more) {
            store_tile((it + 1) & 1);
            if (tile + 2 * g.S < g.ntiles) load_tile(tile + 2 * g.S);
        }
        k_loop(it & 1);
        if (!early && more) {
            store_tile((it + 1) & 1);
            if (tile + 2 * g.S < g.ntiles) load_tile(tile + 2 * g.S);
        }
        __syncthreads();
    }

    // slabs are per 32-channel chunk ([CO_PAD][288]): column tile nt of this workgroup is tile nt % 9 of chunk
    // (CT / 32) * blockIdx.y + nt / 9
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int p = wave + NW * q;
        if (p >= PAIRS) continue;
        const int m = p / NTILES, nt = p - m * NTILES;
        float* slab = part + ((size_t)(g.split0 + split) * g.nchunks + (CT / 32) * chunk + nt / 9) * CO_PAD * 288;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[(size_t)co * 288 + (nt % 9) * 32 + l31] = acc[q][r];
        }
    }
    __syncthreads();  // the last tile's bias cells are written
    if (chunk == 0 && tid < CO_PAD) {
        float v = 0.f;
        for (int j = 0; j < 16; ++j) v += bsl[tid * 16 + j];  // fixed order
        partb[(size_t)(g.split0 + split) * CO_PAD + tid] = v;
    }
}

template <int MT, int CT = 32>
int launchw_p(const GW& g, const float* x, const float* dz, float* part, float* partb, hipStream_t s) {
    constexpr size_t lds = (size_t)(2 * (MT * 32 * kWPitch + CT * 3 * 68) + MT * 32 * 16) * 4;  // two tile images
    static_assert(lds <= 160 * 1024, "two tile images fit the LDS");
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3p_kernel<MT, CT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wgrad3x3: %s", hipGetErrorString(e));
        attr.mark();
    }
    hipLaunchKernelGGL((wgrad3x3p_kernel<MT, CT>), dim3(g.S, g.nchunks * 32 / CT), dim3(512), lds, s, g, x, dz, part, partb);
    return afd::check_launch("wgrad3x3p_kernel");
}

template <int MT, int CT, int TPW, int TW>
int launchw_t(const GW& g, const float* x, const float* dz, float* part, float* partb, hipStream_t s) {
    constexpr size_t lds = (size_t)(MT * 32 * kWPitch + CT * (kWPix / TW + 2) * (TW + 4) + MT * 32 * 16) * 4;
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_kernel<MT, CT, TPW, TW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wgrad3x3: %s", hipGetErrorString(e));
        attr.mark();
    }
    hipLaunchKernelGGL((wgrad3x3_kernel<MT, CT, TPW, TW>), dim3(g.S, g.nchunks), dim3(256), lds, s, g, x, dz,
                       part, partb);
    return afd::check_launch("wgrad3x3_kernel");
}

template <int MT, int CT, int TPW>
int launchw(const GW& g, const float* x, const float* dz, float* part, float* partb, hipStream_t s) {
    if (g.W >= 1024) return launchw_t<MT, CT, TPW, 64>(g, x, dz, part, partb, s);
    return launchw_t<MT, CT, TPW, 32>(g, x, dz, part, partb, s);
}

// tile shape for an image width: one row of 64 pixels, or two rows of 32
int wgrad_tw(int W) { return W >= 1024 ? 64 : 32; }

// channels per chunk (64 with 32 output channels measured slower: 58 vs 68 TF/s on 128 -> 32)
int wgrad_ct(int, int) { return 32; }

}  // namespace

namespace afd {

bool wgrad3x3_applicable(int Cin, int H, int W, int Cout, int K, int pad, int dil) {
    if (getenv("AFD_NO_WGRAD3X3")) return false;
    if (K != 3 || dil != 1 || pad != 1 || W < 24 || H < 2 || Cout > 128) return false;
    const int mt = (Cout + 31) / 32;
    if (Cin % wgrad_ct(mt, Cin) != 0) return false;
    return (size_t)128 * H * W < 0x7fffffffULL;
}

// Tile columns of a wide image (64-pixel tiles): columns 1 .. last are interior (their patch and dz
// loads stay inside the image rows), column 0 and the ones after `last` cross an edge.
struct WgCols {
    bool split;   // two launches: wgrad3x3p_kernel on the interior columns, wgrad3x3_kernel on the rest
    int tilesX, ninner, nborder, border[4];
};

// mt = 32-channel tiles of Cout.  Measured at B = 32 (new vs general kernel, 32 input channels per workgroup): mt 4 1.70 vs 1.80 ms (the
// general kernel spills 116 registers there), mt 3 3.73 vs 3.46, mt 2 0.34 vs 0.31, mt 1 0.87 vs 0.72 ms --
// two workgroups of four waves cover each other's loads better than one of eight covers its own, so
// only the 128-channel layer takes the split.
WgCols wgrad_cols(int W, int wc, int mt, int cin) {
    WgCols c{};
    c.tilesX = (wc + 63) / 64;
    // the 96-channel layer on 64 input channels per workgroup (54 pairs on 8 waves, dz staged once): 12.4 -> 11.1 ms
    // at level 14; for one channel tile of Cout the same change measured level (2.9 ms)
    const bool ct64 = mt == 3 && cin % 64 == 0;
    if (W < 1024 || getenv("AFD_NO_WGRAD3X3P") || (mt != 4 && !ct64)) return c;
    int last = 0;
    for (int tx = 1; tx < c.tilesX; ++tx)
        if (tx * 64 + 64 + 3 <= W) last = tx;
    c.ninner = last;
    c.border[c.nborder++] = 0;
    for (int tx = last + 1; tx < c.tilesX && c.nborder < 4; ++tx) c.border[c.nborder++] = tx;
    c.split = c.ninner >= 8 && 1 + (c.tilesX - 1 - last) == c.nborder;
    return c;
}

long coprime_splits(long s, long tx) {
    // A split count that shares a factor with the tiles per row makes every workgroup walk down
    // one tile column in lock step (measured 80 instead of 109 TF/s at 512 splits x 128 tiles per
    // row): take the next smaller count coprime to it.
    auto gcd = [](long a, long b) { while (b) { const long t = a % b; a = b; b = t; } return a; };
    while (s > 1 && gcd(s, tx) != 1) --s;
    return s;
}

// split counts of the two launches (S2 = 0: single launch of wgrad3x3_kernel over all columns)
void wgrad_splits(int N, int tilesY, int nchunks, const WgCols& c, int* S1, int* S2) {
    if (!c.split) {
        const long tiles = (long)N * tilesY * c.tilesX;
        long s = 1024 / nchunks;
        if (s < 1) s = 1;
        if (s > tiles) s = tiles;
        *S1 = (int)coprime_splits(s, c.tilesX);
        *S2 = 0;
        return;
    }
    const long ti = (long)N * tilesY * c.ninner, tb = (long)N * tilesY * c.nborder;
    // one to two workgroups per CU over the whole launch: 1024 / nchunks measured the same kernel time with twice
    // the slabs for the reduction to read
    long s = 512 / nchunks;
    if (s < 1) s = 1;
    if (s > ti) s = ti;
    *S1 = (int)coprime_splits(s, c.ninner);
    // the edge tile columns are a separate launch of the general kernel: enough splits to fill the chip (with
    // 128 / nchunks half the CUs sat idle: the backward-weight class 21.6 -> 21.0 ms)
    long sb = 512 / nchunks;
    if (sb < 1) sb = 1;
    if (sb > tb) sb = tb;
    *S2 = (int)sb;
}

// slab geometry for conv.hip's reduction: S splits x nchunks slabs of [CO_PAD][NCOL]
void wgrad3x3_geometry(int N, int Cin, int H, int W, int Cout, int dz_rows, int dz_cols, int* S,
                       int* nchunks, int* CI_T, int* CO_PAD, int* NCOL) {
    const int mt = (Cout + 31) / 32;
    const int ct = wgrad_ct(mt, Cin);
    *CI_T = ct;
    *nchunks = Cin / ct;
    *CO_PAD = mt * 32;
    *NCOL = ct * 9;
    const int hc = dz_rows < H ? dz_rows : H, wc = dz_cols < W ? dz_cols : W;
    const int tw = wgrad_tw(W), th = kWPix / tw;
    const int tilesY = (hc + th - 1) / th;
    WgCols c = wgrad_cols(W, wc, mt, Cin);
    if (tw != 64) {
        c = WgCols{};
        c.tilesX = (wc + tw - 1) / tw;
    }
    int s1, s2;
    wgrad_splits(N, tilesY, *nchunks, c, &s1, &s2);
    *S = s1 + s2;
}

int wgrad3x3_launch(const float* x, const float* dz, float* part, float* partb, int N, int Cin, int H,
                    int W, int Cout, int dz_rows, int dz_cols, hipStream_t s) {
    GW g{};
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout;
    g.Hc = dz_rows < H ? dz_rows : H;
    g.Wc = dz_cols < W ? dz_cols : W;
    const int tw = wgrad_tw(W), th = kWPix / tw;
    g.tilesX = (g.Wc + tw - 1) / tw;
    g.tilesY = (g.Hc + th - 1) / th;
    int ct, co_pad, ncol, s_total;
    wgrad3x3_geometry(N, Cin, H, W, Cout, dz_rows, dz_cols, &s_total, &g.nchunks, &ct, &co_pad, &ncol);
    WgCols c = wgrad_cols(W, g.Wc, co_pad / 32, Cin);
    if (tw != 64) {
        c = WgCols{};
        c.tilesX = g.tilesX;
    }
    int s1, s2;
    wgrad_splits(N, g.tilesY, g.nchunks, c, &s1, &s2);
    const long tiles = (long)N * g.tilesY * g.tilesX;
    if (tiles > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "wgrad3x3: too many tiles");
    afd::ScopedTiming timing(AFD_K_CONV_WGRAD, 2.0 * N * Cout * (double)g.Hc * g.Wc * Cin * 9, s);
    // GEMM M = padded Cout, N = the chunk's 9 ci columns (32-wide tiles), K = pixels in 64-pixel tiles
    timing.issued(2.0 * co_pad * (double)((ncol + 31) / 32 * 32) * g.nchunks * (double)tiles * kWPix);
    timing.bytes(4.0 * N * ((double)Cin * H * W + (double)Cout * g.Hc * g.Wc));
    const int mt = co_pad / 32;
    auto run_all = [&](const GW& gg) {
        if (mt == 1) return launchw<1, 32, 3>(gg, x, dz, part, partb, s);          //  9 pairs
        if (mt == 2) return launchw<2, 32, 5>(gg, x, dz, part, partb, s);          // 18 pairs
        if (mt == 3) return launchw<3, 32, 7>(gg, x, dz, part, partb, s);          // 27 pairs
        return launchw<4, 32, 9>(gg, x, dz, part, partb, s);                       // 36 pairs
    };
    if (!c.split) {
        g.S = s1; g.ncols = g.tilesX; g.col0 = 0; g.ncl = 0; g.split0 = 0;
        g.ntiles = (int)tiles;
        return run_all(g);
    }
    // interior columns: prefetching kernel; edge columns: the general kernel
    g.S = s1; g.ncols = c.ninner; g.col0 = 1; g.ncl = 0; g.split0 = 0;
    g.ntiles = N * g.tilesY * c.ninner;
    int rc;
    if (mt == 1) rc = launchw_p<1>(g, x, dz, part, partb, s);
    else if (mt == 2) rc = launchw_p<2>(g, x, dz, part, partb, s);
    else if (mt == 3) rc = (Cin % 64 == 0) ? launchw_p<3, 64>(g, x, dz, part, partb, s)
                                                                            : launchw_p<3>(g, x, dz, part, partb, s);
    else rc = launchw_p<4>(g, x, dz, part, partb, s);
    if (rc) return rc;
    g.S = s2; g.ncols = c.nborder; g.col0 = 0; g.ncl = c.nborder; g.split0 = s1;
    for (int i = 0; i < c.nborder; ++i) g.cl[i] = c.border[i];
    g.ntiles = N * g.tilesY * c.nborder;
    return run_all(g);
}

}  // namespace afd

namespace afd {

// (Cin, Cout, ...) of the convolution that is actually computed (for backward-data: swapped)
bool conv3x3_applicable(int Cin, int H, int W, int Cout, int K, int pad, int dil) {
    if (getenv("AFD_NO_CONV3X3")) return false;
    if (K != 3 || dil != 1 || pad != 1) return false;
    if (Cin % kCT != 0 || Cout > 128 || Cin < kCT) return false;
    if (W < 24 || H < 2) return false;  // 32-column tiles would be mostly padding
    return (size_t)kCT * H * W < 0x7fffffffULL;
}

size_t conv3x3_workspace_bytes(int Cin, int Cout) {
    const size_t co_pad = (size_t)(Cout + 31) / 32 * 32;
    const size_t direct = (size_t)(Cin / kCT) * kKsteps * 2 * co_pad * sizeof(float);
    size_t wino = wino_workspace_bytes(Cin, Cout);
    if (Cin % 8 == 0 && wino44_workspace_bytes(Cin, Cout) > wino) wino = wino44_workspace_bytes(Cin, Cout);
    return direct > wino ? direct : wino;
}

// dgrad = 0: y = conv(x, w) + bias, w [Cout][Cin][3][3].
// dgrad = 1: x is dy [N][Cin=forward Cout][H][W], y is dx [N][Cout=forward Cin][H][W],
//            w is the forward weight [Cin][Cout][3][3].
int conv3x3_run(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H,
                int W, int Cout, int dgrad, int out_rows, int out_cols, void* ws, size_t ws_bytes,
                hipStream_t s) {
    if (!ws || ws_bytes < conv3x3_workspace_bytes(Cin, Cout))
        return afd::fail(AFD_ERR_WORKSPACE, "conv3x3: workspace too small");
    // backward-data only: the forward layers with 64 output channels are pooled ones, whose one-launch form
    // (wino.hip's pooling epilogue) and two-launch form are kept bit-identical
    if ((dgrad || Cout == 128 || Cout == 32) && wino44_applicable(Cin, H, W, Cout))
        return wino44_run(x, w, bias, y, N, Cin, H, W, Cout, dgrad, out_rows, out_cols, ws, ws_bytes, s);
    if (wino_applicable(Cin, H, W, Cout))
        return wino_run(x, w, bias, y, N, Cin, H, W, Cout, dgrad, out_rows, out_cols, ws, ws_bytes, s);
    G3 g{};
    g.N = N; g.Cin = Cin; g.H = H; g.W = W; g.Cout = Cout; g.pad = 1;
    g.Hout = H; g.Wout = W;
    g.Hc = out_rows < H ? out_rows : H;
    g.Wc = out_cols < W ? out_cols : W;
    g.nchunks = Cin / kCT;
    float* wp = static_cast<float*>(ws);
    const int co_pad = (Cout + 31) / 32 * 32;
    const int total = g.nchunks * kKsteps * 2 * co_pad;
    hipLaunchKernelGGL(repack3_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, wp, Cin, Cout, co_pad,
                       g.nchunks, dgrad);
    int rc = afd::check_launch("repack3_kernel");
    if (rc) return rc;
    afd::ScopedTiming timing(AFD_K_CONV_IGEMM, 2.0 * N * Cout * (double)g.Hc * g.Wc * Cin * 9, s);
    timing.issued(2.0 * N * co_pad * (double)g.Hc * g.Wc * Cin * 9);  // lower bound: pixel-tile padding not counted
    timing.bytes(4.0 * N * ((double)Cin * H * W + (double)Cout * g.Hc * g.Wc));
    return run3(g, x, wp, bias, y, s);
}

}  // namespace afd
