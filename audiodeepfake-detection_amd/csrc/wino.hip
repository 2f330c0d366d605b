// 3x3 / pad 1 / stride 1 convolutions, forward and backward-data, as Winograd F(2x2, 3x3) on the
// f32 MFMA: 16 multiplies per 2x2 output tile and channel pair instead of 36 (2.25x fewer matrix
// instructions than the direct implicit GEMM of conv3x3.hip), arithmetic still fp32 end to end.
//
// Reference: the cuDNN launches behind nn.Conv2d(k=3, padding=1) of DCNN blocks 3-6
// (src/audiofakedetect/models.py:263-278) and their backward-data passes.
//
//   V = B^T d B   (4x4 input patch d of a tile, per input channel)           vector ALU
//   U = G g G^T   (3x3 filter g, per (cout, cin))                            wino_weights_kernel
//   M[p] = sum_cin U[p][cout][cin] V[p][cin][tile]   for the 16 positions p  16 GEMMs on the MFMA
//   Y = A^T M A   (2x2 outputs of the tile, per output channel)              vector ALU via LDS
//
// workgroup = 8 waves = one row of 32 tiles (2 output rows x 64 output columns) x all output
// channels.  Wave w owns positions 2w, 2w+1 for every 32-channel tile of Cout (accumulators:
// 2 x MT x 16 registers).  Per chunk of 8 input channels: thread (channel, tile) loads its 4x4
// patch straight from global memory (interior workgroups: four unaligned 16-byte loads), transforms
// it and writes its 8 or 16 values of V into a double-buffered LDS image [position][channel][tile]
// -- B fragments are rows of 32 consecutive floats; the U fragments come from a table in fragment
// order (one 256-byte load each, L2 resident) issued before the transform.  One barrier per chunk.
// Epilogue: per 32-channel tile the 16 positions meet in LDS, thread (cout, tile) applies A^T . A,
// adds the bias and stores two 8-byte pairs.
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

constexpr int kCh = 8;        // input channels per chunk (4 k-steps)
constexpr int kTiles = 32;    // tiles per workgroup (one MFMA column tile)
constexpr int kThreads = 512;

struct GW {
    int N, Cin, Cout, H, W;
    int rows, cols;      // output rows / columns that are wanted (crop for a following 2x2 pool)
    int tilesX, tilesY;  // 2x2 tiles covering rows x cols
    int wgX;             // workgroups per tile row
    int wxCount;         // workgroup columns covered by this launch
    // pooled epilogue (afd_conv3x3_prelu_pool_forward): PReLU + MaxPool2d(2, 2) of the tile, which IS
    // the pooling window; u / idx [N][Cout][H/2][W/2] are written instead of y
    const float* slope;
    float* u;
    unsigned char* idx;
    int nchunks;
    // backward-data launches whose result is the gradient of a BatchNorm output (afd_conv3x3_backward_data_bnstats):
    // the epilogue also sums, per channel, g and g * xhat over its outputs, xhat the BatchNorm output (the forward
    // convolution's input) at the same position -- one partial row [sum g | sum g xhat] per workgroup, rows of
    // this launch from part_row0
    const float* bn_in;
    float* stat_part;
    int part_row0;
    const float* x_end;  // one past the input tensor (border patches: see load_patch)
};

// U table: [chunk][position][mt][kstep][lane] = U_p[32 mt + (lane & 31)][8 chunk + 2 kstep + (lane >> 5)]
// dgrad: the GEMM's output channels are the forward's input channels, taps flipped
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout,
                                    int MT, int nchunks, int dgrad) {
    const int total = nchunks * 16 * MT * 4 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63;
        const int ks = (i >> 6) & 3;
        int r = i >> 8;
        const int mt = r % MT;
        r /= MT;
        const int p = r & 15;
        const int chunk = r >> 4;
        const int co = 32 * mt + (lane & 31);
        const int ci = kCh * chunk + 2 * ks + (lane >> 5);
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            float g[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    g[ky][kx] = dgrad ? w[((size_t)ci * Cout + co) * 9 + (8 - (ky * 3 + kx))]
                                      : w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
            // row xi of G g, then column nu of (G g) G^T;  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
            const int xi = p >> 2, nu = p & 3;
            float t[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float a = g[0][kx], b = g[1][kx], c = g[2][kx];
                t[kx] = xi == 0 ? a : xi == 1 ? 0.5f * (a + b + c) : xi == 2 ? 0.5f * (a - b + c) : c;
            }
            v = nu == 0 ? t[0] : nu == 1 ? 0.5f * (t[0] + t[1] + t[2]) : nu == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2];
        }
        U[i] = v;
    }
}

// MT = 32-channel tiles of Cout, NT = 32-tile column groups per workgroup (1: 32 tiles, two threads
// share a patch transform; 2: 64 tiles, one thread per (channel, tile) patch -- every U fragment
// then feeds two column tiles, halving the fragment traffic per MFMA)
// BORDER: the launch covers the first and last workgroup column of every tile row (patches that
// reach outside the image: element loads with clamped addresses); the other launch covers the
// columns in between with one unaligned 16-byte load per patch row.  No divergent paths inside.
template <int MT, int NT, bool BORDER, bool POOL, bool BST = false>
__global__ void __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2)))
wino_conv_kernel(const GW g, const float* __restrict__ x, const float* __restrict__ U,
                 const float* __restrict__ bias, float* __restrict__ y) {
    constexpr int TILES = kTiles * NT;
    constexpr int VBUF = 16 * kCh * TILES;  // floats per V buffer
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* V = lds;  // [2][16][kCh][TILES]; the epilogue reuses the allocation as E[16][32][kTiles]
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int id = blockIdx.x;
    const int wi = id % g.wxCount;
    id /= g.wxCount;
    const int wx = BORDER ? (wi == 0 ? 0 : g.wgX - 1) : wi + 1;
    const int ty = id % g.tilesY;
    const int n = id / g.tilesY;
    const int tx0 = wx * TILES;

    // transform role: (channel ch of the chunk, tile tl); NT == 1: part 0 writes rows xi = 0, 1 and
    // part 1 rows 2, 3 of the pair's V; NT == 2: the thread writes all four rows
    const int tl = tid & (TILES - 1);
    const int ch = (tid / TILES) & 7;
    const int part = NT == 1 ? (tid >> 8) : 0;
    const int tx = tx0 + tl;
    const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
    const bool tile_ok = tx < g.tilesX;
    const size_t plane = (size_t)g.H * g.W;
    const float* xn = x + (size_t)n * g.Cin * plane;

    // branch-free: elements outside the image are read from a clamped address and zeroed by a select
    // when the patch is CONSUMED (store_v) -- a branch per row, or a select right after the load,
    // makes the wave wait for the loads where they are issued instead of a chunk later
    unsigned okmask = 0;  // bit 4 r + j: patch element (r, j) lies inside the image
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = iy0 + r, ix = ix0 + j;
            const bool ok = iy >= 0 && iy < g.H && (!BORDER || (tile_ok && ix >= 0 && ix < g.W));
            okmask |= ok ? (1u << (4 * r + j)) : 0u;
        }
    float d[4][4];
    auto load_patch = [&](int c) {
        const float* xc = xn + (size_t)(c * kCh + ch) * plane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = iy0 + r;
            const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
            const float* row = xc + (size_t)iyc * g.W;
            if (!BORDER) {
                const f4u v = *reinterpret_cast<const f4u*>(row + ix0);
                d[r][0] = v.x; d[r][1] = v.y; d[r][2] = v.z; d[r][3] = v.w;
            } else {
                // a patch that sticks out of its row reads the neighbouring row's elements (inside the tensor, and
                // zeroed by `okmask` when the patch is consumed): the same unaligned 16-byte load as in the interior.
                // Only where that load would leave the tensor itself (first row of the first plane, last row of the
                // last one) the elements are fetched one by one from clamped addresses -- a branch two waves of the
                // launch take.  (Element loads for every border patch cost the level-8 / STFT images, where every
                // workgroup is a border workgroup, four times the load instructions.)
                const float* p4 = row + ix0;
                if (p4 >= x && p4 + 4 <= g.x_end) {
                    const f4u v = *reinterpret_cast<const f4u*>(p4);
                    d[r][0] = v.x; d[r][1] = v.y; d[r][2] = v.z; d[r][3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ix = ix0 + j;
                        d[r][j] = row[ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix)];
                    }
                }
            }
        }
    };
    auto dv = [&](int r, int j) { return (okmask >> (4 * r + j)) & 1u ? d[r][j] : 0.f; };
    // B^T d B: rows t = B^T d (t0 = r0 - r2, t1 = r1 + r2, t2 = r2 - r1, t3 = r1 - r3), then
    // (a, b, c, e) -> (a - c, b + c, c - b, b - e) along the row
    auto store_v = [&](int buf) {
        float* vb = V + buf * VBUF + ch * TILES + tl + (2 * part * 4) * (kCh * TILES);
#pragma unroll
        for (int q = 0; q < (NT == 1 ? 2 : 4); ++q) {
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (NT == 1) {
                    // both parts' rows are formed and one is picked: a select between d[][] elements
                    // themselves would be compiled into a dynamically indexed (scratch) array
                    const float lo = q == 0 ? dv(0, j) - dv(2, j) : dv(1, j) + dv(2, j);
                    const float hi = q == 0 ? dv(2, j) - dv(1, j) : dv(1, j) - dv(3, j);
                    t[j] = part ? hi : lo;
                } else {
                    t[j] = q == 0 ? dv(0, j) - dv(2, j) : q == 1 ? dv(1, j) + dv(2, j) : q == 2 ? dv(2, j) - dv(1, j) : dv(1, j) - dv(3, j);
                }
            }
            float* o = vb + (q * 4) * (kCh * TILES);
            o[0 * kCh * TILES] = t[0] - t[2];
            o[1 * kCh * TILES] = t[1] + t[2];
            o[2 * kCh * TILES] = t[2] - t[1];
            o[3 * kCh * TILES] = t[1] - t[3];
        }
    };

    f32x16 acc[2][MT][NT];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pi][m][nt][r] = 0.f;

    const float* Uw = U + (size_t)(2 * wave) * MT * 256 + lane;  // + chunk * 16 * MT * 256
    float uf[2][MT][4];
    auto load_u = [&](int c) {
        const float* uc = Uw + (size_t)c * 16 * MT * 256;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int s = 0; s < 4; ++s) uf[pi][m][s] = uc[((pi * MT + m) * 4 + s) * 64];
    };
    auto mfma_phase = [&](int buf) {
        const float* vb = V + buf * VBUF + half * TILES + l31;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const float* vp = vb + (2 * wave + pi) * (kCh * TILES);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float b[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b[nt] = vp[2 * s * TILES + nt * 32];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[pi][m][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(uf[pi][m][s], b[nt], acc[pi][m][nt], 0, 0, 0);
            }
        }
    };
    // The two waves of a SIMD run the phases of an iteration in opposite order, so one is in its
    // matrix phase while the other transforms the next chunk (vector ALU + LDS writes):
    //   waves 0-3: U(c) loads, transform(c+1), patch loads (c+2), MFMAs(c)
    //   waves 4-7: MFMAs(c) on fragments loaded an iteration ago, U(c+1) loads, transform(c+1), ...
    // Iteration c reads V[c & 1] (written before the previous barrier) and writes V[(c+1) & 1]
    // (last read before the previous barrier): one barrier per chunk.
    const bool early = wave < 4;
    load_patch(0);
    store_v(0);
    if (g.nchunks > 1) load_patch(1);
    if (!early) load_u(0);
    __syncthreads();
    for (int c = 0; c < g.nchunks; ++c) {
        const bool more = c + 1 < g.nchunks;
        if (early) {
            load_u(c);
            if (more) store_v((c + 1) & 1);
            if (c + 2 < g.nchunks) load_patch(c + 2);
        }
        // the late wave's MFMAs go first: it still has its transform to do before the barrier,
        // which then runs under the early wave's remaining MFMAs
        if (!early) __builtin_amdgcn_s_setprio(2);
        mfma_phase(c & 1);  // one copy of the matrix phase: the accumulators stay in place
        if (!early) {
            __builtin_amdgcn_s_setprio(0);
            if (more) {
                load_u(c + 1);
                store_v((c + 1) & 1);
            }
            if (c + 2 < g.nchunks) load_patch(c + 2);
        }
        __syncthreads();
    }

    // output transform: Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1]; one (channel tile, column tile) per round
    // two images, used alternately: a round's writes need not wait for the previous round's reads
    // (the main loop's last barrier covers the first round: every fragment read of V is done)
    const int tl_e = tid & 31;
    const int co_l = tid >> 5;  // 0..15 (+16 for the second pair)
    const int oy = 2 * ty;
    float sg[BST ? MT : 1][2], sgv[BST ? MT : 1][2];  // BST: sums over this thread's outputs of channel 32 m + co_l + 16 j
    f2u zn[BST ? 2 : 1][2];
    // BST: the BatchNorm outputs at a round's output positions: two 8-byte loads per channel from clamped (always
    // valid) addresses; the products are masked where they are used
    auto load_xhat = [&](int m, int nt) {
        const int txq = tx0 + nt * 32 + tl_e;
        const int oxq = min(2 * txq, g.W - 2);
        const int r1 = oy + 1 < g.H ? g.W : 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int co = min(32 * m + co_l + 16 * j, g.Cout - 1);
            const float* zi = g.bn_in + (((size_t)n * g.Cout + co) * g.H + oy) * g.W + oxq;
            zn[j][0] = *reinterpret_cast<const f2u*>(zi);
            zn[j][1] = *reinterpret_cast<const f2u*>(zi + r1);
        }
    };
    if constexpr (BST) {
#pragma unroll
        for (int m = 0; m < MT; ++m) sg[m][0] = sg[m][1] = sgv[m][0] = sgv[m][1] = 0.f;
        load_xhat(0, 0);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float* E = lds + ((m * NT + nt) & 1) * (16 * 32 * kTiles);  // [16][32][kTiles]
            // BST: the BatchNorm inputs at this round's output positions are requested before the LDS round trip
            f2u zq[BST ? 2 : 1][2];
            if constexpr (BST) {
                // this round's values were requested a round ago; the next round's are requested now
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    zq[j][0] = zn[j][0];
                    zq[j][1] = zn[j][1];
                }
                if (m * NT + nt + 1 < MT * NT) load_xhat((m * NT + nt + 1) / NT, (m * NT + nt + 1) % NT);
            }
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                float* e = E + (size_t)(2 * wave + pi) * (32 * kTiles) + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    e[row * kTiles] = acc[pi][m][nt][r];
                }
            }
            __syncthreads();
            const int txe = tx0 + nt * 32 + tl_e;
            const int ox = 2 * txe;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cl = co_l + 16 * j;
                const int co = 32 * m + cl;
                const float* e = E + cl * kTiles + tl_e;
                float mm[4][4];
#pragma unroll
                for (int p = 0; p < 16; ++p) mm[p >> 2][p & 3] = e[(size_t)p * (32 * kTiles)];
                float s0[4], s1[4];  // rows of A^T M
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s0[q] = mm[0][q] + mm[1][q] + mm[2][q];
                    s1[q] = mm[1][q] - mm[2][q] - mm[3][q];
                }
                const float bv = (bias && co < g.Cout) ? bias[co] : 0.f;
                const float y00 = s0[0] + s0[1] + s0[2] + bv, y01 = s0[1] - s0[2] - s0[3] + bv;
                const float y10 = s1[0] + s1[1] + s1[2] + bv, y11 = s1[1] - s1[2] - s1[3] + bv;
                if (POOL) {
                    // same order and tie rule as prelu_pool_fwd_kernel (nn.hip): first maximum wins
                    if (co < g.Cout && txe < g.tilesX) {
                        const float a = g.slope[0];
                        auto act = [a](float z) { return z > 0.f ? z : a * z; };
                        float best = act(y00), zb = y00;
                        int bi = 0;
                        float v = act(y01);
                        if (v > best) { best = v; bi = 1; zb = y01; }
                        v = act(y10);
                        if (v > best) { best = v; bi = 2; zb = y10; }
                        v = act(y11);
                        if (v > best) { best = v; bi = 3; zb = y11; }
                        const size_t o = (((size_t)n * g.Cout + co) * g.tilesY + ty) * g.tilesX + txe;
                        g.u[o] = best;
                        g.idx[o] = (unsigned char)(bi | (zb <= 0.f ? 4 : 0));
                    }
                } else if (co < g.Cout && txe < g.tilesX) {
                    float* yo = y + (((size_t)n * g.Cout + co) * g.H + oy) * g.W + ox;
                    const bool two = ox + 1 < g.cols;
                    if constexpr (BST) {
                        // (the clamped column only differs from ox where `two` is false: W odd, last column, whose
                        // value then sits in the pair's second element)
                        const bool row2 = oy + 1 < g.rows;
                        const float z00 = two ? zq[j][0].x : zq[j][0].y, z10 = two ? zq[j][1].x : zq[j][1].y;
                        const float g01 = two ? y01 : 0.f, g10 = row2 ? y10 : 0.f, g11 = (two && row2) ? y11 : 0.f;
                        sg[m][j] += (y00 + g01) + (g10 + g11);
                        sgv[m][j] += fmaf(y00, z00, g01 * zq[j][0].y) + fmaf(g10, z10, g11 * zq[j][1].y);
                    }
                    if (two) {
                        f2u v0 = {y00, y01};
                        *reinterpret_cast<f2u*>(yo) = v0;
                        if (oy + 1 < g.rows) {
                            f2u v1 = {y10, y11};
                            *reinterpret_cast<f2u*>(yo + g.W) = v1;
                        }
                    } else {
                        yo[0] = y00;
                        if (oy + 1 < g.rows) yo[g.W] = y10;
                    }
                }
            }
        }
    }
    if constexpr (BST) {
        // a channel's outputs of this workgroup sit in the 32 lanes of one wave half
        constexpr int CO_PAD = MT * 32;
        float* row = g.stat_part + ((size_t)g.part_row0 + blockIdx.x) * (2 * CO_PAD);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float a1 = sg[m][j], a2 = sgv[m][j];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) {
                    a1 += __shfl_xor(a1, off, 64);
                    a2 += __shfl_xor(a2, off, 64);
                }
                if (tl_e == 0) {
                    row[32 * m + co_l + 16 * j] = a1;
                    row[CO_PAD + 32 * m + co_l + 16 * j] = a2;
                }
            }
    }
}

// partial rows [rows][2 co_pad] -> part2 [gridDim.x][2 co_pad] doubles (thread = slot, rows strided over the blocks;
// four independent chains per thread)
__global__ void __launch_bounds__(256)
wino_bnstats_reduce1_kernel(const float* __restrict__ part, int rows, int slots, double* __restrict__ part2) {
    const int e = threadIdx.x;
    if (e >= slots) return;
    const int G = gridDim.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int r = blockIdx.x;
    for (; r + 3 * G < rows; r += 4 * G) {
        s0 += (double)part[(size_t)r * slots + e];
        s1 += (double)part[(size_t)(r + G) * slots + e];
        s2 += (double)part[(size_t)(r + 2 * G) * slots + e];
        s3 += (double)part[(size_t)(r + 3 * G) * slots + e];
    }
    for (; r < rows; r += G) s0 += (double)part[(size_t)r * slots + e];
    part2[(size_t)blockIdx.x * slots + e] = (s0 + s1) + (s2 + s3);
}

// sums[c] = sum g, sums[C + c] = sum g xhat: what afd_bn_backward_means takes.  Block = 8 slots x 32 row lanes
// (a single block walking the rows2 rows one after the other took 68 us of load latency)
__global__ void __launch_bounds__(256)
wino_bnstats_reduce2_kernel(const double* __restrict__ part2, int rows2, int co_pad, int C,
                            double* __restrict__ sums) {
    const int lane = threadIdx.x & 31;
    const int slot = blockIdx.x * 8 + (threadIdx.x >> 5);  // 0 .. 2 co_pad - 1
    double s = 0.0;
    if (slot < 2 * co_pad)
        for (int r = lane; r < rows2; r += 32) s += part2[(size_t)r * 2 * co_pad + slot];
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0 && slot < 2 * co_pad) {
        const int c = slot < co_pad ? slot : slot - co_pad;
        if (c < C) sums[(slot < co_pad ? 0 : C) + c] = s;
    }
}

constexpr int kStatBlocks = 256;

template <int MT, int NT, bool POOL, bool BST = false>
int launch_wino_t(GW g, const float* x, const float* U, const float* bias, float* y, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 16 * 32 * kTiles * sizeof(float);  // 128 KB: two epilogue images (V double buffer inside)
    static_assert(2 * 16 * kCh * kTiles * NT * sizeof(float) <= lds, "V double buffer fits the epilogue image");
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_kernel<MT, NT, false, POOL, BST>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_conv_kernel<MT, NT, true, POOL, BST>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "winograd conv: %s", hipGetErrorString(e));
        attr.mark();
    }
    g.wgX = (g.tilesX + kTiles * NT - 1) / (kTiles * NT);
    const long rows = (long)g.N * g.tilesY;
    // 16 GEMMs [32 MT x Cin] x [Cin x 32 NT tiles] per workgroup, every tile computed in full
    afd::timing_annotate(2.0 * 16 * (32.0 * MT) * ((double)kTiles * NT * g.wgX) * (double)rows * g.Cin, -1.0);
    const int inner = g.wgX > 2 ? g.wgX - 2 : 0;
    const int edge = g.wgX >= 2 ? 2 : 1;
    if (rows * (inner > edge ? inner : edge) > 0x7fffffffL)
        return afd::fail(AFD_ERR_UNSUPPORTED, "winograd conv: grid too large");
    g.part_row0 = 0;
    if (inner > 0) {
        g.wxCount = inner;
        hipLaunchKernelGGL((wino_conv_kernel<MT, NT, false, POOL, BST>), dim3((unsigned)(rows * inner)), dim3(kThreads), lds,
                           s, g, x, U, bias, y);
        g.part_row0 = (int)(rows * inner);
    }
    g.wxCount = edge;
    hipLaunchKernelGGL((wino_conv_kernel<MT, NT, true, POOL, BST>), dim3((unsigned)(rows * edge)), dim3(kThreads), lds, s,
                       g, x, U, bias, y);
    return afd::check_launch("wino_conv_kernel");
}

template <int MT, int NT>
int launch_wino(const GW& g, const float* x, const float* U, const float* bias, float* y, hipStream_t s) {
    if constexpr (NT == 2 && (MT == 2 || MT == 3)) {
        if (g.stat_part) return launch_wino_t<MT, NT, false, true>(g, x, U, bias, y, s);
    }
    if (g.stat_part) return afd::fail(AFD_ERR_UNSUPPORTED, "winograd conv: statistics epilogue not built for this shape");
    return g.u ? launch_wino_t<MT, NT, true>(g, x, U, bias, y, s) : launch_wino_t<MT, NT, false>(g, x, U, bias, y, s);
}

// workgroups (= partial rows) of a launch pair over N images of H x W outputs
long wino_stat_rows(int N, int H, int W) {
    const int tilesX = (W + 1) / 2, tilesY = (H + 1) / 2;
    const int wgX = (tilesX + kTiles * 2 - 1) / (kTiles * 2);
    return (long)N * tilesY * wgX;
}

}  // namespace

namespace afd {

bool wino_applicable(int Cin, int H, int W, int Cout) {
    if (getenv("AFD_NO_WINOGRAD")) return false;
    // one 32-channel tile of Cout leaves 8 MFMAs per wave between barriers: the direct kernel wins
    if (Cin % kCh != 0 || Cin < kCh || Cout > 128 || Cout <= 32) return false;
    if (W < 64 || H < 2) return false;
    return (size_t)H * W < 0x7fffffffULL;
}

size_t wino_workspace_bytes(int Cin, int Cout) {
    const size_t mt = (size_t)(Cout + 31) / 32;
    return (size_t)(Cin / kCh) * 16 * mt * 4 * 64 * sizeof(float);
}

// same contract as conv3x3_run (conv3x3.hip)
int wino_run(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
             int Cout, int dgrad, int out_rows, int out_cols, void* ws, size_t ws_bytes, hipStream_t s,
             const float* slope, float* u, unsigned char* idx, const float* bn_in, float* stat_part) {
    if (!ws || ws_bytes < wino_workspace_bytes(Cin, Cout))
        return afd::fail(AFD_ERR_WORKSPACE, "winograd conv: workspace too small");
    GW g{};
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.H = H; g.W = W;
    g.rows = out_rows < H ? out_rows : H;
    g.cols = out_cols < W ? out_cols : W;
    g.tilesX = (g.cols + 1) / 2;
    g.tilesY = (g.rows + 1) / 2;
    g.wgX = (g.tilesX + kTiles - 1) / kTiles;
    g.nchunks = Cin / kCh;
    g.x_end = x + (size_t)N * Cin * H * W;
    g.slope = slope; g.u = u; g.idx = idx;
    g.bn_in = bn_in; g.stat_part = stat_part;
    if (u && (g.rows != 2 * (H / 2) || g.cols != 2 * (W / 2) || !slope || !idx))
        return afd::fail(AFD_ERR_ARG, "winograd conv + pool: bad arguments");
    const int MT = (Cout + 31) / 32;
    float* U = static_cast<float*>(ws);
    const int total = g.nchunks * 16 * MT * 4 * 64;
    afd::ScopedTiming timing(AFD_K_CONV_WINOGRAD, 2.0 * N * Cout * (double)g.rows * g.cols * Cin * 9, s);
    hipLaunchKernelGGL(wino_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, U, Cin, Cout, MT,
                       g.nchunks, dgrad);
    int rc = afd::check_launch("wino_weights_kernel");
    if (rc) return rc;
    timing.bytes(4.0 * N * ((double)Cin * H * W + (double)Cout * g.rows * g.cols / (u ? 4.0 : 1.0)));
    switch (MT) {
        case 1: return launch_wino<1, 2>(g, x, U, bias, y, s);
        case 2: return launch_wino<2, 2>(g, x, U, bias, y, s);
        case 3: return launch_wino<3, 2>(g, x, U, bias, y, s);
        case 4: return launch_wino<4, 1>(g, x, U, bias, y, s);
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "winograd conv: Cout %d > 128", Cout);
}

}  // namespace afd

// ---- backward-data of a 3x3 convolution whose input was a BatchNorm output, with that BatchNorm's backward sums ----
extern "C" int afd_conv3x3_backward_data_bnstats_applicable(int Cin, int H, int W, int Cout) {
    // the backward-data GEMM has the forward's Cout as its input channels; only the wide 32x32x2 Winograd kernel
    // with two or three channel tiles carries the statistics epilogue
    if (getenv("AFD_NO_BWD_BNSTATS")) return 0;
    if (afd::wino44_applicable(Cout, H, W, Cin)) return 1;
    if (!afd::wino_applicable(Cout, H, W, Cin)) return 0;
    const int mt = (Cin + 31) / 32;
    return mt == 2 || mt == 3;
}

// 0: the launch can leave sum(g * xhat) to the caller (xhat = nullptr; sums[Cin .. 2 Cin) come back as zeros and the
// caller adds afd_conv_weight_dot of the layer's weights and weight gradient) -- the F(4x4) kernel
extern "C" int afd_conv3x3_backward_data_bnstats_needs_input(int Cin, int H, int W, int Cout) {
    if (getenv("AFD_BNSTATS_FROM_INPUT")) return 1;
    return afd::wino44_applicable(Cout, H, W, Cin) ? 0 : 1;
}

// out[ci] = sum over co and the KK taps of w[co][ci][k] * dw[co][ci][k], in double precision: for a convolution
// y = w * x (any padding), sum_px dx[ci][px] x[ci][px] = sum_{co,k} w[co][ci][k] dw[co][ci][k] -- the second backward sum
// of a BatchNorm whose output x feeds the convolution, without a pass over the activations
__global__ void __launch_bounds__(256)
conv_weight_dot_kernel(const float* __restrict__ w, const float* __restrict__ dw, int Cout, int Cin, int KK,
                       double* __restrict__ out) {
    __shared__ double red[256];
    const int ci = blockIdx.x;
    double s = 0.0;
    for (int e = threadIdx.x; e < Cout * KK; e += 256) {
        const int co = e / KK, k = e - co * KK;
        const size_t o = ((size_t)co * Cin + ci) * KK + k;
        s += (double)w[o] * (double)dw[o];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[ci] = red[0];
}

extern "C" int afd_conv_weight_dot(const float* w, const float* dw, int Cout, int Cin, int KK, double* out,
                                   afd_stream_t stream) {
    if (!w || !dw || !out || Cout < 1 || Cin < 1 || KK < 1) return afd::fail(AFD_ERR_ARG, "conv weight dot: bad argument");
    afd::ScopedBytes timing(AFD_K_BATCHNORM, 8.0 * Cout * Cin * KK, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(conv_weight_dot_kernel, dim3(Cin), dim3(256), 0, static_cast<hipStream_t>(stream), w, dw, Cout, Cin,
                       KK, out);
    return afd::check_launch("conv_weight_dot_kernel");
}

extern "C" size_t afd_conv3x3_backward_data_bnstats_workspace_bytes(int N, int Cin, int H, int W) {
    const size_t slots = 2 * ((size_t)(Cin + 31) / 32 * 32);
    long rows = wino_stat_rows(N, H, W);
    if (afd::wino44_stat_rows(N, H, W) > rows) rows = afd::wino44_stat_rows(N, H, W);
    return (size_t)rows * slots * sizeof(float) + (size_t)kStatBlocks * slots * sizeof(double) + 64;
}

extern "C" int afd_conv3x3_backward_data_bnstats(const float* dy, const float* w, float* dx, const float* xhat,
                                                 double* sums, int N, int Cin, int H, int W, int Cout, void* ws,
                                                 size_t ws_bytes, void* stat_ws, size_t stat_ws_bytes,
                                                 afd_stream_t stream) {
    if (!dy || !w || !dx || !sums || !ws || !stat_ws)
        return afd::fail(AFD_ERR_ARG, "conv3x3 dgrad + bn sums: null pointer");
    if (!xhat && afd_conv3x3_backward_data_bnstats_needs_input(Cin, H, W, Cout))
        return afd::fail(AFD_ERR_ARG, "conv3x3 dgrad + bn sums: this shape's kernel needs the BatchNorm output");
    if (N < 1 || W < 2 || !afd_conv3x3_backward_data_bnstats_applicable(Cin, H, W, Cout))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3 dgrad + bn sums: shape not on the Winograd kernel");
    if (stat_ws_bytes < afd_conv3x3_backward_data_bnstats_workspace_bytes(N, Cin, H, W))
        return afd::fail(AFD_ERR_WORKSPACE, "conv3x3 dgrad + bn sums: statistics workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int co_pad = (Cin + 31) / 32 * 32;
    const int slots = 2 * co_pad;
    const bool f44 = afd::wino44_applicable(Cout, H, W, Cin);
    const long rows = f44 ? afd::wino44_stat_rows(N, H, W) : wino_stat_rows(N, H, W);
    float* part = static_cast<float*>(stat_ws);
    double* part2 = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(part + (size_t)rows * slots) + 63) & ~(uintptr_t)63);
    int rc = f44 ? afd::wino44_run(dy, w, nullptr, dx, N, Cout, H, W, Cin, 1, H, W, ws, ws_bytes, s, xhat, part)
                 : afd::wino_run(dy, w, nullptr, dx, N, Cout, H, W, Cin, 1, H, W, ws, ws_bytes, s, nullptr, nullptr,
                                 nullptr, xhat, part);
    if (rc) return rc;
    const int blocks = rows < kStatBlocks ? (int)rows : kStatBlocks;
    afd::ScopedBytes red_timing(AFD_K_BATCHNORM, 4.0 * (double)rows * slots, s);
    hipLaunchKernelGGL(wino_bnstats_reduce1_kernel, dim3(blocks), dim3(256), 0, s, part, (int)rows, slots, part2);
    hipLaunchKernelGGL(wino_bnstats_reduce2_kernel, dim3((2 * co_pad + 7) / 8), dim3(256), 0, s, part2, blocks, co_pad, Cin,
                       sums);
    return afd::check_launch("wino_bnstats_reduce kernels");
}

// ---- the same launch pair for a convolution followed by PReLU + MaxPool2d(2, 2), from the POOLED gradient ----
// gg [N][Cout][H/2][W/2] (afd_prelu_pool_backward_compact) and the pool's codes stand for the dense gradient dz
// [N][Cout][H][W] (gg at position code & 3 of each 2x2 window, zero elsewhere, zero in an odd last row / column): the
// backward-data and backward-weight kernels build their patches / tiles from them while loading, dz is never stored.
extern "C" int afd_conv3x3_pooled_backward_applicable(int Cin, int H, int W, int Cout) {
    if (getenv("AFD_NO_POOLED_BWD")) return 0;
    if (Cin != 64 && Cin != 32) return 0;  // result channels of the backward-data launch built with pooled input
    if (H < 2 || W < 2) return 0;
    if (!afd::wino44_applicable(Cout, H, W, Cin)) return 0;
    if (!afd::wino44_wgrad_applicable(Cin, H, W, Cout, 3, 1, 1)) return 0;
    return afd::wino44_wgrad_crop_ok(H, W, 2 * (H / 2), 2 * (W / 2)) ? 1 : 0;
}

extern "C" int afd_conv3x3_backward_data_bnstats_pooled(const float* gg, const uint8_t* codes, const float* w, float* dx,
                                                        double* sums, int N, int Cin, int H, int W, int Cout, void* ws,
                                                        size_t ws_bytes, void* stat_ws, size_t stat_ws_bytes,
                                                        afd_stream_t stream) {
    if (!gg || !codes || !w || !dx || !sums || !ws || !stat_ws)
        return afd::fail(AFD_ERR_ARG, "conv3x3 dgrad from the pooled gradient: null pointer");
    if (N < 1 || !afd_conv3x3_pooled_backward_applicable(Cin, H, W, Cout))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3 dgrad from the pooled gradient: shape not built");
    if (stat_ws_bytes < afd_conv3x3_backward_data_bnstats_workspace_bytes(N, Cin, H, W))
        return afd::fail(AFD_ERR_WORKSPACE, "conv3x3 dgrad from the pooled gradient: statistics workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int co_pad = (Cin + 31) / 32 * 32;
    const int slots = 2 * co_pad;
    const long rows = afd::wino44_stat_rows(N, H, W);
    float* part = static_cast<float*>(stat_ws);
    double* part2 = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(part + (size_t)rows * slots) + 63) & ~(uintptr_t)63);
    int rc = afd::wino44_run(gg, w, nullptr, dx, N, Cout, H, W, Cin, 1, H, W, ws, ws_bytes, s, nullptr, part, nullptr, nullptr,
                             nullptr, 0, codes);
    if (rc) return rc;
    const int blocks = rows < kStatBlocks ? (int)rows : kStatBlocks;
    afd::ScopedBytes red_timing(AFD_K_BATCHNORM, 4.0 * (double)rows * slots, s);
    hipLaunchKernelGGL(wino_bnstats_reduce1_kernel, dim3(blocks), dim3(256), 0, s, part, (int)rows, slots, part2);
    hipLaunchKernelGGL(wino_bnstats_reduce2_kernel, dim3((2 * co_pad + 7) / 8), dim3(256), 0, s, part2, blocks, co_pad, Cin,
                       sums);
    return afd::check_launch("wino_bnstats_reduce kernels");
}

extern "C" int afd_conv3x3_backward_weight_pooled(const float* x, const float* gg, const uint8_t* codes, float* dw,
                                                  float* dbias, int N, int Cin, int H, int W, int Cout, void* ws,
                                                  size_t ws_bytes, afd_stream_t stream) {
    if (!x || !gg || !codes || !dw || !ws) return afd::fail(AFD_ERR_ARG, "conv3x3 wgrad from the pooled gradient: null pointer");
    if (N < 1 || !afd_conv3x3_pooled_backward_applicable(Cin, H, W, Cout))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3 wgrad from the pooled gradient: shape not built");
    return afd::wino44_wgrad_run(x, gg, dw, dbias, N, Cin, H, W, Cout, H, W, ws, ws_bytes, static_cast<hipStream_t>(stream), codes);
}

// ---- forward 3x3 convolution whose result feeds a training-mode BatchNorm, with that BatchNorm's batch sums ----
// (the F(4x4) kernel's epilogue, wino44.hip: block 3's pooled forward and block 4's forward at level 14)
extern "C" int afd_conv3x3_forward_stats_applicable(int Cin, int H, int W, int Cout, int pooled) {
    if (getenv("AFD_NO_FWD_STATS")) return 0;
    if (pooled) return Cout == 96 && H >= 2 && W >= 2 && afd::wino44_pool_applicable(Cin, H, W, Cout) ? 1 : 0;
    return (Cout == 128 || Cout == 32) && afd::wino44_applicable(Cin, H, W, Cout) ? 1 : 0;
}

// (the F(4x4) launch pair's workgroups over the rows x cols outputs it computes: wino44.hip knows its own forms)
static long fwd_stat_rows(int N, int rows, int cols) { return afd::wino44_stat_rows(N, rows, cols); }

extern "C" size_t afd_conv3x3_forward_stats_workspace_bytes(int N, int H, int W, int Cout) {
    const size_t slots = 2 * ((size_t)(Cout + 31) / 32 * 32);
    return (size_t)fwd_stat_rows(N, H, W) * slots * sizeof(float) + (size_t)kStatBlocks * slots * sizeof(double) + 64;
}

static int forward_stats_impl(const float* x, const float* w, const float* bias, const float* slope,
                              float* y, float* u, uint8_t* idx, double* sums, int N, int Cin, int H,
                              int W, int Cout, void* ws, size_t ws_bytes, void* stat_ws,
                              size_t stat_ws_bytes, afd_stream_t stream, const float* in_aff, const float* in_slope) {
    if (!x || !w || !sums || !ws || !stat_ws || (!y && !u) || (u && (!idx || !slope)))
        return afd::fail(AFD_ERR_ARG, "conv3x3 + bn sums: null pointer");
    const int pooled = u != nullptr;
    if (N < 1 || !afd_conv3x3_forward_stats_applicable(Cin, H, W, Cout, pooled))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3 + bn sums: layer not on the F(4x4) Winograd kernel");
    if (stat_ws_bytes < afd_conv3x3_forward_stats_workspace_bytes(N, H, W, Cout))
        return afd::fail(AFD_ERR_WORKSPACE, "conv3x3 + bn sums: statistics workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rows_out = pooled ? 2 * (H / 2) : H, cols_out = pooled ? 2 * (W / 2) : W;
    const int co_pad = (Cout + 31) / 32 * 32;
    const int slots = 2 * co_pad;
    const long rows = fwd_stat_rows(N, rows_out, cols_out);
    float* part = static_cast<float*>(stat_ws);
    double* part2 = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(part + (size_t)rows * slots) + 63) & ~(uintptr_t)63);
    int rc = afd::wino44_run(x, w, bias, y, N, Cin, H, W, Cout, 0, rows_out, cols_out, ws, ws_bytes, s, nullptr, part, slope,
                             u, idx, 1, nullptr, in_aff, in_slope);
    if (rc) return rc;
    const int blocks = rows < kStatBlocks ? (int)rows : kStatBlocks;
    afd::ScopedBytes red_timing(AFD_K_BATCHNORM, 4.0 * (double)rows * slots, s);
    hipLaunchKernelGGL(wino_bnstats_reduce1_kernel, dim3(blocks), dim3(256), 0, s, part, (int)rows, slots, part2);
    hipLaunchKernelGGL(wino_bnstats_reduce2_kernel, dim3((2 * co_pad + 7) / 8), dim3(256), 0, s, part2, blocks, co_pad, Cout,
                       sums);
    return afd::check_launch("conv3x3 forward statistics reduce kernels");
}

extern "C" int afd_conv3x3_forward_stats(const float* x, const float* w, const float* bias, const float* slope,
                                         float* y, float* u, uint8_t* idx, double* sums, int N, int Cin, int H,
                                         int W, int Cout, void* ws, size_t ws_bytes, void* stat_ws,
                                         size_t stat_ws_bytes, afd_stream_t stream) {
    return forward_stats_impl(x, w, bias, slope, y, u, idx, sums, N, Cin, H, W, Cout, ws, ws_bytes, stat_ws, stat_ws_bytes,
                              stream, nullptr, nullptr);
}

// ---- backward of the BatchNorm in front of a 3x3 convolution inside that convolution's backward-data launch (round 4) ----
// g = dL/d(xhat) = w^T * dy is the gradient of a training-mode BatchNorm(affine=False) output.  Its backward needs the batch
// sums of g and of g * xhat BEFORE it can touch an element -- which is why it used to be a pass of its own after the
// backward-data launch (bn_bwd_apply: read z, read g, write dz).  Both sums can be had without g:
//   sum_px g[ci] xhat[ci]  = sum_{co,k} w[co][ci][k] dw[co][ci][k]                          (afd_conv_weight_dot)
//   sum_px g[ci]           = sum_{co,k} w[co][ci][k] R[co][k],  R[co][ky][kx] = sum of dy[co] over the rows / columns tap
//                            (ky, kx) reaches inside the image: everything, less row 0 for ky = 0, the last row for ky = 2,
//                            column 0 for kx = 0, the last column for kx = 2, plus the corner counted twice
// so with the backward-weight launch run FIRST the backward-data epilogue applies the BatchNorm and PReLU backward to its
// own result and writes dz: one read and one write of the activation tensor less per BatchNorm (4.8 GB behind block 4).
namespace {

// border sums of dy per channel: out[c][8] = row 0, last row, column 0, last column, and the four corners
// (0,0), (0,W-1), (H-1,0), (H-1,W-1) of the H x W image; dy is dense [N][C][H][W] with a live region rows x cols, or
// (codes != null) the pooled gradient [N][C][H/2][W/2] with the pool's codes (dense value at 2 pr + bit 1, 2 pc + bit 0)
__global__ void __launch_bounds__(256)
conv_border_sums_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ codes, int C, int H, int W, int rows,
                        int cols, double* __restrict__ out) {
    const int c = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (codes) {
        const int Hp = H / 2, Wp = W / 2;
        const size_t base = ((size_t)n * C + c) * Hp * Wp;
        auto at = [&](int pr, int pc, int& y, int& x) {
            const unsigned cd = codes[base + (size_t)pr * Wp + pc];
            y = 2 * pr + ((cd >> 1) & 1);
            x = 2 * pc + (cd & 1);
            return dy[base + (size_t)pr * Wp + pc];
        };
        for (int pc = tid; pc < Wp; pc += 256) {
            int y, x;
            float v = at(0, pc, y, x);
            if (y == 0) {
                s[0] += v;
                if (x == 0) s[4] += v;
                if (x == W - 1) s[5] += v;
            }
            v = at(Hp - 1, pc, y, x);  // (Hp == 1: the same windows again -- each lands in row 0 or row 1 = H - 1, not both)
            if (y == H - 1) {
                s[1] += v;
                if (x == 0) s[6] += v;
                if (x == W - 1) s[7] += v;
            }
        }
        for (int pr = tid; pr < Hp; pr += 256) {
            int y, x;
            float v = at(pr, 0, y, x);
            if (x == 0) s[2] += v;
            v = at(pr, Wp - 1, y, x);
            if (x == W - 1) s[3] += v;
        }
    } else {
        const float* p = dy + ((size_t)n * C + c) * H * W;
        for (int x = tid; x < cols; x += 256) {
            if (rows > 0) s[0] += p[x];
            if (H - 1 < rows) s[1] += p[(size_t)(H - 1) * W + x];
        }
        for (int y = tid; y < rows; y += 256) {
            s[2] += p[(size_t)y * W];
            if (W - 1 < cols) s[3] += p[(size_t)y * W + W - 1];
        }
        if (tid == 0 && rows > 0) {
            s[4] = p[0];
            if (W - 1 < cols) s[5] = p[W - 1];
            if (H - 1 < rows) {
                s[6] = p[(size_t)(H - 1) * W];
                if (W - 1 < cols) s[7] = p[(size_t)(H - 1) * W + W - 1];
            }
        }
    }
    __shared__ float red[4][8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = s[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((tid & 63) == 0) red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < 8) atomicAdd(out + (size_t)c * 8 + tid, (double)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])));
}

// sums[ci] = sum_{co,k} w[co][ci][k] R[co][k] (see above); total[co] = sum of dy[co] (double, or the float bias gradient)
__global__ void __launch_bounds__(256)
conv_input_grad_sums_kernel(const float* __restrict__ w, const double* __restrict__ total_d, const float* __restrict__ total_f,
                            const double* __restrict__ bs, int Cout, int Cin, double* __restrict__ sums) {
    __shared__ double red[256];
    const int ci = blockIdx.x;
    double acc = 0.0;
    for (int e = threadIdx.x; e < Cout * 9; e += 256) {
        const int co = e / 9, k = e - 9 * co, ky = k / 3, kx = k - 3 * ky;
        const double* b = bs + (size_t)co * 8;
        double r = total_d ? total_d[co] : (double)total_f[co];
        if (ky == 0) r -= b[0];
        if (ky == 2) r -= b[1];
        if (kx == 0) r -= b[2];
        if (kx == 2) r -= b[3];
        if (ky == 0 && kx == 0) r += b[4];
        if (ky == 0 && kx == 2) r += b[5];
        if (ky == 2 && kx == 0) r += b[6];
        if (ky == 2 && kx == 2) r += b[7];
        acc += (double)w[((size_t)co * Cin + ci) * 9 + k] * r;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[ci] = red[0];
}

}  // namespace

extern "C" int afd_conv3x3_backward_data_bnapply_applicable(int Cin, int H, int W, int Cout, int pooled) {
    if (getenv("AFD_NO_BWD_BNAPPLY")) return 0;
    if (H < 2 || W < 2 || !afd::wino44_applicable(Cout, H, W, Cin)) return 0;
    if (afd_conv3x3_backward_data_bnstats_needs_input(Cin, H, W, Cout)) return 0;
    return pooled ? afd_conv3x3_pooled_backward_applicable(Cin, H, W, Cout) : 1;
}

// sums[0 .. Cin) = sum over the batch and the pixels of g = w^T * dy per input channel, from the weights and the border
// sums of dy (sums must hold Cin + 8 Cout doubles: the tail is scratch); dy dense with its live region dy_rows x dy_cols,
// or pooled with codes; dy_sums (double) or dbias (float): the per-channel sums of dy, whichever the caller has
extern "C" int afd_conv3x3_input_grad_sums(const float* dy, const uint8_t* codes, const float* w, const double* dy_sums,
                                           const float* dbias, double* sums, int N, int Cin, int H, int W, int Cout,
                                           int dy_rows, int dy_cols, afd_stream_t stream) {
    if (!dy || !w || !sums || (!dy_sums && !dbias) || N < 1 || Cin < 1 || Cout < 1 || H < 2 || W < 2)
        return afd::fail(AFD_ERR_ARG, "conv3x3 input-gradient sums: bad argument");
    if (N > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3 input-gradient sums: N > 65535");
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* bs = sums + Cin;
    hipError_t e = hipMemsetAsync(bs, 0, sizeof(double) * 8 * Cout, s);
    if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "conv3x3 input-gradient sums: memset: %s", hipGetErrorString(e));
    if (codes) {
        dy_rows = 2 * (H / 2);
        dy_cols = 2 * (W / 2);
    }
    const int rows = dy_rows < H ? dy_rows : H, cols = dy_cols < W ? dy_cols : W;
    // two rows and two columns of every dy plane (the pooled form: of the pooled plane and its codes)
    afd::ScopedBytes timing(AFD_K_BATCHNORM, (codes ? 5.0 : 4.0) * (double)N * Cout * (double)(codes ? (W / 2 + H / 2) : 2 * (W + H)), s);
    hipLaunchKernelGGL(conv_border_sums_kernel, dim3(Cout, N), dim3(256), 0, s, dy, codes, Cout, H, W, rows, cols, bs);
    hipLaunchKernelGGL(conv_input_grad_sums_kernel, dim3(Cin), dim3(256), 0, s, w, dy_sums, dbias, bs, Cout, Cin, sums);
    return afd::check_launch("conv3x3 input-gradient sums kernels");
}

// backward-data with the BatchNorm (+ PReLU) backward of its result in the epilogue: dz [N][Cin][H][W] instead of g; z the
// BatchNorm's input, bn_tab [Cin][4] = (mean, invstd, mean of g, mean of g * xhat), bn_slope the PReLU slope or null;
// sums[0 .. Cin) = sum(dz) per channel, sums[Cin .. 2 Cin) = the slope's partial gradients (their sum is dslope);
// dy dense, or (codes != null) the pooled gradient of the convolution's PReLU + max-pool.  bn_codes != null: the
// BatchNorm sits right behind PReLU + MaxPool2d(2, 2) -- z is the pooled tensor, bn_codes the pool's codes, bn_slope that
// PReLU's slope -- and dz is the pooled gradient afd_prelu_pool_backward_compact would leave
extern "C" int afd_conv3x3_backward_data_bnapply(const float* dy, const uint8_t* codes, const float* w, const float* z,
                                                 const float* bn_tab, const float* bn_slope, const uint8_t* bn_codes,
                                                 float* dz, double* sums, int N, int Cin, int H, int W, int Cout, void* ws,
                                                 size_t ws_bytes, void* stat_ws, size_t stat_ws_bytes, afd_stream_t stream) {
    if (!dy || !w || !z || !bn_tab || !dz || !sums || !ws || !stat_ws)
        return afd::fail(AFD_ERR_ARG, "conv3x3 dgrad + bn backward: null pointer");
    if (N < 1 || !afd_conv3x3_backward_data_bnapply_applicable(Cin, H, W, Cout, codes != nullptr))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3 dgrad + bn backward: shape not on the F(4x4) kernel");
    if (stat_ws_bytes < afd_conv3x3_backward_data_bnstats_workspace_bytes(N, Cin, H, W))
        return afd::fail(AFD_ERR_WORKSPACE, "conv3x3 dgrad + bn backward: statistics workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int co_pad = (Cin + 31) / 32 * 32;
    const int slots = 2 * co_pad;
    const long rows = afd::wino44_stat_rows(N, H, W);
    float* part = static_cast<float*>(stat_ws);
    double* part2 = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(part + (size_t)rows * slots) + 63) & ~(uintptr_t)63);
    int rc = afd::wino44_run(dy, w, nullptr, dz, N, Cout, H, W, Cin, 1, H, W, ws, ws_bytes, s, z, part, nullptr, nullptr, nullptr,
                             0, codes, nullptr, nullptr, bn_tab, bn_slope, bn_codes);
    if (rc) return rc;
    const int blocks = rows < kStatBlocks ? (int)rows : kStatBlocks;
    afd::ScopedBytes red_timing(AFD_K_BATCHNORM, 4.0 * (double)rows * slots, s);
    hipLaunchKernelGGL(wino_bnstats_reduce1_kernel, dim3(blocks), dim3(256), 0, s, part, (int)rows, slots, part2);
    hipLaunchKernelGGL(wino_bnstats_reduce2_kernel, dim3((2 * co_pad + 7) / 8), dim3(256), 0, s, part2, blocks, co_pad, Cin,
                       sums);
    return afd::check_launch("wino_bnstats_reduce kernels");
}

// ---- the BatchNorm in FRONT of a 3x3 convolution applied while the convolution loads (round 4) ----
// A training-mode BatchNorm(affine=False) whose only consumer is a 3x3 / pad 1 convolution on the F(4x4) kernels need
// not write its result: the convolution's forward and backward-weight launches take the BatchNorm's INPUT (z and the
// PReLU slope between them, or the tensor itself) and (mean, invstd) per channel and build (PReLU(z) - mean) * invstd
// in registers, with the arithmetic of afd_bn_apply_forward (bit-identical values): one read and one write of the
// activation tensor less per BatchNorm and step (7 GB for DCNN block 2 -> 3 at level 14), and the tensor is never
// stored.  The backward-data launch does not read the convolution's input (afd_conv_weight_dot supplies the BatchNorm's
// second backward sum), so it is the launch of the unfolded layer.
//   pooled = the convolution is followed by PReLU + MaxPool2d(2, 2) in the same launch; want_stats = the launch also
//   sums its result for the BatchNorm behind it (afd_conv3x3_forward_stats)
extern "C" int afd_conv3x3_input_fold_applicable(int Cin, int H, int W, int Cout, int pooled, int want_stats) {
    if (getenv("AFD_NO_INPUT_FOLD")) return 0;
    if (H < 2 || W < 2) return 0;
    if (want_stats ? !afd_conv3x3_forward_stats_applicable(Cin, H, W, Cout, pooled)
                   : !(pooled && afd::wino44_pool_applicable(Cin, H, W, Cout)))
        return 0;
    if (!afd_conv3x3_backward_data_bnstats_applicable(Cin, H, W, Cout) ||
        afd_conv3x3_backward_data_bnstats_needs_input(Cin, H, W, Cout))
        return 0;
    if (!afd::wino44_wgrad_applicable(Cin, H, W, Cout, 3, 1, 1)) return 0;
    return afd::wino44_wgrad_crop_ok(H, W, pooled ? 2 * (H / 2) : H, pooled ? 2 * (W / 2) : W) ? 1 : 0;
}

// forward: afd_conv3x3_forward_stats (sums != null) or afd_conv3x3_prelu_pool_forward (sums == null, u != null) with x
// the BatchNorm's input; in_aff [Cin][2] = (mean, invstd), in_slope the PReLU slope in front of the BatchNorm or null
extern "C" int afd_conv3x3_forward_fold(const float* x, const float* in_aff, const float* in_slope, const float* w,
                                        const float* bias, const float* slope, float* y, float* u, uint8_t* idx,
                                        double* sums, int N, int Cin, int H, int W, int Cout, void* ws, size_t ws_bytes,
                                        void* stat_ws, size_t stat_ws_bytes, afd_stream_t stream) {
    if (!in_aff) return afd::fail(AFD_ERR_ARG, "conv3x3 with input fold: null pointer");
    const int pooled = u != nullptr;
    if (N < 1 || !afd_conv3x3_input_fold_applicable(Cin, H, W, Cout, pooled, sums != nullptr))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3 with input fold: layer not on the F(4x4) Winograd kernels");
    if (sums)
        return forward_stats_impl(x, w, bias, slope, y, u, idx, sums, N, Cin, H, W, Cout, ws, ws_bytes, stat_ws,
                                  stat_ws_bytes, stream, in_aff, in_slope);
    if (!x || !w || !ws || !idx || !slope) return afd::fail(AFD_ERR_ARG, "conv3x3 with input fold: null pointer");
    return afd::wino44_run(x, w, bias, nullptr, N, Cin, H, W, Cout, 0, 2 * (H / 2), 2 * (W / 2), ws, ws_bytes,
                           static_cast<hipStream_t>(stream), nullptr, nullptr, slope, u, idx, 0, nullptr, in_aff, in_slope);
}

// backward-weight: dy the dense gradient (codes == null; rows / columns past dy_rows x dy_cols are not read) or the
// pooled gradient with the pool's codes; dbias as afd_conv2d_backward_weight_sums (dy_sums: the per-channel sums of dy
// when its producer has them)
__global__ void fold_bias_from_sums_kernel(const double* __restrict__ sums, float* __restrict__ db, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) db[c] = (float)sums[c];
}

extern "C" int afd_conv3x3_backward_weight_fold(const float* x, const float* in_aff, const float* in_slope,
                                                const float* dy, const uint8_t* codes, float* dw, float* dbias,
                                                const double* dy_sums, int N, int Cin, int H, int W, int Cout,
                                                int dy_rows, int dy_cols, void* ws, size_t ws_bytes,
                                                afd_stream_t stream) {
    if (!x || !in_aff || !dy || !dw || !ws) return afd::fail(AFD_ERR_ARG, "conv3x3 wgrad with input fold: null pointer");
    if (codes) {
        dy_rows = 2 * (H / 2);
        dy_cols = 2 * (W / 2);
    }
    if (N < 1 || !afd::wino44_wgrad_applicable(Cin, H, W, Cout, 3, 1, 1) || !afd::wino44_wgrad_crop_ok(H, W, dy_rows, dy_cols))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv3x3 wgrad with input fold: shape not on the Winograd-domain kernel");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = afd::wino44_wgrad_run(x, dy, dw, (dbias && !dy_sums) ? dbias : nullptr, N, Cin, H, W, Cout, dy_rows, dy_cols, ws,
                                   ws_bytes, s, codes, in_aff, in_slope);
    if (rc) return rc;
    if (dbias && dy_sums) {
        hipLaunchKernelGGL(fold_bias_from_sums_kernel, dim3((Cout + 63) / 64), dim3(64), 0, s, dy_sums, dbias, Cout);
        return afd::check_launch("fold_bias_from_sums_kernel");
    }
    return AFD_OK;
}
