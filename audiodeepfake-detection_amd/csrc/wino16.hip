// Winograd F(2x2, 3x3) forward / backward-data, second layout: 16x16x4 MFMA tiles with all 16
// transform positions of an output element in ONE lane, so the output transform A^T M A runs in
// registers (wino.hip splits the positions over waves and meets them in LDS: ~17 % of its time).
//
// Same contract and arithmetic as wino.hip (reference: the nn.Conv2d(k=3, padding=1) launches of
// DCNN blocks 3-6, src/audiofakedetect/models.py:263-278, and their backward-data passes).
//
//   wave   = 16 output channels x 32 tiles (two 16-column MFMA tiles) x 16 positions:
//            accumulators 16 x 2 x 4 = 128 registers; D fragment: lane = tile, 4 registers = 4
//            consecutive channels -> every (channel, tile) has its 16 positions in one lane.
//   workgroup = Cout / 16 waves over the same 32 tiles of one tile row.  Per chunk of 8 input
//            channels thread (channel, tile) transforms its 4x4 patch into the double-buffered LDS
//            image V[position][channel][tile] (as in wino.hip); every wave reads all of V (B
//            fragments: 4 channels x 16 tiles, requested four positions at a time, one block ahead
//            of their MFMAs) and its own U fragments (table in fragment order, the chunk's 32
//            requested up front).  One barrier per chunk.
//   epilogue = registers only: 8 (channel, tile) pairs per lane, bias, two 8-byte stores each.
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

constexpr int kCh = 8;      // input channels per chunk = 2 k-steps of 4
constexpr int kTiles = 32;  // tiles per workgroup

struct GV {
    int N, Cin, Cout, H, W;
    int rows, cols;
    int tilesX, tilesY, wgX, wxCount, nchunks;
    const float* x_end;  // one past the input tensor (border patches: see load_patch)
};

// U table: [chunk][position][cg][kstep (2)][lane] = U_p[16 cg + (lane & 15)][8 chunk + 4 kstep + (lane >> 4)]
__global__ void wino16_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout,
                                      int CG, int nchunks, int dgrad) {
    const int total = nchunks * 16 * CG * 2 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63;
        const int ks = (i >> 6) & 1;
        int r = i >> 7;
        const int cg = r % CG;
        r /= CG;
        const int p = r & 15;
        const int chunk = r >> 4;
        const int co = 16 * cg + (lane & 15);
        const int ci = kCh * chunk + 4 * ks + (lane >> 4);
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            float g[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    g[ky][kx] = dgrad ? w[((size_t)ci * Cout + co) * 9 + (8 - (ky * 3 + kx))]
                                      : w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
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

// CG = waves = 16-channel groups of Cout; WPE = waves per SIMD the register budget is set for
template <int CG, int WPE, bool BORDER>
__global__ void __launch_bounds__(CG * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
wino16_conv_kernel(const GV g, const float* __restrict__ x, const float* __restrict__ U,
                   const float* __restrict__ bias, float* __restrict__ y) {
    constexpr int NT = CG * 64;
    constexpr int VBUF = 16 * kCh * kTiles;
    __shared__ __attribute__((aligned(16))) float V[2 * VBUF];  // [2][16][kCh][kTiles], 32 KB
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int id = blockIdx.x;
    const int wi = id % g.wxCount;
    id /= g.wxCount;
    const int wx = BORDER ? (wi == 0 ? 0 : g.wgX - 1) : wi + 1;
    const int ty = id % g.tilesY;
    const int n = id / g.tilesY;
    const int tx0 = wx * kTiles;

    // transform role: thread owns (channel ch [+ CHS], tile tl) of every chunk: with >= 4 waves the first
    // 256 threads take one patch each, the 2-wave workgroup (Cout <= 32) two patches per thread
    constexpr int PPT = NT >= kCh * kTiles ? 1 : 2;   // patches per thread
    constexpr int CHS = NT >= kCh * kTiles ? 0 : 4;   // channel step between a thread's patches
    static_assert(PPT == 1 || NT == 128, "two patches per thread: exactly two waves");
    const bool xf = tid < kCh * kTiles;
    const int tl = tid & 31, ch = (tid >> 5) & 7;
    const int txp = tx0 + tl;
    const int iy0 = 2 * ty - 1, ix0 = 2 * txp - 1;
    const bool tile_ok = txp < g.tilesX;
    const size_t plane = (size_t)g.H * g.W;
    const float* xn = x + (size_t)n * g.Cin * plane;
    unsigned okmask = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = iy0 + r, ix = ix0 + j;
            const bool ok = iy >= 0 && iy < g.H && (!BORDER || (tile_ok && ix >= 0 && ix < g.W));
            okmask |= ok ? (1u << (4 * r + j)) : 0u;
        }
    float d[PPT][4][4];
    auto load_patch = [&](int c) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const float* xc = xn + (size_t)(c * kCh + ch + q * CHS) * plane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int iy = iy0 + r;
                const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
                const float* row = xc + (size_t)iyc * g.W;
                if (!BORDER) {
                    const f4u v = *reinterpret_cast<const f4u*>(row + ix0);
                    d[q][r][0] = v.x; d[q][r][1] = v.y; d[q][r][2] = v.z; d[q][r][3] = v.w;
                } else {
                    // as in wino.hip: the interior's 16-byte load (what it reads past a row end is inside the tensor
                    // and masked when consumed); element loads only where the load would leave the tensor
                    const float* p4 = row + ix0;
                    if (p4 >= x && p4 + 4 <= g.x_end) {
                        const f4u v = *reinterpret_cast<const f4u*>(p4);
                        d[q][r][0] = v.x; d[q][r][1] = v.y; d[q][r][2] = v.z; d[q][r][3] = v.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int ix = ix0 + j;
                            d[q][r][j] = row[ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix)];
                        }
                    }
                }
            }
        }
    };
    auto store_v = [&](int buf) {
#pragma unroll
        for (int p2 = 0; p2 < PPT; ++p2) {
            auto dv = [&](int r, int j) { return (okmask >> (4 * r + j)) & 1u ? d[p2][r][j] : 0.f; };
            float* vb = V + buf * VBUF + (ch + p2 * CHS) * kTiles + tl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[j] = q == 0 ? dv(0, j) - dv(2, j) : q == 1 ? dv(1, j) + dv(2, j) : q == 2 ? dv(2, j) - dv(1, j) : dv(1, j) - dv(3, j);
                float* o = vb + (q * 4) * (kCh * kTiles);
                o[0 * kCh * kTiles] = t[0] - t[2];
                o[1 * kCh * kTiles] = t[1] + t[2];
                o[2 * kCh * kTiles] = t[2] - t[1];
                o[3 * kCh * kTiles] = t[1] - t[3];
            }
        }
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) acc[p][sb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int kq = lane >> 4, l15 = lane & 15;
    // U fragments of chunk c, position p: Uw[((c * 16 + p) * CG) * 128 + ks * 64]
    const float* Uw = U + (size_t)wave * 128 + lane;
    if (xf) {
        load_patch(0);
        store_v(0);
        if (g.nchunks > 1) load_patch(1);
    }
    __syncthreads();
    for (int c = 0; c < g.nchunks; ++c) {
        const float* uc = Uw + (size_t)c * 16 * CG * 128;
        const float* vb = V + (c & 1) * VBUF + kq * kTiles + l15;
        // operands in blocks: the chunk's 32 U fragments are requested up front (consumed from the
        // second block on), the V fragments of four positions at a time, one block ahead of their MFMAs
        float u[16][2];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            u[p][0] = uc[(size_t)p * CG * 128];
            u[p][1] = uc[(size_t)p * CG * 128 + 64];
        }
        float bf[2][4][4];
        auto load_b = [&](int blk, int slot) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* vp = vb + (4 * blk + q) * (kCh * kTiles);
                bf[slot][q][0] = vp[0];
                bf[slot][q][1] = vp[16];
                bf[slot][q][2] = vp[4 * kTiles];
                bf[slot][q][3] = vp[4 * kTiles + 16];
            }
        };
        load_b(0, 0);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            if (blk + 1 < 4) load_b(blk + 1, (blk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = 4 * blk + q;
                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[p][0], bf[blk & 1][q][0], acc[p][0], 0, 0, 0);
                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[p][0], bf[blk & 1][q][1], acc[p][1], 0, 0, 0);
                acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[p][1], bf[blk & 1][q][2], acc[p][0], 0, 0, 0);
                acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[p][1], bf[blk & 1][q][3], acc[p][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (xf) {
            if (c + 1 < g.nchunks) store_v((c + 1) & 1);
            if (c + 2 < g.nchunks) load_patch(c + 2);
        }
        __syncthreads();
    }

    // output transform in registers: Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1]
    // D fragment: column (tile) = lane & 15, rows (channels) = 4 (lane >> 4) + j
    const int oy = 2 * ty;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const int txe = tx0 + sb * 16 + l15;
        const int ox = 2 * txe;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = 16 * wave + 4 * kq + j;
            float s0[4], s1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s0[q] = acc[q][sb][j] + acc[4 + q][sb][j] + acc[8 + q][sb][j];
                s1[q] = acc[4 + q][sb][j] - acc[8 + q][sb][j] - acc[12 + q][sb][j];
            }
            if (co < g.Cout && txe < g.tilesX) {
                const float bv = bias ? bias[co] : 0.f;
                const float y00 = s0[0] + s0[1] + s0[2] + bv, y01 = s0[1] - s0[2] - s0[3] + bv;
                const float y10 = s1[0] + s1[1] + s1[2] + bv, y11 = s1[1] - s1[2] - s1[3] + bv;
                float* yo = y + (((size_t)n * g.Cout + co) * g.H + oy) * g.W + ox;
                if (ox + 1 < g.cols) {
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

template <int CG, int WPE>
int launch16(GV g, const float* x, const float* U, const float* bias, float* y, hipStream_t s) {
    g.wgX = (g.tilesX + kTiles - 1) / kTiles;
    const long rows = (long)g.N * g.tilesY;
    // 16 GEMMs [16 CG x Cin] x [Cin x kTiles tiles] per workgroup, every tile computed in full
    afd::timing_annotate(2.0 * 16 * (16.0 * CG) * ((double)kTiles * g.wgX) * (double)rows * g.Cin, -1.0);
    const int inner = g.wgX > 2 ? g.wgX - 2 : 0;
    const int edge = g.wgX >= 2 ? 2 : 1;
    if (rows * (inner > edge ? inner : edge) > 0x7fffffffL)
        return afd::fail(AFD_ERR_UNSUPPORTED, "winograd conv: grid too large");
    if (inner > 0) {
        g.wxCount = inner;
        hipLaunchKernelGGL((wino16_conv_kernel<CG, WPE, false>), dim3((unsigned)(rows * inner)), dim3(CG * 64), 0, s,
                           g, x, U, bias, y);
    }
    g.wxCount = edge;
    hipLaunchKernelGGL((wino16_conv_kernel<CG, WPE, true>), dim3((unsigned)(rows * edge)), dim3(CG * 64), 0, s, g, x,
                       U, bias, y);
    return afd::check_launch("wino16_conv_kernel");
}

}  // namespace

namespace afd {

bool wino16_applicable(int Cin, int H, int W, int Cout) {
    if (getenv("AFD_NO_WINOGRAD") || getenv("AFD_NO_WINO16")) return false;
    // measured against wino.hip: ahead with 8 waves (Cout 113..128: block 4 forward 1.06 -> 0.96 ms,
    // block 5 backward-data 0.43 -> 0.38 ms at B = 32), level or behind with 4, and 6 waves do not
    // fit three to a SIMD without spilling; AFD_WINO16=1 takes every Cout >= 49 (A/B runs)
    if (Cin % kCh != 0 || Cin < kCh || Cout > 128 || Cout < 17) return false;
    if (Cout > 32 && Cout < 49) return false;  // three waves: no transform mapping built
    // 2 waves (Cout 17..32: block 5 forward, block 6 backward-data, the layers wino.hip leaves to the
    // direct kernel) and 8 waves take this kernel; the sizes in between stay on wino.hip
    if (Cout > 32 && Cout <= 112 && !getenv("AFD_WINO16")) return false;
    if (W < 64 || H < 2) return false;
    return (size_t)H * W < 0x7fffffffULL;
}

size_t wino16_workspace_bytes(int Cin, int Cout) {
    const size_t cg = (size_t)(Cout + 15) / 16;
    return (size_t)(Cin / kCh) * 16 * cg * 2 * 64 * sizeof(float);
}

int wino16_run(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
               int Cout, int dgrad, int out_rows, int out_cols, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < wino16_workspace_bytes(Cin, Cout))
        return afd::fail(AFD_ERR_WORKSPACE, "winograd conv: workspace too small");
    GV g{};
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.H = H; g.W = W;
    g.rows = out_rows < H ? out_rows : H;
    g.cols = out_cols < W ? out_cols : W;
    g.tilesX = (g.cols + 1) / 2;
    g.tilesY = (g.rows + 1) / 2;
    g.nchunks = Cin / kCh;
    g.x_end = x + (size_t)N * Cin * H * W;
    const int CG = (Cout + 15) / 16;
    float* U = static_cast<float*>(ws);
    const int total = g.nchunks * 16 * CG * 2 * 64;
    hipLaunchKernelGGL(wino16_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, U, Cin, Cout, CG,
                       g.nchunks, dgrad);
    int rc = afd::check_launch("wino16_weights_kernel");
    if (rc) return rc;
    afd::ScopedTiming timing(AFD_K_CONV_WINOGRAD, 2.0 * N * Cout * (double)g.rows * g.cols * Cin * 9, s);
    timing.bytes(4.0 * N * ((double)Cin * H * W + (double)Cout * g.rows * g.cols));
    switch (CG) {
        case 2: return launch16<2, 2>(g, x, U, bias, y, s);
        case 4: return launch16<4, 2>(g, x, U, bias, y, s);
        case 5: return launch16<5, 2>(g, x, U, bias, y, s);
        case 6: return launch16<6, 2>(g, x, U, bias, y, s);
        case 7: return launch16<7, 2>(g, x, U, bias, y, s);
        case 8: return launch16<8, 2>(g, x, U, bias, y, s);
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "winograd conv: Cout %d", Cout);
}

}  // namespace afd
