// First DCNN / LCNN block for single-channel inputs, fused: Conv2d(1 -> Cout, 3x3, pad p) +
// PReLU + MaxPool2d(2,2) forward, and the matching backward (pool routing + PReLU + weight,
// bias and slope gradients) without ever materialising the pre-pool tensor.
//
// Replaces cnn[0..2] of the reference DCNN (src/audiofakedetect/models.py:255-259) for
// args.input_dim[1] == 1.  At level 14 (B = 128) the pre-pool activation is 14 GB: the
// unfused path writes it, reads it for the pool, writes a 14 GB gradient in the backward pool
// and reads that again for the weight gradient.  Here both passes touch only the pooled
// tensors (u 3.5 GB, 3-bit codes 0.9 GB) and the 0.2 GB input.  With K*K = 9 taps the MFMA
// K dimension would be 90 % padding; this is f32 VALU work, memory bound.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

constexpr int kT = 256;
constexpr int kCG = 8;  // channels per backward workgroup

__device__ __forceinline__ float prelu1(float z, float a) { return z > 0.f ? z : a * z; }

// 4x4 input patch of pooled pixel (py, px): rows 2py-pad .. +3, cols 2px-pad .. +3, zero padded
__device__ __forceinline__ void load_patch(const float* __restrict__ xn, int H, int W, int py, int px,
                                           int pad, float (&p)[4][4]) {
    const int r0 = 2 * py - pad, c0 = 2 * px - pad;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = r0 + j;
        const bool rok = (r >= 0) && (r < H);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + i;
            p[j][i] = (rok && c >= 0 && c < W) ? xn[(size_t)r * W + c] : 0.f;
        }
    }
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4u1 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32u1 __attribute__((aligned(1)));

constexpr int kFwdPixels = 4;  // pooled pixels per thread of the forward kernel's row form

// A thread owns kFP consecutive pooled pixels of one row: its results leave as one 16-byte store of u and one
// 4-byte store of the codes per channel (a wave writes 1 KB + 256 B per instruction instead of 256 B + 64 B --
// the kernel moves 4.4 GB of results against 0.2 GB of input and was bound by its narrow stores), and the
// 4 x (2 kFP + 2) input patch is shared by the pixels.  The two outputs of a window row share a weight and take
// horizontally adjacent samples: one packed FMA (v_pk_fma_f32, weight broadcast from a scalar register) does both.
// Sum of a value over the 64 lanes of a wave on the vector ALU alone (DPP: quad permutes, row mirrors, row broadcasts);
// the total ends in lane 63.  (A shuffle tree through ds_bpermute is a chain of LDS round trips: round 3 measured the
// statistics epilogue of this kernel at +0.9 ms with it.)
#define AFD_DPP_ADD(v, ctrl, rows) \
    (v) += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rows), 0xF, false))
__device__ __forceinline__ float wave_sum63(float v) {
    AFD_DPP_ADD(v, 0xB1, 0xF);   // quad_perm [1,0,3,2]
    AFD_DPP_ADD(v, 0x4E, 0xF);   // quad_perm [2,3,0,1]
    AFD_DPP_ADD(v, 0x141, 0xF);  // row_half_mirror
    AFD_DPP_ADD(v, 0x140, 0xF);  // row_mirror: every lane holds its row's sum
    AFD_DPP_ADD(v, 0x142, 0xA);  // row_bcast15 into rows 1 and 3
    AFD_DPP_ADD(v, 0x143, 0xC);  // row_bcast31 into rows 2 and 3: lane 63 holds the wave's sum
    return v;
}
#undef AFD_DPP_ADD

// STATS (round 5): the kernel also sums, per channel, the pooled values and their squares over its workgroup's pixels --
// the batch statistics of the BatchNorm behind the pool (reference models.py:258-260) -- into one partial row
// [sum | sum of squares] per workgroup: the statistics pass over the 3.5 GB pooled tensor (0.6 ms at level 14) is gone
// FLAT (round 6; narrow images, one pooled pixel per thread): the threads run over the FLATTENED pooled pixels of an image
// (f = py * Wp + px) instead of over one row -- on the 129-column rows of the level-8 / STFT model the row form filled
// 33 lanes of one wave per row (the kernel is bound by its vector instructions: 0.18 -> 0.10 ms at batch 128).
template <bool STATS, int kFP = 4, bool FLAT = false>
__global__ void __launch_bounds__(kT)
conv1_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                      const float* __restrict__ bias, const float* __restrict__ slope,
                      float* __restrict__ u, unsigned char* __restrict__ idx, int H, int W, int Cout,
                      int pad, int Hp, int Wp, float* __restrict__ stat_part) {
    static_assert(!FLAT || (kFP == 1 && !STATS), "the flat form takes one pixel per thread and no statistics");
    constexpr int PC = 2 * kFP + 2;  // patch columns
    __shared__ float wsum[STATS ? kT / 64 : 1][STATS ? 2 * 128 : 1];  // [wave][sum | squares][channel <= 128]
    int px0, py;
    const int n = blockIdx.z;
    if constexpr (FLAT) {
        const unsigned f = blockIdx.x * kT + threadIdx.x;
        if (f >= (unsigned)(Hp * Wp)) return;
        py = (int)(f / (unsigned)Wp);
        px0 = (int)f - py * Wp;
    } else {
        px0 = (blockIdx.x * kT + threadIdx.x) * kFP;
        py = blockIdx.y;
    }
    if (!STATS && px0 >= Wp) return;
    const bool live = px0 < Wp;  // STATS: the threads stay for the reductions; dead lanes of a live wave compute on zeros
    // a wave entirely past the row (narrow images: 129 pooled columns use 33 of a workgroup's 256 threads) only zeroes its
    // cells of the cross-wave sum
    const bool wave_live = (int)((blockIdx.x * kT + (threadIdx.x & ~63)) * kFP) < Wp;  // uniform per wave
    if (STATS && !wave_live) {
        for (int e = threadIdx.x & 63; e < 2 * 128; e += 64) wsum[threadIdx.x >> 6][e] = 0.f;
    }
    const float a = slope[0];
    const bool mono = a > 0.f;  // uniform
    const float* xn = x + (size_t)n * H * W;
    float p[4][PC];
    {
        const int r0 = 2 * py - pad, c0 = 2 * px0 - pad;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = r0 + j;
            const bool rok = (r >= 0) && (r < H);
#pragma unroll
            for (int i = 0; i < PC; ++i) {
                const int c = c0 + i;
                p[j][i] = (rok && c >= 0 && c < W) ? xn[(size_t)r * W + c] : 0.f;
            }
        }
    }
    f2 pp[4][PC - 1];  // pp[r][i] = (p[r][i], p[r][i + 1])
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < PC - 1; ++i) pp[r][i] = f2{p[r][i], p[r][i + 1]};
    const size_t plane = (size_t)Hp * Wp;
    size_t o = ((size_t)n * Cout * Hp + py) * Wp + px0;
    const bool whole = px0 + kFP <= Wp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int co = 0; co < (STATS && !wave_live ? 0 : Cout); ++co, o += plane) {
        const float* wc = w + co * 9;  // uniform address: scalar loads
        const float b = bias ? bias[co] : 0.f;
        float best[kFP];
        unsigned code = 0;
#pragma unroll
        for (int j = 0; j < kFP; ++j) {
            f2 z01 = {b, b}, z23 = {b, b};  // (z0, z1): window row 0, (z2, z3): window row 1
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wv = wc[ky * 3 + kx];
                    const f2 ww = {wv, wv};
                    z01 = __builtin_elementwise_fma(ww, pp[ky][2 * j + kx], z01);
                    z23 = __builtin_elementwise_fma(ww, pp[ky + 1][2 * j + kx], z23);
                }
            const float z0 = z01.x, z1 = z01.y, z2 = z23.x, z3 = z23.y;
            float bst, zb;
            int bi;
            if (mono) {
                // slope > 0: PReLU is strictly increasing, the window's first maximum of PReLU(z) is the first
                // maximum of z -- two max instructions and an equality chain instead of four PReLUs and three
                // compare / select rounds (this kernel's vector instructions and its stores add up: 1.60 -> 1.45 ms)
                zb = fmaxf(fmaxf(z0, z1), fmaxf(z2, z3));
                bi = z0 == zb ? 0 : (z1 == zb ? 1 : (z2 == zb ? 2 : 3));
                bst = prelu1(zb, a);
            } else {
                bst = prelu1(z0, a); zb = z0; bi = 0;
                float v = prelu1(z1, a);
                if (v > bst) { bst = v; bi = 1; zb = z1; }
                v = prelu1(z2, a);
                if (v > bst) { bst = v; bi = 2; zb = z2; }
                v = prelu1(z3, a);
                if (v > bst) { bst = v; bi = 3; zb = z3; }
            }
            best[j] = bst;
            code |= (unsigned)(bi | (zb <= 0.f ? 4 : 0)) << (8 * j);
        }
        if constexpr (STATS) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < kFP; ++j) {
                const float v = (live && px0 + j < Wp) ? best[j] : 0.f;
                s1 += v;
                s2 = fmaf(v, v, s2);
            }
            s1 = wave_sum63(s1);
            s2 = wave_sum63(s2);
            if (lane == 63) {
                wsum[wave][co] = s1;
                wsum[wave][128 + co] = s2;
            }
        }
        if constexpr (kFP == 4) {
            if (whole) {
                f4u1 v4 = {best[0], best[1], best[2], best[3]};
                *reinterpret_cast<f4u1*>(u + o) = v4;
                *reinterpret_cast<u32u1*>(idx + o) = code;
                continue;
            }
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < kFP; ++j)
                if (px0 + j < Wp) {
                    u[o + j] = best[j];
                    idx[o + j] = (unsigned char)((code >> (8 * j)) & 0xff);
                }
        }
    }
    if constexpr (STATS) {
        __syncthreads();
        // one partial row [sum(Cout) | squares(Cout)] per workgroup, waves added in a fixed order
        const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        for (int e = threadIdx.x; e < 2 * Cout; e += kT) {
            const int which = e / Cout, c = e - which * Cout;
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < kT / 64; ++wv) v += wsum[wv][which * 128 + c];
            stat_part[(size_t)wg * 2 * Cout + e] = v;
        }
    }
}

// sums[e] (e < 2 Cout) = sum over the workgroups' partial rows, in double precision: one workgroup per output
__global__ void __launch_bounds__(256)
conv1_stats_reduce_kernel(const float* __restrict__ part, int rows, int C2, double* __restrict__ sums) {
    __shared__ double red[256];
    const int e = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)part[(size_t)r * C2 + e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[e] = red[0];
}

// partial[split][cg][kCG][11]: 9 weight gradients, bias gradient, slope gradient
//
// A workgroup takes every S-th tile (256 pooled pixels of one pooled row) of the flat tile list: at any moment the
// chip works on S consecutive tiles, i.e. a contiguous band of every channel plane (few, wide streams).  Round 6
// (same-box A/B of five builds, profiles/r06_conv1_bwd_ab.txt; class conv_first = this launch + the forward one):
//   as it was (two integer divisions per tile on the scalar unit, 31 scalar instructions per pixel and channel)  3.61 ms
//   tile coordinates advanced by carries instead, wave sums through DPP (kept)                                    3.50
//   + the PReLU branch as arithmetic                                                                               3.58
//   + patch rows as one 16-byte load (or two 8-byte loads) per lane instead of four 4-byte loads                    3.97
//   a workgroup walking DOWN the pooled rows of a column tile (two patch rows carried over, constant address steps) 3.89
// The wide patch loads lose although they are the same bytes in a quarter of the instructions: neighbouring lanes'
// 16-byte windows overlap by half, and the one-channel image is the only operand every channel group re-reads.
__global__ void __launch_bounds__(kT)
conv1_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ du,
                      const unsigned char* __restrict__ idx, const float* __restrict__ u,
                      const float* __restrict__ slope, float* __restrict__ partial, int N, int H,
                      int W, int Cout, int pad, int Hp, int Wp, int tilesX, long totalUnits,
                      const float* __restrict__ aff_alpha, const float* __restrict__ aff_beta, int xcd_split,
                      unsigned wp_magic) {
    __shared__ float red[kT / 64][kCG * 11];
    // The channel groups of one split read the same input patches (the one-channel image, 4 rows x 2 KB per tile).  As
    // grid (cg, split) the eight groups of a split had consecutive block ids -- one per XCD, the image fetched once per
    // XCD (3.2 GB of the launch's 13.6 GB, profiles/r05_pmc_traffic_*).  Linear ids in groups of 64: XCD x = id & 7 takes
    // split 8 g + x with its eight channel groups in consecutive dispatch slots, so seven of the eight patch reads hit L2.
    int cg, split;
    const int S = gridDim.y;
    if (xcd_split) {
        const int id = blockIdx.y * gridDim.x + blockIdx.x;  // gridDim.x == 8 (checked by the host)
        split = (id >> 6) * 8 + (id & 7);
        cg = (id >> 3) & 7;
    } else {
        cg = blockIdx.x;
        split = blockIdx.y;
    }
    const float a = slope[0];
    const float inva = a != 0.f ? 1.f / a : 0.f;
    float acc[kCG][9];
    float accb[kCG], accs[kCG];
    // AFFINE: the gradient of the pooled tensor is du + alpha[c] * u + beta[c] (uniform addresses: scalar loads)
    float al[kCG], be[kCG];
#pragma unroll
    for (int c = 0; c < kCG; ++c) {
        const bool ok = aff_alpha && cg * kCG + c < Cout;
        al[c] = ok ? aff_alpha[cg * kCG + c] : 0.f;
        be[c] = ok ? aff_beta[cg * kCG + c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < kCG; ++c) {
        accb[c] = 0.f;
        accs[c] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[c][k] = 0.f;
    }
    const size_t plane = (size_t)Hp * Wp;
    const int ntiles = (int)totalUnits;  // N * tiles per image < 2^31 (checked by the host): 32-bit tile arithmetic
    const int nch = min(kCG, Cout - cg * kCG);  // live channels of this group (uniform)
    // Tiles run over the FLATTENED pooled pixels of an image (f = py * Wp + px, contiguous in du / u / idx): kT consecutive
    // f per tile whatever the row length -- a 129-column row (level 8 / STFT) filled half of a 256-column tile, the 8 193
    // columns of level 14 left a 33rd tile with one pixel.  tile t = (image n, tile tf of the image); t advances by S:
    // (n, tf) by (S / tilesX, S % tilesX) with a carry -- no division in the loop; tilesX = tiles per image here.
    const int dN = S / tilesX, dTf = S - dN * tilesX;
    int n = split / tilesX, tf = split - n * tilesX;
    for (int t = split; t < ntiles; t += S) {
        const unsigned f = (unsigned)tf * kT + threadIdx.x;
        if (f < (unsigned)plane) {
            const float* xn = x + (size_t)n * H * W;
            const int py = (int)(wp_magic ? __umulhi(f, wp_magic) : f / (unsigned)Wp);
            const int px = (int)f - py * Wp;
            float p[4][4];
            load_patch(xn, H, W, py, px, pad, p);
            const size_t o = ((size_t)n * Cout + cg * kCG) * plane + f;
            // all 3 * kCG loads of the pixel in flight before the first use (a load -> FMA chain
            // per channel pays one HBM latency per channel)
            int code[kCG];
            float gv[kCG], uv[kCG];
#pragma unroll
            for (int c = 0; c < kCG; ++c) {
                const bool ok = c < nch;
                const size_t oc = o + (size_t)c * plane;
                code[c] = ok ? idx[oc] : 0;
                gv[c] = ok ? du[oc] : 0.f;
                uv[c] = ok ? u[oc] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < kCG; ++c) {
                float g = fmaf(al[c], uv[c], gv[c] + be[c]);
                if (code[c] & 4) {  // winner was <= 0: the slope's gradient takes g * z = g * u / a, the gradient passes scaled by a
                    accs[c] += g * uv[c] * inva;
                    g *= a;
                }
                accb[c] += g;
                const int pos = code[c] & 3;
                const float g0 = pos == 0 ? g : 0.f, g1 = pos == 1 ? g : 0.f;
                const float g2 = pos == 2 ? g : 0.f, g3 = pos == 3 ? g : 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float sacc = acc[c][ky * 3 + kx];
                        sacc = fmaf(g0, p[ky][kx], sacc);
                        sacc = fmaf(g1, p[ky][kx + 1], sacc);
                        sacc = fmaf(g2, p[ky + 1][kx], sacc);
                        sacc = fmaf(g3, p[ky + 1][kx + 1], sacc);
                        acc[c][ky * 3 + kx] = sacc;
                    }
            }
        }
        tf += dTf;
        n += dN;
        if (tf >= tilesX) { tf -= tilesX; ++n; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < kCG; ++c) {
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            float v = k < 9 ? acc[c][k] : (k == 9 ? accb[c] : accs[c]);
            v = wave_sum63(v);
            if (lane == 63) red[wave][c * 11 + k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < kCG * 11) {
        float v = 0.f;
        for (int wv = 0; wv < kT / 64; ++wv) v += red[wv][threadIdx.x];
        partial[((size_t)split * gridDim.x + cg) * (kCG * 11) + threadIdx.x] = v;
    }
}

__global__ void conv1_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                        float* __restrict__ dbias, float* __restrict__ dslope, int Cout,
                                        int CG, int S) {
    // one thread per (cg, c, k)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = kCG * 11;
    float sl = 0.f;
    if (i < CG * per) {
        // eight loads in flight (the splits are a chain of 256 dependent L2 round trips otherwise: 43 us)
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = 0.f;
        const size_t pitch = (size_t)CG * per;
        int sp = 0;
        for (; sp + 8 <= S; sp += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += partial[(size_t)(sp + u) * pitch + i];
        }
        for (; sp < S; ++sp) a[0] += partial[(size_t)sp * pitch + i];
        const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        const int cg = i / per, r = i % per, c = r / 11, k = r % 11;
        const int co = cg * kCG + c;
        if (co < Cout) {
            if (k < 9) dw[co * 9 + k] = s;
            else if (k == 9) { if (dbias) dbias[co] = s; }
            else sl = s;
        }
    }
    // slope gradient: sum over channels
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sl += __shfl_down(sl, off, 64);
    if ((threadIdx.x & 63) == 0 && sl != 0.f) atomicAdd(dslope, sl);
}

int bwd_splits(long totalTiles, int CG) {
    long S = 2048 / CG;
    if (S > totalTiles) S = totalTiles;
    if (S < 1) S = 1;
    return (int)S;
}

}  // namespace

#define AFD_STREAM static_cast<hipStream_t>(stream)

extern "C" size_t afd_conv1_pool_workspace_bytes(int N, int H, int W, int Cout, int pad) {
    const int Hp = (H + 2 * pad - 2) / 2, Wp = (W + 2 * pad - 2) / 2;
    if (Hp < 1 || Wp < 1) return 0;
    const int CG = (Cout + kCG - 1) / kCG;
    const long tiles = (long)N * (((long)Hp * Wp + kT - 1) / kT);
    return (size_t)bwd_splits(tiles, CG) * CG * kCG * 11 * sizeof(float);
}

extern "C" size_t afd_conv1_pool_stats_workspace_bytes(int N, int H, int W, int Cout, int pad) {
    const int Hp = (H + 2 * pad - 2) / 2, Wp = (W + 2 * pad - 2) / 2;
    if (Hp < 1 || Wp < 1 || Cout < 1 || Cout > 128) return 0;
    return (size_t)N * Hp * ((Wp + kT * kFwdPixels - 1) / (kT * kFwdPixels)) * 2 * Cout * sizeof(float);
}

// Whether the sums are worth taking in the forward launch.  On a row narrower than one workgroup (level 8 / STFT: 129
// pooled columns, 33 live threads of 256) the per-channel wave sums cost more than the statistics pass they replace
// (conv_first 0.48 -> 0.64 ms against batchnorm 0.66 -> 0.61 ms at batch 128); at level 14 (8193 columns) the launch
// saves 0.5 ms of the step.  The kernel itself is correct at every width.
extern "C" int afd_conv1_pool_stats_applicable(int N, int H, int W, int Cout, int pad) {
    const int Wp = (W + 2 * pad - 2) / 2;
    return afd_conv1_pool_stats_workspace_bytes(N, H, W, Cout, pad) > 0 && Wp >= kT * kFwdPixels;
}

extern "C" int afd_conv1_pool_forward(const float* x, const float* w, const float* bias,
                                      const float* slope, float* u, uint8_t* idx, double* sums, void* stat_ws,
                                      size_t stat_ws_bytes, int N, int H, int W, int Cout, int pad,
                                      afd_stream_t stream) {
    if (!x || !w || !slope || !u || !idx) return afd::fail(AFD_ERR_ARG, "conv1 fwd: null pointer");
    const int Hp = (H + 2 * pad - 2) / 2, Wp = (W + 2 * pad - 2) / 2;
    if (N < 1 || Cout < 1 || Hp < 1 || Wp < 1 || pad < 0) return afd::fail(AFD_ERR_ARG, "conv1 fwd: bad geometry");
    if (Hp > 65535 || N > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1 fwd: grid too large");
    if (sums) {
        if (Cout > 128) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1 fwd + sums: Cout %d > 128", Cout);
        if (!stat_ws || stat_ws_bytes < afd_conv1_pool_stats_workspace_bytes(N, H, W, Cout, pad))
            return afd::fail(AFD_ERR_WORKSPACE, "conv1 fwd + sums: statistics workspace too small");
    }
    // bytes: the one-channel image in, pooled values + argmax codes out
    const double fwd_bytes = (double)N * (4.0 * H * W + 5.0 * Cout * Hp * Wp);
    afd::ScopedTiming timing(AFD_K_CONV_FIRST, fwd_bytes, AFD_STREAM);
    timing.bytes(fwd_bytes);
    if (!sums && Wp < 512 && (long)Hp * Wp < 0x7fffffffL) {
        // narrow rows: threads over the flattened pooled pixels of an image
        const dim3 fgrid((unsigned)(((long)Hp * Wp + kT - 1) / kT), 1, N);
        hipLaunchKernelGGL((conv1_pool_fwd_kernel<false, 1, true>), fgrid, dim3(kT), 0, AFD_STREAM, x, w, bias, slope, u, idx, H, W,
                           Cout, pad, Hp, Wp, nullptr);
        return afd::check_launch("conv1_pool_fwd_kernel(flat)");
    }
    constexpr int kFP = 4;
    const dim3 grid((Wp + kT * kFP - 1) / (kT * kFP), Hp, N);
    if (!sums) {
        hipLaunchKernelGGL(conv1_pool_fwd_kernel<false>, grid, dim3(kT), 0, AFD_STREAM, x, w, bias, slope, u, idx, H, W, Cout, pad,
                           Hp, Wp, nullptr);
        return afd::check_launch("conv1_pool_fwd_kernel");
    }
    float* part = static_cast<float*>(stat_ws);
    hipLaunchKernelGGL(conv1_pool_fwd_kernel<true>, grid, dim3(kT), 0, AFD_STREAM, x, w, bias, slope, u, idx, H, W, Cout, pad, Hp,
                       Wp, part);
    hipLaunchKernelGGL(conv1_stats_reduce_kernel, dim3(2 * Cout), dim3(256), 0, AFD_STREAM, part,
                       (int)(grid.x * grid.y * grid.z), 2 * Cout, sums);
    return afd::check_launch("conv1_pool_fwd_kernel(stats)");
}

extern "C" int afd_conv1_pool_backward(const float* x, const float* du, const uint8_t* idx,
                                       const float* u, const float* slope, float* dw, float* dbias,
                                       float* dslope, int N, int H, int W, int Cout, int pad, void* ws,
                                       size_t ws_bytes, afd_stream_t stream) {
    return afd_conv1_pool_backward_affine(x, du, idx, u, slope, nullptr, nullptr, dw, dbias, dslope, N, H, W,
                                          Cout, pad, ws, ws_bytes, stream);
}

extern "C" int afd_conv1_pool_backward_affine(const float* x, const float* du, const uint8_t* idx,
                                              const float* u, const float* slope, const float* alpha,
                                              const float* beta, float* dw, float* dbias, float* dslope,
                                              int N, int H, int W, int Cout, int pad, void* ws,
                                              size_t ws_bytes, afd_stream_t stream) {
    if ((alpha == nullptr) != (beta == nullptr)) return afd::fail(AFD_ERR_ARG, "conv1 bwd: alpha and beta come together");
    if (!x || !du || !idx || !u || !slope || !dw || !dslope) return afd::fail(AFD_ERR_ARG, "conv1 bwd: null pointer");
    const int Hp = (H + 2 * pad - 2) / 2, Wp = (W + 2 * pad - 2) / 2;
    if (N < 1 || Cout < 1 || Hp < 1 || Wp < 1) return afd::fail(AFD_ERR_ARG, "conv1 bwd: bad geometry");
    const int CG = (Cout + kCG - 1) / kCG;
    const int tilesX = (int)(((long)Hp * Wp + kT - 1) / kT);  // tiles of kT consecutive pooled pixels per image
    const long tiles = (long)N * tilesX;
    if (tiles > 0x7fffffffL || (long)Hp * Wp > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1 bwd: too many tiles");
    // row of a flat pixel index by multiply-high: exact while f * (magic * Wp - 2^32) < 2^32 for every f < Hp * Wp
    unsigned wp_magic = (unsigned)((0x100000000ULL + (unsigned)Wp - 1) / (unsigned)Wp);
    if (((unsigned long long)wp_magic * Wp - 0x100000000ULL) * (unsigned long long)((long)Hp * Wp) >= 0x100000000ULL) wp_magic = 0;
    const int S = bwd_splits(tiles, CG);
    if (!ws || ws_bytes < (size_t)S * CG * kCG * 11 * sizeof(float))
        return afd::fail(AFD_ERR_WORKSPACE, "conv1 bwd: workspace too small");
    float* partial = static_cast<float*>(ws);
    // bytes: the image, the pooled gradient, the pooled values and the codes in (the affine form's residual IS the
    // pooled value tensor it reads anyway); nothing but the small gradient tensors out
    const double bwd_bytes = (double)N * (4.0 * H * W + 9.0 * Cout * Hp * Wp);
    afd::ScopedTiming timing(AFD_K_CONV_FIRST, bwd_bytes, AFD_STREAM);
    timing.bytes(bwd_bytes);
    hipLaunchKernelGGL(conv1_pool_bwd_kernel, dim3(CG, S), dim3(kT), 0, AFD_STREAM, x, du, idx, u, slope,
                       partial, N, H, W, Cout, pad, Hp, Wp, tilesX, tiles, alpha, beta,
                       (CG == 8 && S % 8 == 0) ? 1 : 0, wp_magic);
    const int total = CG * kCG * 11;
    hipLaunchKernelGGL(conv1_bwd_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, AFD_STREAM, partial,
                       dw, dbias, dslope, Cout, CG, S);
    return afd::check_launch("conv1_pool_bwd kernels");
}
