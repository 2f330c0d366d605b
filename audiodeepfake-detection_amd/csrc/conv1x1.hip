// 1x1 convolutions (pad 0) as plain GEMMs over the flattened pixels of an image.
//
// Reference models.py:262 (DCNN block 2: Conv2d(64, 64, 1) on [B, 64, 13, 8193] at level 14) and
// the network-in-network layers of LCNN (models.py:88-107).  On the implicit-GEMM path a 1x1
// layer pays the patch staging and two barriers per 8-16 channels of K; here
//   forward / backward-data: the weights live in LDS for the whole (persistent) workgroup, each
//       wave streams its B fragments -- 32 consecutive pixels of two input channels -- straight
//       from global memory into the MFMA (no LDS, no barriers after the first), keeps all
//       output channels of its 32*NW pixels in accumulators and stores rows of 128 bytes;
//   backward-weight: dw[co][ci] = sum_p dy[co][p] x[ci][p].  64-pixel tiles of dy and x are
//       staged into a double-buffered LDS image ([row][pixel], pitch 65: conflict-free A and B
//       fragment reads) while the previous tile is in the MFMAs; one partial slab per
//       workgroup, summed by a second kernel (deterministic).
// Bound: HBM for 64 -> 64 channels (256 B/pixel read+written against 8192 MACs), MFMA above.
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct G1 {
    int N, Cin, Cout;  // of THIS GEMM: out[n][Cout][HW] = W[Cout][Cin] . in[n][Cin][HW]
    int HW;
    int tiles_per_img, ntiles;
    int trans;  // 0: W[m][k] = w[m * Cin + k] (forward); 1: W[m][k] = w[k * Cout + m] (dgrad)
    int xcd_m;  // workgroups per XCD group (afd::xcd_grouped_id): neighbouring tiles share the lines their rows end in
};

// NW consecutive floats with 4-byte alignment (image rows of 13 x 8193 floats start anywhere):
// one global_load / global_store of NW dwords per lane
template <int NW>
struct VecU;
template <>
struct VecU<2> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };
template <>
struct VecU<4> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); };

// MW = output-channel tiles of 32 (all of them in one wave); a wave owns 32*NW consecutive
// pixels.  Lane l of the B fragment holds pixels NW*l .. NW*l+NW-1 (one wide load per channel
// row): MFMA pixel tile i is the pixels NW*l + i, so the NW accumulator tiles of a lane are
// NW consecutive pixels again and leave as one wide store per output channel.
// AFF: the epilogue adds a per-output-channel affine function of a second tensor of the output's
// shape, y = W.x + alpha[c] * res + beta[c] (the BatchNorm backward folded into the backward-data
// GEMM of the 1x1 convolution that follows the normalisation, see afd_conv1x1_bn_backward_data).
// STATS: the epilogue also accumulates, per output channel, the sum and the sum of squares of PReLU(y) (slope
// `res[0]`) -- the batch statistics of the BatchNorm that follows the activation -- into one partial row per
// wave (`stat_part` [wave][2][MW*32]); a second kernel adds the rows in double precision.
template <int MW, int NW, bool AFF, bool STATS = false>
__global__ void __launch_bounds__(256)
conv1x1_kernel(const G1 g, const float* __restrict__ x, const float* __restrict__ w,
               const float* __restrict__ bias, float* __restrict__ y, const float* __restrict__ res,
               const float* __restrict__ alpha, float* __restrict__ stat_part = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [Kpad][MW*32], k-major
    typedef typename VecU<NW>::type vec_t;
    constexpr int CO_PAD = MW * 32;
    constexpr int KC = 8;  // k-steps (of two channels) per register chunk
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int Kpad = (g.Cin + 31) / 32 * 32;
    for (int e = tid; e < Kpad * CO_PAD; e += 256) {
        const int k = e / CO_PAD, m = e - k * CO_PAD;
        float v = 0.f;
        if (k < g.Cin && m < g.Cout) v = g.trans ? w[(size_t)k * g.Cout + m] : w[(size_t)m * g.Cin + k];
        Ws[e] = v;
    }
    __syncthreads();
    const int nchunks = Kpad / (2 * KC);
    const size_t HW = (size_t)g.HW;
    const int tstride = gridDim.x * 4;

    // chunk c of tile t: KC wide loads per lane (two channel rows per k-step, lanes 32.. take
    // the odd one)
    auto load_chunk = [&](int t, int c, vec_t (&b)[KC]) {
        const int n = t / g.tiles_per_img;
        const int p = (t - n * g.tiles_per_img) * (32 * NW) + NW * l31;  // first pixel of the lane
        const float* xn = x + (size_t)n * g.Cin * HW;
        const bool full = p + NW <= g.HW;
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
            // uniform base + 32-bit lane offset (an image is far below 4 GB)
            const int k = (c * KC + ks) * 2 + half;
            const unsigned off = (unsigned)k * (unsigned)g.HW + (unsigned)p;
            vec_t v;
#pragma unroll
            for (int i = 0; i < NW; ++i) v[i] = 0.f;
            if (k < g.Cin) {
                if (full) {
                    v = *reinterpret_cast<const vec_t*>(xn + off);
                } else {
#pragma unroll
                    for (int i = 0; i < NW; ++i)
                        if (p + i < g.HW) v[i] = xn[off + i];
                }
            }
            b[ks] = v;
        }
    };

    int t = afd::xcd_grouped_id((int)blockIdx.x, (int)gridDim.x, g.xcd_m) * 4 + wave;
    float s1[STATS ? MW : 1][16], s2[STATS ? MW : 1][16];
    float slope_a = 0.f;
    if constexpr (STATS) {
        slope_a = res[0];
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) s1[m][r] = s2[m][r] = 0.f;
        if (t >= g.ntiles) {  // a wave without tiles still owns a (zero) partial row
            for (int e = lane; e < 2 * CO_PAD; e += 64)
                stat_part[((size_t)blockIdx.x * 4 + wave) * (2 * CO_PAD) + e] = 0.f;
        }
    }
    if (t >= g.ntiles) return;
    f32x16 acc[MW][NW];
    // Tile loop outside, chunk loop inside: the accumulators are carried (as whole 16-register tuples) through the
    // chunk loop only and are fresh per tile -- carried through ONE loop together with the epilogue's element-wise
    // reads they lived as single registers and were copied into tuples and back around every chunk (256
    // v_accvgpr moves per 64 matrix instructions).  Chunks in pairs on two buffers (the count is even: Kpad % 32 == 0):
    // the next chunk -- of this tile or of the wave's next tile -- is in flight during the MFMAs of the present one.
    vec_t b0[KC], b1[KC];
    auto mma = [&](int c, const vec_t (&b)[KC]) {
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
            const float* arow = Ws + ((c * KC + ks) * 2 + half) * CO_PAD + l31;
            float a[MW];
#pragma unroll
            for (int m = 0; m < MW; ++m) a[m] = arow[m * 32];
#pragma unroll
            for (int m = 0; m < MW; ++m)
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    acc[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[ks][i], acc[m][i], 0, 0, 0);
        }
    };
    load_chunk(t, 0, b0);
    while (true) {
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int i = 0; i < NW; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][i][r] = 0.f;
        const int tn = t + tstride;
        const bool more = tn < g.ntiles;
        for (int c = 0; c < nchunks; c += 2) {
            load_chunk(t, c + 1, b1);
            mma(c, b0);
            if (c + 2 < nchunks) load_chunk(t, c + 2, b0);
            else if (more) load_chunk(tn, 0, b0);
            mma(c + 1, b1);
        }
        {

            // D layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
            const int n = t / g.tiles_per_img;
            const int p = (t - n * g.tiles_per_img) * (32 * NW) + NW * l31;
            const bool full = p + NW <= g.HW;
            if (p < g.HW) {
                float* yn = y + (size_t)n * g.Cout * HW + p;
                if constexpr (AFF) {
                    const float* rn = res + (size_t)n * g.Cout * HW + p;
#pragma unroll
                    for (int m = 0; m < MW; ++m) {
#pragma unroll
                        for (int r0 = 0; r0 < 16; r0 += 8) {
                            // eight rows of the second operand in flight, then their FMAs and stores
                            vec_t u[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int co = m * 32 + ((r0 + j) & 3) + 8 * ((r0 + j) >> 2) + 4 * half;
#pragma unroll
                                for (int i = 0; i < NW; ++i) u[j][i] = 0.f;
                                if (co < g.Cout) {
                                    const float* q = rn + (size_t)co * HW;
                                    if (full) {
                                        u[j] = *reinterpret_cast<const vec_t*>(q);
                                    } else {
#pragma unroll
                                        for (int i = 0; i < NW; ++i)
                                            if (p + i < g.HW) u[j][i] = q[i];
                                    }
                                }
                            }
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int r = r0 + j;
                                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                                if (co >= g.Cout) continue;
                                const float bv = bias[co], av = alpha[co];
                                float* o = yn + (size_t)co * HW;
                                if (full) {
                                    vec_t v;
#pragma unroll
                                    for (int i = 0; i < NW; ++i) v[i] = fmaf(av, u[j][i], acc[m][i][r] + bv);
                                    *reinterpret_cast<vec_t*>(o) = v;
                                } else {
#pragma unroll
                                    for (int i = 0; i < NW; ++i)
                                        if (p + i < g.HW) o[i] = fmaf(av, u[j][i], acc[m][i][r] + bv);
                                }
                            }
                        }
                    }
                } else
#pragma unroll
                for (int m = 0; m < MW; ++m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (co >= g.Cout) continue;
                        const float bv = bias ? bias[co] : 0.f;
                        float* o = yn + (size_t)co * HW;
                        if (full) {
                            vec_t v;
#pragma unroll
                            for (int i = 0; i < NW; ++i) v[i] = acc[m][i][r] + bv;
                            *reinterpret_cast<vec_t*>(o) = v;
                            if constexpr (STATS) {
#pragma unroll
                                for (int i = 0; i < NW; ++i) {
                                    const float pv = v[i] > 0.f ? v[i] : slope_a * v[i];
                                    s1[m][r] += pv;
                                    s2[m][r] = fmaf(pv, pv, s2[m][r]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < NW; ++i)
                                if (p + i < g.HW) {
                                    const float zv = acc[m][i][r] + bv;
                                    o[i] = zv;
                                    if constexpr (STATS) {
                                        const float pv = zv > 0.f ? zv : slope_a * zv;
                                        s1[m][r] += pv;
                                        s2[m][r] = fmaf(pv, pv, s2[m][r]);
                                    }
                                }
                        }
                    }
                }
            }
        }
        if (!more) break;
        t = tn;
    }
    if constexpr (STATS) {
        // a register row is one output channel over the 32 lanes of a wave half
        float* row = stat_part + ((size_t)blockIdx.x * 4 + wave) * (2 * CO_PAD);
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a1 = s1[m][r], a2 = s2[m][r];
#pragma unroll
                for (int off = 16; off >= 1; off >>= 1) {
                    a1 += __shfl_xor(a1, off, 64);
                    a2 += __shfl_xor(a2, off, 64);
                }
                if (l31 == 0) {
                    const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    row[co] = a1;
                    row[CO_PAD + co] = a2;
                }
            }
    }
}

// The 64 x 64 forward with statistics (DCNN block 2), stores under the matrix instructions (round 5).
// conv1x1_kernel<2, 4, false, true> writes a tile's 32 output rows (32 KB per wave) in one burst behind the tile's last
// matrix instruction; a CU drains ~6 bytes per cycle towards memory, so its four waves sat ~20k cycles in that burst
// with nothing else to issue -- as long as the tile's matrix instructions take (the store burst was 1.0 of the launch's
// 2.2 ms).  Here a tile is two passes over the SAME B fragments -- output channels 0..31, then 32..63 -- with the whole
// tile's 64 x 128 inputs resident in registers (128; 32 KB per wave in flight instead of 8-16): the rows of a finished
// pass leave one at a time, every second k-step, under the matrix instructions of the next pass (of this tile or the
// wave's next one), and a chunk of B is reloaded for the next tile as soon as the second pass is through with it.
// Arithmetic, summation order of the statistics and the partial-row layout are those of conv1x1_kernel (bit-identical y).
__global__ void __launch_bounds__(256)
conv1x1_split_stats_kernel(const G1 g, const float* __restrict__ x, const float* __restrict__ w,
                           const float* __restrict__ bias, float* __restrict__ y, const float* __restrict__ slope,
                           float* __restrict__ stat_part) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [64][64], k-major; then the 64 biases
    typedef typename VecU<4>::type vec_t;
    constexpr int C = 64, KC = 8, NCH = 4, NW = 4;  // exactly 64 channels on both sides (the host checks)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    for (int e = tid; e < C * C; e += 256) {
        const int k = e / C, m = e - k * C;
        Ws[e] = g.trans ? w[(size_t)k * C + m] : w[(size_t)m * C + k];
    }
    if (tid < C) Ws[C * C + tid] = bias ? bias[tid] : 0.f;
    __syncthreads();
    const float* Bs = Ws + C * C;
    const unsigned HW = (unsigned)g.HW;
    const int tstride = gridDim.x * 4;
    const float slope_a = slope[0];
    float s1[2][16], s2[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[m][r] = s2[m][r] = 0.f;
    int t = afd::xcd_grouped_id((int)blockIdx.x, (int)gridDim.x, g.xcd_m) * 4 + wave;  // uniform
    float* srow = stat_part + ((size_t)blockIdx.x * 4 + wave) * (2 * C);
    if (t >= g.ntiles) {  // a wave without tiles still owns a (zero) partial row
        for (int e = lane; e < 2 * C; e += 64) srow[e] = 0.f;
        return;
    }
    // a tile: image n, first pixel px0 (uniform); the lane's pixels px0 + 4 l31 .. + 3 of the channel rows 2 ks + half
    // (loads) and 4 half + (r & 3) + 8 (r >> 2) + 32 m (stores)
    struct Tile {
        const float* xin;
        float* yout;
        unsigned px0;
        bool whole;
    };
    auto tile_of = [&](int tt) {
        const int n = tt / g.tiles_per_img;
        Tile q;
        q.px0 = (unsigned)(tt - n * g.tiles_per_img) * (32u * NW);
        q.xin = x + (size_t)n * C * HW + q.px0;
        q.yout = y + (size_t)n * C * HW + q.px0;
        q.whole = q.px0 + 32u * NW <= HW;
        return q;
    };
    const unsigned lane_in = (unsigned)half * HW + 4u * (unsigned)l31;
    const unsigned lane_out = 4u * (unsigned)half * HW + 4u * (unsigned)l31;
    vec_t b[NCH][KC];
    auto load_chunk = [&](const Tile& q, int c) {
        if (q.whole) {  // uniform
#pragma unroll
            for (int ks = 0; ks < KC; ++ks)
                b[c][ks] = *reinterpret_cast<const vec_t*>(q.xin + ((unsigned)((c * KC + ks) * 2) * HW + lane_in));
        } else {
            const unsigned p = q.px0 + 4u * (unsigned)l31;
#pragma unroll
            for (int ks = 0; ks < KC; ++ks) {
                const float* src = q.xin + ((unsigned)((c * KC + ks) * 2) * HW + lane_in);
                vec_t v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    if (p + i < HW) v[i] = src[i];
                b[c][ks] = v;
            }
        }
    };
    // row r of a finished pass (block m of tile q): bias, store, statistics -- conv1x1_kernel's epilogue, row by row
    auto put_row = [&](const f32x16 (&a)[NW], int m, int r, const Tile& q) {
        const int cou = m * 32 + (r & 3) + 8 * (r >> 2);  // + 4 half
        const float bv = Bs[cou + 4 * half];
        float* o = q.yout + ((unsigned)cou * HW + lane_out);
        if (q.whole) {  // uniform
            vec_t v;
#pragma unroll
            for (int i = 0; i < NW; ++i) v[i] = a[i][r] + bv;
            *reinterpret_cast<vec_t*>(o) = v;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const float pv = v[i] > 0.f ? v[i] : slope_a * v[i];
                s1[m][r] += pv;
                s2[m][r] = fmaf(pv, pv, s2[m][r]);
            }
        } else {
            const unsigned p = q.px0 + 4u * (unsigned)l31;
#pragma unroll
            for (int i = 0; i < NW; ++i)
                if (p + i < HW) {
                    const float zv = a[i][r] + bv;
                    o[i] = zv;
                    const float pv = zv > 0.f ? zv : slope_a * zv;
                    s1[m][r] += pv;
                    s2[m][r] = fmaf(pv, pv, s2[m][r]);
                }
        }
    };
    f32x16 acc0[NW], acc1[NW];
    Tile cur = tile_of(t), prev = cur, nxt = cur;
    bool have_prev = false;
#pragma unroll
    for (int c = 0; c < NCH; ++c) load_chunk(cur, c);
    while (true) {
        const int tn = t + tstride;
        const bool more = tn < g.ntiles;  // uniform
        if (more) nxt = tile_of(tn);
        // pass 0: output channels 0..31 of this tile; the rows of the previous tile's channels 32..63 leave meanwhile
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[i][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int ks = 0; ks < KC; ++ks) {
                const float a = Ws[((c * KC + ks) * 2 + half) * C + l31];
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    acc0[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[c][ks][i], acc0[i], 0, 0, 0);
                if ((ks & 1) && have_prev) put_row(acc1, 1, (c * KC + ks) >> 1, prev);
                __builtin_amdgcn_sched_barrier(0);
            }
        // pass 1: channels 32..63; the rows of pass 0 leave meanwhile; a chunk of b is requested for the next tile as soon
        // as this pass is through with it
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int ks = 0; ks < KC; ++ks) {
                const float a = Ws[((c * KC + ks) * 2 + half) * C + 32 + l31];
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    acc1[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[c][ks][i], acc1[i], 0, 0, 0);
                if (ks & 1) put_row(acc0, 0, (c * KC + ks) >> 1, cur);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) load_chunk(nxt, c);
        }
        prev = cur;
        have_prev = true;
        if (!more) break;
        cur = nxt;
        t = tn;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) put_row(acc1, 1, r, prev);
    // a register row is one output channel over the 32 lanes of a wave half
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a1 = s1[m][r], a2 = s2[m][r];
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1) {
                a1 += __shfl_xor(a1, off, 64);
                a2 += __shfl_xor(a2, off, 64);
            }
            if (l31 == 0) {
                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                srow[co] = a1;
                srow[C + co] = a2;
            }
        }
}

// sums[c] = sum over the waves' partial rows (double), sums[C + c] likewise for the squares: one wave per output,
// lanes stride over the rows
__global__ void __launch_bounds__(256)
conv1x1_stats_reduce_kernel(const float* __restrict__ part, int rows, int co_pad, int C, double* __restrict__ sums) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= 2 * C) return;
    const int which = e / C, c = e - which * C;
    double s = 0.0;
    for (int r = lane; r < rows; r += 64) s += (double)part[(size_t)r * 2 * co_pad + which * co_pad + c];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) sums[e] = s;
}

// ---------------------------------------------------------------------------------------
// backward-weight
// ---------------------------------------------------------------------------------------
constexpr int kTP = 64;      // pixels per tile
constexpr int kPitch = 65;   // LDS row pitch (floats)

// COT / CIT = 32-channel tiles of dy / x.  4 waves share the COT*CIT output tiles.
template <int COT, int CIT>
__global__ void __launch_bounds__(256)
conv1x1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                     float* __restrict__ partial, int N, int Cin, int Cout, int HW,
                     int tiles_per_img, int ntiles) {
    constexpr int ROWS = (COT + CIT) * 32;  // dy rows first, then x rows
    constexpr int RPT = ROWS / 4;           // rows staged per thread (one pixel column each)
    constexpr int TILES = COT * CIT;
    constexpr int TPW = (TILES + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 x [ROWS][kPitch]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const size_t sHW = (size_t)HW;

    f32x16 acc[TPW];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float bsum[COT * 8];  // per-thread sums of the dy rows it stages (rows wave + 4u < COT*32)
#pragma unroll
    for (int u = 0; u < COT * 8; ++u) bsum[u] = 0.f;

    float st[RPT];
    auto load_tile = [&](int t) {
        const int n = t / tiles_per_img;
        const int p = (t - n * tiles_per_img) * kTP + lane;
        const bool pok = p < HW;
        const float* dyn = dy + (size_t)n * Cout * sHW + p;
        const float* xn = x + (size_t)n * Cin * sHW + p;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int row = wave + 4 * u;
            float v = 0.f;
            if (row < COT * 32) {
                if (pok && row < Cout) v = dyn[(size_t)row * sHW];
            } else {
                const int ci = row - COT * 32;
                if (pok && ci < Cin) v = xn[(size_t)ci * sHW];
            }
            st[u] = v;
        }
    };
    auto store_tile = [&](float* buf) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            buf[(wave + 4 * u) * kPitch + lane] = st[u];
            if (u < COT * 8) bsum[u] += st[u];
        }
    };

    int t = blockIdx.x;
    int cur = 0;
    if (t < ntiles) {
        load_tile(t);
        store_tile(lds);
    }
    __syncthreads();
    for (; t < ntiles; t += gridDim.x) {
        const int tn = t + gridDim.x;
        if (tn < ntiles) load_tile(tn);  // in flight while this tile is in the MFMAs
        const float* buf = lds + cur * ROWS * kPitch;
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int tile = wave + 4 * q;
            if (tile < TILES) {
                const int m = tile / CIT, j = tile - m * CIT;
                const float* ap = buf + (m * 32 + l31) * kPitch + half;
                const float* bp = buf + ((COT + j) * 32 + l31) * kPitch + half;
#pragma unroll 8
                for (int ks = 0; ks < kTP / 2; ++ks)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ks], bp[2 * ks], acc[q], 0, 0, 0);
            }
        }
        if (tn < ntiles) store_tile(lds + (cur ^ 1) * ROWS * kPitch);
        __syncthreads();
        cur ^= 1;
    }
    // slab: [COT*32][CIT*32] then [COT*32] bias sums
    constexpr int SLAB = COT * 32 * CIT * 32 + COT * 32;
    float* slab = partial + (size_t)blockIdx.x * SLAB;
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
        const int tile = wave + 4 * q;
        if (tile < TILES) {
            const int m = tile / CIT, j = tile - m * CIT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                slab[co * (CIT * 32) + j * 32 + l31] = acc[q][r];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < COT * 8; ++u) {
        float v = bsum[u];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
        if (lane == 0) slab[COT * 32 * CIT * 32 + wave + 4 * u] = v;
    }
}

// dw[co][ci] / dbias[co] = sum over slabs, fixed order
__global__ void __launch_bounds__(256)
conv1x1_reduce_kernel(const float* __restrict__ partial, int nslabs, int slab, int ci_pad, int co_pad,
                      int Cin, int Cout, float* __restrict__ dw, float* __restrict__ dbias) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= slab) return;
    float s = 0.f;
    for (int b = 0; b < nslabs; ++b) s += partial[(size_t)b * slab + e];
    if (e < co_pad * ci_pad) {
        const int co = e / ci_pad, ci = e - co * ci_pad;
        if (co < Cout && ci < Cin) dw[(size_t)co * Cin + ci] = s;
    } else if (dbias) {
        const int co = e - co_pad * ci_pad;
        if (co < Cout) dbias[co] = s;
    }
}

int g_cus = 0;
int num_cus() {
    if (!g_cus) {
        int dev = 0, c = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0)
            g_cus = c;
        else
            g_cus = 256;
    }
    return g_cus;
}

template <int MW, int NW>
int launch_gemm_stats(G1 g, const float* x, const float* w, const float* bias, const float* slope, float* y,
                      double* sums, float* part, size_t part_bytes, hipStream_t s) {
    g.tiles_per_img = (g.HW + 32 * NW - 1) / (32 * NW);
    const long nt = (long)g.N * g.tiles_per_img;
    if (nt > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1: too many tiles");
    g.ntiles = (int)nt;
    g.xcd_m = 16;  // measured level at 4 / 16 / 64 and against plain order (2.17-2.24 ms): the setting of the fused backward
    const int Kpad = (g.Cin + 31) / 32 * 32;
    const size_t lds = (size_t)Kpad * MW * 32 * sizeof(float);
    long blocks = (nt + 3) / 4;
    const long cap = (long)num_cus();  // the kernel holds one wave per SIMD: one persistent workgroup per CU
    if (blocks > cap) blocks = cap;
    if (part_bytes < (size_t)blocks * 4 * 2 * MW * 32 * sizeof(float))
        return afd::fail(AFD_ERR_WORKSPACE, "conv1x1 stats: workspace too small");
    if (MW == 2 && NW == 4 && g.Cin == 64 && g.Cout == 64 && (long)g.HW * 64 < 0x7fffffffL && !getenv("AFD_CONV1X1_ONE_PASS"))
        hipLaunchKernelGGL(conv1x1_split_stats_kernel, dim3((unsigned)blocks), dim3(256), lds + 64 * sizeof(float), s, g, x, w,
                           bias, y, slope, part);
    else
        hipLaunchKernelGGL((conv1x1_kernel<MW, NW, false, true>), dim3((unsigned)blocks), dim3(256), lds, s, g, x, w,
                           bias, y, slope, nullptr, part);
    int rc = afd::check_launch("conv1x1_kernel(stats)");
    if (rc) return rc;
    hipLaunchKernelGGL(conv1x1_stats_reduce_kernel, dim3((2 * g.Cout + 3) / 4), dim3(256), 0, s, part,
                       (int)blocks * 4, MW * 32, g.Cout, sums);
    return afd::check_launch("conv1x1_stats_reduce_kernel");
}

template <int MW, int NW>
int launch_gemm(G1 g, const float* x, const float* w, const float* bias, float* y, hipStream_t s,
                const float* res = nullptr, const float* alpha = nullptr) {
    g.tiles_per_img = (g.HW + 32 * NW - 1) / (32 * NW);
    const long nt = (long)g.N * g.tiles_per_img;
    if (nt > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1: too many tiles");
    g.ntiles = (int)nt;
    g.xcd_m = 16;
    const int Kpad = (g.Cin + 31) / 32 * 32;
    const size_t lds = (size_t)Kpad * MW * 32 * sizeof(float);
    long blocks = (nt + 3) / 4;
    const long cap = (long)num_cus() * 4;
    if (blocks > cap) blocks = cap;
    if (res)
        hipLaunchKernelGGL((conv1x1_kernel<MW, NW, true>), dim3((unsigned)blocks), dim3(256), lds, s, g, x, w,
                           bias, y, res, alpha);
    else
        hipLaunchKernelGGL((conv1x1_kernel<MW, NW, false>), dim3((unsigned)blocks), dim3(256), lds, s, g, x, w,
                           bias, y, res, alpha);
    return afd::check_launch("conv1x1_kernel");
}

int run_gemm(const G1& g, const float* x, const float* w, const float* bias, float* y, hipStream_t s,
             const float* res = nullptr, const float* alpha = nullptr) {
    switch ((g.Cout + 31) / 32) {
        case 1: return launch_gemm<1, 4>(g, x, w, bias, y, s, res, alpha);
        case 2: return launch_gemm<2, 4>(g, x, w, bias, y, s, res, alpha);
        case 3: return launch_gemm<3, 2>(g, x, w, bias, y, s, res, alpha);
        case 4: return launch_gemm<4, 2>(g, x, w, bias, y, s, res, alpha);
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1: Cout %d > 128", g.Cout);
}

constexpr int kWgBlocksPerCu = 2;

template <int COT, int CIT>
int launch_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int Cin, int Cout,
                 int HW, float* partial, hipStream_t s) {
    const int tiles_per_img = (HW + kTP - 1) / kTP;
    const long nt = (long)N * tiles_per_img;
    if (nt > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1 wgrad: too many tiles");
    constexpr int ROWS = (COT + CIT) * 32;
    const size_t lds = (size_t)2 * ROWS * kPitch * sizeof(float);
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_wgrad_kernel<COT, CIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "conv1x1 wgrad: %s", hipGetErrorString(e));
        attr.mark();
    }
    long blocks = (long)num_cus() * kWgBlocksPerCu;
    if (blocks > nt) blocks = nt;
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<COT, CIT>), dim3((unsigned)blocks), dim3(256), lds, s, x, dy,
                       partial, N, Cin, Cout, HW, tiles_per_img, (int)nt);
    int rc = afd::check_launch("conv1x1_wgrad_kernel");
    if (rc) return rc;
    const int slab = COT * 32 * CIT * 32 + COT * 32;
    hipLaunchKernelGGL(conv1x1_reduce_kernel, dim3((slab + 255) / 256), dim3(256), 0, s, partial,
                       (int)blocks, slab, CIT * 32, COT * 32, Cin, Cout, dw, dbias);
    return afd::check_launch("conv1x1_reduce_kernel");
}

}  // namespace

namespace afd {

bool conv1x1_applicable(int Cin, int Cout, int K, int pad, int dil) {
    if (getenv("AFD_NO_CONV1X1")) return false;
    return K == 1 && pad == 0 && Cin <= 128 && Cout <= 128;
}

// backward-weight is built for channel counts padded to 32, 64 or 128
bool conv1x1_wgrad_applicable(int Cin, int Cout) {
    const int a = (Cin + 31) / 32, b = (Cout + 31) / 32;
    return a != 3 && b != 3 && a <= 4 && b <= 4;
}

static double pad32(int c) { return (double)((c + 31) / 32 * 32); }

size_t conv1x1_workspace_bytes(int Cin, int Cout) {
    const size_t ci = (size_t)(Cin + 31) / 32 * 32, co = (size_t)(Cout + 31) / 32 * 32;
    return (size_t)num_cus() * kWgBlocksPerCu * (co * ci + co) * sizeof(float);
}

int conv1x1_forward(const float* x, const float* w, const float* bias, float* y, int N, int Cin,
                    int Cout, long HW, hipStream_t s) {
    G1 g{};
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.HW = (int)HW; g.trans = 0;
    afd::ScopedTiming timing(AFD_K_CONV1X1, 2.0 * N * Cout * (double)HW * Cin, s);
    timing.issued(2.0 * N * pad32(Cout) * (double)HW * Cin);
    timing.bytes(4.0 * N * (double)HW * (Cin + Cout));
    return run_gemm(g, x, w, bias, y, s);
}

// dx[n][ci][p] = sum_co w[co][ci] dy[n][co][p]
size_t conv1x1_stats_workspace_bytes(int Cout) {
    return (size_t)num_cus() * 4 * 2 * ((size_t)(Cout + 31) / 32 * 32) * sizeof(float);
}

// y = W x + bias, and sums[c] = sum PReLU(y[c]), sums[Cout + c] = sum PReLU(y[c])^2 over all pixels
int conv1x1_forward_stats(const float* x, const float* w, const float* bias, const float* slope, float* y,
                          double* sums, int N, int Cin, int Cout, long HW, void* ws, size_t ws_bytes,
                          hipStream_t s) {
    G1 g{};
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.HW = (int)HW; g.trans = 0;
    afd::ScopedTiming timing(AFD_K_CONV1X1, 2.0 * N * Cout * (double)HW * Cin, s);
    timing.issued(2.0 * N * pad32(Cout) * (double)HW * Cin);
    timing.bytes(4.0 * N * (double)HW * (Cin + Cout));
    float* part = static_cast<float*>(ws);
    switch ((Cout + 31) / 32) {
        case 1: return launch_gemm_stats<1, 4>(g, x, w, bias, slope, y, sums, part, ws_bytes, s);
        case 2: return launch_gemm_stats<2, 4>(g, x, w, bias, slope, y, sums, part, ws_bytes, s);
        case 3: return launch_gemm_stats<3, 2>(g, x, w, bias, slope, y, sums, part, ws_bytes, s);
        // 4 channel tiles: the statistics registers do not fit next to 128 accumulators (58 spills): not built
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1 stats: Cout %d > 96", Cout);
}

int conv1x1_backward_data(const float* dy, const float* w, float* dx, int N, int Cin, int Cout,
                          long HW, hipStream_t s) {
    G1 g{};
    g.N = N; g.Cin = Cout; g.Cout = Cin; g.HW = (int)HW; g.trans = 1;
    afd::ScopedTiming timing(AFD_K_CONV1X1, 2.0 * N * Cout * (double)HW * Cin, s);
    timing.issued(2.0 * N * pad32(Cin) * (double)HW * Cout);
    timing.bytes(4.0 * N * (double)HW * (Cin + Cout));
    return run_gemm(g, dy, w, nullptr, dx, s);
}

// dx[n][ci][p] = sum_co w[co][ci] dy[n][co][p] + alpha[ci] * res[n][ci][p] + beta[ci]
int conv1x1_backward_data_affine(const float* dy, const float* w, const float* res, const float* alpha,
                                 const float* beta, float* dx, int N, int Cin, int Cout, long HW,
                                 hipStream_t s) {
    G1 g{};
    g.N = N; g.Cin = Cout; g.Cout = Cin; g.HW = (int)HW; g.trans = 1;
    afd::ScopedTiming timing(AFD_K_CONV1X1, 2.0 * N * Cout * (double)HW * Cin, s);
    timing.issued(2.0 * N * pad32(Cin) * (double)HW * Cout);
    timing.bytes(4.0 * N * (double)HW * (2.0 * Cin + Cout));
    return run_gemm(g, dy, w, beta, dx, s, res, alpha);
}

int conv1x1_backward_weight(const float* x, const float* dy, float* dw, float* dbias, int N, int Cin,
                            int Cout, long HW, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!ws || ws_bytes < conv1x1_workspace_bytes(Cin, Cout))
        return afd::fail(AFD_ERR_WORKSPACE, "conv1x1 wgrad: workspace too small");
    float* partial = static_cast<float*>(ws);
    afd::ScopedTiming timing(AFD_K_CONV_WGRAD_1X1, 2.0 * N * Cout * (double)HW * Cin, s);
    timing.issued(2.0 * N * pad32(Cout) * (double)HW * pad32(Cin));
    timing.bytes(4.0 * N * (double)HW * (Cin + Cout));
    const int cot = (Cout + 31) / 32, cit = (Cin + 31) / 32;
    const int key = (cot == 3 ? 0 : cot) * 10 + (cit == 3 ? 0 : cit);
    switch (key) {
        case 11: return launch_wgrad<1, 1>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
        case 12: return launch_wgrad<1, 2>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
        case 14: return launch_wgrad<1, 4>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
        case 21: return launch_wgrad<2, 1>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
        case 22: return launch_wgrad<2, 2>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
        case 24: return launch_wgrad<2, 4>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
        case 41: return launch_wgrad<4, 1>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
        case 42: return launch_wgrad<4, 2>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
        case 44: return launch_wgrad<4, 4>(x, dy, dw, dbias, N, Cin, Cout, (int)HW, partial, s);
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1 wgrad: %d x %d channels not built", Cout, Cin);
}

}  // namespace afd
