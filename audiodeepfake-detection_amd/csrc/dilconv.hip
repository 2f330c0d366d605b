// Direct (VALU) convolutions for the DCNN's dilated stack when it has very few channels.
//
// Reference models.py:286-300: three Conv2d(time_dim, time_dim, k, padding, dilation) with
// (k, pad, dil) = (3,1,1), (5,2,2), (7,2,4) on [B, time_dim, 64, P/8].  For level-14 packets
// time_dim = 24 // 8 = 3: on the MFMA implicit-GEMM path 3 of 32 output-channel rows are live
// and the three layers cost 21 ms of a 178 ms step.  Here every lane owns one image column and
// ROWS output rows of one residue class modulo the dilation (rows y, y+d, .. share all but one
// of their input rows) and reads its input samples from an LDS tile (zero-filled halo, no bounds
// checks in the loop).
//
// Round 3: everything is packed arithmetic over PAIRS OF KERNEL COLUMNS (kx = 2q, 2q + 1): the two
// samples come from one ds_read2_b32 (offsets q*2*DIL and +DIL), the two weights are neighbours in
// memory, the two halves of a v_pk_fma_f32 accumulate the even-kx and the odd-kx partial sums, which
// are added at the end (odd K: the pair of the last column has a zero weight / a discarded sum).
//   forward / backward-data (weights transposed + flipped, pad' = (K-1)d - pad): accumulators
//       [ROWS][C] pairs; a real loop over (input channel, column pair) whose body holds the
//       K*C weight pairs of that step in scalar registers.  The first generation unrolled everything:
//       441 weights do not fit the scalar file, the compiler parked them in vector-register lanes and
//       every FMA came with a v_readlane (4 400 vector instructions per tile-wave; now 1 260).
//   backward-weight: one wave per kernel row ky, [C][C][K/2] pair accumulators per lane over a strided
//       tile list; the next tile's rows are requested before the current tile's arithmetic and land in
//       the other LDS image afterwards (one barrier per tile; the first generation fetched, waited and
//       computed in turn with a branch per load: 12 us per tile against 1 us of arithmetic); one partial
//       slab per workgroup, summed by a second kernel (deterministic, no float atomics).
// Staging is by rows: a wave takes whole tile rows, so row index, channel and bounds are scalar
// arithmetic, a lane only adds its column; out-of-image loads read a clamped address and are zeroed.
// Bound: fp32 VALU (441 FMA per 3 x 4-byte pixel at C = 3, K = 7); algorithmic flops as for
// the implicit-GEMM path: 2 N C C K K Hout Wout.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

struct DirGeom {
    int N, Hin, Win, Hout, Wout, pad;  // out = in + 2 pad - (K-1) dil
    int tilesX, groups;                // groups = row groups per residue class
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// Rows `first, first + NW, ..` (< NROWS) of a tile of NROWS = C * R rows x PC columns, row = ci * R + r at image
// row iy0 + r * DIL, columns ix0 .. ix0 + PC: all loads of the thread first (`v`), the LDS stores afterwards
// (`put`), so that the caller can put arithmetic in between.  first is wave-uniform.
template <int C, int R, int PC, int DIL, int NW>
struct RowStage {
    static constexpr int NROWS = C * R;
    static constexpr int RU = (NROWS + NW - 1) / NW;  // rows per wave
    static constexpr int CC = (PC + 63) / 64;         // 64-column chunks per row
    float v[RU * CC];

    __device__ __forceinline__ void get(const float* __restrict__ img, int H, int W, int iy0, int ix0, int first,
                                        int lane) {
        const int plane = H * W;  // < 2^31 / C: checked by the host
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = first + u * NW;
            const int ci = row / R, r = row - ci * R;
            const int iy = iy0 + r * DIL;
            const bool rok = row < NROWS && iy >= 0 && iy < H;
            const float* src = img + (rok ? ci * plane + iy * W : 0);
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const int ix = ix0 + c * 64 + lane;
                const bool ok = rok && ix >= 0 && ix < W;
                const float t = src[ok ? ix : 0];
                v[u * CC + c] = ok ? t : 0.f;
            }
        }
    }
    __device__ __forceinline__ void put(float* tile, int first, int lane) const {
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = first + u * NW;
            if (row < NROWS) {
#pragma unroll
                for (int c = 0; c < CC; ++c) {
                    const int col = c * 64 + lane;
                    if (col < PC) tile[row * PC + col] = v[u * CC + c];
                }
            }
        }
    }
};

template <int C, int K, int DIL, int ROWS, bool FLIP>
__global__ void __launch_bounds__(256)
dilconv_direct_kernel(const DirGeom g, const float* __restrict__ in, const float* __restrict__ w,
                      const float* __restrict__ bias, float* __restrict__ out) {
    constexpr int TX = 256;
    constexpr int R = ROWS + K - 1;
    constexpr int KP = (K + 1) / 2;                 // kernel-column pairs
    constexpr int PC = TX + (2 * KP - 1) * DIL;     // the pad column of an odd K is read too
    constexpr int KK = K * K;
    __shared__ float tile[C * R * PC];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * TX;
    const int cls = blockIdx.y % DIL, grp = blockIdx.y / DIL;
    const int y0 = cls + DIL * ROWS * grp;
    const int n = blockIdx.z;
    {
        const float* img = in + (size_t)n * C * g.Hin * g.Win;
        // two batches (the row list of a wave is split in halves) keep the registers of the loads in flight low
        RowStage<C, R, PC, DIL, 8> st;
        st.get(img, g.Hin, g.Win, y0 - g.pad, x0 - g.pad, wv, lane);
        st.put(tile, wv, lane);
        st.get(img, g.Hin, g.Win, y0 - g.pad, x0 - g.pad, wv + 4, lane);
        st.put(tile, wv + 4, lane);
    }
    __syncthreads();
    f2 acc[ROWS][C];
#pragma unroll
    for (int j = 0; j < ROWS; ++j)
#pragma unroll
        for (int o = 0; o < C; ++o) acc[j][o] = f2{bias ? bias[o] : 0.f, 0.f};
#pragma unroll 1
    for (int it = 0; it < C * KP; ++it) {
        const int i = it / KP, q = it - i * KP;
        const bool odd_ok = 2 * q + 1 < K;  // uniform
        const float* col = tile + i * R * PC + tid + 2 * q * DIL;
        f2 xp[R];
#pragma unroll
        for (int r = 0; r < R; ++r) xp[r] = f2{col[r * PC], col[r * PC + DIL]};
#pragma unroll
        for (int o = 0; o < C; ++o) {
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                // uniform addresses: scalar loads; the pad column's weight is read from the pair's first address
                const int e0 = FLIP ? (i * C + o) * KK + (K - 1 - ky) * K + (K - 1 - 2 * q)
                                    : (o * C + i) * KK + ky * K + 2 * q;
                const int e1 = odd_ok ? (FLIP ? e0 - 1 : e0 + 1) : e0;
                const float w0 = w[e0];
                const float w1 = odd_ok ? w[e1] : 0.f;
                const f2 ww = {w0, w1};
#pragma unroll
                for (int j = 0; j < ROWS; ++j) acc[j][o] = __builtin_elementwise_fma(xp[ky + j], ww, acc[j][o]);
            }
        }
    }
    const int x = x0 + tid;
    if (x >= g.Wout) return;
    const size_t plane = (size_t)g.Hout * g.Wout;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int y = y0 + j * DIL;
        if (y < g.Hout) {
#pragma unroll
            for (int o = 0; o < C; ++o)
                out[((size_t)n * C + o) * plane + (size_t)y * g.Wout + x] = acc[j][o].x + acc[j][o].y;
        }
    }
}

// One image row segment global -> LDS without passing registers (buffer_load_dword ... lds): lane l's dword lands at
// lds_addr + 4 l; `bytes` is the row's extent from `base` -- lanes whose byte offset `voff` falls outside it (negative
// offsets are huge unsigned ones) write ZERO (tools/micro/buf_lds.hip checks exactly this on the hardware), so the
// zero halo costs no vector instruction; bytes = 0 clears the row.  Inline assembly because the compiler, which cannot
// tell LDS-DMA destinations from other LDS data, would otherwise wait for ALL loads in flight before every LDS read.
// (m0 is a reserved register to the compiler: it writes it immediately before each of its own uses and keeps nothing
// in it across statements, so the block may overwrite it -- naming it as a clobber only draws a warning.)
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void row_to_lds(const float* base, int bytes, unsigned lds_addr, int voff) {
    const unsigned long long b = (unsigned long long)base;
    const i4 r = {(int)(unsigned)b, (int)((b >> 32) & 0xffffu), bytes, 0x00020000};
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(r)
                 : "memory");
}
// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

constexpr size_t wgrad_lds_bytes(int C, int K, int DIL, int ROWS) {
    const int R = ROWS + K - 1, KP = (K + 1) / 2, PC = 64 + (2 * KP - 1) * DIL;
    const int XS = (C * R + K - 1) / K, DS = (C * ROWS + K - 1) / K;
    return (size_t)3 * (XS * K * PC + DS * K * 64) * sizeof(float);
}

// partial[block][C*C*K*K + C]: weight gradient sums of this block's tiles, then the bias sums.
// One wave per kernel row ky.  Three LDS images: while tile t is consumed, the rows of tiles t+1 and t+2 are in
// flight (row_to_lds; every wave issues the same number of loads per tile -- surplus row slots and tiles past the end
// are loads from an empty descriptor -- so "tile t has landed" is s_waitcnt vmcnt(loads per tile)).
template <int C, int K, int DIL, int ROWS, int ABL = 0>  // ABL: ablation builds (1 no loads, 2 no arithmetic), never launched
__global__ void __launch_bounds__(64 * K)
dilconv_wgrad_kernel(const DirGeom g, const float* __restrict__ x, const float* __restrict__ dy,
                     float* __restrict__ partial, int ntiles) {
    constexpr int TX = 64;
    constexpr int R = ROWS + K - 1;
    constexpr int KP = (K + 1) / 2;
    constexpr int PC = TX + (2 * KP - 1) * DIL;
    constexpr int CCX = (PC + 63) / 64;        // 64-column chunks of an x row (the last one with its surplus lanes off)
    constexpr int PX = PC;                     // LDS pitch of an x row
    constexpr int XS = (C * R + K - 1) / K;    // x row slots per wave
    constexpr int DS = (C * ROWS + K - 1) / K; // dy row slots per wave
    constexpr int L = XS * CCX + DS;           // loads per tile and wave
    constexpr int D = 3;
    static_assert(L <= 31, "two tiles of loads in flight must fit the 6-bit counter");
    constexpr int XSZ = XS * K * PX, DSZ = DS * K * TX;
    static_assert(wgrad_lds_bytes(C, K, DIL, ROWS) == (size_t)D * (XSZ + DSZ) * sizeof(float), "host and kernel disagree");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [D] x images, then [D] dy images
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int ky = __builtin_amdgcn_readfirstlane(tid >> 6);
    f2 acc[C][C][KP];
    float accb[C];
#pragma unroll
    for (int o = 0; o < C; ++o) {
        accb[o] = 0.f;
#pragma unroll
        for (int i = 0; i < C; ++i)
#pragma unroll
            for (int q = 0; q < KP; ++q) acc[o][i][q] = f2{0.f, 0.f};
    }
    const int rgs = DIL * g.groups;
    auto issue = [&](int t, int b) {
        if (ABL == 1) return;
        const bool live = t < ntiles;  // uniform
        const int tt = live ? t : 0;
        const int xtile = tt % g.tilesX;
        const int rest = tt / g.tilesX;
        const int rg = rest % rgs, n = rest / rgs;
        const int cls = rg % DIL, grp = rg / DIL;
        const int y0 = cls + DIL * ROWS * grp;
        const int x0 = xtile * TX;
        const float* xn = x + (size_t)n * C * g.Hin * g.Win;
        const float* dn = dy + (size_t)n * C * g.Hout * g.Wout;
        const unsigned xb = (unsigned)(size_t)(smem + b * XSZ), db = (unsigned)(size_t)(smem + D * XSZ + b * DSZ);
#pragma unroll
        for (int u = 0; u < XS; ++u) {
            const int row = ky + u * K;
            const int ci = row / R, r = row - ci * R;
            const int iy = y0 - g.pad + r * DIL;
            const bool ok = live && row < C * R && iy >= 0 && iy < g.Hin;
            const float* src = xn + (ok ? (ci * g.Hin + iy) * g.Win : 0);
#pragma unroll
            for (int c = 0; c < CCX; ++c)
                if (c * 64 + lane < PC)
                    row_to_lds(src, ok ? g.Win * 4 : 0, xb + (row * PX + c * 64) * 4, (x0 - g.pad + c * 64 + lane) * 4);
        }
#pragma unroll
        for (int u = 0; u < DS; ++u) {
            const int row = ky + u * K;  // o * ROWS + j
            const int o = row / ROWS, j = row - o * ROWS;
            const int iy = y0 + j * DIL;
            const bool ok = live && row < C * ROWS && iy < g.Hout;
            const float* src = dn + (ok ? (o * g.Hout + iy) * g.Wout : 0);
            row_to_lds(src, ok ? g.Wout * 4 : 0, db + row * TX * 4, (x0 + lane) * 4);
        }
    };
    const int t0 = blockIdx.x, step = gridDim.x;
    issue(t0, 0);
    issue(t0 + step, 1);
    int b = 0;
    for (int t = t0; t < ntiles; t += step) {
        if (ABL != 1) wait_vm<L>();     // this wave's rows of tile t are in LDS (tile t + 1 may still be in flight)
        __syncthreads();  // everybody's are; and everybody is done with the image of tile t - 1
        issue(t + 2 * step, b == 0 ? 2 : b - 1);
        const float* xb = smem + b * XSZ;
        const float* db = smem + D * XSZ + b * DSZ;
        if (ABL == 2) {
            accb[0] += db[lane] + xb[lane];
            b = b == 2 ? 0 : b + 1;
            continue;
        }
        // Row j + 1's operands are requested before row j's arithmetic (two register sets; the fences keep the
        // scheduler from sinking the reads to their uses -- left alone it issued read, wait, three FMAs in turn and
        // the LDS latency of every one of the 45 - 60 reads of a tile showed: 6 000 cycles per tile against 1 000)
        float d[2][C];
        f2 v[2][C][KP];
        auto request = [&](int j, int s) {
#pragma unroll
            for (int o = 0; o < C; ++o) d[s][o] = db[(o * ROWS + j) * TX + lane];
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const float* row = xb + (i * R + j + ky) * PX + lane;
#pragma unroll
                for (int q = 0; q < KP; ++q) v[s][i][q] = f2{row[2 * q * DIL], row[(2 * q + 1) * DIL]};
            }
        };
        request(0, 0);
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int s = j & 1;
            if (j + 1 < ROWS) request(j + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            if (ky == 0) {
#pragma unroll
                for (int o = 0; o < C; ++o) accb[o] += d[s][o];
            }
#pragma unroll
            for (int i = 0; i < C; ++i)
#pragma unroll
                for (int q = 0; q < KP; ++q)
#pragma unroll
                    for (int o = 0; o < C; ++o)
                        acc[o][i][q] = __builtin_elementwise_fma(f2{d[s][o], d[s][o]}, v[s][i][q], acc[o][i][q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        b = b == 2 ? 0 : b + 1;
    }
    wait_vm<0>();  // no load may land in LDS after the workgroup has gone
    // wave sums -> lane 0 -> this block's slab
    constexpr int KK = K * K;
    float* slab = partial + (size_t)blockIdx.x * (C * C * KK + C);
#pragma unroll
    for (int o = 0; o < C; ++o) {
#pragma unroll
        for (int i = 0; i < C; ++i) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                float v = (kx & 1) ? acc[o][i][kx / 2].y : acc[o][i][kx / 2].x;
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                if (lane == 0) slab[(o * C + i) * KK + ky * K + kx] = v;
            }
        }
    }
    if (ky == 0) {
#pragma unroll
        for (int o = 0; o < C; ++o) {
            float v = accb[o];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) slab[C * C * KK + o] = v;
        }
    }
}

// dw[e] = sum over slabs (fixed order); e >= n_w are the bias sums
__global__ void __launch_bounds__(256)
dilconv_reduce_kernel(const float* __restrict__ partial, int nslabs, int n_w, int stride,
                      float* __restrict__ dw, float* __restrict__ dbias) {
    __shared__ float red[256];
    const int e = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nslabs; b += 256) s += partial[(size_t)b * stride + e];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (e < n_w) dw[e] = red[0];
        else if (dbias) dbias[e - n_w] = red[0];
    }
}

// rows per thread: 5 when that wastes no more of the last row group than 4
int pick_rows(int Hout, int dil) {
    const int per = (Hout + dil - 1) / dil;
    const int waste4 = (per + 3) / 4 * 4 - per, waste5 = (per + 4) / 5 * 5 - per;
    return waste5 <= waste4 ? 5 : 4;
}

template <int C, int K, int DIL, int ROWS, bool FLIP>
void launch_direct(const DirGeom& g, const float* in, const float* w, const float* bias, float* out,
                   hipStream_t s) {
    hipLaunchKernelGGL((dilconv_direct_kernel<C, K, DIL, ROWS, FLIP>),
                       dim3(g.tilesX, DIL * g.groups, g.N), dim3(256), 0, s, g, in, w, bias, out);
}

template <int C, int K, int DIL, bool FLIP>
int run_direct(DirGeom g, const float* in, const float* w, const float* bias, float* out,
               hipStream_t s) {
    const int rows = pick_rows(g.Hout, DIL);
    const int per = (g.Hout + DIL - 1) / DIL;
    g.tilesX = (g.Wout + 255) / 256;
    g.groups = (per + rows - 1) / rows;
    if ((long)DIL * g.groups > 65535 || g.N > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: grid too large");
    if (rows == 5) launch_direct<C, K, DIL, 5, FLIP>(g, in, w, bias, out, s);
    else launch_direct<C, K, DIL, 4, FLIP>(g, in, w, bias, out, s);
    return afd::check_launch("dilconv_direct_kernel");
}

template <int C, int K, int DIL, int ROWS>
int launch_wgrad(const DirGeom& g, const float* x, const float* dy, float* partial, int ntiles, int blocks,
                 hipStream_t s) {
    constexpr size_t lds = wgrad_lds_bytes(C, K, DIL, ROWS);
    static const hipError_t attr = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&dilconv_wgrad_kernel<C, K, DIL, ROWS>), hipFuncAttributeMaxDynamicSharedMemorySize,
        (int)lds);
    if (attr != hipSuccess) return afd::fail(AFD_ERR_HIP, "dilconv wgrad: %zu bytes of LDS refused", lds);
    hipLaunchKernelGGL((dilconv_wgrad_kernel<C, K, DIL, ROWS>), dim3(blocks), dim3(64 * K), lds, s, g, x, dy, partial,
                       ntiles);
    return 0;
}

constexpr int kWgradBlocks = 1021;  // prime: a count sharing a factor with the tiles per row walks tile columns in lock step

template <int C, int K, int DIL>
int run_wgrad(DirGeom g, const float* x, const float* dy, float* dw, float* dbias, float* partial,
              hipStream_t s) {
    const int rows = pick_rows(g.Hout, DIL);
    const int per = (g.Hout + DIL - 1) / DIL;
    g.tilesX = (g.Wout + 63) / 64;
    g.groups = (per + rows - 1) / rows;
    const long ntiles = (long)g.N * DIL * g.groups * g.tilesX;
    if (ntiles > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv wgrad: too many tiles");
    const int blocks = ntiles < kWgradBlocks ? (int)ntiles : kWgradBlocks;
    if (rows == 5) {
        if (int rc = launch_wgrad<C, K, DIL, 5>(g, x, dy, partial, (int)ntiles, blocks, s)) return rc;
    } else {
        if (int rc = launch_wgrad<C, K, DIL, 4>(g, x, dy, partial, (int)ntiles, blocks, s)) return rc;
    }
    int rc = afd::check_launch("dilconv_wgrad_kernel");
    if (rc) return rc;
    const int n_w = C * C * K * K;
    hipLaunchKernelGGL(dilconv_reduce_kernel, dim3(n_w + C), dim3(256), 0, s, partial, blocks, n_w,
                       n_w + C, dw, dbias);
    return afd::check_launch("dilconv_reduce_kernel");
}

// dispatch over the (K, dil) pairs of the reference's dilated stack and C = 1..4
template <int C>
int dispatch(int mode, int K, int dil, const DirGeom& g, const float* a, const float* b,
             const float* bias, float* out, float* dbias, float* partial, hipStream_t s) {
#define AFD_DIL_CASE(KK, DD)                                                      \
    if (K == KK && dil == DD) {                                                   \
        if (mode == 0) return run_direct<C, KK, DD, false>(g, a, b, bias, out, s); \
        if (mode == 1) return run_direct<C, KK, DD, true>(g, a, b, nullptr, out, s); \
        return run_wgrad<C, KK, DD>(g, a, b, out, dbias, partial, s);             \
    }
    AFD_DIL_CASE(3, 1)
    AFD_DIL_CASE(5, 2)
    AFD_DIL_CASE(7, 4)
#undef AFD_DIL_CASE
    return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: K=%d dil=%d not built", K, dil);
}

int dispatch_c(int C, int mode, int K, int dil, const DirGeom& g, const float* a, const float* b,
               const float* bias, float* out, float* dbias, float* partial, hipStream_t s) {
    switch (C) {
        case 1: return dispatch<1>(mode, K, dil, g, a, b, bias, out, dbias, partial, s);
        case 2: return dispatch<2>(mode, K, dil, g, a, b, bias, out, dbias, partial, s);
        case 3: return dispatch<3>(mode, K, dil, g, a, b, bias, out, dbias, partial, s);
        case 4: return dispatch<4>(mode, K, dil, g, a, b, bias, out, dbias, partial, s);
    }
    return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: C=%d not built", C);
}

}  // namespace

namespace afd {

bool dilconv_applicable(int Cin, int Cout, int K, int dil) {
    if (getenv("AFD_NO_DIRECT_CONV")) return false;
    if (Cin != Cout || Cin < 1 || Cin > 4) return false;
    return (K == 3 && dil == 1) || (K == 5 && dil == 2) || (K == 7 && dil == 4);
}

size_t dilconv_workspace_bytes(int C, int K) {
    return (size_t)kWgradBlocks * (C * C * K * K + C) * sizeof(float);
}

// the kernels index one image with 32-bit offsets
static bool image_fits(int C, int H, int W) { return (long)C * H * W < (1L << 31); }

int dilconv_forward(const float* x, const float* w, const float* bias, float* y, int N, int C, int H,
                    int W, int K, int pad, int dil, hipStream_t s) {
    if (!image_fits(C, H, W)) return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: image too large");
    DirGeom g{};
    g.N = N; g.Hin = H; g.Win = W; g.pad = pad;
    g.Hout = H + 2 * pad - dil * (K - 1);
    g.Wout = W + 2 * pad - dil * (K - 1);
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, 2.0 * N * C * C * K * K * (double)g.Hout * g.Wout, s);
    timing.bytes(4.0 * N * C * ((double)H * W + (double)g.Hout * g.Wout));  // x once, y once
    return dispatch_c(C, 0, K, dil, g, x, w, bias, y, nullptr, nullptr, s);
}

int dilconv_backward_data(const float* dy, const float* w, float* dx, int N, int C, int H, int W,
                          int K, int pad, int dil, hipStream_t s) {
    if (!image_fits(C, H, W)) return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: image too large");
    DirGeom g{};
    g.N = N;
    g.Hin = H + 2 * pad - dil * (K - 1);
    g.Win = W + 2 * pad - dil * (K - 1);
    g.pad = dil * (K - 1) - pad;
    g.Hout = H;
    g.Wout = W;
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, 2.0 * N * C * C * K * K * (double)g.Hin * g.Win, s);
    timing.bytes(4.0 * N * C * ((double)H * W + (double)g.Hin * g.Win));  // dy once, dx once
    return dispatch_c(C, 1, K, dil, g, dy, w, nullptr, dx, nullptr, nullptr, s);
}

int dilconv_backward_weight(const float* x, const float* dy, float* dw, float* dbias, int N, int C,
                            int H, int W, int K, int pad, int dil, void* ws, size_t ws_bytes,
                            hipStream_t s) {
    if (!ws || ws_bytes < dilconv_workspace_bytes(C, K))
        return afd::fail(AFD_ERR_WORKSPACE, "dilconv wgrad: workspace too small");
    if (!image_fits(C, H, W)) return afd::fail(AFD_ERR_UNSUPPORTED, "dilconv: image too large");
    DirGeom g{};
    g.N = N; g.Hin = H; g.Win = W; g.pad = pad;
    g.Hout = H + 2 * pad - dil * (K - 1);
    g.Wout = W + 2 * pad - dil * (K - 1);
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, 2.0 * N * C * C * K * K * (double)g.Hout * g.Wout, s);
    timing.bytes(4.0 * N * C * ((double)H * W + (double)g.Hout * g.Wout));  // x once, dy once
    return dispatch_c(C, 2, K, dil, g, x, dy, nullptr, dw, dbias, static_cast<float*>(ws), s);
}

}  // namespace afd
