// Backward-weight of the 3x3 / pad 1 convolutions in the Winograd F(4x4, 3x3) domain on the f32 MFMA.
//
// Reference: the cuDNN backward-filter launches behind nn.Conv2d(k=3, padding=1) of DCNN blocks 3-6
// (src/audiofakedetect/models.py:263-276).
//
// With the forward's matrices (wino44.hip: Y = A^T [(G g G^T) . (B^T d B)] A per 4x4 output tile),
//     dW[co][ci] = G^T [ sum over tiles  (A dY_tile A^T) . (B^T X_tile B) ] G        (. = element-wise, 6x6)
// so per transform position p (36 of them) the sum over tiles is a GEMM  M_p[co][ci] = sum_t DY_p[co][t] X_p[ci][t]
// with K = the 4x4 tiles of the batch: 36 MACs per (co, ci, tile) = 2.25 per output pixel against 9 in direct form.
//
//   wave      = one 16 (co) x 16 (ci) block of all 36 positions on v_mfma_f32_16x16x4_f32: 144 accumulator
//               registers; a k-step is 4 tiles, lane = (channel = lane & 15, tile = lane >> 4) for both operands
//   workgroup = 4 waves = 2 x 2 blocks (32 co x 32 ci), TWO per CU (74 KB of LDS each).  Round 4: the eight-wave
//               forms of round 3 (2 x 4 / 4 x 2 blocks, one workgroup per CU) ran their transform and matrix phases
//               in lockstep over the whole CU, so nothing covered a phase's load and barrier waits; two independent
//               workgroups drift apart and fill each other's gaps -- block 3 at level 14 6.70 -> 6.06 ms, block 4
//               4.31 -> 3.92 ms, although the 32 x 32 block transforms every x patch Cout / 32 times and every dy
//               tile Cin / 32 times (eight waves: 3 x and 1 x on block 3)
//   round     = 2 k-steps (8 tiles).  Transform phase: the round has 4 x sets and 4 dy sets (a set = one
//               16-channel block of one k-step: 64 (channel, tile) pairs, one per lane); every wave transforms one
//               x patch (B^T d B, 6x6 from 6 rows x 24 bytes) and one dy tile (A dy A^T, 4x4 -> 6x6) and writes
//               them to LDS as [set][position group of 4][lane][4]: the producer's lane layout IS the MFMA operand
//               layout, so a consumer reads its own lane's 16 bytes.
//               Matrix phase: per k-step and position group one ds_read_b128 per operand, four MFMAs.
//               Single-buffered (f32 matrix instructions and vector instructions do not overlap on a SIMD --
//               tools/micro/coexec.hip -- so within a workgroup the phases run back to back);
//               the next round's patches are requested from memory before the matrix phase.
//   grid      = (channel groups) x S splits of the tile sequence (n, column group, tile row), tile row fastest so
//               that consecutive k-steps share their two halo rows in L1 / L2; every workgroup writes its partial
//               M [36][co][ci] to a slab; wino44_wgrad_reduce_kernel sums the splits (fixed order: deterministic)
//               and wino44_wgrad_g_kernel applies G^T . G.
// Bias gradient: the dy-transform lanes also sum their 16 values; one partial per split (channel group 0 only).
//
// What bounds it (round 3, block 3 at level 14: 6.8-8.6 ms by box against 2.6 ms of matrix instructions): the patch
// loads.  Ablations: without the x loads 4.2-4.8 ms; the six 8-byte halo loads of a patch cost as much as the six
// 16-byte loads (2.5 ms each class), also with 48 of their 64 lanes switched off -- a load costs by the instruction /
// the cache lines it asks for, not by its bytes; TCP_PENDING_STALL_CYCLES 39 % of the kernel's cycles, average
// L1 -> L2 read latency 540 cycles, L2 hit rate 71 %, HBM traffic 7.5 GB (about the tensors once).  Tried and
// measured level or slower: the dy tile requested a round ahead, quads /
// octets of lanes on consecutive addresses (jobs of 8 channels x 8 tiles), all 16-byte loads before the 8-byte
// ones, touching the lines two rounds ahead with dummy loads.
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// v_max_f32 as is: through fmaxf the compiler first canonicalises an operand it cannot prove quiet (a value straight
// from memory) with a v_max_f32 x, x of its own -- two instructions per element in the input fold's PReLU
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

constexpr int kPos = 36;
constexpr int kSetFloats = 9 * 64 * 4;  // one operand set: [position group][lane][4]
constexpr int kThreadsW = 256;
constexpr int COB = 2, CIB = 2, NW = COB * CIB;  // blocks of a workgroup along co / ci, waves
constexpr int kSets = 2 * (COB + CIB);           // operand sets of a round

struct GW {
    int N, Cin, Cout, H, W;
    int rows, cols;             // valid region of dy (a crop of H x W)
    int tilesX, tilesY, groupsX;  // groupsX = k-step groups (4 tiles) per tile row
    long units;                 // N * groupsX * tilesY k-steps, ordered (n, column group, tile row)
    int S;                      // splits of the unit sequence
    long units_per_split;       // even
    int cig;                    // channel groups along Cin
    float* slab;                // [S][36][Cout][Cin]
    float* partb;               // [S][Cout] (bias partials) or null
    // pooled dy (PDY): dy is the pooled-resolution gradient gg [N][Cout][Hp][Wp] of a PReLU + 2x2 max-pool behind the
    // convolution and didx its argmax codes (position = code & 3): the 4 x 4 dy tile of a lane is built from 2 x 2
    // pooled values while it loads (rows = 2 Hp, cols = 2 Wp)
    const unsigned char* didx;
    int Hp, Wp;
    // input fold (round 4, as wino44.hip's G4::in_aff): x is the INPUT of the training-mode BatchNorm(affine=False) in
    // front of the convolution (z with the PReLU slope in_slope between them, or the tensor itself), in_aff [Cin][2] =
    // (mean, invstd): the patch becomes (PReLU(z) - mean) * invstd before it is transformed
    const float* in_aff;
    const float* in_slope;
};

// The transforms run on packed FMAs (v_pk_fma_f32: two floats per lane and instruction).  A 6 x 6 patch is held as
// three column pairs per row.  The vertical pass is the usual 12-operation form on pairs (both halves do the same);
// the horizontal pass works INSIDE a row, where the two halves of an instruction take different constants:
//     (a, c) = (-4, -1) t2 + t4,   (b, e) = (-4, -1) t1 + t3,   (o1, o3) = (a, c) + (1, 2) (b, e),
//     (o2, o4) = (a, c) - (1, 2) (b, e),   (o0, o5) = 4 (t0, t1) - 5 (t2, t3) + (t4, t5)
// -- 6 instructions per row instead of 12, every source an aligned register pair with its halves picked by op_sel.
// Its results come out as the pairs (o0, o5), (o1, o3), (o2, o4): the operand sets, the accumulators and the products
// use that SLOT order inside a transform row (slot 6 i + j' holds position 6 i + kSlotPos[j']); only the slab
// store at the end of the kernel maps back.  B^T d B: 72 instructions (144 unpacked), A dy A^T: 46 (80).
__device__ constexpr int kSlotPos[6] = {0, 5, 1, 3, 2, 4};

// B^T d along the rows of a column pair
__device__ __forceinline__ void bt6v(const f32x2 d0, const f32x2 d1, const f32x2 d2, const f32x2 d3, const f32x2 d4,
                                     const f32x2 d5, f32x2* t) {
    const f32x2 a = -4.f * d2 + d4, b = -4.f * d1 + d3;
    const f32x2 c = d4 - d2, e = d3 - d1;
    t[0] = 4.f * d0 + (-5.f * d2 + d4);
    t[1] = a + b;
    t[2] = a - b;
    t[3] = 2.f * e + c;
    t[4] = -2.f * e + c;
    t[5] = 4.f * d1 + (-5.f * d3 + d5);
}

// (t B) inside one row held as (t0, t1), (t2, t3), (t4, t5) -> slots (o0, o5), (o1, o3), (o2, o4)
__device__ __forceinline__ void bt6h(const f32x2 p0, const f32x2 p1, const f32x2 p2, f32x2* o) {
    const f32x2 k41 = {-4.f, -1.f}, k12 = {1.f, 2.f};
    const f32x2 ac = k41 * p1.xx + p2.xx;
    const f32x2 be = k41 * p0.yy + p1.yy;
    o[0] = 4.f * p0 + (-5.f * p1 + p2);
    o[1] = k12 * be + ac;
    o[2] = -k12 * be + ac;
}

// A y along the rows of a column pair: A = (A^T)^T of wino44.hip's at6, rows [1 0 0 0], [1 1 1 1], [1 -1 1 -1],
// [1 2 4 8], [1 -2 4 -8], [0 0 0 1]
__device__ __forceinline__ void a6v(const f32x2 y0, const f32x2 y1, const f32x2 y2, const f32x2 y3, f32x2* t) {
    const f32x2 s02 = y0 + y2, s13 = y1 + y3;
    const f32x2 e = 4.f * y2 + y0, o = 4.f * y3 + y1;
    t[0] = y0;
    t[1] = s02 + s13;
    t[2] = s02 - s13;
    t[3] = 2.f * o + e;
    t[4] = -2.f * o + e;
    t[5] = y3;
}

// (t A^T) inside one row held as (y0, y1), (y2, y3) -> slots (o0, o5), (o1, o3), (o2, o4)
__device__ __forceinline__ void a6h(const f32x2 p0, const f32x2 p1, f32x2* o) {
    const f32x2 k14 = {1.f, 4.f}, k12 = {1.f, 2.f};
    const f32x2 se = k14 * p1.xx + p0.xx;  // (y0 + y2, y0 + 4 y2)
    const f32x2 so = k14 * p1.yy + p0.yy;  // (y1 + y3, y1 + 4 y3)
    o[0] = f32x2{p0.x, p1.y};
    o[1] = k12 * so + se;
    o[2] = -k12 * so + se;
}

struct Unit {
    int n, xg, ty;
    bool live;
};

// (uniform arguments: the compiler keeps the result in scalar registers; units < 2^31, checked by the host)
__device__ __forceinline__ Unit decode_unit(long u, long end, const GW& g) {
    Unit r;
    r.live = u < end;
    // (round 5 measured an order in which the two k-steps of a round are neighbouring column groups of one tile row --
    // whole 128-byte lines per patch row instead of 72 bytes of a line whose rest is fetched again tilesY k-steps later;
    // the launch moves 1.6-1.8 x its algorithmic bytes, profiles/r05_pmc_traffic_* -- with super-groups of 2 / 4 / 8
    // column groups: 12.48-12.55 ms for the class either way, so the plain order stays)
    const unsigned uu = r.live ? (unsigned)u : 0u;
    const unsigned q = uu / (unsigned)g.tilesY;
    r.ty = (int)(uu - q * (unsigned)g.tilesY);
    const unsigned n = q / (unsigned)g.groupsX;
    r.xg = (int)(q - n * (unsigned)g.groupsX);
    r.n = (int)n;
    return r;
}

typedef unsigned short u16u __attribute__((aligned(1)));

template <bool PDY = false>
__global__ void __launch_bounds__(kThreadsW) __attribute__((amdgpu_waves_per_eu(2, 2)))
wino44_wgrad_kernel(const GW g, const float* __restrict__ x, const float* __restrict__ dy) {
    extern __shared__ __attribute__((aligned(16))) float sets[];  // [kSets][9][64][4]
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave / CIB, wb = wave % CIB;  // this wave's (co block, ci block)
    const int cgroups = g.cig * (g.Cout / (16 * COB));
    // the channel groups of one split read the same x / dy tiles: they get block ids 8 apart (one XCD under the
    // observed round-robin placement, so its L2 serves the second and third reader: HBM fetch of the block-3 launch
    // 23 -> 7.2 GB, the tensors once; the kernel's time is set by its L1 requests and does not change.  Speed only)
    const int bq = blockIdx.x >> 3, br = blockIdx.x & 7;
    const int cg = bq % cgroups;
    const int split = (bq / cgroups) * 8 + br;
    if (split >= g.S) return;
    const int co0 = (cg / g.cig) * 16 * COB, ci0 = (cg % g.cig) * 16 * CIB;
    const long u_begin = (long)split * g.units_per_split;
    long u_end = u_begin + g.units_per_split;
    u_end = u_end < g.units ? u_end : g.units;

    // transform jobs of this wave: the x set and the dy set (block wave & 1 of k-step wave >> 1)
    static_assert(COB == 2 && CIB == 2 && NW == 4, "one x set and one dy set per wave and round");
    const int xb = wave & 1, xks = wave >> 1;
    const int db_ = wave & 1, dks = wave >> 1;
    // set indices: x sets [ks][b] first, then dy sets [ks][a]
    float* my_xset = sets + (size_t)(xks * CIB + xb) * kSetFloats + lane * 4;
    float* my_dset = sets + (size_t)(2 * CIB + dks * COB + db_) * kSetFloats + lane * 4;

    const int lc = lane & 15, lt = lane >> 4;
    const size_t plane = (size_t)g.H * g.W;
    // every address is (uniform part: image, channel block, row, column group) + one loop-invariant lane offset
    // (channel lc of the block, tile lt of the k-step): scalar registers do the bookkeeping
    const unsigned loff = (unsigned)lc * (unsigned)plane + 4u * (unsigned)lt;
    const float* xblk = x + (size_t)(ci0 + 16 * xb) * plane;    // + n * Cin * plane
    const size_t pplane = PDY ? (size_t)g.Hp * g.Wp : plane;    // plane of dy as it is stored
    const float* dblk = dy + (size_t)(co0 + 16 * db_) * pplane;  // + n * Cout * plane
    const unsigned char* iblk = PDY ? g.didx + (size_t)(co0 + 16 * db_) * pplane : nullptr;
    const unsigned poff = (unsigned)lc * (unsigned)pplane + 2u * (unsigned)lt;  // PDY: channel lc, pooled column pair lt
    float pg[2][2];      // PDY: the tile's 2 x 2 pooled gradients and codes of the coming round
    unsigned pcd[2];
    int pleft = 2;       // PDY: valid pooled columns of the tile (edge groups)

    const bool fold = g.in_aff != nullptr;  // uniform
    const bool in_act = fold && g.in_slope != nullptr;
    const float in_a = in_act ? g.in_slope[0] : 1.f;
    const bool in_fast = in_a >= 0.f && in_a <= 1.f;
    // (mean, invstd) of this lane's x channel: loop-invariant
    const f32x2 aff = fold ? *reinterpret_cast<const f32x2*>(g.in_aff + 2 * (ci0 + 16 * xb + lc)) : f32x2{0.f, 1.f};
    float d[6][6];   // x patch of the coming round
    float e[4][4];   // dy tile of the coming round
    bool x_edge = false, d_edge = false;
    Unit ux{}, ud{};
    // Image borders.  Rows: a patch row outside the image is loaded from a clamped (valid) row and zeroed before
    // the transform -- per row and uniform over the wave.  Columns: only the first column group holds a patch
    // that starts one column left of the image, and (the host checks 4 tilesX <= W) a patch can pass the right
    // border by one column at most; in those groups (uniform branch) the two loads of a row start one column
    // in / one column early and the six values are picked by lane selects.  Tiles past tilesX (ragged last
    // group) read tile 0 of their group and are zeroed.
    // one row of an interior patch (two loads); `load_x` below requests a whole patch, the matrix phase deals the six
    // rows of the coming round's patch out over its first position groups (see there)
    const float* xc_next = nullptr;
    auto load_x_row = [&](int r, int half = 2) {
        const int iy = 4 * ux.ty - 1 + r;
        const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
        const float* row = xc_next + (size_t)iyc * g.W;  // uniform
        if (half != 1) {
            const f4u v = *reinterpret_cast<const f4u*>(row + loff);
            d[r][0] = v.x; d[r][1] = v.y; d[r][2] = v.z; d[r][3] = v.w;
        }
        if (half != 0) {
            const f2u w2 = *reinterpret_cast<const f2u*>(row + loff + 4);
            d[r][4] = w2.x; d[r][5] = w2.y;
        }
    };
    auto load_x = [&](long u0, bool rows_later) {
        ux = decode_unit(u0 + xks, u_end, g);
        const int iy0 = 4 * ux.ty - 1;
        x_edge = ux.xg == 0 || 16 * ux.xg + 17 > g.W || 4 * ux.xg + 3 >= g.tilesX;
        const float* xc = xblk + (size_t)ux.n * g.Cin * plane + (16 * ux.xg - 1);
        xc_next = xc;
        if (!x_edge) {
            if (rows_later) return;
#pragma unroll
            for (int r = 0; r < 6; ++r) load_x_row(r);
        } else {
            const int tx = 4 * ux.xg + lt;
            const bool tile_ok = tx < g.tilesX;
            const int ix0 = tile_ok ? 4 * tx - 1 : 16 * ux.xg - 1;
            const bool L = ix0 < 0, R = ix0 + 5 >= g.W;
            const unsigned o1 = (unsigned)lc * (unsigned)plane + (unsigned)(ix0 - (16 * ux.xg - 1) + (L ? 1 : 0));
            const unsigned o2 = (unsigned)lc * (unsigned)plane + (unsigned)(ix0 - (16 * ux.xg - 1) + 4 - (R ? 1 : 0));
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int iy = iy0 + r;
                const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
                const float* row = xc + (size_t)iyc * g.W;
                const f4u v = *reinterpret_cast<const f4u*>(row + o1);
                const f2u w2 = *reinterpret_cast<const f2u*>(row + o2);
                d[r][0] = L ? 0.f : v.x;
                d[r][1] = L ? v.x : v.y;
                d[r][2] = L ? v.y : v.z;
                d[r][3] = L ? v.z : v.w;
                d[r][4] = R ? w2.y : w2.x;
                d[r][5] = R ? 0.f : w2.y;
                if (!tile_ok) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) d[r][j] = 0.f;
                }
            }
        }
    };
    auto load_d = [&](long u0) {
        ud = decode_unit(u0 + dks, u_end, g);
        const int oy0 = 4 * ud.ty;
        d_edge = 16 * ud.xg + 16 > g.cols;
        if constexpr (PDY) {
            const size_t nb = (size_t)ud.n * g.Cout * pplane + 8 * ud.xg;
            const int tx = 4 * ud.xg + lt;
            const bool tile_ok = !d_edge || tx < g.tilesX;
            const unsigned o1 = d_edge ? (unsigned)lc * (unsigned)pplane + (tile_ok ? 2u * (unsigned)lt : 0u) : poff;
            pleft = d_edge ? (tile_ok ? (g.cols - 4 * tx) >> 1 : 0) : 2;  // cols is even: whole windows
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int pr = 2 * ud.ty + r < g.Hp ? 2 * ud.ty + r : g.Hp - 1;
                const size_t ro = nb + (size_t)pr * g.Wp;  // uniform
                // an edge tile with one window left reads its second value from the next row / plane / the slack the
                // host leaves behind the tensors; it is masked in transform_d
                const f2u v = *reinterpret_cast<const f2u*>(dblk + ro + o1);
                pg[r][0] = v.x; pg[r][1] = v.y;
                pcd[r] = *reinterpret_cast<const u16u*>(iblk + ro + o1);
            }
            return;
        }
        const float* dc = dblk + (size_t)ud.n * g.Cout * plane + 16 * ud.xg;
        if (!d_edge) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = oy0 + r < g.H ? oy0 + r : g.H - 1;
                const float* row = dc + (size_t)oy * g.W;  // uniform
                const f4u v = *reinterpret_cast<const f4u*>(row + loff);
                e[r][0] = v.x; e[r][1] = v.y; e[r][2] = v.z; e[r][3] = v.w;
            }
        } else {
            const int tx = 4 * ud.xg + lt;
            const bool tile_ok = tx < g.tilesX;
            const unsigned o1 = (unsigned)lc * (unsigned)plane + (tile_ok ? 4u * (unsigned)lt : 0u);
            const int left = tile_ok ? g.cols - 4 * tx : 0;  // valid columns of this tile
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = oy0 + r < g.H ? oy0 + r : g.H - 1;
                const float* row = dc + (size_t)oy * g.W;
                const f4u v = *reinterpret_cast<const f4u*>(row + o1);
                e[r][0] = left > 0 ? v.x : 0.f;
                e[r][1] = left > 1 ? v.y : 0.f;
                e[r][2] = left > 2 ? v.z : 0.f;
                e[r][3] = left > 3 ? v.w : 0.f;
            }
        }
    };
    float bsum = 0.f;  // bias partial of this lane's channel (dy lanes)
    auto transform_x = [&]() {
        const int iy0 = 4 * ux.ty - 1;
        if (fold) {  // uniform
            auto norm = [&](auto&& act) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        f32x2 v = act(f32x2{d[r][2 * c], d[r][2 * c + 1]});
                        v = (v - aff.x) * aff.y;
                        d[r][2 * c] = v.x;
                        d[r][2 * c + 1] = v.y;
                    }
            };
            // PReLU as nn.hip computes it (z > 0 ? z : a z); with 0 <= a <= 1 that is max(z, a z), bit for bit
            if (!in_act) norm([](f32x2 v) { return v; });
            else if (in_fast) norm([&](f32x2 v) { const f32x2 av = v * in_a; return f32x2{vmax(v.x, av.x), vmax(v.y, av.y)}; });
            else norm([&](f32x2 v) { return f32x2{v.x > 0.f ? v.x : in_a * v.x, v.y > 0.f ? v.y : in_a * v.y}; });
            if (x_edge) {  // the zeros load_x put at the image's left / right border and in tiles past tilesX
                const int tx = 4 * ux.xg + lt;
                const bool tile_ok = tx < g.tilesX;
                const int ix0 = tile_ok ? 4 * tx - 1 : 16 * ux.xg - 1;
                const bool L = ix0 < 0, R = ix0 + 5 >= g.W;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    d[r][0] = (L || !tile_ok) ? 0.f : d[r][0];
                    d[r][5] = (R || !tile_ok) ? 0.f : d[r][5];
#pragma unroll
                    for (int j = 1; j < 5; ++j) d[r][j] = tile_ok ? d[r][j] : 0.f;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = iy0 + r;
            if (!ux.live || iy < 0 || iy >= g.H) {  // uniform
#pragma unroll
                for (int j = 0; j < 6; ++j) d[r][j] = 0.f;
            }
        }
        f32x2 t[6][3];  // t = B^T d, column pairs
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x2 o[6];
            bt6v(f32x2{d[0][2 * c], d[0][2 * c + 1]}, f32x2{d[1][2 * c], d[1][2 * c + 1]},
                 f32x2{d[2][2 * c], d[2][2 * c + 1]}, f32x2{d[3][2 * c], d[3][2 * c + 1]},
                 f32x2{d[4][2 * c], d[4][2 * c + 1]}, f32x2{d[5][2 * c], d[5][2 * c + 1]}, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) t[r][c] = o[r];
        }
        // second pass two rows at a time: 12 slots = three 16-byte stores, so only 12 results are live
#pragma unroll
        for (int r2 = 0; r2 < 3; ++r2) {
            f32x2 o[6];
            bt6h(t[2 * r2][0], t[2 * r2][1], t[2 * r2][2], o);
            bt6h(t[2 * r2 + 1][0], t[2 * r2 + 1][1], t[2 * r2 + 1][2], o + 3);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<f32x4*>(my_xset + (3 * r2 + q) * 256) =
                    f32x4{o[2 * q].x, o[2 * q].y, o[2 * q + 1].x, o[2 * q + 1].y};
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto transform_d = [&](bool want_bias) {
        const int oy0 = 4 * ud.ty;
        if constexpr (PDY) {
            // window (r, c) of the tile: its gradient sits at position code & 3 = 2 dy + dx, the other three are zero
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const unsigned pos = (pcd[r] >> (8 * c)) & 3u;
                    const float v = c < pleft ? pg[r][c] : 0.f;
                    e[2 * r][2 * c] = pos == 0u ? v : 0.f;
                    e[2 * r][2 * c + 1] = pos == 1u ? v : 0.f;
                    e[2 * r + 1][2 * c] = pos == 2u ? v : 0.f;
                    e[2 * r + 1][2 * c + 1] = pos == 3u ? v : 0.f;
                }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!ud.live || oy0 + r >= g.rows) {  // uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) e[r][j] = 0.f;
            }
        }
        if (want_bias) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) s += (e[r][0] + e[r][1]) + (e[r][2] + e[r][3]);
            bsum += s;
        }
        f32x2 t[6][2];  // t = A e, column pairs
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x2 o[6];
            a6v(f32x2{e[0][2 * c], e[0][2 * c + 1]}, f32x2{e[1][2 * c], e[1][2 * c + 1]},
                f32x2{e[2][2 * c], e[2][2 * c + 1]}, f32x2{e[3][2 * c], e[3][2 * c + 1]}, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) t[r][c] = o[r];
        }
#pragma unroll
        for (int r2 = 0; r2 < 3; ++r2) {
            f32x2 o[6];
            a6h(t[2 * r2][0], t[2 * r2][1], o);
            a6h(t[2 * r2 + 1][0], t[2 * r2 + 1][1], o + 3);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<f32x4*>(my_dset + (3 * r2 + q) * 256) =
                    f32x4{o[2 * q].x, o[2 * q].y, o[2 * q + 1].x, o[2 * q + 1].y};
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4 acc[kPos];
#pragma unroll
    for (int p = 0; p < kPos; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool want_bias = g.partb != nullptr && (cg % g.cig) == 0;
    // A dy tile with at most three live rows (the last tile row of an image whose rows are not a multiple of 4: 6-row
    // images at level 14) has A dy A^T = 0 at the positions 30..35 (row 5 of A picks dy row 3): those six products of
    // its k-step are skipped (uniform: the tile rows of the round's two k-steps)
    const int ty_short = (g.rows & 3) ? g.tilesY - 1 : -1;
    auto ty_of = [&](long u) { return (int)((unsigned)u % (unsigned)g.tilesY); };  // scalar, as decode_unit
    load_x(u_begin, false);
    for (long u0 = u_begin; u0 < u_end; u0 += 2) {
        const bool short0 = ty_of(u0) == ty_short, short1 = ty_of(u0 + 1) == ty_short;
        // the dy tile (four 16-byte loads) is requested here and arrives during the x transform; only the x patch
        // (36 registers) is held across the matrix phase.  The pooled form's tile (6 registers) is requested inside the
        // matrix phase, after the patch's loads
        if (!PDY || u0 == u_begin) load_d(u0);
        transform_x();
        transform_d(want_bias);
        __syncthreads();
        // the coming round's patch travels during the matrix phase: its unit is decoded here, its loads are dealt out
        // over the position groups below (an edge group's select-and-mask loads stay in one piece)
        const bool next = u0 + 2 < u_end;
        if (next) load_x(u0 + 2, true);
        const bool rows_here = next && !x_edge;  // uniform
        // operands of position group pg + 1 are requested before the four matrix instructions of group pg
        const float* as0 = sets + (size_t)(2 * CIB + wa) * kSetFloats + lane * 4;
        const float* bs0 = sets + (size_t)wb * kSetFloats + lane * 4;
        f32x4 av = *reinterpret_cast<const f32x4*>(as0), bv = *reinterpret_cast<const f32x4*>(bs0);
#pragma unroll
        for (int it = 0; it < 18; ++it) {
            const int ks = it / 9, pg = it % 9;
            f32x4 an = av, bn = bv;
            if (it + 1 < 18) {
                const int ks1 = (it + 1) / 9, pg1 = (it + 1) % 9;
                an = *reinterpret_cast<const f32x4*>(as0 + (size_t)ks1 * COB * kSetFloats + pg1 * 256);
                bn = *reinterpret_cast<const f32x4*>(bs0 + (size_t)ks1 * CIB * kSetFloats + pg1 * 256);
            }
            // the coming round's patch, ONE load per position group: a load instruction takes the CU's address pipeline
            // ~16 cycles whatever its size or hit rate (the kernel taken apart), and twelve of them issued back
            // to back by all four waves right behind the barrier kept every wave from its first matrix instruction until
            // the pipeline had taken them: 12.6 -> 11.8 ms for the class (round 5)
            if (it < 12 && rows_here) load_x_row(it >> 1, it & 1);
            // (the pooled dy tile -- 2 x 2 values and their codes, 6 registers -- follows; the dense tile's 16 registers held
            // across the matrix phase cost more than its loads at the top of the round: 12.07 against 11.76 ms for the class)
            if (PDY && it == 12 && next) load_d(u0 + 2);
            __builtin_amdgcn_sched_barrier(0);
            const bool short_ks = ks ? short1 : short0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (4 * pg + q >= 30 && short_ks) continue;  // uniform
                acc[4 * pg + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[4 * pg + q], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            av = an;
            bv = bn;
        }
        __syncthreads();
    }

    // partial M of this workgroup: acc[p][j] = M_p[co0 + 16 wa + 4 (lane >> 4) + j][ci0 + 16 wb + (lane & 15)]
    float* sl = g.slab + (size_t)split * kPos * g.Cout * g.Cin;
    const int co = co0 + 16 * wa + 4 * lt, ci = ci0 + 16 * wb + lc;
#pragma unroll
    for (int p = 0; p < kPos; ++p) {
        const int pos = 6 * (p / 6) + kSlotPos[p % 6];  // accumulator slot -> transform position
#pragma unroll
        for (int j = 0; j < 4; ++j) sl[((size_t)pos * g.Cout + co + j) * g.Cin + ci] = acc[p][j];
    }
    if (want_bias) {
        // this wave summed dy over the tiles (lane >> 4) of its k-steps for channel lane & 15 of block db_;
        // the partials of the waves holding the same block (other k-step) are added by the reduce kernel
        float s = bsum;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lt == 0) g.partb[((size_t)split * 2 + dks) * g.Cout + co0 + 16 * db_ + lc] = s;
    }
}

// sum of the S partial slabs, element-wise: red[i] = sum_s slab[s][i], i over 36 * Cout * Cin (fixed order)
__global__ void __launch_bounds__(256) wino44_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ red,
                                                                  long total, int S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 3 < S; s += 4) {
        const float v0 = slab[(size_t)s * total + i], v1 = slab[(size_t)(s + 1) * total + i];
        const float v2 = slab[(size_t)(s + 2) * total + i], v3 = slab[(size_t)(s + 3) * total + i];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; s < S; ++s) a0 += slab[(size_t)s * total + i];
    red[i] = (a0 + a1) + (a2 + a3);
}

// dW = G^T M G per (co, ci); G (6x3) of wino44.hip.  Also the bias gradient from the per-split partials.
__global__ void __launch_bounds__(256) wino44_wgrad_g_kernel(const float* __restrict__ red, const float* __restrict__ partb,
                                                             float* __restrict__ dw, float* __restrict__ db, int Cout,
                                                             int Cin, int S2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int pairs = Cout * Cin;
    if (i < pairs) {
        float m[6][6];
#pragma unroll
        for (int p = 0; p < kPos; ++p) m[p / 6][p % 6] = red[(size_t)p * pairs + i];
        // t = G^T m (3x6): G^T rows: [1/4 -1/6 -1/6 1/24 1/24 0], [0 -1/6 1/6 1/12 -1/12 0], [0 -1/6 -1/6 1/6 1/6 1]
        float t[3][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float s12 = m[1][j] + m[2][j], d12 = m[1][j] - m[2][j];
            const float s34 = m[3][j] + m[4][j], d34 = m[3][j] - m[4][j];
            t[0][j] = 0.25f * m[0][j] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
            t[1][j] = -(1.f / 6.f) * d12 + (1.f / 12.f) * d34;
            t[2][j] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + m[5][j];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float s12 = t[r][1] + t[r][2], d12 = t[r][1] - t[r][2];
            const float s34 = t[r][3] + t[r][4], d34 = t[r][3] - t[r][4];
            dw[(size_t)i * 9 + r * 3 + 0] = 0.25f * t[r][0] - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
            dw[(size_t)i * 9 + r * 3 + 1] = -(1.f / 6.f) * d12 + (1.f / 12.f) * d34;
            dw[(size_t)i * 9 + r * 3 + 2] = -(1.f / 6.f) * s12 + (1.f / 6.f) * s34 + t[r][5];
        }
    }
    // bias gradient: the last Cout blocks sum one channel's partials each (a tree over the block: fixed order)
    if (db && partb && (int)blockIdx.x >= (pairs + 255) / 256) {
        const int c = blockIdx.x - (pairs + 255) / 256;
        float a = 0.f;
        for (int s = threadIdx.x; s < S2; s += 256) a += partb[(size_t)s * Cout + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        __shared__ float red4[4];
        if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) db[c] = (red4[0] + red4[1]) + (red4[2] + red4[3]);
    }
}

bool shape_ok(int Cin, int Cout) { return Cout % (16 * COB) == 0 && Cin % (16 * CIB) == 0; }

// workgroups aimed at in all: two are resident per CU (74 KB of LDS each) and all do the same work, so the grid
// should be a whole number of rounds over the 512 slots -- three rounds (measured on block 3 / block 4 at level 14:
// 1024 6.32 / 4.11 ms, 1536 6.09 / 3.93, 2048 6.14 / 3.98, 3072 6.11 / 3.98).  The workspace bound follows it.
int target_wgs() { return 1536; }

void plan(GW& g, int N, int Cin, int H, int W, int Cout, int dy_rows, int dy_cols) {
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.H = H; g.W = W;
    g.rows = dy_rows < H ? dy_rows : H;
    g.cols = dy_cols < W ? dy_cols : W;
    g.tilesX = (g.cols + 3) / 4;
    g.tilesY = (g.rows + 3) / 4;
    g.groupsX = (g.tilesX + 3) / 4;
    g.units = (long)N * g.groupsX * g.tilesY;
    g.cig = Cin / (16 * CIB);
    const int cgroups = g.cig * (Cout / (16 * COB));
    // three, two or one rounds of workgroups over the CUs (see target_wgs), each workgroup with at least 8 rounds
    // of its own; a smaller problem gets as many splits as that allows
    const long max_s = g.units / 16 > 0 ? g.units / 16 : 1;
    long S = 0;
    // ... and, round 6, with at least kMinUnits units of its own where the problem allows it: every workgroup writes a
    // [36][32][32] slab (147 KB) that the reduce kernel reads again, and on the small level-8 / STFT layers three rounds
    // of workgroups meant 28 rounds of work per slab (level-8 step: conv_wgrad 1.13 -> 1.05 ms with one round)
    // (same box, conv_wgrad per step with no floor / 128 / 256 units: level 8 1.141 / 1.050 / 1.054 ms, level 14 coif4 11.77 /
    // 11.73 / 11.67, sym5 6.70 / 6.59 / -)
    constexpr long kMinUnits = 256;
    // pass 0: the most rounds that leave kMinUnits per workgroup; pass 1 (none does): the fewest rounds that fit at all
    for (int pass = 0; pass < 2 && !S; ++pass)
        for (int k = 0; k < 3 && !S; ++k) {
            const int rounds = pass == 0 ? 3 - k : 1 + k;
            long cand = (long)target_wgs() * rounds / 3 / cgroups;
            if (cand >= 8) cand = cand / 8 * 8;  // the grid is cgroups * ceil(S / 8) * 8 workgroups: do not pass the aim
            if (cand >= 1 && cand <= max_s && (pass == 1 || g.units / cand >= kMinUnits)) S = cand;
        }
    if (!S) S = max_s;
    // never more splits than the workspace bound holds slabs for (wino44_wgrad_workspace_floats: with more channel
    // groups than workgroups aimed at, every candidate above is 0 and max_s could pass it)
    const long smax = target_wgs() / cgroups > 0 ? target_wgs() / cgroups : 1;
    if (S > smax) S = smax;
    long ups = (g.units + S - 1) / S;
    ups += ups & 1;
    g.units_per_split = ups;
    g.S = (int)((g.units + ups - 1) / ups);
}

}  // namespace

namespace afd {

// the crop of dy decides the right border: a patch may pass it by one column at most (load_x)
bool wino44_wgrad_crop_ok(int H, int W, int dy_rows, int dy_cols) {
    const int cols = dy_cols < W ? dy_cols : W;
    const int rows = dy_rows < H ? dy_rows : H;
    return rows >= 1 && cols >= 1 && 4 * ((cols + 3) / 4) <= W;
}

bool wino44_wgrad_applicable(int Cin, int H, int W, int Cout, int K, int pad, int dil) {
    if (getenv("AFD_NO_WINOGRAD") || getenv("AFD_NO_WINO44_WGRAD")) return false;
    if (K != 3 || pad != 1 || dil != 1) return false;
    if (!shape_ok(Cin, Cout)) return false;
    if (W < 8 || H < 2) return false;
    return (size_t)H * W * 16 < 0x7fffffffULL;  // 32-bit lane offsets inside a 16-channel block
}

// floats: S slabs + the reduced M + bias partials.  An upper bound over every crop of dy (the split count of a
// cropped launch can come out a few higher than the uncropped one's through the rounding of units per split).
size_t wino44_wgrad_workspace_floats(int N, int Cin, int H, int W, int Cout, int dy_rows, int dy_cols) {
    (void)N; (void)H; (void)W; (void)dy_rows; (void)dy_cols;
    if (!shape_ok(Cin, Cout)) return 0;
    const int cgroups = (Cin / (16 * CIB)) * (Cout / (16 * COB));
    const size_t smax = (size_t)(target_wgs() / cgroups > 0 ? target_wgs() / cgroups : 1);
    const size_t m = (size_t)kPos * Cout * Cin;
    return smax * m + m + smax * 2 * Cout;
}

int wino44_wgrad_run(const float* x, const float* dy, float* dw, float* dbias, int N, int Cin, int H, int W, int Cout,
                     int dy_rows, int dy_cols, void* ws, size_t ws_bytes, hipStream_t s, const unsigned char* pooled_codes,
                     const float* in_aff, const float* in_slope) {
    GW g{};
    if (pooled_codes) {  // dy is the pooled gradient [N][Cout][H / 2][W / 2]: the crop is the pooled region
        dy_rows = 2 * (H / 2);
        dy_cols = 2 * (W / 2);
    }
    plan(g, N, Cin, H, W, Cout, dy_rows, dy_cols);
    g.didx = pooled_codes;
    g.in_aff = in_aff;
    g.in_slope = in_slope;
    g.Hp = H / 2;
    g.Wp = W / 2;
    if (!wino44_wgrad_crop_ok(H, W, dy_rows, dy_cols) || g.units >= 0x7fffffffL)
        return afd::fail(AFD_ERR_UNSUPPORTED, "winograd backward-weight: image %d x %d, crop %d x %d", H, W, dy_rows, dy_cols);
    const size_t m = (size_t)kPos * Cout * Cin;
    // the bound the caller sized the workspace by, and what this launch writes: S slabs + the reduced M + bias partials
    if (!ws || ws_bytes < wino44_wgrad_workspace_floats(N, Cin, H, W, Cout, dy_rows, dy_cols) * sizeof(float) ||
        ws_bytes < ((size_t)g.S * m + m + (size_t)g.S * 2 * Cout) * sizeof(float))
        return afd::fail(AFD_ERR_WORKSPACE, "winograd backward-weight: workspace too small");
    g.slab = static_cast<float*>(ws);
    float* red = g.slab + (size_t)g.S * m;
    g.partb = dbias ? red + m : nullptr;
    const int cgroups = g.cig * (Cout / (16 * COB));
    const size_t lds = (size_t)kSets * kSetFloats * sizeof(float);
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino44_wgrad_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino44_wgrad_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "winograd backward-weight: %s", hipGetErrorString(e));
        attr.mark();
    }
    afd::ScopedTiming timing(AFD_K_CONV_WGRAD, 2.0 * N * Cout * (double)g.rows * g.cols * Cin * 9, s);
    // 36 GEMMs [Cout x Cin] with K = every tile of every k-step (tile padding included)
    // (less the six positions skipped per k-step of a short tile row)
    const double short_units = (g.rows & 3) ? (double)N * g.groupsX : 0.0;
    timing.issued(2.0 * (double)Cout * Cin * 4.0 * (kPos * (double)g.units - 6.0 * short_units));
    timing.bytes(4.0 * N * ((double)Cin * H * W + (double)Cout * g.rows * g.cols * (pooled_codes ? 0.3125 : 1.0)));
    const unsigned grid = (unsigned)(cgroups * ((g.S + 7) / 8) * 8);
    if (pooled_codes) hipLaunchKernelGGL(wino44_wgrad_kernel<true>, dim3(grid), dim3(kThreadsW), lds, s, g, x, dy);
    else hipLaunchKernelGGL(wino44_wgrad_kernel<false>, dim3(grid), dim3(kThreadsW), lds, s, g, x, dy);
    int rc = afd::check_launch("wino44_wgrad_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(wino44_wgrad_reduce_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, g.slab, red, (long)m, g.S);
    rc = afd::check_launch("wino44_wgrad_reduce_kernel");
    if (rc) return rc;
    const int pairs = Cout * Cin;
    hipLaunchKernelGGL(wino44_wgrad_g_kernel, dim3((pairs + 255) / 256 + (g.partb ? Cout : 0)), dim3(256), 0, s, red, g.partb,
                       dw, dbias, Cout, Cin, 2 * g.S);
    return afd::check_launch("wino44_wgrad_g_kernel");
}

}  // namespace afd
