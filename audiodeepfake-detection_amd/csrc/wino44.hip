// Winograd F(4x4, 3x3) forward / backward-data on the f32 MFMA: 36 multiplies per 4x4 output tile and channel
// pair instead of 64 for four F(2x2, 3x3) tiles (wino.hip) or 144 direct; arithmetic fp32 end to end
// (transform constants up to 8: measured 7e-6 of the largest output against float64 on a block-3 shaped layer,
// F(2x2): 4e-7).
//
// Reference: the cuDNN launches behind nn.Conv2d(k=3, padding=1) of DCNN block 3
// (src/audiofakedetect/models.py:263-278) and its backward-data pass.
//
//   wave      = 16 output channels x 16 tiles x 36 positions on 16x16x4 tiles: 144 accumulator registers; every
//               (channel, tile) has its 36 positions in one lane, so A^T M A runs in registers
//   workgroup = CG waves (Cout / 16) over the same 16 tiles of a tile row (4 output rows x 64 columns); 74 KB of
//               LDS, two workgroups per CU
//   chunk     = 16 input channels: thread (channel, tile) of the first 256 threads transforms its 6x6 patch
//               (B^T d B, compile-time constants) into V[position][lane of the B fragment][k-step]: one
//               ds_read_b128 per position brings a wave's B operands of all four k-steps
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>
#include <type_traits>

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

constexpr int kCh = 16;     // input channels per chunk = 4 k-steps of 4
constexpr int kTiles = 16;  // tiles per workgroup
constexpr int kPos = 36;

struct G4 {
    int N, Cin, Cout, H, W;
    int rows, cols;
    int tilesX, tilesY, wgX, wxCount, nchunks;
    // backward-data launches whose result is the gradient of a BatchNorm output (afd_conv3x3_backward_data_bnstats):
    // the epilogue also sums, per channel, g and g * xhat over its outputs (xhat = bn_in, the forward convolution's
    // input): one partial row [sum g | sum g xhat] per workgroup, rows of this launch from part_row0
    const float* bn_in;
    float* stat_part;
    int part_row0;
    // pooled epilogue (afd_conv3x3_prelu_pool_forward): PReLU + MaxPool2d(2, 2) of the tile's four windows;
    // u / idx [N][Cout][rows / 2][cols / 2] are written instead of y
    const float* slope;
    float* u;
    unsigned char* idx;
    const float* x_end;  // one past the input tensor (border patches: see load_patch)
    // pooled input (PIN; backward-data only): x is the POOLED gradient gg [N][Cin][Hp][Wp] of the PReLU + 2x2 max-pool
    // behind the convolution, pidx the pool's argmax codes (position = code & 3); the dense gradient -- gg at that
    // position of each window, zero elsewhere -- is built in registers when a patch is consumed
    const unsigned char* pidx;
    int Hp, Wp;
    // input fold (forward launches; round 4): x is the INPUT of a training-mode BatchNorm(affine=False) in front of the
    // convolution -- z with the PReLU slope in_slope between them, or the tensor itself (in_slope null) -- and
    // in_aff [Cin][2] = (mean, invstd): a patch becomes (PReLU(z) - mean) * invstd while it is consumed, with
    // bn_apply_fwd_kernel's arithmetic (bit-identical values), so the normalised tensor is never written or read
    const float* in_aff;
    const float* in_slope;
    // border launch, round 6: the last workgroup column holds ONE live tile (129-column images: 33 tile columns = 2 x 16 + 1;
    // 8 193 columns: 128 x 16 + 1) -- instead of one workgroup per tile row with 15 dead tiles, `tail_groups` workgroups per
    // image take that tile column of 16 tile rows each (tile slot = tile row).  0 = off.
    int tail_groups;
    // round 6: a last tile row with at most TWO live output rows (the second tile row of the 6-row images of blocks 4-6
    // at level 14) runs F(2x4): the 4-point vertical transform of F(2, 3) with the 6-point horizontal one -- 24 positions
    // (transform rows 0..3) instead of the 30 the F(4x4) form needs there.  Its filter table [chunk][24][cg][lane][k-step]
    // follows the F(4x4) one at float offset u_short_off (0 = no such tile row in this launch).
    int u_short_off;
    int ty0, tyCount;  // tile rows of this launch (the F(2x4) tile row is launched apart)
    int noflip;  // (development) keep the helper waves of the second workgroup of a CU last: see the kernel's wave roles
    int xcd_map;  // workgroup columns per XCD group (16; 0 / 1 = plain row-major order): see the kernel
    // BatchNorm backward in the epilogue (backward-data launches with the statistics epilogue; round 4): the result g is the
    // gradient of a training-mode BatchNorm(affine=False) output whose backward sums are known BEFORE the launch
    // (afd_conv3x3_input_grad_sums, afd_conv_weight_dot): bn_tab [C][4] = (mean, invstd, mean of g, mean of g * xhat),
    // bn_in the BatchNorm's INPUT z, bn_slope the PReLU slope in front of it or null.  The launch then writes
    // dz = PReLU'(z) * invstd * (g - mdy - xhat * mdyx) -- bn_bwd_apply_kernel's arithmetic -- instead of g, and its two
    // sums per channel are sum(dz) (the bias gradient of the convolution in front) and the PReLU slope's partial gradient
    const float* bn_tab;
    const float* bn_slope;
    // ... of a BatchNorm right behind PReLU + MaxPool2d(2, 2) (block 3 -> 4): bn_in is the pooled tensor u, bn_codes the
    // pool's codes (bit 2: the window's winner was <= 0) -- the launch then writes the POOLED gradient of the convolution
    // in front of the pool (what afd_prelu_pool_backward_compact leaves), the slope factor decided by the code
    const unsigned char* bn_codes;
};
__device__ __forceinline__ bool getenv_noflip(const G4& g) { return g.noflip != 0; }

// slot j' of a transform row holds position kSlotPos[j'] (see bt6h)
__device__ constexpr int kSlotPos[6] = {0, 5, 1, 3, 2, 4};

// U = G g G^T, G (6x3)
__device__ __forceinline__ float g_row(int i, float a, float b, float c) {
    switch (i) {
        case 0: return 0.25f * a;
        case 1: return (-1.f / 6.f) * (a + b + c);
        case 2: return (-1.f / 6.f) * (a - b + c);
        case 3: return (1.f / 24.f) * a + (1.f / 12.f) * b + (1.f / 6.f) * c;
        case 4: return (1.f / 24.f) * a - (1.f / 12.f) * b + (1.f / 6.f) * c;
        default: return c;
    }
}

// F(2, 3) vertical transform of the filter (G2, 4 x 3)
__device__ __forceinline__ float g2_row(int i, float a, float b, float c) {
    switch (i) {
        case 0: return a;
        case 1: return 0.5f * (a + b + c);
        case 2: return 0.5f * (a - b + c);
        default: return c;
    }
}

// U table: [chunk][position][cg][lane][kstep (KS)] = U_p[16 cg + (lane & 15)][4 KS chunk + 4 kstep + (lane >> 4)];
// behind it (short_total > 0) the F(2x4) table of the short last tile row: the same with 24 positions, vertical G2
__global__ void wino44_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int Cin, int Cout,
                                      int CG, int nchunks, int dgrad, int KS, int short_total) {
    const int main_total = nchunks * kPos * CG * KS * 64;
    const int total = main_total + short_total;
    for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += gridDim.x * blockDim.x) {
        const bool shrt = i0 >= main_total;
        const int i = shrt ? i0 - main_total : i0;
        const int npos = shrt ? 24 : kPos;
        const int ks = i % KS;
        const int lane = (i / KS) & 63;
        int r = i / (64 * KS);
        const int cg = r % CG;
        r /= CG;
        const int p = r % npos;
        const int chunk = r / npos;
        const int co = 16 * cg + (lane & 15);
        const int ci = 4 * KS * chunk + 4 * ks + (lane >> 4);
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            float g[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    g[ky][kx] = dgrad ? w[((size_t)ci * Cout + co) * 9 + (8 - (ky * 3 + kx))]
                                      : w[((size_t)co * Cin + ci) * 9 + ky * 3 + kx];
            const int xi = p / 6, nu = kSlotPos[p - 6 * xi];  // p is a SLOT of the transform row (see bt6h)
            float t[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                t[kx] = shrt ? g2_row(xi, g[0][kx], g[1][kx], g[2][kx]) : g_row(xi, g[0][kx], g[1][kx], g[2][kx]);
            v = g_row(nu, t[0], t[1], t[2]);
        }
        U[i0] = v;
    }
}

// The transforms run on packed FMAs (v_pk_fma_f32: two floats per lane and instruction; round 4).  A 6 x 6 patch is
// three column pairs per row.  The vertical pass is the usual 12-operation form on pairs; the horizontal pass works
// INSIDE a row, the two halves of an instruction taking different constants:
//     (a, c) = (-4, -1) t2 + t4,   (b, e) = (-4, -1) t1 + t3,   (o1, o3) = (a, c) + (1, 2) (b, e),
//     (o2, o4) = (a, c) - (1, 2) (b, e),   (o0, o5) = 4 (t0, t1) - 5 (t2, t3) + (t4, t5)
// -- 6 instructions per row instead of 12 (B^T d B: 72 instead of 144), every source an aligned register pair whose
// halves op_sel picks.  The results come out as the pairs (o0, o5), (o1, o3), (o2, o4): the V image, the U table and
// the accumulators use that SLOT order inside a transform row (slot 6 i + j' = position 6 i + kSlotPos[j']); the
// output transform reads its columns through kPosSlot.
__device__ constexpr int kPosSlot[6] = {0, 2, 4, 3, 5, 1};

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

// A^T m along one axis, for the two channels of a register pair of the D fragment
__device__ __forceinline__ void at6v(const f32x2 m0, const f32x2 m1, const f32x2 m2, const f32x2 m3, const f32x2 m4,
                                     const f32x2 m5, f32x2* y) {
    const f32x2 s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    y[0] = m0 + s1 + s2;
    y[1] = 2.f * d2 + d1;
    y[2] = 4.f * s2 + s1;
    y[3] = (8.f * d2 + d1) + m5;
}

// HELP = 2 (six matrix waves): waves 6, 7 only load and transform patches, together with waves 2, 3 -- the waves of
// the two SIMDs that carry one matrix wave each, so the SIMDs with two matrix waves do nothing but multiply
// FST (forward launches whose result feeds a training-mode BatchNorm): the epilogue also sums, per channel, v and
// v^2 over its outputs -- v the pooled value, or PReLU(y) with g.slope (y itself when it is null) -- one partial
// row [sum v | sum v^2] per workgroup, in the rows the BST form uses
// KS = k-steps of 4 input channels per chunk: 4, or 2 for the two-wave form (32 output channels: 128 threads
// transform the 128 patches of an 8-channel chunk; 37 KB of LDS, four workgroups per CU)
typedef unsigned u32b __attribute__((aligned(1)));

template <int CG, bool BORDER, bool BST, bool POOL = false, int HELP = 0, bool FST = false, int KS = 4, bool PIN = false,
          bool TAIL = false, bool SHORT2 = false>
__global__ void __launch_bounds__((CG + HELP) * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
wino44_conv_kernel(const G4 g, const float* __restrict__ x, const float* __restrict__ U,
                   const float* __restrict__ bias, float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float V[];  // [2][36][64 lanes][4 k-steps]
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (HELP == 2 && !getenv_noflip(g)) {
        // Six matrix waves on four SIMDs (wave w runs on SIMD w & 3): SIMDs 0, 1 carry two of them, SIMDs 2, 3 one and a
        // helper.  Two workgroups share a CU; if the second one takes its helpers FIRST (physical waves 0, 1 = logical
        // 6, 7) its matrix waves double up on SIMDs 2, 3 and every SIMD carries three matrix waves instead of four on
        // two of them.  Workgroups 256 apart are the ones that meet on a CU (8 XCDs x 32 CUs, dispatched in order).
        if ((blockIdx.x >> 8) & 1) wave = (wave + 6) & 7;
    }
    // Workgroup -> (image, tile row, workgroup column).  Workgroups go to the 8 XCDs round-robin (blockIdx % 8) and each
    // XCD has its own L2: in plain row-major order a workgroup's column neighbours -- which share the 128-byte lines its
    // 264-byte row segments start and end in -- sit on other XCDs, and every shared line is fetched once per XCD (round 5:
    // the read requests by size, profiles/r05_pmc_traffic_*, show the class moving 2.0 x its algorithmic bytes).
    // xcd_map = m: in every group of 8 m consecutive ids XCD x takes the m consecutive logical ids x m .. x m + m - 1,
    // i.e. m neighbouring workgroup columns of one tile row.  Measured on the level-14 step (same box, alternating runs,
    // tools/ab_env.py): row-major 20.26-20.41 ms for the class, m = 2 / 4 / 16 / 32 / 64: 20.21 / 20.18 / 20.08-20.15 /
    // 20.12 / 20.29 -- the default is 16.  The order that also keeps the tile rows of a column block on one XCD (negative
    // values: tile row fastest inside the group, or over a whole eighth of the grid) costs 1.5 ms at every group size:
    // 21.7-22.0 ms -- the launch then reads four times as many image rows at once in windows a quarter as wide; what the
    // memory side rewards is few, wide streams, not fewer bytes (the class draws 2.4 TB/s, far from the HBM's limit).
    // (afd::xcd_grouped_id; the tile-row-fastest orders of the measurement above are not in the tree)
    int id = afd::xcd_grouped_id((int)blockIdx.x, (int)gridDim.x, g.xcd_map);
    const int tl = lane & 15;  // tile slot of this lane (transform role and D fragment alike)
    // TAIL (its own instantiation and launch, for images whose last workgroup column holds ONE live tile): tile slot tl of
    // workgroup (n, tg) is tile row 16 tg + tl of tile column tilesX - 1.  There the tile row is per lane -- what is marked
    // "uniform" below in terms of ty is uniform in the other instantiations only (as a per-lane value in every border
    // workgroup it cost the level-8 step 0.25 ms: every workgroup is a border workgroup there) -- and the two uniform
    // shortcuts of the matrix loop (skip5, row0_only: both skip products with zero operands) are off.
    static_assert(!TAIL || BORDER, "tail workgroups take the border form's masks");
    constexpr bool tail = TAIL;
    int ty, n, tx_lane;
    if constexpr (TAIL) {
        n = id / g.tail_groups;
        const int tyl = (id - n * g.tail_groups) * kTiles + tl;
        ty = tyl < g.tilesY ? tyl : g.tilesY - 1;
        tx_lane = tyl < g.tilesY ? g.tilesX - 1 : g.tilesX;  // (a slot past the last tile row is a dead tile)
    } else {
        const int wi = id % g.wxCount;
        id /= g.wxCount;
        ty = g.ty0 + id % g.tyCount;  // the launch covers tile rows ty0 .. ty0 + tyCount - 1
        n = id / g.tyCount;
        const int wx = BORDER ? (wi == 0 ? 0 : g.wgX - 1) : wi + 1;
        tx_lane = wx * kTiles + tl;
    }
    const bool skip5 = !tail && !SHORT2 && g.rows - 4 * ty <= 3;  // uniform: see the matrix loop
    // PIN: the dense gradient has 2 Hp live rows; a tile row whose patches start on the last of them (4 ty - 1 =
    // 2 Hp - 1: the fourth tile row of block 3's 13-row backward-data at level 14, one output row) has a single
    // non-zero patch row, row 0, and column 0 of B^T is (4, 0, 0, 0, 0, 0): only transform row 0 -- positions 0..5 --
    // is non-zero.  Those workgroups run 6 of the 36 products per chunk (they ran 30) and fetch only those operands:
    // block 3's backward-data 5.50 -> 5.23 ms (a shorter transform for them on top measured level, with spills).
    const bool row0_only = !tail && !SHORT2 && PIN && 4 * ty - 1 == 2 * g.Hp - 1;  // uniform
    // F(2x4) tile row (G4::u_short_off): a last tile row with at most two live output rows -- its own instantiation and
    // launches (the kernel sits at its 256-register budget: with both forms in one instantiation every instance spilled
    // 600-970 bytes per thread)
    static_assert(!SHORT2 || !TAIL, "the tail workgroups multiply F(4x4) in every tile slot");
    constexpr bool short2 = SHORT2;

    // transform role (threads 0..255): wave w holds the channels 4 ks + w of the chunk, lane = (ks, tile): its 36
    // values go to V[position][(w * 16 + tile) * 4 + ks] -- a wave's 64 lanes write 64 consecutive floats
    static_assert(HELP == 0 || (HELP == 2 && CG == 6), "helper waves: the six-wave form");
    constexpr int CHK = 4 * KS;          // input channels per chunk
    constexpr int PS = 64 * KS;          // floats per position of a V image / U fragment block
    constexpr int VB = kPos * PS;        // floats per V buffer
    typedef float vk __attribute__((ext_vector_type(KS)));
    static_assert(KS == 4 || (KS == 2 && CG == 2 && HELP == 0), "two k-steps per chunk: the two-wave form");
    const bool xf = HELP ? (wave & 2) != 0 : tid < CHK * kTiles;
    const bool mm = wave < CG;  // matrix wave
    const int tq = HELP ? (wave & 1) + ((wave >> 2) << 1) : (wave & 3);  // which quarter of a chunk's channels
    // the patch of channel 4 ks_t + kq_t of the chunk: KS = 4: wave = kq_t, lane = (ks_t, tile); KS = 2: wave = ks_t,
    // lane = (kq_t, tile)
    const int kq_t = KS == 4 ? tq : (lane >> 4);
    const int ksx = KS == 4 ? (lane >> 4) : (wave & 1);
    const int ch = 4 * ksx + kq_t;
    const int txp = tx_lane;
    const int iy0 = 4 * ty - 1, ix0 = 4 * txp - 1;
    const size_t plane = (size_t)g.H * g.W;
    const float* xn = x + (size_t)n * g.Cin * plane;
    const bool rows_in = iy0 >= 0 && iy0 + 5 < g.H;  // uniform
    float d[6][6];
    const bool fold = !PIN && !BST && g.in_aff != nullptr;      // uniform (forward launches only)
    const bool in_act = fold && g.in_slope != nullptr;
    const float in_a = in_act ? g.in_slope[0] : 1.f;
    const bool in_fast = in_a >= 0.f && in_a <= 1.f;
    f32x2 aff = {0.f, 1.f};  // (mean, invstd) of the channel whose patch is in d
    // PIN: the patch's rows 4 ty - 1 .. + 4 lie in the pooled rows 2 ty - 1 .. + 2, its columns in the pooled columns
    // 2 tx - 1 .. + 2: 4 x 4 pooled values and codes travel (20 registers across the matrix loop instead of 36)
    float pq[PIN ? 4 : 1][4];
    unsigned pcd[PIN ? 4 : 1];
    auto load_patch = [&](int c) {
        if constexpr (PIN) {
            const size_t pplane = (size_t)g.Hp * g.Wp;
            const size_t co = ((size_t)n * g.Cin + c * CHK + ch) * pplane;
            const int pr0 = 2 * ty - 1, pc0 = 2 * txp - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pr = pr0 + r;
                const size_t ro = co + (size_t)(pr < 0 ? 0 : (pr >= g.Hp ? g.Hp - 1 : pr)) * g.Wp;
                if (!BORDER) {
                    // (the fourth value of the last interior tile may lie one past the row: the next row or the slack
                    // the host leaves behind the tensors; it is masked in store_v)
                    const f4u v = *reinterpret_cast<const f4u*>(x + ro + pc0);
                    pq[r][0] = v.x; pq[r][1] = v.y; pq[r][2] = v.z; pq[r][3] = v.w;
                    pcd[r] = *reinterpret_cast<const u32b*>(g.pidx + ro + pc0);
                } else {
                    pcd[r] = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int pc = pc0 + j;
                        const size_t o = ro + (pc < 0 ? 0 : (pc >= g.Wp ? g.Wp - 1 : pc));
                        pq[r][j] = x[o];
                        pcd[r] |= (unsigned)g.pidx[o] << (8 * j);
                    }
                }
            }
            return;
        }
        const float* xc = xn + (size_t)(c * CHK + ch) * plane;
        if (fold) aff = *reinterpret_cast<const f32x2*>(g.in_aff + 2 * (c * CHK + ch));
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const int iy = iy0 + r;
            const int iyc = iy < 0 ? 0 : (iy >= g.H ? g.H - 1 : iy);
            const float* row = xc + (size_t)iyc * g.W;
            if (!BORDER) {
                const f4u v = *reinterpret_cast<const f4u*>(row + ix0);
                const f2u u = *reinterpret_cast<const f2u*>(row + ix0 + 4);
                d[r][0] = v.x; d[r][1] = v.y; d[r][2] = v.z; d[r][3] = v.w; d[r][4] = u.x; d[r][5] = u.y;
            } else {
                // as in wino.hip: the interior's two loads (what they read past a row end is inside the tensor and
                // zeroed in store_v); element loads only where they would leave the tensor
                const float* p6 = row + ix0;
                if (p6 >= x && p6 + 6 <= g.x_end) {
                    const f4u v = *reinterpret_cast<const f4u*>(p6);
                    const f2u u = *reinterpret_cast<const f2u*>(p6 + 4);
                    d[r][0] = v.x; d[r][1] = v.y; d[r][2] = v.z; d[r][3] = v.w; d[r][4] = u.x; d[r][5] = u.y;
                } else {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        const int ix = ix0 + j;
                        d[r][j] = row[ix < 0 ? 0 : (ix >= g.W ? g.W - 1 : ix)];
                    }
                }
            }
        }
    };
    auto store_v = [&](int buf) {
        if constexpr (PIN) {
            const int pr0 = 2 * ty - 1, pc0 = 2 * txp - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool rok = pr0 + r >= 0 && pr0 + r < g.Hp;  // uniform
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bool ok = rok;
                    if (BORDER) ok = ok && txp < g.tilesX && pc0 + j >= 0 && pc0 + j < g.Wp;
                    else if (j == 3) ok = ok && pc0 + 3 < g.Wp;
                    if (BORDER || j == 3) pq[r][j] = ok ? pq[r][j] : 0.f;
                    else if (!rok) pq[r][j] = 0.f;
                }
            }
            // patch row r = image row 4 ty - 1 + r: pooled row (r + 1) >> 1 of the four, window row bit (r + 1) & 1
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int pr = (r + 1) >> 1, pc = (j + 1) >> 1;
                    const unsigned want = (unsigned)((((r + 1) & 1) << 1) | ((j + 1) & 1));
                    const unsigned pos = (pcd[pr] >> (8 * pc)) & 3u;
                    d[r][j] = pos == want ? pq[pr][pc] : 0.f;
                }
        } else {
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
        }
        if (!rows_in || BORDER) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int iy = iy0 + r, ix = ix0 + j;
                    const bool ok = iy >= 0 && iy < g.H && (!BORDER || (txp < g.tilesX && ix >= 0 && ix < g.W));
                    d[r][j] = ok ? d[r][j] : 0.f;
                }
        }
        }
        f32x2 t[6][3];  // t = B^T d, column pairs
        float* vb = V + buf * VB + (kq_t * 16 + tl) * KS + ksx;
        if constexpr (SHORT2) {  // F(2, 3) down the columns (patch rows 0..3), transform rows 0..3 = positions 0..23
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x2 d0 = {d[0][2 * c], d[0][2 * c + 1]}, d1 = {d[1][2 * c], d[1][2 * c + 1]};
                const f32x2 d2 = {d[2][2 * c], d[2][2 * c + 1]}, d3 = {d[3][2 * c], d[3][2 * c + 1]};
                t[0][c] = d0 - d2;
                t[1][c] = d1 + d2;
                t[2][c] = d2 - d1;
                t[3][c] = d1 - d3;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x2 o[3];
                bt6h(t[r][0], t[r][1], t[r][2], o);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    vb[(r * 6 + 2 * q) * PS] = o[q].x;
                    vb[(r * 6 + 2 * q + 1) * PS] = o[q].y;
                }
            }
            return;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            f32x2 o[6];
            bt6v(f32x2{d[0][2 * c], d[0][2 * c + 1]}, f32x2{d[1][2 * c], d[1][2 * c + 1]},
                 f32x2{d[2][2 * c], d[2][2 * c + 1]}, f32x2{d[3][2 * c], d[3][2 * c + 1]},
                 f32x2{d[4][2 * c], d[4][2 * c + 1]}, f32x2{d[5][2 * c], d[5][2 * c + 1]}, o);
#pragma unroll
            for (int r = 0; r < 6; ++r) t[r][c] = o[r];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            f32x2 o[3];
            bt6h(t[r][0], t[r][1], t[r][2], o);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                vb[(r * 6 + 2 * q) * PS] = o[q].x;
                vb[(r * 6 + 2 * q + 1) * PS] = o[q].y;
            }
        }
    };

    f32x4 acc[kPos];
#pragma unroll
    for (int p = 0; p < kPos; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    // U fragments of chunk c, position p (four k-steps): Uw[((c * 36 + p) * CG) * 256]
    // Buffer loads: the table is the buffer, the lane's 4 KS bytes-offset the vector offset, (chunk, position, wave)
    // the scalar offset -- the 36 offsets of a chunk cost scalar adds.  As 64-bit lane addresses (global_load) they
    // cost two vector instructions per load, 72 per chunk INSIDE the matrix loop, where vector and f32 matrix
    // instructions share the issue.  Reads past the table (there are none) would return zero.
    const unsigned ubytes = (unsigned)g.nchunks * (kPos + (g.u_short_off ? 24 : 0)) * CG * PS * 4u;
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, (int)ubytes, 0x00020000);
    const unsigned ulane = (unsigned)lane * KS * 4u;
    // scalar byte offset of this wave's fragments in a position block (of the F(2x4) table in a short tile row) and the
    // positions per chunk of that table
    const unsigned Uw = (unsigned)wave * PS * 4u + (short2 ? (unsigned)g.u_short_off * 4u : 0u);
    const unsigned upos = short2 ? 24u : (unsigned)kPos;
    if (xf) {
        load_patch(0);
        store_v(0);
    }
    __syncthreads();
    // positions in groups of 3: U fragments (L2) are requested two groups ahead of their MFMAs -- the first two groups
    // of a chunk during the previous chunk's last groups, i.e. before the transform -- and V fragments (LDS) one ahead
    vk u[3][3], b[2][3];
    auto load_u = [&](unsigned uc, int grp, int slot) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const unsigned so = uc + (unsigned)(3 * grp + q) * CG * PS * 4u;
            if constexpr (KS == 4) u[slot][q] = __builtin_bit_cast(vk, __builtin_amdgcn_raw_buffer_load_b128(urs, ulane, so, 0));
            else u[slot][q] = __builtin_bit_cast(vk, __builtin_amdgcn_raw_buffer_load_b64(urs, ulane, so, 0));
        }
    };
    if (mm) {
        load_u(Uw, 0, 0);
        load_u(Uw, 1, 1);
    }
    for (int c = 0; c < g.nchunks; ++c) {
        const bool more = c + 1 < g.nchunks;
        // (Round 5: dealt out over the position groups, one load each -- what gained 6 % in the backward-weight kernel --
        // the class went 20.0 -> 21.4 ms: the patch loads then sit between the filter-fragment loads in the in-order return
        // queue and every fragment waits for the patch load in front of it; in one piece at position group 0 / 2 / 10
        // instead of here: 19.89 / 20.37 / 20.73 against 19.84-19.87 ms)
        if (xf && more) load_patch(c + 1);
        const unsigned uc = Uw + (unsigned)c * upos * CG * PS * 4u;
        const unsigned un = Uw + (unsigned)(more ? c + 1 : c) * upos * CG * PS * 4u;  // (the last chunk re-reads its own)
        const float* vb = V + (c & 1) * VB + lane * KS;
        auto load_b = [&](int grp, int slot) {
#pragma unroll
            for (int q = 0; q < 3; ++q) b[slot][q] = *reinterpret_cast<const vk*>(vb + (3 * grp + q) * PS);
        };
        // NR position groups of 3: 12 for F(4x4), 8 for the F(2x4) tile row (its own copy of the loop: the counts are
        // compile-time constants in both).  The filter fragments rotate through three register slots (group g in slot
        // g % 3, requested two groups ahead, the next chunk's first two groups behind the last ones): the walk has NG
        // steps, a multiple of 3 -- for F(2x4) a ninth, empty step -- so that group 0 of the next chunk lands in slot 0.
        auto matrix_loop = [&](auto nr_tag) {
            constexpr int NR = decltype(nr_tag)::value, NG = (NR + 2) / 3 * 3;
            load_b(0, 0);
#pragma unroll
            for (int grp = 0; grp < NG; ++grp) {
                // (row0_only: groups 0 and 1 are all that is multiplied -- the other groups' operands are not fetched)
                if (grp + 2 < NR) {
                    if (!row0_only) load_u(uc, grp + 2, (grp + 2) % 3);
                } else if (grp + 2 >= NG) {
                    load_u(un, grp + 2 - NG, (grp + 2) % 3);
                }
                if (grp + 1 < NR && (grp + 1 < 2 || !row0_only)) load_b(grp + 1, (grp + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                // positions 30..35 (transform row 5) only enter output row 3 of the tile (A^T row 3): a tile row with
                // at most three live output rows -- the last one of the 13- and 6-row level-14 images -- skips them
                if (grp < NR && (grp < 2 || !row0_only) && (grp < 10 || !skip5)) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const int p = 3 * grp + q;
                            acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[grp % 3][q][ks], b[grp & 1][q][ks], acc[p], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (HELP == 0 || mm) matrix_loop(std::integral_constant<int, SHORT2 ? 8 : 12>{});
        if (xf && more) store_v((c + 1) & 1);
        __syncthreads();
    }
    if (HELP && !mm) return;  // helper waves: no accumulators, nothing to write

    // output transform in registers: Y = A^T M A; D fragment: column (tile) = lane & 15, rows (channels) = 4 (lane >> 4) + j
    const int kq = lane >> 4;
    const int oy = 4 * ty;
    const int txe = tx_lane;
    const int ox = 4 * txe;
    // BST: the BatchNorm outputs at channel j's output positions, requested one channel ahead from clamped (always
    // valid) addresses; products are masked where they are used
    // (two channels ahead: behind one channel's 100 vector instructions a 2 us load does not hide)
    f4u zn[BST ? 2 : 1][BST ? 4 : 1];
    unsigned cn[BST ? 2 : 1][BST ? 4 : 1];
    auto load_xhat = [&](int j) {
        const int co = min(16 * wave + 4 * kq + j, g.Cout - 1);
        const int oxq = min(ox, g.W - 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oyr = min(oy + r, g.H - 1);
            const size_t o = (((size_t)n * g.Cout + co) * g.H + oyr) * g.W + oxq;
            zn[j & 1][r] = *reinterpret_cast<const f4u*>(g.bn_in + o);
            if (g.bn_codes) cn[j & 1][r] = *reinterpret_cast<const u32b*>(g.bn_codes + o);
        }
    };
    static_assert(!(BST && FST), "one statistics epilogue per launch");
    float sg[(BST || FST) ? 4 : 1], sgv[(BST || FST) ? 4 : 1];
    // bn_in == nullptr: only the sums of the result are wanted (the caller takes sum(g * x) from the weight gradient:
    // sum_px g[ci] x[ci] = sum_{co,k} w[co][ci][k] dw[co][ci][k], afd_conv_weight_dot) -- no loads, 3.5 GB less per launch
    const bool have_x = BST && g.bn_in != nullptr;  // uniform
    const bool bnb = BST && have_x && g.bn_tab != nullptr;  // uniform: see G4::bn_tab
    const bool bn_act = bnb && g.bn_slope != nullptr;
    const float bn_a = bn_act ? g.bn_slope[0] : 1.f;
    const bool bn_pool = bnb && g.bn_codes != nullptr;
    const float bn_inva = (bn_act && bn_a != 0.f) ? 1.f / bn_a : 0.f;
    if constexpr (BST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) zn[0][r] = zn[1][r] = f4u{0.f, 0.f, 0.f, 0.f};
        if (have_x) {
            load_xhat(0);
            load_xhat(1);
        }
    }
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
    // Y = A^T M A for the channels 2 jp, 2 jp + 1 of the D fragment at once: acc[p][2 jp], acc[p][2 jp + 1] are a
    // register pair, so both passes are packed instructions (100 per pair instead of 100 per channel)
    f32x2 o2[4][4];
    {
        f32x2 sp[4][6];  // A^T M: rows of outputs x 6 slot columns
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            f32x2 o[4];
            auto A = [&](int p) { return f32x2{acc[p][2 * jp], acc[p][2 * jp + 1]}; };
            if constexpr (SHORT2) {  // A2^T (2 x 4) down the column; output rows 2, 3 of the tile do not exist
                o[0] = A(q) + A(6 + q) + A(12 + q);
                o[1] = A(6 + q) - A(12 + q) - A(18 + q);
                o[2] = o[3] = f32x2{0.f, 0.f};
            } else {
                at6v(A(q), A(6 + q), A(12 + q), A(18 + q), A(24 + q), A(30 + q), o);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[r][q] = o[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            at6v(sp[r][kPosSlot[0]], sp[r][kPosSlot[1]], sp[r][kPosSlot[2]], sp[r][kPosSlot[3]], sp[r][kPosSlot[4]],
                 sp[r][kPosSlot[5]], o2[r]);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = 2 * jp + h;
        const int co = 16 * wave + 4 * kq + j;
        f4u zq[BST ? 4 : 1];
        unsigned cq[BST ? 4 : 1];
        if constexpr (BST) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                zq[r] = zn[j & 1][r];
                cq[r] = cn[j & 1][r];
            }
            if (j + 2 < 4 && have_x) load_xhat(j + 2);
            sg[j] = sgv[j] = 0.f;
        }
        if constexpr (FST) sg[j] = sgv[j] = 0.f;
        const float bv = (bias && co < g.Cout) ? bias[co] : 0.f;
        f32x4 bt = {0.f, 1.f, 0.f, 0.f};
        if (BST && bnb) bt = *reinterpret_cast<const f32x4*>(g.bn_tab + 4 * min(co, g.Cout - 1));
        if constexpr (POOL) {
            // same order and tie rule as prelu_pool_fwd_kernel (nn.hip): first maximum wins
            if (co < g.Cout && txe < g.tilesX) {
                const float a = g.slope[0];
                auto act = [a](float z) { return z > 0.f ? z : a * z; };
                const int PH = g.rows >> 1, PW = g.cols >> 1;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    float o0[4], o1[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o0[q] = o2[2 * pr][q][h];
                        o1[q] = o2[2 * pr + 1][q][h];
                    }
                    const int py = 2 * ty + pr;
                    if (py < PH) {
                        float ub[2];
                        unsigned cb[2];
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc) {
                            const float y00 = o0[2 * pc] + bv, y01 = o0[2 * pc + 1] + bv;
                            const float y10 = o1[2 * pc] + bv, y11 = o1[2 * pc + 1] + bv;
                            float best, zb;
                            unsigned bi;
                            if (a > 0.f) {
                                // (uniform) a positive slope makes PReLU strictly increasing: the window's first maximum
                                // of PReLU(y) is the first maximum of y -- two max instructions and an equality chain, one
                                // PReLU, instead of four PReLUs and three compare / select rounds (conv1.hip does the same)
                                zb = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11));
                                bi = y00 == zb ? 0u : (y01 == zb ? 1u : (y10 == zb ? 2u : 3u));
                                best = act(zb);
                            } else {
                                best = act(y00); zb = y00; bi = 0;
                                float v = act(y01);
                                if (v > best) { best = v; bi = 1; zb = y01; }
                                v = act(y10);
                                if (v > best) { best = v; bi = 2; zb = y10; }
                                v = act(y11);
                                if (v > best) { best = v; bi = 3; zb = y11; }
                            }
                            ub[pc] = best;
                            cb[pc] = bi | (zb <= 0.f ? 4u : 0u);
                        }
                        const int px = 2 * txe;
                        const size_t o = (((size_t)n * g.Cout + co) * PH + py) * PW + px;
                        if constexpr (FST) {
                            const float u1 = px + 1 < PW ? ub[1] : 0.f, u0 = px < PW ? ub[0] : 0.f;
                            sg[j] += u0 + u1;
                            sgv[j] += fmaf(u0, u0, u1 * u1);
                        }
                        if (px + 1 < PW) {
                            f2u uv = {ub[0], ub[1]};
                            *reinterpret_cast<f2u*>(g.u + o) = uv;
                            g.idx[o] = (unsigned char)cb[0];
                            g.idx[o + 1] = (unsigned char)cb[1];
                        } else if (px < PW) {
                            g.u[o] = ub[0];
                            g.idx[o] = (unsigned char)cb[0];
                        }
                    }
                }
            }
        } else
        if (co < g.Cout && txe < g.tilesX) {
            float* yo = y + (((size_t)n * g.Cout + co) * g.H + oy) * g.W + ox;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = o2[r][q][h];
                if (oy + r < g.rows) {
                    if constexpr (BST) {
                        // (interior columns: ox + 3 < cols and the clamped column is ox itself; the edge workgroups'
                        // clamped loads start at W - 4, so output column ox + q sits at element ox + q - (W - 4))
                        float gq[4], zv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            gq[q] = (!BORDER || ox + q < g.cols) ? o[q] + bv : 0.f;
                            if (!BORDER) {
                                zv[q] = zq[r][q];
                            } else {
                                const int e = ox + q - min(ox, g.W - 4);
                                zv[q] = e == 0 ? zq[r][0] : e == 1 ? zq[r][1] : e == 2 ? zq[r][2] : e == 3 ? zq[r][3] : 0.f;
                            }
                        }
                        if (bnb) {  // uniform: the BatchNorm (+ PReLU) backward of the result, as bn_bwd_apply_kernel does it
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float zz = zv[q];
                                // (pooled: zz is the pool's output, already through its PReLU; the code says which side)
                                const int e = BORDER ? ox + q - min(ox, g.W - 4) : q;
                                const bool neg = bn_pool ? ((unsigned)e < 4u && ((cq[r] >> (8 * e)) & 4u) != 0u) : zz <= 0.f;
                                const float v = (bn_act && !bn_pool) ? (zz > 0.f ? zz : bn_a * zz) : zz;
                                const float xh = (v - bt[0]) * bt[1];
                                float gg = bt[1] * (gq[q] - bt[2] - xh * bt[3]);
                                const bool live = !BORDER || ox + q < g.cols;
                                gg = live ? gg : 0.f;
                                if (bn_act && neg) {
                                    sgv[j] = fmaf(gg, bn_pool ? zz * bn_inva : zz, sgv[j]);
                                    gg *= bn_a;
                                }
                                sg[j] += gg;
                                o[q] = gg - bv;  // (stored as o + bv below)
                            }
                        } else {
                            sg[j] += (gq[0] + gq[1]) + (gq[2] + gq[3]);
                            sgv[j] += fmaf(gq[0], zv[0], gq[1] * zv[1]) + fmaf(gq[2], zv[2], gq[3] * zv[3]);
                        }
                    }
                    if constexpr (FST) {
                        const bool act = g.slope != nullptr;
                        const float a = act ? g.slope[0] : 1.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v = o[q] + bv;
                            v = (act && v <= 0.f) ? a * v : v;
                            v = (!BORDER || ox + q < g.cols) ? v : 0.f;
                            sg[j] += v;
                            sgv[j] = fmaf(v, v, sgv[j]);
                        }
                    }
                    if (!BORDER || ox + 3 < g.cols) {
                        f4u v = {o[0] + bv, o[1] + bv, o[2] + bv, o[3] + bv};
                        *reinterpret_cast<f4u*>(yo + (size_t)r * g.W) = v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (ox + q < g.cols) yo[(size_t)r * g.W + q] = o[q] + bv;
                    }
                }
            }
        }
    }
    }
    if constexpr (BST || FST) {
        // a channel's outputs of this workgroup sit in the 16 lanes of one quarter wave
        const int co_pad = (g.Cout + 31) / 32 * 32;
        float* row = g.stat_part + ((size_t)g.part_row0 + blockIdx.x) * (2 * co_pad);
        const bool live = txe < g.tilesX;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a1 = live ? sg[j] : 0.f, a2 = live ? sgv[j] : 0.f;
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                a1 += __shfl_xor(a1, off, 64);
                a2 += __shfl_xor(a2, off, 64);
            }
            const int co = 16 * wave + 4 * kq + j;
            if (tl == 0 && co < g.Cout) {
                row[co] = a1;
                row[co_pad + co] = a2;
            }
        }
    }
}

}  // namespace
namespace afd {
// Tail workgroups per image (0 = the plain border form): the last workgroup column holds exactly one live tile and there
// are at least two tile rows to pack (AFD_NO_WINO44_TAIL=1 keeps one workgroup per tile row there)
int wino44_tail_groups(int tilesX, int tilesY) {
    static const bool off = getenv("AFD_NO_WINO44_TAIL") != nullptr;
    const int wgX = (tilesX + kTiles - 1) / kTiles;
    if (off || wgX < 2 || tilesX - kTiles * (wgX - 1) != 1 || tilesY < 2) return 0;
    return (tilesY + kTiles - 1) / kTiles;
}
}  // namespace afd
namespace {

template <int CG, bool BST, bool POOL = false, int HELP = 0, bool FST = false, int KS = 4, bool PIN = false>
int launch44(G4 g, const float* x, const float* U, const float* bias, float* y, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * kPos * 64 * KS * sizeof(float);
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino44_conv_kernel<CG, false, BST, POOL, HELP, FST, KS, PIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino44_conv_kernel<CG, true, BST, POOL, HELP, FST, KS, PIN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino44_conv_kernel<CG, true, BST, POOL, HELP, FST, KS, PIN, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino44_conv_kernel<CG, false, BST, POOL, HELP, FST, KS, PIN, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino44_conv_kernel<CG, true, BST, POOL, HELP, FST, KS, PIN, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "winograd 4x4 conv: %s", hipGetErrorString(e));
        attr.mark();
    }
    g.wgX = (g.tilesX + kTiles - 1) / kTiles;
    g.tail_groups = afd::wino44_tail_groups(g.tilesX, g.tilesY);
    // the last tile row as F(2x4) (wino44_run decided: u_short_off): never the row0_only tile row of a pooled-input launch
    const bool last_row0 = PIN && 4 * (g.tilesY - 1) - 1 == 2 * g.Hp - 1;
    if (last_row0) g.u_short_off = 0;
    const bool has_short = g.u_short_off != 0;
    const int ty_main = has_short ? g.tilesY - 1 : g.tilesY;
    const int inner = g.wgX > 2 ? g.wgX - 2 : 0;
    const int edge = g.wgX >= 2 ? 2 : 1;
    const int cols_main = g.tail_groups ? g.wgX - 1 : g.wgX;  // workgroup columns outside the tail launch
    // issued flops: 36 GEMMs [16 CG x Cin] x [Cin x 16 tiles] per workgroup (tail workgroups: all 36 in every slot)
    {
        double pos_main = 0.0;
        for (int ty = 0; ty < ty_main; ++ty) {
            const bool row0_only = PIN && 4 * ty - 1 == 2 * g.Hp - 1;
            const bool skip5 = g.rows - 4 * ty <= 3;
            pos_main += row0_only ? 6.0 : (skip5 ? 30.0 : (double)kPos);
        }
        if (has_short) pos_main += 24.0;
        const double slots = pos_main * kTiles * cols_main + (double)kPos * kTiles * g.tail_groups;
        afd::timing_annotate(2.0 * slots * (16.0 * CG) * (double)g.N * g.Cin, -1.0);
    }
    if ((long)g.N * g.tilesY * (inner > edge ? inner : edge) > 0x7fffffffL)
        return afd::fail(AFD_ERR_UNSUPPORTED, "winograd 4x4 conv: grid too large");
    const dim3 block((CG + HELP) * 64);
    g.part_row0 = 0;
    // tile rows 0 .. ty_main - 1: interior workgroup columns (every patch column inside the image: 6 columns from 4 tx - 1),
    // then the border ones -- the left one and the right one, or, when the right one holds a single live tile, the left
    // one here and that tile column in its own launch, 16 tile rows per workgroup
    if (ty_main > 0) {
        const long rows = (long)g.N * ty_main;
        g.ty0 = 0;
        g.tyCount = ty_main;
        if (inner > 0) {
            g.wxCount = inner;
            hipLaunchKernelGGL((wino44_conv_kernel<CG, false, BST, POOL, HELP, FST, KS, PIN>), dim3((unsigned)(rows * inner)), block, lds, s, g, x, U,
                               bias, y);
            g.part_row0 += (int)(rows * inner);
        }
        g.wxCount = g.tail_groups ? 1 : edge;
        hipLaunchKernelGGL((wino44_conv_kernel<CG, true, BST, POOL, HELP, FST, KS, PIN>), dim3((unsigned)(rows * g.wxCount)), block, lds, s, g, x, U,
                           bias, y);
        g.part_row0 += (int)(rows * g.wxCount);
    }
    if (g.tail_groups) {  // (all tile rows, the short one included: F(4x4) there)
        g.ty0 = 0;
        g.tyCount = g.tilesY;
        hipLaunchKernelGGL((wino44_conv_kernel<CG, true, BST, POOL, HELP, FST, KS, PIN, true>), dim3((unsigned)((long)g.N * g.tail_groups)), block, lds,
                           s, g, x, U, bias, y);
        g.part_row0 += (int)((long)g.N * g.tail_groups);
    }
    if (has_short) {  // the last tile row, F(2x4)
        g.ty0 = g.tilesY - 1;
        g.tyCount = 1;
        if (inner > 0) {
            g.wxCount = inner;
            hipLaunchKernelGGL((wino44_conv_kernel<CG, false, BST, POOL, HELP, FST, KS, PIN, false, true>), dim3((unsigned)((long)g.N * inner)), block, lds,
                               s, g, x, U, bias, y);
            g.part_row0 += (int)((long)g.N * inner);
        }
        g.wxCount = g.tail_groups ? 1 : edge;
        hipLaunchKernelGGL((wino44_conv_kernel<CG, true, BST, POOL, HELP, FST, KS, PIN, false, true>), dim3((unsigned)((long)g.N * g.wxCount)), block, lds,
                           s, g, x, U, bias, y);
    }
    return afd::check_launch("wino44_conv_kernel");
}

}  // namespace

namespace afd {

// narrowest image the F(4x4) kernels take (a workgroup is 16 tiles = 64 columns of a tile row).  Until round 3
// this was 256: border workgroups fetched their patches element by element, and on the 64- / 129-wide level-8 and
// STFT images every workgroup is a border workgroup.  With the interior's vector loads there (load_patch) the
// level-8 / STFT steps gain 10 % from F(4x4): 7.6 / 6.8 / 7.3 -> 6.8 / 6.2 / 6.6 ms.
static int min_width44() { return 48; }

bool wino44_applicable(int Cin, int H, int W, int Cout) {
    if (getenv("AFD_NO_WINOGRAD") || getenv("AFD_NO_WINO44")) return false;
    // four waves (64 output channels): block 3's backward-data at level 14, 7.7 -> 5.9 ms; six waves (96 channels)
    // would leave two SIMDs with one wave.  Rows: ceil(H / 4) * 36 matrix products against ceil(H / 2) * 32
    // eight waves (128 output channels): block 4 forward 3.7 -> 2.8 ms, block 5 backward-data 1.6 -> 1.4 ms
    // (measured on the level-14 geometries: coif4, 13 / 6 rows, and sym5, 6 / 3 rows -- the sym5 step 33.3 -> 30.8 ms)
    // 96 channels (block 4's backward-data; conv3x3_run sends forward layers of that width elsewhere): six matrix
    // waves + two helper waves, 3.57 -> 3.38 ms with the BatchNorm sums (level without the helpers)
    // 32 channels (block 5 forward, block 6 backward-data): two waves, 8-channel chunks: 1.52 -> 1.38 and 0.86 -> 0.78 ms
    const bool c32 = Cout == 32 && Cin % 8 == 0;
    if (!c32 && (Cin % kCh != 0 || (Cout != 64 && Cout != 96 && Cout != 128))) return false;
    if (W < min_width44() || H < 3) return false;
    return (size_t)H * W < 0x7fffffffULL;
}

// the pooled forward layer with 96 output channels (block 3): six waves, i.e. two SIMDs carry two waves -- still
// ahead of the F(2x2) kernel there (6.8 -> ? ms at level 14)
bool wino44_pool_applicable(int Cin, int H, int W, int Cout) {
    if (getenv("AFD_NO_WINOGRAD") || getenv("AFD_NO_WINO44")) return false;
    // (and block 6, 32 -> 64 channels on four waves: 0.82 -> 0.69 ms)
    if (Cin % kCh != 0 || (Cout != 96 && Cout != 64)) return false;
    if (W < min_width44() || H < 4) return false;
    return (size_t)H * W < 0x7fffffffULL;
}

size_t wino44_workspace_bytes(int Cin, int Cout) {
    const size_t cg = (size_t)(Cout + 15) / 16;
    return (size_t)(Cin / 4) * (kPos + 24) * cg * 64 * sizeof(float);  // (either chunk depth; F(4x4) table + F(2x4) table)
}

// workgroups (= partial rows of the statistics epilogue) of a launch pair over N images of H x W outputs
long wino44_stat_rows(int N, int H, int W) {
    const int tilesX = (W + 3) / 4, tilesY = (H + 3) / 4;
    const int wgX = (tilesX + kTiles - 1) / kTiles;
    const int tg = wino44_tail_groups(tilesX, tilesY);
    return tg ? (long)N * (tilesY * (wgX - 1) + tg) : (long)N * tilesY * wgX;
}

int wino44_run(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
               int Cout, int dgrad, int out_rows, int out_cols, void* ws, size_t ws_bytes, hipStream_t s,
               const float* bn_in, float* stat_part, const float* slope, float* u, unsigned char* idx, int fwd_stats,
               const unsigned char* pooled_codes, const float* in_aff, const float* in_slope, const float* bn_tab,
               const float* bn_slope, const unsigned char* bn_codes) {
    if (!ws || ws_bytes < wino44_workspace_bytes(Cin, Cout))
        return afd::fail(AFD_ERR_WORKSPACE, "winograd 4x4 conv: workspace too small");
    G4 g{};
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.H = H; g.W = W;
    g.rows = out_rows < H ? out_rows : H;
    g.cols = out_cols < W ? out_cols : W;
    g.tilesX = (g.cols + 3) / 4;
    g.tilesY = (g.rows + 3) / 4;
    const int KS = Cout <= 32 ? 2 : 4;
    g.nchunks = Cin / (4 * KS);
    g.bn_in = bn_in; g.stat_part = stat_part;
    g.x_end = x + (size_t)N * Cin * H * W;
    g.slope = slope; g.u = u; g.idx = idx;
    g.pidx = pooled_codes; g.Hp = H / 2; g.Wp = W / 2;
    g.in_aff = in_aff; g.in_slope = in_slope;
    g.noflip = 0;
    g.xcd_map = 16;
    g.bn_tab = bn_tab; g.bn_slope = bn_slope; g.bn_codes = bn_tab ? bn_codes : nullptr;
    if (bn_tab && (!dgrad || !bn_in || !stat_part || fwd_stats))
        return afd::fail(AFD_ERR_ARG, "winograd 4x4 conv: the BatchNorm backward epilogue belongs to backward-data launches "
                                      "with the statistics epilogue and the BatchNorm's input");
    if (in_aff && (dgrad || pooled_codes))
        return afd::fail(AFD_ERR_ARG, "winograd 4x4 conv: the input fold is built for forward launches");
    if (pooled_codes && (!dgrad || !stat_part || fwd_stats || u || g.rows != H || g.cols != W || (Cout != 64 && Cout != 32)))
        return afd::fail(AFD_ERR_UNSUPPORTED, "winograd 4x4 conv: pooled input is built for the backward-data launches with "
                                              "BatchNorm sums of 64 / 32 result channels");
    if (u && (g.rows != 2 * (H / 2) || g.cols != 2 * (W / 2) || !slope || !idx || (stat_part && !fwd_stats)))
        return afd::fail(AFD_ERR_ARG, "winograd 4x4 conv + pool: bad arguments");
    if (stat_part && !fwd_stats && (g.rows != H || g.cols != W))
        return afd::fail(AFD_ERR_ARG, "winograd 4x4 conv: statistics epilogue on a cropped output");
    if (fwd_stats && (!stat_part || dgrad || (!u && (g.rows != H || g.cols != W))))
        return afd::fail(AFD_ERR_ARG, "winograd 4x4 conv: forward statistics: bad arguments");
    const int CG = (Cout + 15) / 16;
    if (!x || !w || (!y && !u)) return afd::fail(AFD_ERR_ARG, "winograd 4x4 conv: null pointer");
    float* U = static_cast<float*>(ws);
    const int main_total = g.nchunks * kPos * CG * KS * 64;
    // a last tile row with at most two live output rows runs F(2x4) from a table of its own (AFD_NO_WINO44_SHORT=1: F(4x4))
    static const bool no_short = getenv("AFD_NO_WINO44_SHORT") != nullptr;
    const bool has_short = !no_short && g.tilesY >= 1 && g.rows - 4 * (g.tilesY - 1) <= 2;
    const int short_total = has_short ? g.nchunks * 24 * CG * KS * 64 : 0;
    g.u_short_off = has_short ? main_total : 0;
    const int total = main_total + short_total;
    afd::ScopedTiming timing(AFD_K_CONV_WINOGRAD, 2.0 * N * Cout * (double)g.rows * g.cols * Cin * 9, s);
    hipLaunchKernelGGL(wino44_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, U, Cin, Cout, CG,
                       g.nchunks, dgrad, KS, short_total);
    int rc = afd::check_launch("wino44_weights_kernel");
    if (rc) return rc;
    {   // every tensor the launch touches, once: the pooled forms move 4 + 1 bytes per 2x2 window, the epilogues that
        // read the BatchNorm's input (statistics with xhat, BatchNorm backward) add that tensor (and its pool codes)
        const double px_in = (double)H * W, px_out = (double)g.rows * g.cols;
        double b = pooled_codes ? 5.0 * Cin * (double)(H / 2) * (W / 2) : 4.0 * Cin * px_in;
        b += u ? 5.0 * Cout * (double)(g.rows / 2) * (g.cols / 2) : 4.0 * Cout * px_out;
        if (bn_in) b += (bn_tab && bn_codes ? 5.0 : 4.0) * Cout * px_out;
        timing.bytes(b * N);
    }
    if (u) {  // pooled forward: u / idx are written, y is not used
        if (CG == 6 && fwd_stats) return launch44<6, false, true, 2, true>(g, x, U, bias, y, s);
        if (fwd_stats) return afd::fail(AFD_ERR_UNSUPPORTED, "winograd 4x4 conv + pool: statistics for Cout %d", Cout);
        if (CG == 6)
            return launch44<6, false, true, 2>(g, x, U, bias, y, s);
        if (CG == 4) return launch44<4, false, true>(g, x, U, bias, y, s);
        return afd::fail(AFD_ERR_UNSUPPORTED, "winograd 4x4 conv + pool: Cout %d", Cout);
    }
    if (!y) return afd::fail(AFD_ERR_ARG, "winograd 4x4 conv: null output");
    if (fwd_stats) {
        if (CG == 8) return launch44<8, false, false, 0, true>(g, x, U, bias, y, s);
        if (CG == 2) return launch44<2, false, false, 0, true, 2>(g, x, U, bias, y, s);
        return afd::fail(AFD_ERR_UNSUPPORTED, "winograd 4x4 conv: forward statistics for Cout %d", Cout);
    }
    if (pooled_codes)
        return CG == 4 ? launch44<4, true, false, 0, false, 4, true>(g, x, U, bias, y, s)
                       : launch44<2, true, false, 0, false, 2, true>(g, x, U, bias, y, s);
    if (CG == 2)
        return stat_part ? launch44<2, true, false, 0, false, 2>(g, x, U, bias, y, s)
                         : launch44<2, false, false, 0, false, 2>(g, x, U, bias, y, s);
    if (CG == 4) return stat_part ? launch44<4, true>(g, x, U, bias, y, s) : launch44<4, false>(g, x, U, bias, y, s);
    if (CG == 6) return stat_part ? launch44<6, true, false, 2>(g, x, U, bias, y, s) : launch44<6, false, false, 2>(g, x, U, bias, y, s);
    if (CG == 8) return stat_part ? launch44<8, true>(g, x, U, bias, y, s) : launch44<8, false>(g, x, U, bias, y, s);
    return afd::fail(AFD_ERR_UNSUPPORTED, "winograd 4x4 conv: Cout %d", Cout);
}

}  // namespace afd
