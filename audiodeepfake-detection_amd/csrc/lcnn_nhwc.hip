// LCNN evaluation forward with bf16 STORAGE (BASELINE.json configs[4]: "STFT + LCNN, bf16"): activations travel
// between the layers as channels-last bf16 tensors [N][H][W][C], weights are converted once per model state.
//
// Reference: LCNN.forward of src/audiofakedetect/models.py:68-131 -- nine times Conv2d -> MaxFeatureMap2D
// (:161-209) [-> MaxPool2d(2, 2)] [-> BatchNorm2d(affine=False), evaluation mode], then two BLSTM layers.
//
// Why channels-last: with k ordered (tap, input channel) the eight consecutive k a lane feeds to
// v_mfma_f32_32x32x16_bf16 are eight consecutive channels of ONE input pixel -- a single 16-byte load, no
// per-element index arithmetic (lcnn_bf16.hip gathers its im2col tile element by element from fp32 NCHW: its
// convolutions ran at 4 % of the bf16 peak).  Evaluation-mode BatchNorm is a positive per-channel scale and a
// shift, which commute with the channel-pair maximum and with max-pooling: they are folded into the weights and
// bias of the convolution in front (both halves of the pair), so no BatchNorm pass is left.
//
//   afd_lcnn_prep_conv_bf16     w [Cout][Cin][K][K] f32 (+ bias, + the BatchNorm that follows) -> bf16 rows in MFMA
//                               tile order (the two halves of the feature-map pairs on separate 32-row tiles),
//                               k = (ky K + kx) Cin + ci, and the folded fp32 bias
//   afd_lcnn_conv1_nhwc_bf16    first layer (Cin = 1, 5x5): fp32 image in, bf16 [N][H][W][32] out
//   afd_lcnn_conv_nhwc_bf16     1x1 / 3x3 layers: workgroup = 128 output pixels x all rows; per tap the weight slice
//                               is staged in LDS (shared by the four waves), the pixel fragments come straight from
//                               global memory as 16-byte loads; epilogue: bias, pair maximum, four channels = one
//                               8-byte store
//   afd_lcnn_pool_nhwc_bf16     MaxPool2d(2, 2) on [N][H][W][C] bf16, 8 channels per thread; optionally writes fp32
//                               (the last pool feeds the LSTM, whose input is [N][H'][W' C] -- the input weights'
//                               columns are permuted to that order by the host)
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThr = 256;
constexpr int kPixT = 128;  // output pixels per workgroup (32 per wave)

inline int round_up_i(int v, int m) { return (v + m - 1) / m * m; }

// rows: [0, HP) first halves, [HP, 2 HP) second halves of the feature-map pairs (HP = half padded to 32)
__global__ void lcnn_prep_conv_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                      const float* __restrict__ bn_mean, const float* __restrict__ bn_var, float eps,
                                      __bf16* __restrict__ wb, float* __restrict__ bb, int Cout, int Cin, int K,
                                      int Kpad, int HP) {
    const int half = Cout / 2;
    const int total = 2 * HP * Kpad;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int row = e / Kpad, k = e - row * Kpad;
        const int r = row < HP ? row : row - HP;
        const int co = row < HP ? r : half + r;
        float v = 0.f;
        if (r < half && k < K * K * Cin) {
            const int t = k / Cin, ci = k - t * Cin;
            const float scale = bn_var ? rsqrtf(bn_var[r] + eps) : 1.f;
            v = w[((size_t)co * Cin + ci) * K * K + t] * scale;
        }
        wb[e] = (__bf16)v;
        if (k == 0) {
            float b = 0.f;
            if (r < half) {
                const float scale = bn_var ? rsqrtf(bn_var[r] + eps) : 1.f;
                b = ((bias ? bias[co] : 0.f) - (bn_mean ? bn_mean[r] : 0.f)) * scale;
            }
            bb[row] = b;
        }
    }
}

struct LG {
    int N, H, W, Cin, Cout, pad, Ho, Wo, tiles, half, HP, Kpad;
    int PH, PW;  // POOL forms: the pooled image (floor(Ho / 2) x floor(Wo / 2)); tiles then counts 32-window tiles
};

// POOL forms (round 5): MaxPool2d(2, 2) in the convolution's epilogue.  A lane's pixel is position (r & 3) of window
// (r >> 2) of its wave -- the four pixels of a 2 x 2 window sit in the four lanes of a quad, so the pool is two DPP
// quad permutes + maxima per value in the D fragment, and the pooled pixel's channels leave from all four lanes (lane
// `pos` stores channel group `pos`): the pre-pool tensor (1.7 GB behind the first layer at B = 1024) is never written.
// max(round(a), round(b)) = round(max(a, b)): bit-identical to the separate pool on the rounded tensor.
__device__ __forceinline__ float quad_max(float v) {
    const int i = __builtin_bit_cast(int, v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    const float m = a > v ? a : v;
    const int j = __builtin_bit_cast(int, m);
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(j, j, 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    return b > m ? b : m;
}

// y: bf16 (or, OUT_F32, bf16-rounded fp32) [window][half] of this pooled pixel
template <int MT, bool OUT_F32>
__device__ __forceinline__ void mfm_pool_store(const f32x16 (&acc)[MT], const float* __restrict__ bb, void* __restrict__ yv,
                                               int half, int HP, int h, int pos) {
#pragma unroll
    for (int i = 0; i < MT / 2; ++i) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int c = 32 * i + 8 * (q >> 2) + 4 * h + (q & 3);
            const float a = acc[i][q] + bb[c];
            const float b = acc[i + MT / 2][q] + bb[HP + c];
            v[q] = quad_max(b > a ? b : a);
        }
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pos == 0 ? v[j] : (pos == 1 ? v[4 + j] : (pos == 2 ? v[8 + j] : v[12 + j]));
        const int c0 = 32 * i + 8 * pos + 4 * h;
        if (c0 < half) {
            bf16x4 ob;
#pragma unroll
            for (int j = 0; j < 4; ++j) ob[j] = (__bf16)o[j];
            if (OUT_F32) {
                float* y = static_cast<float*>(yv) + c0;
                *reinterpret_cast<float4*>(y) = make_float4((float)ob[0], (float)ob[1], (float)ob[2], (float)ob[3]);
            } else {
                *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(yv) + c0) = ob;
            }
        }
    }
}

// pixel of a lane: plain forms p = tile * 128 + 32 wave + r; POOL forms window = tile * 32 + 8 wave + (r >> 2)
template <bool POOL>
__device__ __forceinline__ bool lane_pixel(const LG& g, int tile, int wave, int r, int& oy, int& ox, size_t& out_px) {
    if (POOL) {
        const int wdw = tile * 32 + 8 * wave + (r >> 2);
        const bool pv = wdw < g.PH * g.PW;
        const int py = pv ? wdw / g.PW : 0, px = pv ? wdw - py * g.PW : 0;
        oy = 2 * py + ((r >> 1) & 1);
        ox = 2 * px + (r & 1);
        out_px = (size_t)wdw;
        return pv;
    }
    const int p = tile * kPixT + 32 * wave + r;
    const bool pv = p < g.Ho * g.Wo;
    oy = pv ? p / g.Wo : 0;
    ox = pv ? p - oy * g.Wo : 0;
    out_px = (size_t)p;
    return pv;
}

// D fragment of a 32x32 tile: column (pixel) = lane & 31, row (channel) = (q & 3) + 8 (q >> 2) + 4 (lane >> 5).
// Pair maximum of tile i (first halves) and tile i + MT/2 (second halves), bias included; four consecutive channels
// of a pixel leave as one 8-byte store.
template <int MT>
__device__ __forceinline__ void mfm_store(const f32x16 (&acc)[MT], const float* __restrict__ bb, __bf16* __restrict__ yp,
                                          int half, int HP, int h) {
#pragma unroll
    for (int i = 0; i < MT / 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c0 = 32 * i + 8 * g + 4 * h;
            if (c0 < half) {  // half is a multiple of 4 (16 in this model)
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = acc[i][4 * g + j] + bb[c0 + j];
                    const float b = acc[i + MT / 2][4 * g + j] + bb[HP + c0 + j];
                    o[j] = (__bf16)(b > a ? b : a);
                }
                *reinterpret_cast<bf16x4*>(yp + c0) = o;
            }
        }
}

// first layer: Cin = 1, K x K taps (25 -> Kpad 32); x fp32 [N][H][W]
template <int MT, bool POOL>
__global__ void __launch_bounds__(kThr) lcnn_conv1_kernel(const LG g, const float* __restrict__ x,
                                                          const __bf16* __restrict__ wb, const float* __restrict__ bb,
                                                          __bf16* __restrict__ y, int K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / g.tiles;
    int oy, ox;
    size_t opx;
    const bool pv = lane_pixel<POOL>(g, blockIdx.x - n * g.tiles, wave, r, oy, ox, opx);
    const size_t out_plane = POOL ? (size_t)g.PH * g.PW : (size_t)g.Ho * g.Wo;
    const float* xn = x + (size_t)n * g.H * g.W;
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int k0 = 0; k0 < g.Kpad; k0 += 16) {
        bf16x8 b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + 8 * h + e;
            const int ky = k / K, kx = k - ky * K;
            const int iy = oy + ky - g.pad, ix = ox + kx - g.pad;
            float f = 0.f;
            if (pv && k < K * K && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) f = xn[(size_t)iy * g.W + ix];
            b[e] = (__bf16)f;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(wb + (size_t)(32 * mt + r) * g.Kpad + k0 + 8 * h);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mt], 0, 0, 0);
        }
    }
    if (pv) {
        __bf16* yp = y + ((size_t)n * out_plane + opx) * g.half;
        if (POOL) mfm_pool_store<MT, false>(acc, bb, yp, g.half, g.HP, h, r & 3);
        else mfm_store<MT>(acc, bb, yp, g.half, g.HP, h);
    }
}

// First layer with the pool in its epilogue, patch staged in LDS (round 5).  The gathering kernel above pays ~15 vector
// instructions per tap and pixel (k -> (ky, kx) by integer division, four bounds checks, a scattered 4-byte load): 0.94 ms
// for the 26.5 M pixels of B = 1024, all of it index arithmetic.  Here a workgroup owns 4 pooled rows x 8 pooled columns
// (8 x 16 convolution pixels; wave = pooled row, quad = window), its 12 x 20 input patch sits in LDS with the zero padding
// materialised, and a tap is one select between two compile-time offsets (the lane's k half) + one ds_read_b32.
template <int MT, int K>
__global__ void __launch_bounds__(kThr) lcnn_conv1_pool_kernel(const LG g, const float* __restrict__ x,
                                                               const __bf16* __restrict__ wb, const float* __restrict__ bb,
                                                               __bf16* __restrict__ y) {
    constexpr int PR = 8 + K - 1, PC = 16 + K - 1;  // patch rows / columns
    constexpr int KP = (K * K + 15) / 16 * 16;
    __shared__ float patch[PR * PC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int pxb = blockIdx.x, pyb = blockIdx.y, n = blockIdx.z;
    const float* xn = x + (size_t)n * g.H * g.W;
    const int iy0 = 8 * pyb - g.pad, ix0 = 16 * pxb - g.pad;
    for (int e = tid; e < PR * PC; e += kThr) {
        const int pr = e / PC, pc = e - pr * PC;
        const int iy = iy0 + pr, ix = ix0 + pc;
        patch[e] = (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) ? xn[(size_t)iy * g.W + ix] : 0.f;
    }
    __syncthreads();
    const int py = 4 * pyb + wave, px = 8 * pxb + (r >> 2);
    const bool pv = py < g.PH && px < g.PW;
    // top-left tap of this lane's convolution pixel inside the patch
    const float* base = patch + (2 * wave + ((r >> 1) & 1)) * PC + 2 * (r >> 2) + (r & 1);
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
#pragma unroll
    for (int k0 = 0; k0 < KP; k0 += 16) {
        bf16x8 b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // k = k0 + 8 h + e: both halves' offsets are compile-time constants; taps past K * K meet zero weights and
            // read the pixel's first tap (a finite value of the same patch)
            const int ka = k0 + e, kb = k0 + 8 + e;
            const int oa = ka < K * K ? (ka / K) * PC + ka % K : 0;
            const int ob = kb < K * K ? (kb / K) * PC + kb % K : 0;
            b[e] = (__bf16)base[h ? ob : oa];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(wb + (size_t)(32 * mt + r) * g.Kpad + k0 + 8 * h);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[mt], 0, 0, 0);
        }
    }
    if (pv) {
        __bf16* yp = y + (((size_t)n * g.PH + py) * g.PW + px) * g.half;
        mfm_pool_store<MT, false>(acc, bb, yp, g.half, g.HP, h, r & 3);
    }
}

// 1x1 / 3x3 layers on channels-last bf16
template <int K, int MT, int CIN, int POOL = 0>  // POOL: 0 none, 1 pooled bf16, 2 pooled fp32 (bf16-rounded values)
__global__ void __launch_bounds__(kThr) lcnn_conv_nhwc_kernel(const LG g, const __bf16* __restrict__ x,
                                                              const __bf16* __restrict__ wb,
                                                              const float* __restrict__ bb, void* __restrict__ yv) {
    constexpr int PITCH = CIN + 8;            // bf16 per LDS row: 16-byte aligned rows, banks spread
    constexpr int ROWS = 32 * MT;
    constexpr int PIECES = ROWS * (CIN / 8);  // 16-byte pieces of one tap's weight slice
    constexpr int PER = (PIECES + kThr - 1) / kThr;
    constexpr int CB = CIN / 16;              // k-steps per tap
    __shared__ __attribute__((aligned(16))) __bf16 As[2][ROWS][PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / g.tiles;
    int oy, ox;
    size_t opx;
    const bool pv = lane_pixel<POOL != 0>(g, blockIdx.x - n * g.tiles, wave, r, oy, ox, opx);
    const size_t out_plane = POOL ? (size_t)g.PH * g.PW : (size_t)g.Ho * g.Wo;
    const __bf16* xn = x + (size_t)n * g.H * g.W * CIN + 8 * h;

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;

    bf16x8 aw[PER];   // this thread's pieces of the coming tap's weight slice
    bf16x8 bx[CB];    // this lane's pixel fragments of the coming tap
    auto fetch = [&](int t) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int piece = tid + kThr * j;
            const int row = piece / (CIN / 8), c8 = piece - row * (CIN / 8);
            if (PIECES % kThr == 0 || piece < PIECES)
                aw[j] = *reinterpret_cast<const bf16x8*>(wb + (size_t)row * g.Kpad + t * CIN + 8 * c8);
        }
        const int ky = t / K, kx = t - ky * K;
        const int iy = oy + ky - g.pad, ix = ox + kx - g.pad;
        const bool ok = pv && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        const __bf16* px = xn + ((size_t)(ok ? iy : 0) * g.W + (ok ? ix : 0)) * CIN;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(px + 16 * j);
            if (!ok) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            }
            bx[j] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int piece = tid + kThr * j;
            const int row = piece / (CIN / 8), c8 = piece - row * (CIN / 8);
            if (PIECES % kThr == 0 || piece < PIECES) *reinterpret_cast<bf16x8*>(&As[buf][row][8 * c8]) = aw[j];
        }
    };
    fetch(0);
    stash(0);
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < K * K; ++t) {
        bf16x8 bc[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) bc[j] = bx[j];
        if (t + 1 < K * K) fetch(t + 1);  // in flight during this tap's matrix instructions
        const int buf = t & 1;
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&As[buf][32 * mt + r][16 * j + 8 * h]);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bc[j], acc[mt], 0, 0, 0);
            }
        if (t + 1 < K * K) stash(buf ^ 1);
        __syncthreads();
    }
    if (pv) {
        const size_t o = ((size_t)n * out_plane + opx) * g.half;
        if (POOL == 2) mfm_pool_store<MT, true>(acc, bb, static_cast<float*>(yv) + o, g.half, g.HP, h, r & 3);
        else if (POOL == 1) mfm_pool_store<MT, false>(acc, bb, static_cast<__bf16*>(yv) + o, g.half, g.HP, h, r & 3);
        else mfm_store<MT>(acc, bb, static_cast<__bf16*>(yv) + o, g.half, g.HP, h);
    }
}

// MaxPool2d(2, 2), floor mode, 8 channels per thread
template <bool OUT_F32>
__global__ void __launch_bounds__(256) lcnn_pool_kernel(const __bf16* __restrict__ x, void* __restrict__ yv, int N, int H,
                                                        int W, int C) {
    const int C8 = C / 8, PH = H / 2, PW = W / 2;
    const long total = (long)N * PH * PW * C8;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % C8);
    long q = i / C8;
    const int px = (int)(q % PW);
    q /= PW;
    const int py = (int)(q % PH);
    const int n = (int)(q / PH);
    const __bf16* p00 = x + (((size_t)n * H + 2 * py) * W + 2 * px) * C + 8 * c8;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p00), b = *reinterpret_cast<const bf16x8*>(p00 + C);
    const bf16x8 c = *reinterpret_cast<const bf16x8*>(p00 + (size_t)W * C);
    const bf16x8 d = *reinterpret_cast<const bf16x8*>(p00 + (size_t)W * C + C);
    bf16x8 o;
    float of[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float m = fmaxf(fmaxf((float)a[e], (float)b[e]), fmaxf((float)c[e], (float)d[e]));
        o[e] = (__bf16)m;
        of[e] = m;
    }
    const size_t off = (((size_t)n * PH + py) * PW + px) * C + 8 * c8;
    if (OUT_F32) {
        float* y = static_cast<float*>(yv) + off;
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = of[e];
    } else {
        *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(yv) + off) = o;
    }
}

// One step of an LSTM direction, recurrent projection and cell in one launch (reference BLSTMLayer,
// models.py:212-237: nn.LSTM): gates = pre[b][4H] (input projection + both biases, fp32) + h_prev[b][:] . Wh^T with
// bf16 operands, then the fp32 cell.  Wave = 32 gate rows (4 gates x 8 hidden units: the A operand, bf16 rows of Wh)
// x 32 batch columns (the B operand: h_prev rounded to bf16 while it is loaded); in the D fragment a lane holds,
// for its batch row, all four gates of four units -- the cell update is lane-local.  h_prev / h_next are two
// buffers: other workgroups still read h_prev.  (The two-launch form -- a 64x64-tile GEMM over [128 x 1024 x 256]
// on 32 workgroups, then the cell kernel -- took 27 + 5 us per step, 0.76 ms of the 1.5 ms evaluation step.)
struct LstmDir {
    const float* pre;
    const __bf16* wh;
    const float* hprev;
    float* c;
    float* hout;
    float* hnext;
};

__device__ __forceinline__ void lstm_step_body(const float* __restrict__ pre, const __bf16* __restrict__ wh,
                                               const float* __restrict__ hprev, float* __restrict__ c,
                                               float* __restrict__ hout, int ldh, float* __restrict__ hnext, int B, int H) {
    const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
    const int j0 = blockIdx.x * 8;
    const int b = blockIdx.y * 32 + r;
    const bool bv = b < B;
    const __bf16* arow = wh + ((size_t)(r >> 3) * H + j0 + (r & 7)) * H + 8 * hh;
    const float* brow = hprev + (size_t)(bv ? b : 0) * H + 8 * hh;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int k0 = 0; k0 < H; k0 += 16) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + k0);
        const float4 x0 = *reinterpret_cast<const float4*>(brow + k0), x1 = *reinterpret_cast<const float4*>(brow + k0 + 4);
        bf16x8 bb;
        bb[0] = (__bf16)(bv ? x0.x : 0.f); bb[1] = (__bf16)(bv ? x0.y : 0.f); bb[2] = (__bf16)(bv ? x0.z : 0.f);
        bb[3] = (__bf16)(bv ? x0.w : 0.f); bb[4] = (__bf16)(bv ? x1.x : 0.f); bb[5] = (__bf16)(bv ? x1.y : 0.f);
        bb[6] = (__bf16)(bv ? x1.z : 0.f); bb[7] = (__bf16)(bv ? x1.w : 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
    }
    if (!bv) return;
    // D: column = batch row b, row = (q & 3) + 8 (q >> 2) + 4 hh: gate q >> 2, unit 4 hh + (q & 3)
    const int u0 = j0 + 4 * hh;
    const float* pb = pre + (size_t)b * 4 * H + u0;
    const float4 pi = *reinterpret_cast<const float4*>(pb), pf = *reinterpret_cast<const float4*>(pb + H);
    const float4 pg = *reinterpret_cast<const float4*>(pb + 2 * H), po = *reinterpret_cast<const float4*>(pb + 3 * H);
    float* cb = c + (size_t)b * H + u0;
    const float4 cv = *reinterpret_cast<const float4*>(cb);
    const float pia[4] = {pi.x, pi.y, pi.z, pi.w}, pfa[4] = {pf.x, pf.y, pf.z, pf.w};
    const float pga[4] = {pg.x, pg.y, pg.z, pg.w}, poa[4] = {po.x, po.y, po.z, po.w};
    const float cva[4] = {cv.x, cv.y, cv.z, cv.w};
    float cn[4], hn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float ig = 1.f / (1.f + expf(-(acc[j] + pia[j])));
        const float fg = 1.f / (1.f + expf(-(acc[4 + j] + pfa[j])));
        const float gg = tanhf(acc[8 + j] + pga[j]);
        const float og = 1.f / (1.f + expf(-(acc[12 + j] + poa[j])));
        cn[j] = fg * cva[j] + ig * gg;
        hn[j] = og * tanhf(cn[j]);
    }
    *reinterpret_cast<float4*>(cb) = make_float4(cn[0], cn[1], cn[2], cn[3]);
    *reinterpret_cast<float4*>(hnext + (size_t)b * H + u0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) hout[(size_t)b * ldh + u0 + j] = hn[j];
}

__global__ void __launch_bounds__(64) lstm_step_bf16_kernel(const float* __restrict__ pre, const __bf16* __restrict__ wh,
                                                            const float* __restrict__ hprev, float* __restrict__ c,
                                                            float* __restrict__ hout, int ldh, float* __restrict__ hnext,
                                                            int B, int H) {
    lstm_step_body(pre, wh, hprev, c, hout, ldh, hnext, B, H);
}

// both directions of a bidirectional layer in one launch (blockIdx.z): their steps are independent, and a step is
// bound by its launch and its one round of loads, not by the 128 workgroups' arithmetic
__global__ void __launch_bounds__(64) lstm_step_bf16_pair_kernel(const LstmDir d0, const LstmDir d1, int ldh, int B, int H) {
    const LstmDir& d = blockIdx.z ? d1 : d0;
    lstm_step_body(d.pre, d.wh, d.hprev, d.c, d.hout, ldh, d.hnext, B, H);
}

// A whole bidirectional LSTM layer in ONE launch (round 5; the step kernels above take 2 x T launches of 9-17 us, bound by
// launch and one round of loads each).  The recurrence only couples the hidden units of ONE batch row, so a workgroup that
// owns 32 batch rows and ALL hidden units of one direction needs no other workgroup: it walks the T steps by itself with a
// workgroup barrier per step.  8 waves; wave w owns the row tiles w, w + 8, ... (a tile = 4 gates x 8 units, as in
// lstm_step_body, so that a lane holds the four gates of its units); h lives in LDS as bf16 [32][H] (double-buffered: the
// B operand of the next step), c in registers for the whole sequence; Wh -- handed over in matrix-fragment order, see
// afd_blstm_layer_bf16 -- streams from L2 every step (512 KB per direction at H = 256, resident in every XCD's L2), the
// step's `pre` rows are requested before its matrix instructions.
constexpr int kLstmWaves = 8, kLstmTilesMax = 4;
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__global__ void __launch_bounds__(kLstmWaves * 64)
blstm_layer_bf16_kernel(const float* __restrict__ pre_f, const float* __restrict__ pre_r, const __bf16* __restrict__ wh_f,
                        const __bf16* __restrict__ wh_r, float* __restrict__ out, int T, int B, int H) {
    extern __shared__ __attribute__((aligned(16))) __bf16 hbuf[];  // [2][32][H + 8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int dir = blockIdx.y;
    const int b = blockIdx.x * 32 + r;
    const bool bv = b < B;
    const float* pre = dir ? pre_r : pre_f;
    const __bf16* wh = dir ? wh_r : wh_f;
    const int pitch = H + 8;
    const int tiles = H / 8;
    for (int e = tid; e < 2 * 32 * pitch; e += kLstmWaves * 64) hbuf[e] = (__bf16)0.f;
    float c[kLstmTilesMax][4];
#pragma unroll
    for (int i = 0; i < kLstmTilesMax; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[i][j] = 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = dir ? T - 1 - s : s;
        const __bf16* hp = hbuf + (size_t)(s & 1) * 32 * pitch + (size_t)r * pitch + 8 * hh;
        __bf16* hn = hbuf + (size_t)((s + 1) & 1) * 32 * pitch + (size_t)r * pitch;
        const float* pb = pre + ((size_t)t * B + (bv ? b : 0)) * 4 * H;
        float* ob = out + ((size_t)t * B + (bv ? b : 0)) * 2 * H + (size_t)dir * H;
#pragma unroll
        for (int i = 0; i < kLstmTilesMax; ++i) {
            const int tile = wave + kLstmWaves * i;
            if (tile >= tiles) break;  // uniform
            const int u0 = 8 * tile + 4 * hh;
            // the four gates of this lane's four units (order i | f | g | o): in flight during the matrix instructions
            const float4 pi = *reinterpret_cast<const float4*>(pb + u0), pf = *reinterpret_cast<const float4*>(pb + H + u0);
            const float4 pg = *reinterpret_cast<const float4*>(pb + 2 * H + u0), po = *reinterpret_cast<const float4*>(pb + 3 * H + u0);
            // weights in FRAGMENT order [tile][k-step][lane][8]: a wave's load is 1 KB of consecutive bytes (row-major
            // weights made every load touch 32 lines for 32 bytes each -- 22 us per step on the CU's L1 at H = 256)
            const __bf16* arow = wh + ((size_t)tile * (H / 16) * 64 + lane) * 8;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            if (s > 0) {  // h_0 = 0: the first step has no recurrent term
                // eight k-steps of weight rows requested together (one L2 round trip per 128 k, not per 16)
                for (int k0 = 0; k0 < H; k0 += 128) {
                    bf16x8 a[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k0 + 16 * j < H) a[j] = *reinterpret_cast<const bf16x8*>(arow + (size_t)(k0 / 16 + j) * 512);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k0 + 16 * j < H) {
                            const bf16x8 bb = *reinterpret_cast<const bf16x8*>(hp + k0 + 16 * j);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], bb, acc, 0, 0, 0);
                        }
                }
            }
            const float pia[4] = {pi.x, pi.y, pi.z, pi.w}, pfa[4] = {pf.x, pf.y, pf.z, pf.w};
            const float pga[4] = {pg.x, pg.y, pg.z, pg.w}, poa[4] = {po.x, po.y, po.z, po.w};
            float hv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // hardware exp / reciprocal (a 32-row workgroup walks the whole sequence alone: its gate arithmetic is on
                // the critical path of every step); tanh x = 2 sigmoid(2 x) - 1.  ~1e-6 relative, inside the bf16 bar
                const float ig = fast_sigmoid(acc[j] + pia[j]);
                const float fg = fast_sigmoid(acc[4 + j] + pfa[j]);
                const float gg = 2.f * fast_sigmoid(2.f * (acc[8 + j] + pga[j])) - 1.f;
                const float og = fast_sigmoid(acc[12 + j] + poa[j]);
                c[i][j] = fg * c[i][j] + ig * gg;
                hv[j] = og * (2.f * fast_sigmoid(2.f * c[i][j]) - 1.f);
            }
            bf16x4 hb;
#pragma unroll
            for (int j = 0; j < 4; ++j) hb[j] = (__bf16)hv[j];
            *reinterpret_cast<bf16x4*>(hn + u0) = hb;
            if (bv) *reinterpret_cast<float4*>(ob + u0) = make_float4(hv[0], hv[1], hv[2], hv[3]);
        }
        __syncthreads();
    }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (__bf16)x[i];
}

template <int K, int MT, int POOL>
int launch_nhwc_p(const LG& g, const __bf16* x, const __bf16* wb, const float* bb, void* y, hipStream_t s) {
    const unsigned grid = (unsigned)(g.N * g.tiles);
    switch (g.Cin) {
        case 32: hipLaunchKernelGGL((lcnn_conv_nhwc_kernel<K, MT, 32, POOL>), dim3(grid), dim3(kThr), 0, s, g, x, wb, bb, y); break;
        case 48: hipLaunchKernelGGL((lcnn_conv_nhwc_kernel<K, MT, 48, POOL>), dim3(grid), dim3(kThr), 0, s, g, x, wb, bb, y); break;
        case 64: hipLaunchKernelGGL((lcnn_conv_nhwc_kernel<K, MT, 64, POOL>), dim3(grid), dim3(kThr), 0, s, g, x, wb, bb, y); break;
        default: return afd::fail(AFD_ERR_UNSUPPORTED, "lcnn conv bf16: %d input channels", g.Cin);
    }
    return afd::check_launch("lcnn_conv_nhwc_kernel");
}

template <int K, int MT>
int launch_nhwc(const LG& g, const __bf16* x, const __bf16* wb, const float* bb, void* y, int pool, hipStream_t s) {
    if (pool == 2) return launch_nhwc_p<K, MT, 2>(g, x, wb, bb, y, s);
    if (pool == 1) return launch_nhwc_p<K, MT, 1>(g, x, wb, bb, y, s);
    return launch_nhwc_p<K, MT, 0>(g, x, wb, bb, y, s);
}

int fill_geom(LG& g, int N, int H, int W, int Cin, int Cout, int K, int pad, int pool = 0) {
    if (N < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 2 || (Cout & 1)) return afd::fail(AFD_ERR_ARG, "lcnn conv bf16: bad shape");
    if (pad < 0 || pad >= K) return afd::fail(AFD_ERR_ARG, "lcnn conv bf16: padding %d", pad);
    g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.pad = pad;
    g.Ho = H + 2 * pad - (K - 1);
    g.Wo = W + 2 * pad - (K - 1);
    if (g.Ho < 1 || g.Wo < 1) return afd::fail(AFD_ERR_ARG, "lcnn conv bf16: empty output");
    g.tiles = (g.Ho * g.Wo + kPixT - 1) / kPixT;
    g.PH = g.Ho / 2;
    g.PW = g.Wo / 2;
    if (pool) {
        if (pool < 0 || pool > 2 || g.PH < 1 || g.PW < 1) return afd::fail(AFD_ERR_ARG, "lcnn conv bf16: pooled form on a %d x %d output", g.Ho, g.Wo);
        g.tiles = (g.PH * g.PW + 31) / 32;
    }
    g.half = Cout / 2;
    if (g.half % 4) return afd::fail(AFD_ERR_UNSUPPORTED, "lcnn conv bf16: %d feature-map pairs", g.half);
    g.HP = round_up_i(g.half, 32);
    g.Kpad = round_up_i(K * K * Cin, 16);
    if (2 * g.HP > 128) return afd::fail(AFD_ERR_UNSUPPORTED, "lcnn conv bf16: Cout %d", Cout);
    if ((long)N * g.tiles > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "lcnn conv bf16: grid too large");
    return AFD_OK;
}

}  // namespace

#define AFD_STREAM static_cast<hipStream_t>(stream)

extern "C" size_t afd_lcnn_prep_bytes(int Cin, int Cout, int K) {
    if (Cin < 1 || Cout < 2 || K < 1) return 0;
    const int HP = round_up_i(Cout / 2, 32);
    return (size_t)2 * HP * round_up_i(K * K * Cin, 16) * sizeof(__bf16) + (size_t)2 * HP * sizeof(float);
}

// wb_bb: [2 HP][Kpad] bf16 followed by [2 HP] f32 (afd_lcnn_prep_bytes); bn_mean / bn_var: the evaluation-mode
// BatchNorm2d(affine=False) that follows the layer's max-feature-map (and pool), or null
extern "C" int afd_lcnn_prep_conv_bf16(const float* w, const float* bias, const float* bn_mean, const float* bn_var,
                                       float eps, void* wb_bb, int Cin, int Cout, int K, afd_stream_t stream) {
    if (!w || !wb_bb || Cin < 1 || Cout < 2 || (Cout & 1) || K < 1 || (bn_mean == nullptr) != (bn_var == nullptr))
        return afd::fail(AFD_ERR_ARG, "lcnn prep: bad argument");
    const int HP = round_up_i(Cout / 2, 32), Kpad = round_up_i(K * K * Cin, 16);
    __bf16* wb = static_cast<__bf16*>(wb_bb);
    float* bb = reinterpret_cast<float*>(wb + (size_t)2 * HP * Kpad);
    const int total = 2 * HP * Kpad;
    hipLaunchKernelGGL(lcnn_prep_conv_kernel, dim3((total + 255) / 256), dim3(256), 0, AFD_STREAM, w, bias, bn_mean, bn_var,
                       eps, wb, bb, Cout, Cin, K, Kpad, HP);
    return afd::check_launch("lcnn_prep_conv_kernel");
}

extern "C" int afd_lcnn_conv1_nhwc_bf16(const float* x, const void* wb_bb, void* y, int N, int H, int W, int Cout, int K,
                                        int pad, int pool, afd_stream_t stream) {
    if (!x || !wb_bb || !y) return afd::fail(AFD_ERR_ARG, "lcnn conv1 bf16: null pointer");
    if (pool != 0 && pool != 1) return afd::fail(AFD_ERR_ARG, "lcnn conv1 bf16: pool is 0 or 1");
    LG g{};
    int rc = fill_geom(g, N, H, W, 1, Cout, K, pad, pool);
    if (rc) return rc;
    const __bf16* wb = static_cast<const __bf16*>(wb_bb);
    const float* bb = reinterpret_cast<const float*>(wb + (size_t)2 * g.HP * g.Kpad);
    afd::ScopedTiming timing(AFD_K_LCNN_BF16, 2.0 * N * (double)g.Ho * g.Wo * Cout * K * K, AFD_STREAM);
    timing.issued(2.0 * N * (double)g.tiles * kPixT * 2 * g.HP * g.Kpad);
    timing.bytes((double)N * (4.0 * H * W + 2.0 * (pool ? (double)g.PH * g.PW : (double)g.Ho * g.Wo) * g.half));
    const unsigned grid = (unsigned)(g.N * g.tiles);
    __bf16* yb = static_cast<__bf16*>(y);
    if (pool && K == 5 && g.Kpad == 32 && 2 * g.HP == 64 && N <= 65535) {
        // the model's first layer (5 x 5, 64 channels): patch in LDS; its grid is 4 x 8-window blocks of the pooled image
        timing.issued(2.0 * N * (double)((g.PW + 7) / 8) * ((g.PH + 3) / 4) * kPixT * 2 * g.HP * g.Kpad);
        hipLaunchKernelGGL((lcnn_conv1_pool_kernel<2, 5>), dim3((g.PW + 7) / 8, (g.PH + 3) / 4, N), dim3(kThr), 0, AFD_STREAM, g, x,
                           wb, bb, yb);
        return afd::check_launch("lcnn_conv1_pool_kernel");
    }
    if (2 * g.HP == 64) {
        if (pool) hipLaunchKernelGGL((lcnn_conv1_kernel<2, true>), dim3(grid), dim3(kThr), 0, AFD_STREAM, g, x, wb, bb, yb, K);
        else hipLaunchKernelGGL((lcnn_conv1_kernel<2, false>), dim3(grid), dim3(kThr), 0, AFD_STREAM, g, x, wb, bb, yb, K);
    } else {
        if (pool) hipLaunchKernelGGL((lcnn_conv1_kernel<4, true>), dim3(grid), dim3(kThr), 0, AFD_STREAM, g, x, wb, bb, yb, K);
        else hipLaunchKernelGGL((lcnn_conv1_kernel<4, false>), dim3(grid), dim3(kThr), 0, AFD_STREAM, g, x, wb, bb, yb, K);
    }
    return afd::check_launch("lcnn_conv1_kernel");
}

extern "C" int afd_lcnn_conv_nhwc_bf16(const void* x, const void* wb_bb, void* y, int N, int H, int W, int Cin, int Cout,
                                       int K, int pad, int pool, afd_stream_t stream) {
    if (!x || !wb_bb || !y) return afd::fail(AFD_ERR_ARG, "lcnn conv bf16: null pointer");
    if (K != 1 && K != 3) return afd::fail(AFD_ERR_UNSUPPORTED, "lcnn conv bf16: kernel size %d", K);
    if (pool < 0 || pool > 2) return afd::fail(AFD_ERR_ARG, "lcnn conv bf16: pool is 0, 1 (bf16) or 2 (fp32)");
    LG g{};
    int rc = fill_geom(g, N, H, W, Cin, Cout, K, pad, pool);
    if (rc) return rc;
    const __bf16* wb = static_cast<const __bf16*>(wb_bb);
    const float* bb = reinterpret_cast<const float*>(wb + (size_t)2 * g.HP * g.Kpad);
    const __bf16* xb = static_cast<const __bf16*>(x);
    afd::ScopedTiming timing(AFD_K_LCNN_BF16, 2.0 * N * (double)g.Ho * g.Wo * Cout * Cin * K * K, AFD_STREAM);
    timing.issued(2.0 * N * (double)g.tiles * kPixT * 2 * g.HP * g.Kpad);
    timing.bytes(2.0 * N * ((double)H * W * Cin + (pool ? (pool == 2 ? 2.0 : 1.0) * g.PH * g.PW : (double)g.Ho * g.Wo) * g.half));
    const bool two = 2 * g.HP == 64;
    if (K == 1) return two ? launch_nhwc<1, 2>(g, xb, wb, bb, y, pool, AFD_STREAM) : launch_nhwc<1, 4>(g, xb, wb, bb, y, pool, AFD_STREAM);
    return two ? launch_nhwc<3, 2>(g, xb, wb, bb, y, pool, AFD_STREAM) : launch_nhwc<3, 4>(g, xb, wb, bb, y, pool, AFD_STREAM);
}

extern "C" int afd_lcnn_pool_nhwc_bf16(const void* x, void* y, int N, int H, int W, int C, int out_f32,
                                       afd_stream_t stream) {
    if (!x || !y || N < 1 || H < 2 || W < 2 || C < 8 || (C & 7)) return afd::fail(AFD_ERR_ARG, "lcnn pool bf16: bad argument");
    const long total = (long)N * (H / 2) * (W / 2) * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (out_f32) hipLaunchKernelGGL((lcnn_pool_kernel<true>), dim3(grid), dim3(256), 0, AFD_STREAM, static_cast<const __bf16*>(x), y, N, H, W, C);
    else hipLaunchKernelGGL((lcnn_pool_kernel<false>), dim3(grid), dim3(256), 0, AFD_STREAM, static_cast<const __bf16*>(x), y, N, H, W, C);
    return afd::check_launch("lcnn_pool_kernel");
}

extern "C" int afd_f32_to_bf16(const float* x, void* y, size_t n, afd_stream_t stream) {
    if (!x || !y || n < 1) return afd::fail(AFD_ERR_ARG, "f32 -> bf16: bad argument");
    size_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, AFD_STREAM, x, static_cast<__bf16*>(y), n);
    return afd::check_launch("f32_to_bf16_kernel");
}

extern "C" int afd_lstm_step_bf16_pair(const float* const* pre, const void* const* wh_bf16, const float* const* hprev,
                                       float* const* c, float* const* hout, int ldh, float* const* hnext, int B, int H,
                                       afd_stream_t stream) {
    if (!pre || !wh_bf16 || !hprev || !c || !hout || !hnext || B < 1 || H < 16 || (H & 15) || ldh < H)
        return afd::fail(AFD_ERR_ARG, "lstm step bf16 (pair): bad argument");
    LstmDir d[2];
    for (int k = 0; k < 2; ++k) {
        if (!pre[k] || !wh_bf16[k] || !hprev[k] || !c[k] || !hout[k] || !hnext[k] || hprev[k] == hnext[k])
            return afd::fail(AFD_ERR_ARG, "lstm step bf16 (pair): bad pointer of direction %d", k);
        d[k] = LstmDir{pre[k], static_cast<const __bf16*>(wh_bf16[k]), hprev[k], c[k], hout[k], hnext[k]};
    }
    afd::ScopedTiming timing(AFD_K_LCNN_BF16, 2.0 * 2.0 * B * 4.0 * H * H, AFD_STREAM);
    timing.issued(2.0 * 2.0 * ((B + 31) / 32 * 32) * 4.0 * H * H);
    timing.bytes(2.0 * (2.0 * 4 * H * H + 4.0 * B * (4.0 * H + 4.0 * H)));
    hipLaunchKernelGGL(lstm_step_bf16_pair_kernel, dim3((unsigned)(H / 8), (unsigned)((B + 31) / 32), 2), dim3(64), 0, AFD_STREAM,
                       d[0], d[1], ldh, B, H);
    return afd::check_launch("lstm_step_bf16_pair_kernel");
}

extern "C" int afd_blstm_layer_bf16(const float* pre_fwd, const float* pre_rev, const void* wh_fwd_bf16,
                                    const void* wh_rev_bf16, float* out, int T, int B, int H, afd_stream_t stream) {
    if (!pre_fwd || !pre_rev || !wh_fwd_bf16 || !wh_rev_bf16 || !out || T < 1 || B < 1)
        return afd::fail(AFD_ERR_ARG, "blstm layer bf16: bad argument");
    if (H < 16 || (H & 15) || H / 8 > kLstmWaves * kLstmTilesMax)
        return afd::fail(AFD_ERR_UNSUPPORTED, "blstm layer bf16: hidden size %d (multiples of 16 up to %d)", H,
                         8 * kLstmWaves * kLstmTilesMax);
    const size_t lds = (size_t)2 * 32 * (H + 8) * sizeof(__bf16);
    afd::ScopedTiming timing(AFD_K_LCNN_BF16, 2.0 * 2.0 * T * (double)B * 4.0 * H * H, AFD_STREAM);
    // the first step has no recurrent product (h_0 = 0)
    timing.issued(2.0 * 2.0 * (T - 1) * (double)((B + 31) / 32 * 32) * 4.0 * H * H);
    timing.bytes(2.0 * (2.0 * 4 * H * H) + (double)T * B * 4.0 * (2.0 * 4.0 * H + 2.0 * H));
    hipLaunchKernelGGL(blstm_layer_bf16_kernel, dim3((unsigned)((B + 31) / 32), 2), dim3(kLstmWaves * 64), lds, AFD_STREAM,
                       pre_fwd, pre_rev, static_cast<const __bf16*>(wh_fwd_bf16), static_cast<const __bf16*>(wh_rev_bf16), out,
                       T, B, H);
    return afd::check_launch("blstm_layer_bf16_kernel");
}

extern "C" int afd_lstm_step_bf16(const float* pre, const void* wh_bf16, const float* hprev, float* c, float* hout, int ldh,
                                  float* hnext, int B, int H, afd_stream_t stream) {
    if (!pre || !wh_bf16 || !hprev || !c || !hout || !hnext || B < 1 || H < 16 || (H & 15) || ldh < H || hprev == hnext)
        return afd::fail(AFD_ERR_ARG, "lstm step bf16: bad argument");
    afd::ScopedTiming timing(AFD_K_LCNN_BF16, 2.0 * B * 4.0 * H * H, AFD_STREAM);
    timing.issued(2.0 * ((B + 31) / 32 * 32) * 4.0 * H * H);
    timing.bytes(2.0 * 4 * H * H + 4.0 * B * (4.0 * H + 4.0 * H));
    hipLaunchKernelGGL(lstm_step_bf16_kernel, dim3((unsigned)(H / 8), (unsigned)((B + 31) / 32)), dim3(64), 0, AFD_STREAM, pre,
                       static_cast<const __bf16*>(wh_bf16), hprev, c, hout, ldh, hnext, B, H);
    return afd::check_launch("lstm_step_bf16_kernel");
}
