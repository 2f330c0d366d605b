// Pieces shared by the wavelet-packet kernels of wpt3.hip (levels 1..8, vector ALU) and wpt4.hip (levels 9..14,
// lattice form): epilogue, padded-node store, register window.  Reference: src/audiofakedetect/wavelet_math.py:167-263.
#pragma once
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace afd {
namespace wptc {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kMaxTaps = 64;  // the top kernel's limit (coif10: 60, dmey: 62); longer banks take wpt.hip's kernel (128)
constexpr int kTopThreads = 1024;
constexpr int kMaxRun = 46;  // longest run of outputs per work item the top kernel's compile-time instances use
constexpr int kTopLdsFloats = 40960;  // 163 840 B: all of a CU's LDS
constexpr int kKsMax = 8;

struct Epi {
    unsigned flags;
    float power, eps, k1, k0, mean, inv_std, sgn_neg, sgn_pos;
};

static __device__ __noinline__ float pow_log_slow3(float v, float power, float eps) {
    return logf(powf(fabsf(v), power) + eps);
}

// log(|v|^power + eps) and (x - mean) / std.  The kernels are compiled per epilogue mode, so that an
// element costs three vector instructions and no branch (the f32 matrix instructions and the vector ALU
// share the SIMD's FMA lanes: every vector instruction is time the matrix stream does not get):
//   EPI_RAW   v * k1 + k0                      (k1 = 1/std, k0 = -mean/std; 1, 0 without AFD_WPT_NORM)
//   EPI_LOG2  log2(v*v + eps) * k1 + k0        (power == 2: v*v + eps >= 1e-12 is a normal float, the bare
//             v_log_f32 needs no denormal pre-scaling; k1 = ln 2 / std)
//   EPI_SLOW  any other power, library pow / log
enum { EPI_RAW = 0, EPI_LOG2 = 1, EPI_SLOW = 2 };

template <int MODE>
__device__ __forceinline__ float epi_value(float v, const Epi& e) {
    if (MODE == EPI_RAW) return fmaf(v, e.k1, e.k0);
    if (MODE == EPI_LOG2) return fmaf(__builtin_amdgcn_logf(fmaf(v, v, e.eps)), e.k1, e.k0);
    v = pow_log_slow3(v, e.power, e.eps);
    if (e.flags & AFD_WPT_NORM) v = (v - e.mean) * e.inv_std;
    return v;
}

static inline int epi_mode(unsigned flags, float power) {
    if (!(flags & AFD_WPT_LOG)) return EPI_RAW;
    return power == 2.0f ? EPI_LOG2 : EPI_SLOW;
}

// one coefficient of a child node of length n: its own slot and the pad slots that mirror it
template <int L>
__device__ __forceinline__ void put(float* node, int i, int n, float v) {
    constexpr int PAD = L - 2;
    node[i] = v;
    if ((unsigned)(i - 1) < (unsigned)PAD) node[-i] = v;
    if ((unsigned)(n - 2 - i) < (unsigned)(PAD + (n & 1))) node[2 * (n - 1) - i] = v;
}

// A work item is a run of W neighbouring outputs i .. i + W - 1, i even: their windows overlap, so the item reads
// L + 2 (W - 1) samples as 16-byte vectors and both filters run over registers.  W = 2: 7 ds_read_b128 for 24 taps, 3.5
// per output position; W = 6: 9 reads, 1.5 per position (the compile-time-geometry levels of the top kernel, round 5).
template <int L, int W = 2>
struct Window {
    static constexpr int NV = (L + 2 * (W - 1) + 3) / 4;
    float w[4 * NV];
    __device__ __forceinline__ void load(const float* __restrict__ src) {
        const f4* s4 = reinterpret_cast<const f4*>(src);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f4 x = s4[v];
            w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w;
        }
    }
    // filter `taps` (reversed) at output i (O = 0) or i + 1 (O = 1)
    template <int O>
    __device__ __forceinline__ float dot(const float* taps) const {
        f2 acc = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < L / 2; ++t) {
            const f2 tp = {taps[2 * t], taps[2 * t + 1]};
            const f2 xv = {w[2 * t + 2 * O], w[2 * t + 1 + 2 * O]};
            acc = __builtin_elementwise_fma(tp, xv, acc);
        }
        return acc.x + acc.y;
    }
};


// Lattice form of an orthogonal two-channel bank (wpt4.hip: wpt_lattice_coefficients): stage s of an analysis step is
// (A, B)[j] <- (A[j] + ab[s].x B[j-1], A[j] + ab[s].y B[j-1]) (s = 0: on (e_j, o_j) without the delay), and
// cA = A sc.x, cD = B sc.y.  Kernel argument: every pair is an aligned scalar register pair.
constexpr int kMaxStages = kMaxTaps / 2;
struct Lat4 {
    f2 ab[kMaxStages];
    f2 sc;
};
// the lattice of a tap table, cached per process; false when the taps are not an orthogonal bank (wpt4.hip)
bool wpt_get_lattice(const float* lo, const float* hi, int L, Lat4* out, double* oa, double* ob);

// B of the position in front of a lane's run: the neighbouring lane's value through DPP (wave_shr:1).  Inline assembly:
// given a vector element through __builtin_amdgcn_update_dpp, hipcc 7.2 takes the pair's LOW half as the DPP source
// (tools/micro/dpp_subreg.hip); the s_nop covers the two wait states between a vector write and a DPP read of it,
// which the compiler does not track into assembly.  Lane 0 keeps whatever the register held (never used).
__device__ __forceinline__ float lane_below(float v) {
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}

constexpr int refl_c(int j, int n) {
    j = j < 0 ? -j : j;
    return j >= n ? 2 * (n - 1) - j : j;
}

// node lengths of the standard 1 s frame (N = 22 050) at levels 8..14
template <int L> struct Shape3;
template <> struct Shape3<24> { static constexpr int L = 24; static constexpr int n[7] = {109, 66, 44, 33, 28, 25, 24}; };
template <> struct Shape3<10> { static constexpr int L = 10; static constexpr int n[7] = {95, 52, 30, 19, 14, 11, 10}; };
template <> struct Shape3<16> { static constexpr int L = 16; static constexpr int n[7] = {101, 58, 36, 25, 20, 17, 16}; };
template <int L> struct HasShape3 { static constexpr bool value = false; };
template <> struct HasShape3<24> { static constexpr bool value = true; };
template <> struct HasShape3<10> { static constexpr bool value = true; };
template <> struct HasShape3<16> { static constexpr bool value = true; };

// node pitch (floats) of an LDS image whose nodes carry their reflect pads:
// [L-2 left pad | n samples | L-2 (+1) right pad | up to 3 floats an odd node's last item reads], 16-byte aligned
// nodes, pitch / 4 odd (node-strided 16-byte reads hit distinct banks)
// (round 5: a work item of the top kernel covers up to 14 output positions, its window reaches 2 (14 - 1) + 3 floats past
// the last sample an item of a node's last positions needs: 31 floats of slack instead of 5)
constexpr int padded_pitch(int n, int L, int slack = 31) {
    int pitch = n + 2 * (L - 2) + slack;
    while (pitch % 8 != 4) ++pitch;
    return pitch;
}

}  // namespace wptc
}  // namespace afd
