// Wavelet-packet front end, levels 9..14 of the level-14 transform of 1 s frames on the vector ALU in LATTICE form
// (reference src/audiofakedetect/wavelet_math.py:167-263; the ptwt analysis step behind :182,192).
//
// Why not the matrix cores (wpt3.hip's deep kernel, rounds 2-3): as GEMMs the deep levels cost MORE multiply-adds
// than the direct form (banded 32-row tiles of the stepwise levels: 2.8-3x; sym5's 10 -> 14 composite: 1.4x), the
// f32 MFMA runs at the f32 vector rate anyway, and every coefficient left the matrix tile through quad transposes.
// The orthogonal filter bank factors instead into K = L/2 plane rotations with one delay between them
// (tools/wpt_lattice.py, profiles/r04_lattice_study.json):
//
//     (A, B)[j] <- (A[j] + alpha_s B[j-1], A[j] + beta_s B[j-1])        s = 1 .. K-1,   j = position (output index)
//
// ONE v_pk_fma_f32 per position and stage (operand halves picked by op_sel: no data movement), scaling deferred to
// the end of the step: L multiply-adds per output pair (cA[i], cD[i]) instead of 2 L, plus the triangle of
// K (K-1) / 2 positions in front of a run.  With a whole node per thread (deep nodes are 10..66 samples) the
// reflect extension is a compile-time choice of registers and the triangle is paid once per node.
//
// One workgroup (512 threads) = 16 level-8 nodes of one frame -> their 1 024 level-14 packets:
//   S0        level-8 nodes from the hand-off image (wpt3_top_kernel) -> LDS, reflect pads materialised
//   S1 .. S4  8 -> 9 -> 10 -> 11 -> 12 in direct form, work item = two neighbouring output pairs of a node (the
//             register window of the top kernel): 16 .. 128 nodes are too few for a node per thread
//   S5        12 -> 13, lattice, thread = node (256 nodes)
//   S6        13 -> 14, lattice, thread = node (512 nodes), epilogue on the registers, and the two children of a
//             node are neighbouring packets: a wave stores 512 contiguous bytes per time step
// Level images alternate between two LDS regions (81 KB for coif4: two workgroups per CU; 38 KB for sym5).
// float32 error of the lattice against the float64 oracle: <= 1.8e-6 of the largest coefficient for coif4 at
// level 14 (direct form 4.7e-7; bar 5e-6), <= 5.3e-7 for sym5.
#include "wpt_shared.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

using namespace afd::wptc;

namespace {

typedef unsigned u2 __attribute__((ext_vector_type(2)));

struct D4Params {
    const float* ws;  // level-8 hand-off [B][n8][256]
    float* out;       // features [B][C][n14][16384]
    Epi e;
    Lat4 lat;
    // last level: epilogue constants with the lattice's output scales folded in, per channel (0 = cA, 1 = cD)
    //   EPI_RAW   A * fk1[c] + k0        EPI_LOG2  log2(A * A + feps[c]) * k1 + fk0[c]
    f2 fk1, fk0, feps;
};

template <int L, int GRP> struct Plan4 {
    static constexpr int kD4Group = GRP;
    using SH = Shape3<L>;
    static constexpr int K = L / 2;
    // lattice level with G lanes per node: J positions (outputs + the K - 1 in front of them), R per lane
    static constexpr int positions(int n_in) { return (n_in + L - 2 + (n_in & 1)) / 2 + K - 1; }
    static constexpr int run(int n_in, int G) { return (positions(n_in) + G - 1) / G; }
    // padded node image read by a G-lane level: the reflect-padded node, and the 2 G R floats its lanes read
    static constexpr int pitch(int n_in, int G) {
        int p = n_in + 2 * (L - 2) + 2;
        const int need = 2 * G * run(n_in, G);
        p = p > need ? p : need;
        p = (p + 3) & ~3;
        return (p & 4) ? p : p + 4;  // pitch / 4 odd: the nodes of a wave start in different banks
    }
    static constexpr int p8 = pitch(SH::n[0], 32), p9 = pitch(SH::n[1], 16);
    static constexpr int p10 = pitch(SH::n[2], 8), p11 = pitch(SH::n[3], 4);
    // thread = node images: odd pitch, lane-strided 4-byte reads hit distinct banks
    static constexpr int p12 = SH::n[4] | 1, p13 = SH::n[5] | 1;
    static constexpr int mx(int a, int b) { return a > b ? a : b; }
    static constexpr int r0 = (mx(mx(kD4Group * p8, 4 * kD4Group * p10), 16 * kD4Group * p12) + 3) & ~3;
    static constexpr int r1 = (mx(mx(2 * kD4Group * p9, 8 * kD4Group * p11), 32 * kD4Group * p13) + 3) & ~3;
    static constexpr int lds_floats = r0 + r1;
};

// Lattice level with G adjacent lanes per node (G = 32, 16, 8, 4 for the 16, 32, 64, 128 g nodes of levels 8..11 of
// a group: every thread of the workgroup works at every level).  Lane seg of a node holds the R consecutive
// positions [seg R, seg R + R) as (A, B) register pairs; the reflect extension is materialised in the node image
// (position jj reads the pair at float 2 jj of the padded node), and the one value a stage needs from outside the
// lane -- B of the position in front of its run -- comes from the neighbouring lane through DPP (wave_shr:1).
// Positions in front of a stage's valid range and behind the node's last output carry garbage that never reaches
// a valid position (a position only reads positions below it, and those are valid whenever it is).
template <int L, int NIN, int G, int PIN, int POUT, bool PADDED>
__device__ __forceinline__ void lattice_level(float* lds0, const float* src, float* dst, const Lat4& c, int tid) {
    constexpr int K = L / 2, PAD = L - 2;
    constexpr int NOUT = (NIN + L - 2 + (NIN & 1)) / 2;
    constexpr int J = NOUT + K - 1, R = (J + G - 1) / G;
    static_assert(2 * G * R <= PIN, "the lanes of a node read inside its image");
    const int q = tid / G, seg = tid % G;
    const float* img = src + q * PIN + 2 * seg * R;
    f2 P[R];
    {
        const f2 ab = c.ab[0];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f2 eo = *reinterpret_cast<const f2*>(img + 2 * r);
            P[r] = __builtin_elementwise_fma(ab, f2{eo.y, eo.y}, f2{eo.x, eo.x});
        }
    }
#pragma unroll
    for (int s = 1; s < K; ++s) {
        const f2 ab = c.ab[s];
        const float pb = lane_below(P[R - 1].y);  // B of the position in front of this lane's run
#pragma unroll
        for (int r = R - 1; r >= 1; --r)
            P[r] = __builtin_elementwise_fma(ab, f2{P[r - 1].y, P[r - 1].y}, f2{P[r].x, P[r].x});
        P[0] = __builtin_elementwise_fma(ab, f2{pb, pb}, f2{P[0].x, P[0].x});
    }
    const int par = q & 1;  // odd-frequency parents list their children (d, a)
    const int i0 = seg * R - (K - 1);
    // Store addresses: every slot is (a base that holds all lane-dependent terms) + (a compile-time offset in the
    // instruction).  Coefficient i = i0 + r of a child goes to node[i], its left mirror to node[-i] and its right mirror
    // to node[2 (NOUT - 1) - i]: bases node + i0 (offset r) and node - i0 - (R - 1), node + 2 (NOUT - 1) - i0 - (R - 1)
    // (offset R - 1 - r).  The bases are made opaque: left to itself the compiler rebuilt each address from i with a
    // shift and a subtraction per store (the kernel is bound by its vector instruction count).
    const int ca = (int)(dst - lds0) + (PADDED ? PAD : 0) + (2 * q + par) * POUT;
    const int cd = (int)(dst - lds0) + (PADDED ? PAD : 0) + (2 * q + 1 - par) * POUT;
    int ma = ca + i0, md = cd + i0;
    int la = ca - i0 - (R - 1), ld = cd - i0 - (R - 1);
    int ra = la + 2 * (NOUT - 1), rd = ld + 2 * (NOUT - 1);
    asm volatile("" : "+v"(ma), "+v"(md));
    if (PADDED) asm volatile("" : "+v"(la), "+v"(ld), "+v"(ra), "+v"(rd));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = i0 + r;
        if ((unsigned)i < (unsigned)NOUT) {
            const f2 v = P[r] * c.sc;
            lds0[ma + r] = v.x;
            lds0[md + r] = v.y;
            if (PADDED) {
                if ((unsigned)(i - 1) < (unsigned)PAD) {
                    lds0[la + (R - 1 - r)] = v.x;
                    lds0[ld + (R - 1 - r)] = v.y;
                }
                if ((unsigned)(NOUT - 2 - i) < (unsigned)(PAD + (NOUT & 1))) {
                    lds0[ra + (R - 1 - r)] = v.x;
                    lds0[rd + (R - 1 - r)] = v.y;
                }
            }
        }
    }
}

// One analysis step of a whole node held in registers, lattice form.  x[NIN] -> P[K - 1 + i] = (A, B) of output i,
// still to be scaled by (oa, ob).  Positions jj = 0 .. J-1 stand for j = jj - (K - 1); stage s leaves jj >= s valid.
template <int L, int NIN> struct LatticeNode {
    static constexpr int K = L / 2;
    static constexpr int NOUT = (NIN + L - 2 + (NIN & 1)) / 2;
    static constexpr int J = NOUT + K - 1;
    f2 P[J];
    // `last`: the coefficient pair of the last stage -- (alpha, beta), or (beta, alpha) to get the outputs as (B, A)
    __device__ __forceinline__ void run(const float (&x)[NIN], const Lat4& c, const f2 last) {
        {
            const f2 ab = c.ab[0];
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
                const int j = jj - (K - 1);
                const float e = x[refl_c(2 * j, NIN)], o = x[refl_c(2 * j + 1, NIN)];
                P[jj] = __builtin_elementwise_fma(ab, f2{o, o}, f2{e, e});
            }
        }
#pragma unroll
        for (int s = 1; s < K; ++s) {
            const f2 ab = s == K - 1 ? last : c.ab[s];
#pragma unroll
            for (int jj = J - 1; jj >= s; --jj)
                P[jj] = __builtin_elementwise_fma(ab, f2{P[jj - 1].y, P[jj - 1].y}, f2{P[jj].x, P[jj].x});
        }
    }
};

template <int L, int MODE, bool SIGN, int GRP>
__global__ void __launch_bounds__(32 * GRP) wpt4_deep_kernel(const D4Params p) {
    using SH = Shape3<L>;
    using PL = Plan4<L, GRP>;
    constexpr int kD4Group = GRP;        // level-8 nodes per workgroup
    constexpr int kD4Threads = 32 * GRP;  // = its level-13 nodes
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PAD = L - 2;
    constexpr int n8 = SH::n[0], n12 = SH::n[4], n13 = SH::n[5], n14 = SH::n[6];
    const int tid = threadIdx.x;
    constexpr int GPF = 256 / kD4Group;  // groups per frame
    const int b = blockIdx.x / GPF, g = blockIdx.x % GPF;
    float* R0 = lds;
    float* R1 = lds + PL::r0;

    // ---- S0: level-8 nodes of this group, [pos][256] in the hand-off image -> padded nodes ----
    {
        const float* wsb = p.ws + (size_t)b * n8 * 256 + kD4Group * g;
        const int node = tid % kD4Group, pos0 = tid / kD4Group;
        constexpr int PASS = kD4Threads / kD4Group;  // positions per pass
        constexpr int NP = (n8 + PASS - 1) / PASS;
        float v[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int pos = pos0 + u * PASS;
            v[u] = pos < n8 ? wsb[(size_t)pos * 256 + node] : 0.f;
        }
        float* nd = R0 + node * PL::p8 + PAD;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int pos = pos0 + u * PASS;
            if (pos < n8) put<L>(nd, pos, n8, v[u]);
        }
    }
    __syncthreads();
    // ---- S1 .. S4: 8 -> 9 -> 10 -> 11 -> 12, lattice with 32 / 16 / 8 / 4 lanes per node ----
    {
        lattice_level<L, SH::n[0], 32, PL::p8, PL::p9, true>(lds, R0, R1, p.lat, tid);
        __syncthreads();
        lattice_level<L, SH::n[1], 16, PL::p9, PL::p10, true>(lds, R1, R0, p.lat, tid);
        __syncthreads();
        lattice_level<L, SH::n[2], 8, PL::p10, PL::p11, true>(lds, R0, R1, p.lat, tid);
        __syncthreads();
        lattice_level<L, SH::n[3], 4, PL::p11, PL::p12, false>(lds, R1, R0, p.lat, tid);
        __syncthreads();
    }
    // ---- S5: 12 -> 13, thread = level-12 node ----
    if (tid < 16 * kD4Group) {
        float x[n12];
        const float* nd = R0 + tid * PL::p12;
#pragma unroll
        for (int i = 0; i < n12; ++i) x[i] = nd[i];
        LatticeNode<L, n12> lt;
        lt.run(x, p.lat, p.lat.ab[L / 2 - 1]);
        const int par = tid & 1;
        // (float offsets from the start of LDS, opaque to the compiler: the region offset then sits in the base register
        // and the element index in the instruction's 8-bit offset field -- given the constant, it re-adds region + element
        // in a vector instruction per ds_write2)
        int oa = PL::r0 + (2 * tid + par) * PL::p13, od = PL::r0 + (2 * tid + 1 - par) * PL::p13;
        asm volatile("" : "+v"(oa), "+v"(od));
        float* na = lds + oa;
        float* ndd = lds + od;
        const f2 sc = p.lat.sc;
#pragma unroll
        for (int i = 0; i < n13; ++i) {
            const f2 v = lt.P[lt.K - 1 + i] * sc;
            na[i] = v.x;
            ndd[i] = v.y;
        }
    }
    __syncthreads();
    // ---- S6: 13 -> 14, thread = level-13 node; children = packets 2 q13, 2 q13 + 1 ----
    {
        float x[n13];
        int on = PL::r0 + tid * PL::p13;
        asm volatile("" : "+v"(on));
        const float* nd = lds + on;
#pragma unroll
        for (int i = 0; i < n13; ++i) x[i] = nd[i];
        LatticeNode<L, n13> lt;
        // an odd-frequency parent lists its children (d, a): its last stage runs with (beta, alpha), which leaves the
        // pair as (B, A) = packet order, and takes the per-channel epilogue constants in that order too
        const bool par = tid & 1;
        const f2 abl = p.lat.ab[L / 2 - 1];
        lt.run(x, p.lat, par ? f2{abl.y, abl.x} : abl);
        constexpr unsigned P = 16384;
        constexpr unsigned chan = (unsigned)n14 * P;
        const f2 fk1 = par ? f2{p.fk1.y, p.fk1.x} : p.fk1, fk0 = par ? f2{p.fk0.y, p.fk0.x} : p.fk0;
        const f2 feps = par ? f2{p.feps.y, p.feps.x} : p.feps, sc = par ? f2{p.lat.sc.y, p.lat.sc.x} : p.lat.sc;
        auto value = [&](const f2 ab) {
            f2 r;
            if (MODE == EPI_RAW) {
                r = __builtin_elementwise_fma(ab, fk1, fk0);
            } else if (MODE == EPI_LOG2) {
                const f2 t = __builtin_elementwise_fma(ab, ab, feps);
                const f2 lg = {__builtin_amdgcn_logf(t.x), __builtin_amdgcn_logf(t.y)};
                r = __builtin_elementwise_fma(lg, f2{p.e.k1, p.e.k1}, fk0);
            } else {
                const f2 v = ab * sc;
                r.x = epi_value<EPI_SLOW>(v.x, p.e);
                r.y = epi_value<EPI_SLOW>(v.y, p.e);
            }
            return r;
        };
        auto sign = [&](const f2 ab) {
            const f2 v = ab * sc;
            return f2{v.x < 0.f ? p.e.sgn_neg : p.e.sgn_pos, v.y < 0.f ? p.e.sgn_neg : p.e.sgn_pos};
        };
        // Stores: the frame is the buffer, the lane's two packets the vector offset and the time step the scalar offset
        // of a buffer store -- no vector address arithmetic (as 64-bit lane addresses every store cost an add /
        // add-with-carry pair; this kernel is bound by its vector instruction count: measured 1.3 us per instruction
        // and thread at B = 4096).  Measured and not kept: the lanes of a pair trading halves through DPP selects so
        // that a lane stores 16 bytes (12 stores instead of 24, +56 vector instructions): 2 493 -> 2 553 us.
        float* frame = p.out + (size_t)b * (SIGN ? 2 : 1) * chan;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(frame, 0, (int)((SIGN ? 2u : 1u) * chan * 4u), 0x00020000);
        const unsigned vo = 8u * ((unsigned)g * 32u * kD4Group + (unsigned)tid);  // this node's two packets
#pragma unroll
        for (int i = 0; i < n14; ++i) {
            const f2 ab = lt.P[lt.K - 1 + i];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, value(ab)), rs, vo, (unsigned)i * P * 4u, 2);
            if (SIGN)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, sign(ab)), rs, vo, (chan + (unsigned)i * P) * 4u, 2);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host: lattice factorisation of the tap table (double precision)
// ------------------------------------------------------------------------------------------------
struct M2 {
    double a[2][2];
};
M2 mul(const M2& x, const M2& y) {
    M2 z{};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) z.a[i][j] = x.a[i][0] * y.a[0][j] + x.a[i][1] * y.a[1][j];
    return z;
}
M2 rot(double th, bool reflect) {
    const double c = std::cos(th), s = std::sin(th);
    M2 m{};
    m.a[0][0] = c;
    m.a[1][0] = s;
    m.a[0][1] = reflect ? s : -s;
    m.a[1][1] = reflect ? -c : c;
    return m;
}

// taps of R_{K-1} D R_{K-2} ... D R_0, D = diag(1, z^-1):
// polyphase coefficient q = [[lo[2q+1], lo[2q]], [hi[2q+1], hi[2q]]]
void taps_of(const double* th, bool reflect0, int K, double* lo, double* hi) {
    std::vector<M2> H(1, rot(th[0], reflect0));
    for (int s = 1; s < K; ++s) {
        const M2 r = rot(th[s], false);
        std::vector<M2> n(H.size() + 1);
        for (size_t q = 0; q <= H.size(); ++q) {
            M2 d{};
            if (q < H.size()) { d.a[0][0] = H[q].a[0][0]; d.a[0][1] = H[q].a[0][1]; }
            if (q >= 1) { d.a[1][0] = H[q - 1].a[1][0]; d.a[1][1] = H[q - 1].a[1][1]; }
            n[q] = mul(r, d);
        }
        H.swap(n);
    }
    for (int q = 0; q < K; ++q) {
        lo[2 * q + 1] = H[q].a[0][0];
        lo[2 * q] = H[q].a[0][1];
        hi[2 * q + 1] = H[q].a[1][0];
        hi[2 * q] = H[q].a[1][1];
    }
}

}  // namespace

namespace afd {

// Lattice of an orthogonal two-channel bank given by its decomposition taps (pywt convention).  Peeling the
// rotations off the ends divides by the end taps (coif4: 1.8e-6) and amplifies the table's own rounding, so the
// peeled angles only start a Levenberg-Marquardt fit of the K angles to the 2 L taps.  Returns false when the
// taps are not an orthogonal bank to 1e-5 (biorthogonal wavelets, arbitrary filters): the caller keeps the
// direct-form kernels.  alpha, beta: K floats; scales[0..1] = (oa, ob).
bool wpt_lattice_coefficients(const float* dec_lo, const float* dec_hi, int L, double* alpha, double* beta, double* scales,
                              double* fit_residual) {
    if (L < 4 || (L & 1) || L > kMaxTaps) return false;
    const int K = L / 2;
    std::vector<double> lo(dec_lo, dec_lo + L), hi(dec_hi, dec_hi + L);
    std::vector<M2> Pq(K);
    for (int q = 0; q < K; ++q) {
        Pq[q].a[0][0] = lo[2 * q + 1]; Pq[q].a[0][1] = lo[2 * q];
        Pq[q].a[1][0] = hi[2 * q + 1]; Pq[q].a[1][1] = hi[2 * q];
    }
    std::vector<double> th(K, 0.0);
    for (int m = K - 1; m >= 1; --m) {
        const int col = std::fabs(Pq[m].a[0][0]) + std::fabs(Pq[m].a[1][0]) >= std::fabs(Pq[m].a[0][1]) + std::fabs(Pq[m].a[1][1]) ? 0 : 1;
        const double x = Pq[m].a[0][col], y = Pq[m].a[1][col];
        const double r = std::hypot(x, y);
        if (!(r > 0.0)) return false;
        const double c = y / r, s = -x / r;  // R^T = [[c, s], [-s, c]] kills the top row of P_m
        M2 rt{};
        rt.a[0][0] = c; rt.a[0][1] = s; rt.a[1][0] = -s; rt.a[1][1] = c;
        std::vector<M2> Q(m + 1);
        for (int q = 0; q <= m; ++q) Q[q] = mul(rt, Pq[q]);
        for (int q = 0; q < m; ++q) {
            Pq[q].a[0][0] = Q[q].a[0][0]; Pq[q].a[0][1] = Q[q].a[0][1];
            Pq[q].a[1][0] = Q[q + 1].a[1][0]; Pq[q].a[1][1] = Q[q + 1].a[1][1];
        }
        th[m] = std::atan2(s, c);  // R = (R^T)^T = [[c, -s], [s, c]]
    }
    const bool reflect0 = Pq[0].a[0][0] * Pq[0].a[1][1] - Pq[0].a[0][1] * Pq[0].a[1][0] < 0.0;
    th[0] = std::atan2(Pq[0].a[1][0], Pq[0].a[0][0]);

    // Levenberg-Marquardt on r(th) = taps(th) - taps, numeric Jacobian (2 L x K, K <= 16)
    const int R = 2 * L;
    std::vector<double> res(R), tl(L), thi(L), J((size_t)R * K), trial(K), r2(R);
    auto residual = [&](const double* t, double* out) {
        taps_of(t, reflect0, K, tl.data(), thi.data());
        double ss = 0.0;
        for (int i = 0; i < L; ++i) {
            out[i] = tl[i] - lo[i];
            out[L + i] = thi[i] - hi[i];
            ss += out[i] * out[i] + out[L + i] * out[L + i];
        }
        return ss;
    };
    double cost = residual(th.data(), res.data());
    double lambda = 1e-6;
    for (int it = 0; it < 200 && cost > 1e-30; ++it) {
        for (int k = 0; k < K; ++k) {
            const double h = 1e-6;
            std::vector<double> tp(th), tm(th);
            tp[k] += h;
            tm[k] -= h;
            std::vector<double> rp(R), rm(R);
            residual(tp.data(), rp.data());
            residual(tm.data(), rm.data());
            for (int i = 0; i < R; ++i) J[(size_t)i * K + k] = (rp[i] - rm[i]) / (2 * h);
        }
        // normal equations (J^T J + lambda diag) d = -J^T r
        std::vector<double> A((size_t)K * K, 0.0), gvec(K, 0.0);
        for (int i = 0; i < R; ++i)
            for (int a = 0; a < K; ++a) {
                gvec[a] += J[(size_t)i * K + a] * res[i];
                for (int c2 = 0; c2 < K; ++c2) A[(size_t)a * K + c2] += J[(size_t)i * K + a] * J[(size_t)i * K + c2];
            }
        bool improved = false;
        for (int tries = 0; tries < 12 && !improved; ++tries) {
            std::vector<double> Mx(A), rhs(K);
            for (int a = 0; a < K; ++a) {
                Mx[(size_t)a * K + a] += lambda * (A[(size_t)a * K + a] + 1e-300);
                rhs[a] = -gvec[a];
            }
            // Gaussian elimination with partial pivoting
            bool singular = false;
            for (int c2 = 0; c2 < K && !singular; ++c2) {
                int piv = c2;
                for (int r3 = c2 + 1; r3 < K; ++r3)
                    if (std::fabs(Mx[(size_t)r3 * K + c2]) > std::fabs(Mx[(size_t)piv * K + c2])) piv = r3;
                if (Mx[(size_t)piv * K + c2] == 0.0) { singular = true; break; }
                if (piv != c2) {
                    for (int k2 = 0; k2 < K; ++k2) std::swap(Mx[(size_t)piv * K + k2], Mx[(size_t)c2 * K + k2]);
                    std::swap(rhs[piv], rhs[c2]);
                }
                for (int r3 = c2 + 1; r3 < K; ++r3) {
                    const double f = Mx[(size_t)r3 * K + c2] / Mx[(size_t)c2 * K + c2];
                    for (int k2 = c2; k2 < K; ++k2) Mx[(size_t)r3 * K + k2] -= f * Mx[(size_t)c2 * K + k2];
                    rhs[r3] -= f * rhs[c2];
                }
            }
            if (!singular) {
                for (int r3 = K - 1; r3 >= 0; --r3) {
                    double v = rhs[r3];
                    for (int k2 = r3 + 1; k2 < K; ++k2) v -= Mx[(size_t)r3 * K + k2] * trial[k2];
                    trial[r3] = v / Mx[(size_t)r3 * K + r3];
                }
                for (int k = 0; k < K; ++k) trial[k] += th[k];
                const double c3 = residual(trial.data(), r2.data());
                if (c3 < cost) {
                    th.assign(trial.begin(), trial.end());
                    res = r2;
                    improved = true;
                    const bool done = cost - c3 <= 1e-3 * c3 && c3 < 1e-20;
                    cost = c3;
                    lambda = lambda > 1e-12 ? lambda * 0.1 : lambda;
                    if (done) it = 1000;
                    break;
                }
            }
            lambda *= 10.0;
        }
        if (!improved) break;
    }
    double worst = 0.0;
    for (int i = 0; i < R; ++i) worst = std::fabs(res[i]) > worst ? std::fabs(res[i]) : worst;
    if (fit_residual) *fit_residual = worst;
    // a few ulp of the float32 taps: real orthogonal tables fit to <= 2e-7 (tests/test_wpt_lattice.py); a bank that is
    // only approximately orthogonal (learned / hand-edited taps) must run through the direct-form kernels with ITS taps,
    // not through the nearest lattice -- six cascaded levels would turn a 1e-5 tap error into visible feature error
    if (!(worst <= 1e-6)) return false;
    // scaled one-FMA-per-stage form: A = G a, B = G k b (tools/wpt_lattice.py::scaled_form)
    double G = 1.0, k = 1.0;
    for (int s = 0; s < K; ++s) {
        const M2 r = rot(th[s], s == 0 && reflect0);
        if (std::fabs(r.a[0][0]) < 1e-9 || std::fabs(r.a[1][0]) < 1e-9) return false;
        alpha[s] = r.a[0][1] / (r.a[0][0] * k);
        beta[s] = r.a[1][1] / (r.a[1][0] * k);
        G = G / r.a[0][0];
        k = r.a[0][0] / r.a[1][0];
        // conditioning of the float32 recursion: a stage computes A + alpha B, so |alpha| is the factor by which B's
        // rounding error enters A (coif4's largest is 45, every other shipped table's below 10; the fuzz of
        // tools/wpt_fuzz.py holds the 5e-6 bar with them); near-degenerate rotations go to the direct form
        if (!(std::fabs(alpha[s]) < 1e3) || !(std::fabs(beta[s]) < 1e3)) return false;
    }
    scales[0] = 1.0 / G;
    scales[1] = 1.0 / (G * k);
    return std::isfinite(scales[0]) && std::isfinite(scales[1]) && scales[0] != 0.0 && scales[1] != 0.0;
}

}  // namespace afd

namespace {

struct LatEntry {
    int L;
    float lo[kMaxTaps], hi[kMaxTaps];
    bool ok;
    Lat4 lat;
    double oa, ob;
};
std::vector<LatEntry>& lat_cache() {
    static std::vector<LatEntry> v;
    return v;
}
std::mutex& lat_mutex() {
    static std::mutex m;
    return m;
}

}  // namespace

namespace afd {
namespace wptc {
bool wpt_get_lattice(const float* lo, const float* hi, int L, Lat4* out, double* oa, double* ob) {
    std::lock_guard<std::mutex> guard(lat_mutex());
    for (const LatEntry& e : lat_cache())
        if (e.L == L && !memcmp(e.lo, lo, L * sizeof(float)) && !memcmp(e.hi, hi, L * sizeof(float))) {
            *out = e.lat;
            *oa = e.oa;
            *ob = e.ob;
            return e.ok;
        }
    LatEntry e{};
    e.L = L;
    memcpy(e.lo, lo, L * sizeof(float));
    memcpy(e.hi, hi, L * sizeof(float));
    double al[kMaxStages], be[kMaxStages], sc[2];
    e.ok = afd::wpt_lattice_coefficients(lo, hi, L, al, be, sc, nullptr);
    if (e.ok) {
        for (int s = 0; s < L / 2; ++s) e.lat.ab[s] = f2{(float)al[s], (float)be[s]};
        e.lat.sc = f2{(float)sc[0], (float)sc[1]};
        e.oa = sc[0];
        e.ob = sc[1];
    }
    lat_cache().push_back(e);
    *out = e.lat;
    *oa = e.oa;
    *ob = e.ob;
    return e.ok;
}
}  // namespace wptc
}  // namespace afd

namespace {

template <int L, int MODE, bool SIGN, int GRP>
int launch4g(const D4Params& q, int B, hipStream_t stream) {
    constexpr size_t lds = (size_t)Plan4<L, GRP>::lds_floats * 4;
    constexpr int kD4Group = GRP, kD4Threads = 32 * GRP;
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wpt4_deep_kernel<L, MODE, SIGN, GRP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "wpt: hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr.mark();
    }
    hipLaunchKernelGGL((wpt4_deep_kernel<L, MODE, SIGN, GRP>), dim3((unsigned)B * (256 / kD4Group)), dim3(kD4Threads), lds,
                       stream, q);
    return afd::check_launch("wpt4_deep_kernel");
}

template <int L, int MODE, bool SIGN>
int launch4(const D4Params& q, int B, hipStream_t stream) {
    // 16 level-8 nodes per workgroup (512 threads): 4 / 8 measured 3-8 % slower at B = 4096, level at B = 128
    return launch4g<L, MODE, SIGN, 16>(q, B, stream);
}

template <int L>
int deep4(D4Params& q, int B, int mode, bool sign, hipStream_t stream) {
    if (mode == EPI_LOG2) return sign ? launch4<L, EPI_LOG2, true>(q, B, stream) : launch4<L, EPI_LOG2, false>(q, B, stream);
    if (mode == EPI_RAW) return sign ? launch4<L, EPI_RAW, true>(q, B, stream) : launch4<L, EPI_RAW, false>(q, B, stream);
    return sign ? launch4<L, EPI_SLOW, true>(q, B, stream) : launch4<L, EPI_SLOW, false>(q, B, stream);
}

}  // namespace

namespace afd {

bool wpt4_available(const float* dec_lo, const float* dec_hi, int L) {
    if (L != 24 && L != 10 && L != 16) return false;
    wptc::Lat4 lat{};
    double oa = 0.0, ob = 0.0;
    return wptc::wpt_get_lattice(dec_lo, dec_hi, L, &lat, &oa, &ob);
}

// Levels 9..14 from the level-8 hand-off image `ws` [B][n8][256] (wpt3_top_kernel<L, -1>) to the features.
// Returns AFD_OK, an error, or 1 = not this kernel's case (taps without an orthogonal lattice, other tap counts).
int wpt4_deep(const float* ws, float* out, int B, const float* dec_lo, const float* dec_hi, int L, unsigned flags,
              float power, float eps, float k1, float k0, float mean, float inv_std, float sgn_neg, float sgn_pos,
              hipStream_t stream) {
    if (L != 24 && L != 10 && L != 16) return 1;
    if ((long)B * 64 > 0x7fffffffL) return 1;
    D4Params q{};
    double oa = 0.0, ob = 0.0;
    if (!wpt_get_lattice(dec_lo, dec_hi, L, &q.lat, &oa, &ob)) return 1;
    q.ws = ws;
    q.out = out;
    q.e.flags = flags;
    q.e.power = power;
    q.e.eps = eps;
    q.e.k1 = k1;
    q.e.k0 = k0;
    q.e.mean = mean;
    q.e.inv_std = inv_std;
    q.e.sgn_neg = sgn_neg;
    q.e.sgn_pos = sgn_pos;
    const int mode = epi_mode(flags, power);
    const double sc[2] = {oa, ob};
    for (int c = 0; c < 2; ++c) {
        // raw: (A s) k1 + k0;   log2: log2((A s)^2 + eps) k1 + k0 = log2(A^2 + eps / s^2) k1 + (k0 + k1 log2 s^2)
        q.fk1[c] = (float)((double)k1 * sc[c]);
        q.fk0[c] = mode == EPI_LOG2 ? (float)((double)k0 + (double)k1 * std::log2(sc[c] * sc[c])) : k0;
        q.feps[c] = (float)((double)eps / (sc[c] * sc[c]));
    }
    const bool sign = flags & AFD_WPT_SIGN;
    switch (L) {
        case 24: return deep4<24>(q, B, mode, sign, stream);
        case 16: return deep4<16>(q, B, mode, sign, stream);
        case 10: return deep4<10>(q, B, mode, sign, stream);
        default: return 1;
    }
}

}  // namespace afd

// Development / test entry: the lattice of a tap table (host only, no GPU).
extern "C" int afd_wpt_lattice(const float* dec_lo, const float* dec_hi, int L, double* alpha, double* beta, double* scales,
                               double* fit_residual) {
    if (!dec_lo || !dec_hi || !alpha || !beta || !scales) return afd::fail(AFD_ERR_ARG, "wpt lattice: null pointer");
    if (L < 4 || (L & 1) || L > kMaxTaps) return afd::fail(AFD_ERR_ARG, "wpt lattice: %d taps", L);
    return afd::wpt_lattice_coefficients(dec_lo, dec_hi, L, alpha, beta, scales, fit_residual) ? AFD_OK : AFD_ERR_UNSUPPORTED;
}
