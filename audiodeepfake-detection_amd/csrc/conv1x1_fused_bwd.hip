// Backward of DCNN block 2 in ONE pass over the activations (reference src/audiofakedetect/models.py:260-264):
//
//     u --[BatchNorm2d(affine=False), folded into the weights]--> Conv2d(C, C, 1) --> z --PReLU--> P
//       --BatchNorm2d(affine=False)--> xhat
//
// Given g = dL/dxhat, the unfused chain is three passes over [N][C][HW] tensors (3.5 GB each at level 14):
// the BatchNorm backward (reads g, z, writes dz), the 1x1 backward-weight GEMM (reads dz, u) and the 1x1
// backward-data GEMM (reads dz, u, writes du) -- 28 GB.  Here a wave reads g, z and u of its 64 pixels
// once (14 GB in total with the store) and does all three:
//
//   dz[c][p]  = PReLU'(z) * (A[c] g + B[c] P + K[c])      the BatchNorm backward with the batch means folded
//                                                         into per-channel constants (coef, built by the caller:
//                                                         A = invstd, B = -invstd^2 E[g xhat],
//                                                         K = invstd^2 E[g xhat] mean - invstd E[g])
//   t[ci][p]  = sum_co wf[co][ci] dz[co][p]               GEMM 1, B operand straight from the lane's registers
//                                                         (lane = 2 pixels of channels 2 ks + lane / 32)
//   G[co][ci] += sum_p dz[co][p] u[ci][p]                 GEMM 2, K = pixels: both operands go through a
//                                                         wave-private LDS image [channel][pixel] and come back
//                                                         as 16-byte fragments (four k-steps per ds_read_b128)
//   db[co]    += sum_p dz[co][p],  dslope += sum_{z <= 0} dP z
//
// The affine term of the first BatchNorm's backward (alpha[ci] u + beta[ci], whose coefficients need the
// finished G) is added by the consumer of t (afd_conv1_pool_backward_affine reads t and u anyway).
// Waves are independent (no barrier after the weight image is in LDS); a wave keeps two 16-channel chunks
// of loads in flight (24 KB) ahead of its matrix instructions.  Bound: HBM (14 GB against 2 x 112 GFLOP).
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kCh = 64;      // channels on both sides, padded
constexpr int kPx = 64;      // pixels per wave tile (lane = 2 consecutive pixels)
constexpr int kKC = 8;       // k-steps (channel pairs) per load chunk
constexpr int kPitch = 68;   // LDS row pitch in floats: pitch / 4 odd -> conflict-free ds_read_b128 down a column
constexpr int kWaves = 4;
constexpr int kWtFloats = kCh * kPitch;
constexpr int kCoefFloats = kCh * 4;
constexpr int kTileFloats = 2 * kCh * kPitch;  // dz rows, then u rows
constexpr int kLdsFloats = kWtFloats + kCoefFloats + kWaves * kTileFloats;
constexpr int kSlab = kCh * kCh + kCh + 4;  // G, db, dslope (+ pad)

struct FB {
    const float* g;
    const float* z;
    const float* u;
    const float* wf;
    const float* coef;
    const float* slope;
    float* t;
    float* partial;
    int N, Cin, C, HW, tiles_per_img, ntiles;
    int xcd_m;
};

struct Chunk {
    f32x2 g[kKC], z[kKC], u[kKC];
};

// FULLC: both channel counts are exactly 64 (no channel guards in the load path)
template <bool FULLC>
__global__ void __launch_bounds__(kWaves * 64)
conv1x1_fused_bwd_kernel(const FB p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Wt = lds;                      // [ci][slot]: slot 8q + 4h + r <-> co = 2 (4q + r) + h
    float* coef = lds + kWtFloats;        // [c][4]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    float* T = lds + kWtFloats + kCoefFloats + wave * kTileFloats;

    for (int e = tid; e < kCh * kCh; e += kWaves * 64) {
        const int ci = e >> 6, slot = e & 63;
        const int q = slot >> 3, h = (slot >> 2) & 1, r = slot & 3;
        const int co = 2 * (4 * q + r) + h;
        Wt[ci * kPitch + slot] = (ci < p.Cin && co < p.C) ? p.wf[(size_t)co * p.Cin + ci] : 0.f;
    }
    for (int e = tid; e < kCoefFloats; e += kWaves * 64) coef[e] = (e >> 2) < p.C ? p.coef[e] : 0.f;
    __syncthreads();

    const float a = p.slope[0];
    const size_t HW = (size_t)p.HW;
    const int tstride = (int)gridDim.x * kWaves;
    // neighbouring tiles share the cache lines their 256-byte row segments start and end in: m consecutive workgroups
    // (4 m tiles) per XCD at a time (afd::xcd_grouped_id)
    int t = afd::xcd_grouped_id((int)blockIdx.x, (int)gridDim.x, p.xcd_m) * kWaves + wave;

    f32x16 acc2[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][j][r] = 0.f;
    float bs[2] = {0.f, 0.f};
    float ds = 0.f;

    // chunk cc of tile tt: channels 16 cc + 2 ks + half, pixels 2 l31, 2 l31 + 1 of the tile.
    // Addresses are a uniform tile pointer plus one 32-bit lane offset (an image is far below 4 GB).
    const unsigned lane_off = (unsigned)half * (unsigned)p.HW + 2u * (unsigned)l31;
    auto load_chunk = [&](int tt, int cc, Chunk& c) {
        const int n = tt / p.tiles_per_img;
        const int px0 = (tt - n * p.tiles_per_img) * kPx;
        const float* gb = p.g + (size_t)n * p.C * HW + px0;
        const float* zb = p.z + (size_t)n * p.C * HW + px0;
        const float* ub = p.u + (size_t)n * p.Cin * HW + px0;
        const unsigned row0 = (unsigned)(2 * cc * kKC) * (unsigned)p.HW;
        if (FULLC && px0 + kPx <= p.HW) {  // uniform: the whole tile is inside the image
#pragma unroll
            for (int ks = 0; ks < kKC; ++ks) {
                const unsigned o = row0 + (unsigned)(2 * ks) * (unsigned)p.HW + lane_off;
                c.g[ks] = *reinterpret_cast<const f32x2u*>(gb + o);
                c.z[ks] = *reinterpret_cast<const f32x2u*>(zb + o);
                c.u[ks] = *reinterpret_cast<const f32x2u*>(ub + o);
            }
        } else {
            // partial tile or fewer than 64 channels: element loads from clamped (always valid) addresses,
            // masked afterwards -- no divergent branches
            const int px = px0 + 2 * l31;
            const bool ok0 = px < p.HW, ok1 = px + 1 < p.HW;
            const unsigned q0 = (unsigned)min(px, p.HW - 1) - (unsigned)px0;
            const unsigned q1 = (unsigned)min(px + 1, p.HW - 1) - (unsigned)px0;
#pragma unroll
            for (int ks = 0; ks < kKC; ++ks) {
                const int ch = (cc * kKC + ks) * 2 + half;
                const unsigned rc = (unsigned)min(ch, p.C - 1) * (unsigned)p.HW;
                const unsigned ri = (unsigned)min(ch, p.Cin - 1) * (unsigned)p.HW;
                const float g0 = gb[rc + q0], g1 = gb[rc + q1], z0 = zb[rc + q0], z1 = zb[rc + q1];
                const float u0 = ub[ri + q0], u1 = ub[ri + q1];
                const bool cok = ch < p.C, iok = ch < p.Cin;
                c.g[ks] = f32x2{(cok && ok0) ? g0 : 0.f, (cok && ok1) ? g1 : 0.f};
                c.z[ks] = f32x2{(cok && ok0) ? z0 : 0.f, (cok && ok1) ? z1 : 0.f};
                c.u[ks] = f32x2{(iok && ok0) ? u0 : 0.f, (iok && ok1) ? u1 : 0.f};
            }
        }
        asm volatile("" ::: "memory");  // the loads are issued here, not where the scheduler would like them
    };

    if (t >= p.ntiles) return;
    Chunk cb0, cb1;
    load_chunk(t, 0, cb0);
    load_chunk(t, 1, cb1);

    f32x16 acc1[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[m][i][r] = 0.f;

    // one chunk: BatchNorm / PReLU backward in registers, the wave's LDS image, GEMM 1
    auto process = [&](const Chunk& c, int cc, bool ok0, bool ok1) {
        // weight fragments of this chunk's 8 k-steps: two 16-byte reads per row tile
        f32x4 wa[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
                wa[m][qq] = *reinterpret_cast<const f32x4*>(Wt + (m * 32 + l31) * kPitch + 8 * (2 * cc + qq) + 4 * half);
        const float* kp = coef + 4 * (2 * cc * kKC + half);
        float* Tw = T + (2 * cc * kKC + half) * kPitch + 2 * l31;
        // per k-step: the vector work of the step, then its four matrix instructions -- the wave is alone on its
        // SIMD (LDS-limited occupancy), so the vector work of step ks + 1 has to run under the matrix
        // instructions of step ks; the scheduling barriers keep that order (3.87 -> 3.66 ms at level 14)
#pragma unroll
        for (int ks = 0; ks < kKC; ++ks) {
            const f32x4 k = *reinterpret_cast<const f32x4*>(kp + 8 * ks);
            f32x2 d;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float zz = c.z[ks][i], gg = c.g[ks][i];
                const bool neg = zz <= 0.f;
                const float P = neg ? a * zz : zz;
                float dP = fmaf(k.x, gg, fmaf(k.y, P, k.z));
                dP = (i ? ok1 : ok0) ? dP : 0.f;
                ds += neg ? dP * zz : 0.f;
                d[i] = neg ? a * dP : dP;
            }
            // the wave's LDS image, [channel][pixel]
            *reinterpret_cast<f32x2*>(Tw + 2 * ks * kPitch) = d;
            *reinterpret_cast<f32x2*>(Tw + (kCh + 2 * ks) * kPitch) = c.u[ks];
            // GEMM 1: t[ci][px] += wf[co][ci] dz[co][px]
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc1[m][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[m][ks >> 2][ks & 3], d[i], acc1[m][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" ::: "memory");
    };

    // Two chunks per iteration (buffers cb0 / cb1 keep their roles), two iterations per tile; every buffer is
    // refilled with the chunk two ahead -- of this tile or of the wave's next one -- as soon as it is consumed.
#pragma unroll 1
    for (int it = 0;; ++it) {
        const int cc0 = (it & 1) * 2;
        const int tn = t + tstride;
        const bool more = tn < p.ntiles;
        const int n = t / p.tiles_per_img;
        const int px0 = (t - n * p.tiles_per_img) * kPx;
        const int px = px0 + 2 * l31;
        const bool ok0 = px < p.HW, ok1 = px + 1 < p.HW;
        const int tnext = cc0 ? tn : t;  // tile of the chunks two ahead

        process(cb0, cc0, ok0, ok1);
        if (cc0 == 0 || more) load_chunk(tnext, cc0 ^ 2, cb0);
        process(cb1, cc0 + 1, ok0, ok1);
        if (cc0 == 0 || more) load_chunk(tnext, (cc0 ^ 2) + 1, cb1);
        if (cc0 == 0) continue;

        // ---- end of the tile ----
        // t leaves: D column = lane & 31 -> pixels 2 l31 + i, row = (r & 3) + 8 (r >> 2) + 4 half -> ci.  Whole
        // tiles: the 32 row stores are dealt out over GEMM 2's eight rounds (four after each round's matrix
        // instructions) -- a CU drains ~6 B per cycle to HBM, so a burst of 4 x 32 KB would hold its four waves for
        // ~20k cycles with nothing else to issue (measured on the 1x1 forward: the store burst is 1.0 of its 2.3 ms)
        float* tb = p.t + (size_t)n * p.Cin * HW + px0;
        const bool whole = FULLC && px0 + kPx <= p.HW;
        const unsigned lo = 4u * (unsigned)half * (unsigned)p.HW + 2u * (unsigned)l31;
        if (!whole) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float* o = tb + (size_t)ci * HW + 2 * l31;
                    if (ci < p.Cin && ok0) o[0] = acc1[m][0][r];
                    if (ci < p.Cin && ok1) o[1] = acc1[m][1][r];
                }
        }

        // GEMM 2 from the wave's own image (LDS operations of a wave execute in order; the fences keep the
        // compiler from moving the reads above the other lanes' writes, and the next tile's writes above them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < kPx / 8; ++q) {
            f32x4 am[2], bj[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                am[m] = *reinterpret_cast<const f32x4*>(T + (m * 32 + l31) * kPitch + 8 * q + 4 * half);
                bj[m] = *reinterpret_cast<const f32x4*>(T + (kCh + m * 32 + l31) * kPitch + 8 * q + 4 * half);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) bs[m] += (am[m].x + am[m].y) + (am[m].z + am[m].w);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc2[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[m][r], bj[j][r], acc2[m][j], 0, 0, 0);
            if (whole) {
#pragma unroll
                for (int u4 = 0; u4 < 4; ++u4) {
                    const int e = 4 * q + u4, m = e >> 4, r = e & 15;
                    const unsigned row = (unsigned)(m * 32 + (r & 3) + 8 * (r >> 2)) * (unsigned)p.HW;
                    *reinterpret_cast<f32x2u*>(tb + (row + lo)) = f32x2{acc1[m][0][r], acc1[m][1][r]};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[m][i][r] = 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (!more) break;
        t = tn;
    }

    // one partial slab per wave: G [co][ci], db [co], dslope
    float* slab = p.partial + ((size_t)blockIdx.x * kWaves + wave) * kSlab;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                slab[co * kCh + j * 32 + l31] = acc2[m][j][r];
            }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const float v = bs[m] + __shfl_xor(bs[m], 32, 64);
        if (half == 0) slab[kCh * kCh + m * 32 + l31] = v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ds += __shfl_xor(ds, off, 64);
    if (lane == 0) slab[kCh * kCh + kCh] = ds;
}

// 64 outputs per workgroup, 16 slab groups (waves) each: coalesced reads, fixed summation order
__global__ void __launch_bounds__(1024)
conv1x1_fused_bwd_reduce_kernel(const float* __restrict__ partial, int nslabs, int Cin, int C,
                                float* __restrict__ G, float* __restrict__ db, float* __restrict__ dslope) {
    __shared__ float red[16][64];
    const int o = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + o;
    float s = 0.f;
    if (e <= kCh * kCh + kCh)
        for (int b = sg; b < nslabs; b += 16) s += partial[(size_t)b * kSlab + e];
    red[sg][o] = s;
    __syncthreads();
    if (sg != 0 || e > kCh * kCh + kCh) return;
    s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][o];
    if (e < kCh * kCh) {
        const int co = e >> 6, ci = e & 63;
        if (co < C && ci < Cin) G[(size_t)co * Cin + ci] = s;
    } else if (e < kCh * kCh + kCh) {
        const int co = e - kCh * kCh;
        if (co < C) db[co] = s;
    } else if (s != 0.f) {
        atomicAdd(dslope, s);
    }
}

int fused_blocks() {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1)
        c = 256;
    return c;  // one 4-wave workgroup per CU (158 KB of LDS)
}

}  // namespace

extern "C" int afd_conv1x1_prelu_bn_backward_applicable(int Cin, int C) {
    return !getenv("AFD_NO_BLOCK2_FUSE") && Cin >= 1 && Cin <= kCh && C >= 1 && C <= kCh;
}

extern "C" size_t afd_conv1x1_prelu_bn_backward_workspace_bytes(int Cin, int C) {
    (void)Cin;
    (void)C;
    return (size_t)fused_blocks() * kWaves * kSlab * sizeof(float);
}

extern "C" int afd_conv1x1_prelu_bn_backward(const float* g, const float* z, const float* u, const float* wf,
                                             const float* coef, const float* slope, float* t, float* G,
                                             float* db, float* dslope, int N, int Cin, int C, long HW,
                                             void* ws, size_t ws_bytes, afd_stream_t stream) {
    if (!g || !z || !u || !wf || !coef || !slope || !t || !G || !db || !dslope)
        return afd::fail(AFD_ERR_ARG, "conv1x1 fused backward: null pointer");
    if (N < 1 || HW < 1 || HW > 0x7fffffffL) return afd::fail(AFD_ERR_ARG, "conv1x1 fused backward: bad shape");
    if (!afd_conv1x1_prelu_bn_backward_applicable(Cin, C))
        return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1 fused backward: %d -> %d channels (at most 64 each)", Cin, C);
    if (!ws || ws_bytes < afd_conv1x1_prelu_bn_backward_workspace_bytes(Cin, C))
        return afd::fail(AFD_ERR_WORKSPACE, "conv1x1 fused backward: workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    FB p{};
    p.g = g; p.z = z; p.u = u; p.wf = wf; p.coef = coef; p.slope = slope; p.t = t;
    p.partial = static_cast<float*>(ws);
    p.N = N; p.Cin = Cin; p.C = C; p.HW = (int)HW;
    p.tiles_per_img = (int)((HW + kPx - 1) / kPx);
    const long nt = (long)N * p.tiles_per_img;
    if (nt > 0x7fffffffL) return afd::fail(AFD_ERR_UNSUPPORTED, "conv1x1 fused backward: too many tiles");
    p.ntiles = (int)nt;
    p.xcd_m = 16;  // same box, alternating runs: plain order 3.47 ms, m = 4 / 16 / 64: 3.37 / 3.33 / 3.46 ms
    long blocks = (nt + kWaves - 1) / kWaves;
    const int cap = fused_blocks();
    if (blocks > cap) blocks = cap;
    constexpr size_t lds = (size_t)kLdsFloats * sizeof(float);
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_fused_bwd_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_fused_bwd_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "conv1x1 fused backward: %s", hipGetErrorString(e));
        attr.mark();
    }
    // waves past the tile count return before writing a slab: zero the workspace rows they own
    const int nslabs = (int)blocks * kWaves;
    if ((long)nslabs > nt) {
        hipError_t e = hipMemsetAsync(ws, 0, (size_t)nslabs * kSlab * sizeof(float), s);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "conv1x1 fused backward: %s", hipGetErrorString(e));
    }
    afd::ScopedTiming timing(AFD_K_CONV_WGRAD_1X1, 4.0 * N * (double)C * (double)HW * Cin, s);
    timing.issued(4.0 * N * (double)kCh * (double)HW * kCh);
    timing.bytes(4.0 * N * (double)HW * (2.0 * C + 2.0 * Cin));
    if (Cin == kCh && C == kCh)
        hipLaunchKernelGGL(conv1x1_fused_bwd_kernel<true>, dim3((unsigned)blocks), dim3(kWaves * 64), lds, s, p);
    else
        hipLaunchKernelGGL(conv1x1_fused_bwd_kernel<false>, dim3((unsigned)blocks), dim3(kWaves * 64), lds, s, p);
    int rc = afd::check_launch("conv1x1_fused_bwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(conv1x1_fused_bwd_reduce_kernel, dim3((kCh * kCh + kCh + 1 + 63) / 64), dim3(1024), 0, s,
                       p.partial, nslabs, Cin, C, G, db, dslope);
    return afd::check_launch("conv1x1_fused_bwd_reduce_kernel");
}
