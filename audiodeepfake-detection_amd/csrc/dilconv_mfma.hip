// The DCNN's dilated stack at 5..16 channels (the level-8 / STFT models: time_dim = 12 or 13) on the f32 MFMA, with
// the image resident in LDS (round 5).
//
// Reference models.py:286-300: three Conv2d(time_dim, time_dim, k, padding, dilation) with (k, pad, dil) = (3,1,1),
// (5,2,2), (7,2,4) on [B, time_dim, 64, P/8] -- 1.4 MB of activations per layer at B = 128, 0.8-1.8 GFLOP per launch.
// Until round 4 these ran on the table-driven implicit GEMM of conv.hip (83 us per forward / backward-data launch,
// 50-215 us per backward-weight launch: 0.87 of the 5.9 ms of a level-8 step); the few-channel direct kernels of
// dilconv.hip stop at 4 channels, and a vector-ALU kernel for 12 / 13 channels measured level.
//
// Here a workgroup owns a band of output rows of one image.  The rows of the input it needs -- zero padding materialised,
// the real channels only -- sit in LDS as [C][rows][Wp]; a convolution tap is then a constant offset into that image:
//
//   forward / backward-data   GEMM per tap: D[co][px] += W_tap[co][ci] . X[ci][px + tap] on v_mfma_f32_16x16x4_f32, K index
//                             = 4 j + (lane >> 4) over the 16 padded channels: the lane's B value is ONE ds_read_b32 whose
//                             address is a per-lane base (channel group, pixel) + an instruction immediate (kx dil) -- no
//                             vector arithmetic per tap but one add per kernel row; A comes from a fragment-ordered
//                             table [tap][lane][4] (one 16-byte load per tap, built by a 5 us kernel from w; the
//                             backward-data table is transposed and flipped, its padding (K-1) dil - pad).
//   backward-weight           per tap a 16 x 16 accumulator: D_tap[co][ci] += dy[co][px] . X[ci][px + tap], K index = 4
//                             consecutive pixels; dy is read once per pixel group and multiplied against all K^2 taps;
//                             4 waves x K^2 accumulators, summed through LDS, one partial slab per workgroup, a second
//                             kernel adds the slabs in a fixed order (deterministic).
//
// Measured at the level-8 geometry (B = 128, 13 channels; tools/dil_time.py, us per launch incl. the table kernel):
//   forward 38 / 51 / 35 (k = 3 / 5 / 7; implicit GEMM: 49 / 93 / 82), backward-data 35 / 56 / 79 (46 / 92 / 170),
//   backward-weight 41 / 58 / 62 (65 / 231 / 113): the stack's nine launches 943 -> 453 us per step.
// Ablation builds: without the matrix loop a launch is 20-25 us (table kernel 5, band staging 11 -- its loads are two
// rounds of memory latency with nothing to hide them --, stores), the loop itself runs at 70-80 % of the f32 matrix
// rate (k = 7 backward-data: 53 us for 41 us of matrix instructions; 13 of 16 rows and columns are real).
//
// fp32 end to end (exact f32 matrix instructions): the 2e-5 / 3e-5 bars of tests/test_nn_gpu.py::test_conv2d_forward_backward.
#include "afd_common.h"
#include "../../include/afd_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreadsD = 512;  // 8 waves: two per SIMD (a wave holds 16 + K^2 x 4 accumulators at most)
constexpr int kLdsBudget = 150 * 1024;  // bytes of the image band

struct DM {
    int N, C, Hin, Win, Hout, Wout, pad;  // out = in + 2 pad - (K-1) dil
    int Wp;                               // padded row length of the LDS image: Win + 2 pad
    int R, parts;                         // output rows per workgroup, workgroups per image
    int in_rows;                          // staged rows: R + (K-1) dil
    int plane;                            // floats per staged channel: in_rows * Wp
};

// band geometry: the fewest parts per image whose band fits the LDS budget, at least two workgroups per CU's worth of work
template <int K, int DIL>
bool plan(DM& g, int N, int C, int Hin, int Win, int pad) {
    g.N = N; g.C = C; g.Hin = Hin; g.Win = Win; g.pad = pad;
    g.Hout = Hin + 2 * pad - DIL * (K - 1);
    g.Wout = Win + 2 * pad - DIL * (K - 1);
    if (g.Hout < 1 || g.Wout < 1) return false;
    g.Wp = Win + 2 * pad;
    for (int parts = 1; parts <= g.Hout; ++parts) {
        const int R = (g.Hout + parts - 1) / parts;
        const long bytes = (long)C * (R + (K - 1) * DIL) * g.Wp * 4;
        if (bytes > kLdsBudget) continue;
        if ((long)N * parts < 256 && R > 4) continue;  // fill the chip when the image allows it
        g.R = R;
        g.parts = (g.Hout + R - 1) / R;
        g.in_rows = R + (K - 1) * DIL;
        g.plane = g.in_rows * g.Wp;
        return true;
    }
    return false;
}

// the staged band of image n, part q: channel c, padded row pr (image row r0 + pr - pad), padded column pc.  A wave takes
// whole rows (channel and row arithmetic is scalar), its lanes the columns.
__device__ __forceinline__ void stage_band(float* __restrict__ lds, const float* __restrict__ xn, const DM& g, int r0, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = kThreadsD / 64;
    constexpr int U = 16;  // rows in flight per wave: a batch's loads are issued together, then stored (the band arrives
                           // in a few rounds of memory latency instead of one round per row)
    const int live = g.C * g.in_rows;  // only the real channels are staged: the matrix rows / columns of the padding
                                       // channels meet zero weights (conv) or are never stored (backward-weight) and
                                       // read the last real channel's plane
    for (int pc0 = 0; pc0 < g.Wp; pc0 += 64) {
        const int pc = pc0 + lane;
        const int ix = pc - g.pad;
        const bool cok = pc < g.Wp && ix >= 0 && ix < g.Win;
        for (int base = wave; base < live; base += NW * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = base + u * NW;
                const int c = row / g.in_rows, pr = row - c * g.in_rows;
                const int iy = r0 + pr - g.pad;
                const bool rok = row < live && iy >= 0 && iy < g.Hin;  // uniform
                const float* src = xn + ((size_t)(rok ? c : 0) * g.Hin + (rok ? iy : 0)) * g.Win;
                v[u] = (rok && cok) ? src[ix] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int row = base + u * NW;
                if (row < live && pc < g.Wp) lds[(size_t)row * g.Wp + pc] = v[u];
            }
        }
    }
}

// Wt[tap][lane][j] = A[m = lane & 15][k = 4 j + (lane >> 4)] of tap (ky, kx):
//   forward        A[co][ci] = w[co][ci][ky][kx]
//   backward-data  A[ci][co] = w[co][ci][K-1-ky][K-1-kx]      (result channel = the forward's input channel)
__global__ void dilmfma_weights_kernel(const float* __restrict__ w, float* __restrict__ Wt, int C, int K, int dgrad) {
    const int total = K * K * 256;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i & 3, lane = (i >> 2) & 63, tap = i >> 8;
        const int m = lane & 15, k = 4 * j + (lane >> 4);
        const int ky = tap / K, kx = tap - ky * K;
        float v = 0.f;
        if (m < C && k < C)
            v = dgrad ? w[(((size_t)k * C + m) * K + (K - 1 - ky)) * K + (K - 1 - kx)] : w[(((size_t)m * C + k) * K + ky) * K + kx];
        Wt[i] = v;
    }
}

template <int K, int DIL>
__global__ void __launch_bounds__(kThreadsD)
dilmfma_conv_kernel(const DM g, const float* __restrict__ x, const float* __restrict__ Wt, const float* __restrict__ bias,
                    float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float img[];  // [C][in_rows][Wp]
    constexpr int NT = 2;  // 16-pixel tiles per wave and chunk (32-pixel chunks: an 840-pixel band is 3.3 chunks per wave)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / g.parts, q = blockIdx.x - n * g.parts;
    const int r0 = q * g.R;
    const int rows = g.Hout - r0 < g.R ? g.Hout - r0 : g.R;
    stage_band(img, x + (size_t)n * g.C * g.Hin * g.Win, g, r0, tid);
    __syncthreads();
    const int npx = rows * g.Wout;
    const int pl = lane & 15, kg = lane >> 4;
    const f32x4* __restrict__ wt4 = reinterpret_cast<const f32x4*>(Wt) + lane;
    const size_t oplane = (size_t)g.Hout * g.Wout;
    float* yn = y + (size_t)n * g.C * oplane + (size_t)r0 * g.Wout;
    for (int chunk = wave; chunk * (16 * NT) < npx; chunk += kThreadsD / 64) {
        // byte addresses of this lane's pixel in its four channel planes 4 j + kg (the instruction immediates reach the kx taps)
        unsigned bj[NT][4];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            int p = chunk * (16 * NT) + 16 * i + pl;
            p = p < npx ? p : npx - 1;  // lanes past the band read its last pixel, their results are not stored
            const int oy = p / g.Wout, ox = p - oy * g.Wout;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ci = 4 * j + kg < g.C ? 4 * j + kg : g.C - 1;  // padding channels: zero weights, any finite sample
                bj[i][j] = (unsigned)((ci * g.plane + oy * g.Wp + ox) * 4);
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned rowstep = (unsigned)(DIL * g.Wp * 4);  // bytes between two kernel rows
        // the weight fragments of kernel row ky + 1 are requested before the matrix instructions of row ky (they come from
        // L2: one exposed round trip per kernel row and chunk otherwise)
        f32x4 an[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) an[kx] = wt4[kx * 64];
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            f32x4 ac[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) ac[kx] = an[kx];
            if (ky + 1 < K) {
#pragma unroll
                for (int kx = 0; kx < K; ++kx) an[kx] = wt4[((ky + 1) * K + kx) * 64];
            }
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 a = ac[kx];
                const unsigned off = (unsigned)(kx * DIL * 4);
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const float b0 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(img) + bj[i][0] + off);
                    const float b1 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(img) + bj[i][1] + off);
                    const float b2 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(img) + bj[i][2] + off);
                    const float b3 = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(img) + bj[i][3] + off);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, acc[i], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) bj[i][j] += rowstep;
        }
        // D: lane holds rows (output channels) 4 kg + r, column (pixel) pl of each tile
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int p = chunk * (16 * NT) + 16 * i + pl;
            if (p < npx) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 4 * kg + r;
                    if (co < g.C) yn[(size_t)co * oplane + p] = acc[i][r] + (bias ? bias[co] : 0.f);
                }
            }
        }
    }
}

// slab of a workgroup: [K*K][16 co][16 ci] partial weight gradients, then [16] partial bias gradients
template <int K>
constexpr int slab_floats() { return K * K * 256 + 16; }

template <int K, int DIL>
__global__ void __launch_bounds__(kThreadsD)
dilmfma_wgrad_kernel(const DM g, const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float img[];  // [C][in_rows][Wp]; reused for the cross-wave sum
    constexpr int T = K * K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / g.parts, q = blockIdx.x - n * g.parts;
    const int r0 = q * g.R;
    const int rows = g.Hout - r0 < g.R ? g.Hout - r0 : g.R;
    stage_band(img, x + (size_t)n * g.C * g.Hin * g.Win, g, r0, tid);
    __syncthreads();
    const int npx = rows * g.Wout;
    const int lc = lane & 15, pk = lane >> 4;  // A: (co = lc, pixel pk); B: (pixel pk, ci = lc)
    const size_t oplane = (size_t)g.Hout * g.Wout;
    const float* dyn = dy + (size_t)n * g.C * oplane + (size_t)r0 * g.Wout + (size_t)(lc < g.C ? lc : 0) * oplane;
    f32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    const unsigned rowstep = (unsigned)(DIL * g.Wp * 4);
    for (int p0 = 4 * wave; p0 < npx; p0 += 4 * (kThreadsD / 64)) {
        const int p = p0 + pk;
        const bool ok = p < npx;
        const int pc = ok ? p : npx - 1;
        const float a = (ok && lc < g.C) ? dyn[pc] : 0.f;
        bsum += a;
        const int oy = pc / g.Wout, ox = pc - oy * g.Wout;
        unsigned base = (unsigned)(((lc < g.C ? lc : g.C - 1) * g.plane + oy * g.Wp + ox) * 4);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const float b = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(img) + base + (unsigned)(kx * DIL * 4));
                acc[ky * K + kx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[ky * K + kx], 0, 0, 0);
            }
            base += rowstep;
        }
    }
    // sum the four waves through LDS (the image is dead) in wave order -- a fixed order: deterministic -- one slab per
    // workgroup.  (All four accumulator sets at once would be 4 x K^2 KB: 196 KB at K = 7.)
    __syncthreads();
    float* red = img;                      // [T][64 lanes][4]
    float* bred = red + (size_t)T * 256;   // [waves][64 lanes]
    bred[wave * 64 + lane] = bsum;
    for (int w4 = 0; w4 < kThreadsD / 64; ++w4) {
        if (wave == w4) {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                f32x4* cell = reinterpret_cast<f32x4*>(red + ((size_t)t * 64 + lane) * 4);
                *cell = w4 == 0 ? acc[t] : *cell + acc[t];
            }
        }
        __syncthreads();
    }
    float* slab = slabs + (size_t)blockIdx.x * slab_floats<K>();
    for (int e = tid; e < T * 256; e += kThreadsD) {
        // element e = (tap, lane, r): D[co = 4 (lane >> 4) + r][ci = lane & 15]
        const int r = e & 3, l = (e >> 2) & 63, t = e >> 8;
        slab[t * 256 + (4 * (l >> 4) + r) * 16 + (l & 15)] = red[e];
    }
    if (tid < 16) {
        float s = 0.f;
        for (int w4 = 0; w4 < kThreadsD / 64; ++w4)
            for (int k4 = 0; k4 < 4; ++k4) s += bred[w4 * 64 + 16 * k4 + tid];
        slab[T * 256 + tid] = s;
    }
}

// Sum of the slabs in two coalesced stages (fixed order: deterministic).  Stage 1: thread = slab element, block row s adds
// the slabs s, s + kRedSplits, ... into part[s][element]; stage 2 adds the kRedSplits partials and writes
// dw[co][ci][ky][kx] / dbias[co].  (One thread per RESULT walking 512 slabs at a 50 KB stride took 37 us.)
constexpr int kRedSplits = 16;
__global__ void dilmfma_reduce1_kernel(const float* __restrict__ slabs, int nslabs, int slab_len, float* __restrict__ part) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= slab_len) return;
    const int s0 = blockIdx.y;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int s = s0;
    for (; s + 3 * kRedSplits < nslabs; s += 4 * kRedSplits) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] += slabs[(size_t)(s + u * kRedSplits) * slab_len + e];
    }
    for (; s < nslabs; s += kRedSplits) a[0] += slabs[(size_t)s * slab_len + e];
    part[(size_t)s0 * slab_len + e] = (a[0] + a[1]) + (a[2] + a[3]);
}

__global__ void dilmfma_reduce2_kernel(const float* __restrict__ part, int slab_len, int C, int K, float* __restrict__ dw,
                                       float* __restrict__ dbias) {
    const int T = K * K;
    const int total = C * C * T + (dbias ? C : 0);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int src;
    if (i < C * C * T) {
        const int t = i % T, ci = (i / T) % C, co = i / (T * C);
        src = t * 256 + co * 16 + ci;
    } else {
        src = T * 256 + (i - C * C * T);
    }
    float v = 0.f;
#pragma unroll
    for (int s = 0; s < kRedSplits; ++s) v += part[(size_t)s * slab_len + src];
    if (i < C * C * T) dw[i] = v;
    else dbias[i - C * C * T] = v;
}

template <int K, int DIL>
int run_conv(const float* x, const float* w, const float* bias, float* y, int N, int C, int Hin, int Win, int pad, int dgrad,
             void* ws, size_t ws_bytes, hipStream_t s) {
    DM g{};
    if (!plan<K, DIL>(g, N, C, Hin, Win, pad)) return afd::fail(AFD_ERR_UNSUPPORTED, "dilated conv (mfma): geometry does not fit");
    if (!ws || ws_bytes < (size_t)K * K * 256 * sizeof(float)) return afd::fail(AFD_ERR_WORKSPACE, "dilated conv (mfma): workspace too small");
    float* Wt = static_cast<float*>(ws);
    hipLaunchKernelGGL(dilmfma_weights_kernel, dim3((K * K * 256 + 255) / 256), dim3(256), 0, s, w, Wt, C, K, dgrad);
    const size_t lds = (size_t)C * g.plane * sizeof(float);
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dilmfma_conv_kernel<K, DIL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "dilated conv (mfma): %s", hipGetErrorString(e));
        attr.mark();
    }
    hipLaunchKernelGGL((dilmfma_conv_kernel<K, DIL>), dim3((unsigned)(N * g.parts)), dim3(kThreadsD), lds, s, g, x, Wt, bias, y);
    return afd::check_launch("dilmfma_conv_kernel");
}

template <int K, int DIL>
int run_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int Hin, int Win, int pad, void* ws,
              size_t ws_bytes, hipStream_t s) {
    DM g{};
    if (!plan<K, DIL>(g, N, C, Hin, Win, pad)) return afd::fail(AFD_ERR_UNSUPPORTED, "dilated wgrad (mfma): geometry does not fit");
    const int nslabs = N * g.parts;
    if (!ws || ws_bytes < (size_t)(nslabs + kRedSplits) * slab_floats<K>() * sizeof(float))
        return afd::fail(AFD_ERR_WORKSPACE, "dilated wgrad (mfma): workspace too small");
    // the cross-wave sum reuses the image: it needs K^2 x 256 floats + one bias partial per thread
    size_t lds = (size_t)C * g.plane * sizeof(float);
    const size_t red = (size_t)(K * K * 256 + kThreadsD) * sizeof(float);
    if (red > lds) lds = red;
    if (lds > (size_t)afd::kLdsBytes) return afd::fail(AFD_ERR_UNSUPPORTED, "dilated wgrad (mfma): LDS");
    static afd::PerDeviceOnce attr;
    if (!attr.done()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dilmfma_wgrad_kernel<K, DIL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, afd::kLdsBytes);
        if (e != hipSuccess) return afd::fail(AFD_ERR_HIP, "dilated wgrad (mfma): %s", hipGetErrorString(e));
        attr.mark();
    }
    float* slabs = static_cast<float*>(ws);
    hipLaunchKernelGGL((dilmfma_wgrad_kernel<K, DIL>), dim3((unsigned)nslabs), dim3(kThreadsD), lds, s, g, x, dy, slabs);
    float* part = slabs + (size_t)nslabs * slab_floats<K>();
    hipLaunchKernelGGL(dilmfma_reduce1_kernel, dim3((slab_floats<K>() + 255) / 256, kRedSplits), dim3(256), 0, s, slabs, nslabs,
                       slab_floats<K>(), part);
    const int total = C * C * K * K + (dbias ? C : 0);
    hipLaunchKernelGGL(dilmfma_reduce2_kernel, dim3((total + 255) / 256), dim3(256), 0, s, part, slab_floats<K>(), C, K, dw, dbias);
    return afd::check_launch("dilmfma_wgrad kernels");
}

template <int K, int DIL>
size_t ws_bytes_for(int N, int C, int Hin, int Win, int pad) {
    DM g{};
    size_t b = (size_t)K * K * 256 * sizeof(float);
    if (plan<K, DIL>(g, N, C, Hin, Win, pad)) {
        const size_t s = (size_t)(N * g.parts + kRedSplits) * slab_floats<K>() * sizeof(float);
        if (s > b) b = s;
    }
    return b;
}

}  // namespace

namespace afd {

// Cin == Cout in 5..16, the three (K, dilation) pairs of the reference's stack, and a band that fits LDS -- forward
// geometry (H, W, pad); the backward-data launch (input = the forward's output, padding (K-1) dil - pad) must fit too.
// The batch size does not enter feasibility (plan<> asks for at least one row band per workgroup, R = 1 always fits when a
// band fits; N only sets the grid), so the probe plans with a nominal N = 128 while the launches plan with the real one.
bool dilmfma_applicable(int Cin, int Cout, int H, int W, int K, int pad, int dil) {
    if (getenv("AFD_NO_DIRECT_CONV")) return false;
    if (Cin != Cout || Cin < 5 || Cin > 16 || pad < 0) return false;
    const int Ho = H + 2 * pad - dil * (K - 1), Wo = W + 2 * pad - dil * (K - 1);
    if (Ho < 1 || Wo < 1) return false;
    const int padb = dil * (K - 1) - pad;
    if (padb < 0) return false;
    DM a{}, b{};
    if (K == 3 && dil == 1) return plan<3, 1>(a, 128, Cin, H, W, pad) && plan<3, 1>(b, 128, Cin, Ho, Wo, padb);
    if (K == 5 && dil == 2) return plan<5, 2>(a, 128, Cin, H, W, pad) && plan<5, 2>(b, 128, Cin, Ho, Wo, padb);
    if (K == 7 && dil == 4) return plan<7, 4>(a, 128, Cin, H, W, pad) && plan<7, 4>(b, 128, Cin, Ho, Wo, padb);
    return false;
}

size_t dilmfma_workspace_bytes(int N, int C, int H, int W, int K, int pad, int dil) {
    if (K == 3 && dil == 1) return ws_bytes_for<3, 1>(N, C, H, W, pad);
    if (K == 5 && dil == 2) return ws_bytes_for<5, 2>(N, C, H, W, pad);
    if (K == 7 && dil == 4) return ws_bytes_for<7, 4>(N, C, H, W, pad);
    return 0;
}

static double conv_flops(int N, int C, int K, int Ho, int Wo) { return 2.0 * N * C * C * K * K * (double)Ho * Wo; }
static double issued_flops(int N, int K, int Ho, int Wo) { return 2.0 * N * 16.0 * 16.0 * K * K * (double)((Ho * Wo + 15) / 16 * 16); }

int dilmfma_forward(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int K, int pad,
                    int dil, void* ws, size_t ws_bytes, hipStream_t s) {
    const int Ho = H + 2 * pad - dil * (K - 1), Wo = W + 2 * pad - dil * (K - 1);
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, conv_flops(N, C, K, Ho, Wo), s);
    timing.issued(issued_flops(N, K, Ho, Wo));
    timing.bytes(4.0 * N * C * ((double)H * W + (double)Ho * Wo));
    if (K == 3) return run_conv<3, 1>(x, w, bias, y, N, C, H, W, pad, 0, ws, ws_bytes, s);
    if (K == 5) return run_conv<5, 2>(x, w, bias, y, N, C, H, W, pad, 0, ws, ws_bytes, s);
    return run_conv<7, 4>(x, w, bias, y, N, C, H, W, pad, 0, ws, ws_bytes, s);
}

// dx [N][C][H][W] from dy [N][C][Ho][Wo]: the same kernel on dy with the transposed, flipped weights and padding (K-1) dil - pad
int dilmfma_backward_data(const float* dy, const float* w, float* dx, int N, int C, int H, int W, int K, int pad, int dil,
                          void* ws, size_t ws_bytes, hipStream_t s) {
    const int Ho = H + 2 * pad - dil * (K - 1), Wo = W + 2 * pad - dil * (K - 1);
    const int padb = dil * (K - 1) - pad;
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, conv_flops(N, C, K, Ho, Wo), s);
    timing.issued(issued_flops(N, K, H, W));
    timing.bytes(4.0 * N * C * ((double)H * W + (double)Ho * Wo));
    if (K == 3) return run_conv<3, 1>(dy, w, nullptr, dx, N, C, Ho, Wo, padb, 1, ws, ws_bytes, s);
    if (K == 5) return run_conv<5, 2>(dy, w, nullptr, dx, N, C, Ho, Wo, padb, 1, ws, ws_bytes, s);
    return run_conv<7, 4>(dy, w, nullptr, dx, N, C, Ho, Wo, padb, 1, ws, ws_bytes, s);
}

int dilmfma_backward_weight(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int H, int W, int K,
                            int pad, int dil, void* ws, size_t ws_bytes, hipStream_t s) {
    const int Ho = H + 2 * pad - dil * (K - 1), Wo = W + 2 * pad - dil * (K - 1);
    afd::ScopedTiming timing(AFD_K_CONV_DIRECT, conv_flops(N, C, K, Ho, Wo), s);
    timing.issued(2.0 * N * 16.0 * 16.0 * K * K * (double)((Ho * Wo + 3) / 4 * 4));
    timing.bytes(4.0 * N * C * ((double)H * W + (double)Ho * Wo));
    if (K == 3) return run_wgrad<3, 1>(x, dy, dw, dbias, N, C, H, W, pad, ws, ws_bytes, s);
    if (K == 5) return run_wgrad<5, 2>(x, dy, dw, dbias, N, C, H, W, pad, ws, ws_bytes, s);
    return run_wgrad<7, 4>(x, dy, dw, dbias, N, C, H, W, pad, ws, ws_bytes, s);
}

}  // namespace afd
