// Shared helpers for the libafd_hip kernels (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#define AFD_OK 0
#define AFD_ERR_ARG (-1)
#define AFD_ERR_HIP (-2)
#define AFD_ERR_UNSUPPORTED (-3)
#define AFD_ERR_WORKSPACE (-4)

namespace afd {

// thread-local last-error text, read through afd_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AFD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return AFD_OK;
}

// "done once on this device": a function attribute (hipFuncSetAttribute: dynamic LDS above 64 KB) belongs to the device
// the launch goes to, so a process that drives a second GPU has to set it there too.  One bit per device ordinal.
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask{0};
    static unsigned long long bit() {
        int d = 0;
        (void)hipGetDevice(&d);
        return 1ull << (d & 63);
    }
    bool done() const { return (mask.load(std::memory_order_acquire) & bit()) != 0; }
    void mark() { mask.fetch_or(bit(), std::memory_order_release); }
};

// per-kernel HIP-event timing (afd_timing_enable); `work` = algorithmic flops (direct form) or bytes
bool timing_on();
void timing_begin(int id, double work, hipStream_t s);
void timing_annotate(double issued_flops, double algo_bytes);  // negative = leave as is
void timing_end(hipStream_t s);

struct ScopedTiming {
    hipStream_t s;
    bool on;
    ScopedTiming(int id, double work, hipStream_t st) : s(st), on(timing_on()) {
        if (on) timing_begin(id, work, s);
    }
    // flops the launch really issues on the matrix cores (tile padding included; Winograd: 16 GEMMs)
    void issued(double flops) { if (on) timing_annotate(flops, -1.0); }
    // bytes an ideal implementation moves: every input and output tensor once
    void bytes(double b) { if (on) timing_annotate(-1.0, b); }
    ~ScopedTiming() {
        if (on) timing_end(s);
    }
};

// launches bound by HBM or by the vector ALU (no matrix work): work = bytes = every tensor of the launch once
struct ScopedBytes : ScopedTiming {
    ScopedBytes(int id, double b, hipStream_t st) : ScopedTiming(id, b, st) { bytes(b); }
};

// wpt4.hip: levels 9..14 of the level-14 packet transform in lattice form (1 = not its case);
// wpt4_available: the taps are an orthogonal bank with a usable lattice (cached per tap table)
bool wpt4_available(const float* dec_lo, const float* dec_hi, int L);
int wpt4_deep(const float* ws, float* out, int B, const float* dec_lo, const float* dec_hi, int L, unsigned flags,
              float power, float eps, float k1, float k0, float mean, float inv_std, float sgn_neg, float sgn_pos,
              hipStream_t stream);

// dilconv.hip: direct small-channel convolutions of the dilated stack (C = Cin = Cout <= 4)
bool dilconv_applicable(int Cin, int Cout, int K, int dil);
size_t dilconv_workspace_bytes(int C, int K);
int dilconv_forward(const float* x, const float* w, const float* bias, float* y, int N, int C, int H,
                    int W, int K, int pad, int dil, hipStream_t s);
int dilconv_backward_data(const float* dy, const float* w, float* dx, int N, int C, int H, int W,
                          int K, int pad, int dil, hipStream_t s);
int dilconv_backward_weight(const float* x, const float* dy, float* dw, float* dbias, int N, int C,
                            int H, int W, int K, int pad, int dil, void* ws, size_t ws_bytes,
                            hipStream_t s);

// dilconv_mfma.hip: the dilated stack at 5..16 channels on the f32 MFMA, image band resident in LDS
bool dilmfma_applicable(int Cin, int Cout, int H, int W, int K, int pad, int dil);
size_t dilmfma_workspace_bytes(int N, int C, int H, int W, int K, int pad, int dil);
int dilmfma_forward(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int K, int pad,
                    int dil, void* ws, size_t ws_bytes, hipStream_t s);
int dilmfma_backward_data(const float* dy, const float* w, float* dx, int N, int C, int H, int W, int K, int pad, int dil,
                          void* ws, size_t ws_bytes, hipStream_t s);
int dilmfma_backward_weight(const float* x, const float* dy, float* dw, float* dbias, int N, int C, int H, int W, int K,
                            int pad, int dil, void* ws, size_t ws_bytes, hipStream_t s);

// conv1x1.hip: 1x1 convolutions as GEMMs over flattened pixels
bool conv1x1_applicable(int Cin, int Cout, int K, int pad, int dil);
bool conv1x1_wgrad_applicable(int Cin, int Cout);
size_t conv1x1_workspace_bytes(int Cin, int Cout);
int conv1x1_forward(const float* x, const float* w, const float* bias, float* y, int N, int Cin,
                    int Cout, long HW, hipStream_t s);
size_t conv1x1_stats_workspace_bytes(int Cout);
int conv1x1_forward_stats(const float* x, const float* w, const float* bias, const float* slope, float* y,
                          double* sums, int N, int Cin, int Cout, long HW, void* ws, size_t ws_bytes,
                          hipStream_t s);
int conv1x1_backward_data(const float* dy, const float* w, float* dx, int N, int Cin, int Cout,
                          long HW, hipStream_t s);
int conv1x1_backward_data_affine(const float* dy, const float* w, const float* res, const float* alpha,
                                 const float* beta, float* dx, int N, int Cin, int Cout, long HW,
                                 hipStream_t s);
int conv1x1_backward_weight(const float* x, const float* dy, float* dw, float* dbias, int N, int Cin,
                            int Cout, long HW, void* ws, size_t ws_bytes, hipStream_t s);

// conv3x3.hip: 3x3 / dilation 1 / pad 1 on wide images, vector-ALU-free inner loop
bool conv3x3_applicable(int Cin, int H, int W, int Cout, int K, int pad, int dil);
size_t conv3x3_workspace_bytes(int Cin, int Cout);
int conv3x3_run(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H,
                int W, int Cout, int dgrad, int out_rows, int out_cols, void* ws, size_t ws_bytes,
                hipStream_t s);

// wino.hip: the same 3x3 / pad 1 layers as Winograd F(2x2, 3x3) on the MFMA (same contract as conv3x3_run)
bool wino_applicable(int Cin, int H, int W, int Cout);
size_t wino_workspace_bytes(int Cin, int Cout);
int wino_run(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
             int Cout, int dgrad, int out_rows, int out_cols, void* ws, size_t ws_bytes, hipStream_t s,
             const float* slope = nullptr, float* u = nullptr, unsigned char* idx = nullptr,
             const float* bn_in = nullptr, float* stat_part = nullptr);

// wino44.hip: Winograd F(4x4, 3x3), 16x16x4 tiles, register-only output transform
bool wino44_applicable(int Cin, int H, int W, int Cout);
size_t wino44_workspace_bytes(int Cin, int Cout);
int wino44_run(const float* x, const float* w, const float* bias, float* y, int N, int Cin, int H, int W,
               int Cout, int dgrad, int out_rows, int out_cols, void* ws, size_t ws_bytes, hipStream_t s,
               const float* bn_in = nullptr, float* stat_part = nullptr, const float* slope = nullptr,
               float* u = nullptr, unsigned char* idx = nullptr, int fwd_stats = 0,
               const unsigned char* pooled_codes = nullptr,  // backward-data from the pooled gradient: see G4::pidx
               const float* in_aff = nullptr, const float* in_slope = nullptr,  // input fold: see G4::in_aff
               const float* bn_tab = nullptr, const float* bn_slope = nullptr,  // BatchNorm backward epilogue: G4::bn_tab
               const unsigned char* bn_codes = nullptr);
long wino44_stat_rows(int N, int H, int W);
bool wino44_pool_applicable(int Cin, int H, int W, int Cout);
// wino44_wgrad.hip: 3x3 / pad 1 backward-weight in the Winograd F(4x4, 3x3) domain
bool wino44_wgrad_applicable(int Cin, int H, int W, int Cout, int K, int pad, int dil);
bool wino44_wgrad_crop_ok(int H, int W, int dy_rows, int dy_cols);
size_t wino44_wgrad_workspace_floats(int N, int Cin, int H, int W, int Cout, int dy_rows, int dy_cols);
// pooled_codes != nullptr: dy is the POOLED gradient [N][Cout][H/2][W/2] of the PReLU + 2x2 max-pool behind the
// convolution (afd_prelu_pool_backward_compact) and pooled_codes the pool's argmax codes; the crop is 2 (H/2) x 2 (W/2)
int wino44_wgrad_run(const float* x, const float* dy, float* dw, float* dbias, int N, int Cin, int H, int W, int Cout,
                     int dy_rows, int dy_cols, void* ws, size_t ws_bytes, hipStream_t s,
                     const unsigned char* pooled_codes = nullptr, const float* in_aff = nullptr,
                     const float* in_slope = nullptr);  // input fold: x is the input of the BatchNorm in front (GW::in_aff)

bool wgrad3x3_applicable(int Cin, int H, int W, int Cout, int K, int pad, int dil);
void wgrad3x3_geometry(int N, int Cin, int H, int W, int Cout, int dz_rows, int dz_cols, int* S,
                       int* nchunks, int* CI_T, int* CO_PAD, int* NCOL);
int wgrad3x3_launch(const float* x, const float* dz, float* part, float* partb, int N, int Cin, int H,
                    int W, int Cout, int dz_rows, int dz_cols, hipStream_t s);

// Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its own L2.  Kernels whose neighbouring
// work items share cache lines (row segments that do not start on 128-byte boundaries) want neighbours on ONE XCD, in
// flight together: in every group of 8 m consecutive ids XCD x takes the m consecutive logical ids x m .. x m + m - 1
// (the tail of the grid that does not fill a group keeps its ids).  m <= 1: identity.
__device__ __forceinline__ int xcd_grouped_id(int id, int total, int m) {
    if (m <= 1) return id;
    const int grp = 8 * m;
    const int base = id / grp * grp;
    if (base + grp > total) return id;
    const int p = id - base;
    return base + (p & 7) * m + (p >> 3);
}
inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

constexpr int kWave = 64;
constexpr int kLdsBytes = 160 * 1024;  // per-CU LDS on gfx950

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace afd
