// Shared helpers for the libafd_hip kernels (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

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

// per-kernel HIP-event timing (afd_timing_enable); `work` = algorithmic flops or bytes
bool timing_on();
void timing_begin(int id, double work, hipStream_t s);
void timing_end(hipStream_t s);

struct ScopedTiming {
    hipStream_t s;
    bool on;
    ScopedTiming(int id, double work, hipStream_t st) : s(st), on(timing_on()) {
        if (on) timing_begin(id, work, s);
    }
    ~ScopedTiming() {
        if (on) timing_end(s);
    }
};

constexpr int kWave = 64;
constexpr int kLdsBytes = 160 * 1024;  // per-CU LDS on gfx950

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace afd
