// Error reporting and version of libafd_hip.
#include "afd_common.h"

#include <vector>
#include "../../include/afd_hip.h"

namespace afd {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- optional per-kernel timing with HIP events on the launch stream ------------------
namespace {
struct Slot {
    hipEvent_t a, b;
    int id;
    double work, issued, bytes;
};
std::vector<Slot>& slots() {
    static std::vector<Slot> v;
    return v;
}
bool g_timing = false;
}  // namespace

bool timing_on() { return g_timing; }

void timing_begin(int id, double work, hipStream_t s) {
    Slot sl;
    sl.id = id;
    sl.work = work;
    sl.issued = 0.0;
    sl.bytes = 0.0;
    if (hipEventCreate(&sl.a) != hipSuccess || hipEventCreate(&sl.b) != hipSuccess) return;
    (void)hipEventRecord(sl.a, s);
    slots().push_back(sl);
}

void timing_annotate(double issued_flops, double algo_bytes) {
    if (slots().empty()) return;
    if (issued_flops >= 0.0) slots().back().issued = issued_flops;
    if (algo_bytes >= 0.0) slots().back().bytes = algo_bytes;
}

void timing_end(hipStream_t s) {
    if (!slots().empty()) (void)hipEventRecord(slots().back().b, s);
}

}  // namespace afd

extern "C" int afd_timing_enable(int on) {
    afd::g_timing = on != 0;
    return AFD_OK;
}

// Sums the recorded launches of kernel class `id` (AFD_K_*); synchronises on their events.
extern "C" int afd_timing_collect(int id, double* total_ms, long long* count, double* total_work,
                                  double* total_issued, double* total_bytes) {
    double ms = 0.0, work = 0.0, issued = 0.0, bytes = 0.0;
    long long n = 0;
    for (auto& sl : afd::slots()) {
        if (sl.id != id) continue;
        if (hipEventSynchronize(sl.b) != hipSuccess) return afd::fail(AFD_ERR_HIP, "timing: event sync failed");
        float t = 0.f;
        if (hipEventElapsedTime(&t, sl.a, sl.b) != hipSuccess) return afd::fail(AFD_ERR_HIP, "timing: elapsed failed");
        ms += t;
        work += sl.work;
        issued += sl.issued;
        bytes += sl.bytes;
        ++n;
    }
    if (total_ms) *total_ms = ms;
    if (count) *count = n;
    if (total_work) *total_work = work;
    if (total_issued) *total_issued = issued;
    if (total_bytes) *total_bytes = bytes;
    return AFD_OK;
}

extern "C" int afd_timing_reset(void) {
    for (auto& sl : afd::slots()) {
        (void)hipEventDestroy(sl.a);
        (void)hipEventDestroy(sl.b);
    }
    afd::slots().clear();
    return AFD_OK;
}

extern "C" const char* afd_last_error(void) { return afd::error_buffer(); }
extern "C" int afd_version(void) { return 1; }
