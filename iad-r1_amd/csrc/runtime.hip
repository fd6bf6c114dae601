// Library-level plumbing: version, thread-local last error, launch checking.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void iadr1_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int iadr1_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        iadr1_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return IADR1_ERR_LAUNCH;
    }
    return IADR1_OK;
}

static thread_local SideOut g_side = {};
static thread_local hipStream_t g_side_stream = nullptr;

SideOut iadr1_take_side_out(hipStream_t stream) {
    if (!g_side.step || stream != g_side_stream) return SideOut{};
    const SideOut s = g_side;
    g_side = SideOut{};
    return s;
}

extern "C" int iadr1_decode_side_outputs(void* p0, long long ld0, void* p1, long long ld1, void* p2, long long ld2, const unsigned* step, long long base,
                                         long long seq_stride, hipStream_t stream) {
    IADR1_REQUIRE(step != nullptr && base >= 0 && seq_stride >= 0, "decode_side_outputs: a device step counter and non-negative row arithmetic are required");
    IADR1_REQUIRE((ld0 % 8) == 0 && (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0, "decode_side_outputs: 16-byte aligned rows");
    g_side = SideOut{p0, p1, p2, ld0, ld1, ld2, step, base, seq_stride};
    g_side_stream = stream;
    return IADR1_OK;
}

extern "C" const char* iadr1_last_error(void) { return g_err; }
extern "C" int iadr1_version(void) { return 100; }
