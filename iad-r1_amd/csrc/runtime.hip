// Library-level plumbing: version, thread-local last error, launch checking.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

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

int iadr1_side_arg(const void* side, SideOut* out) {
    *out = SideOut{};
    if (!side) return IADR1_OK;
    const SideOut s = *(const SideOut*)side;
    IADR1_REQUIRE(s.step != nullptr && s.base >= 0 && s.seq_stride >= 0, "side outputs: a device step counter and non-negative row arithmetic are required");
    IADR1_REQUIRE((s.ld0 % 8) == 0 && (((uintptr_t)s.p0 | (uintptr_t)s.p1 | (uintptr_t)s.p2) & 15) == 0, "side outputs: 16-byte aligned rows");
    *out = s;
    return IADR1_OK;
}

int iadr1_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

extern "C" const char* iadr1_last_error(void) { return g_err; }
// 104: round 4 -- iadr1_decode_advance gained the all_done / rotary-table arguments, the FP8-MFMA pair (iadr1_quant_rows_fp8, iadr1_gemm_nt_fp8) is gone
extern "C" int iadr1_version(void) { return 104; }

