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

extern "C" const char* iadr1_last_error(void) { return g_err; }
extern "C" int iadr1_version(void) { return 100; }
