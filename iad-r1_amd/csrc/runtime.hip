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

// CUs the decode-step launchers size their persistent grids for: the device's CU count unless the rollout runs on a CU-masked stream
// (iadr1_stream_create_cu_mask) next to the teacher-forced training forward -- a persistent kernel launched with one block per CU of the WHOLE
// device onto a stream that owns fewer would run its blocks in two rounds.  THREAD-LOCAL (like iadr1_last_error): it is launcher configuration of the thread that
// captures / launches a decode step, set right before and reset (0) right after -- the ABI keeps no process-wide mutable state.  It cannot
// be read off the launch stream: the decode graph is CAPTURED on torch's capture stream and only REPLAYED on the masked one.
static thread_local int g_decode_cus = 0;
int iadr1_decode_cus(void) {
    if (g_decode_cus > 0) return g_decode_cus;
    static const int dev_cus = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        const int env = iadr1_env_int("IADR1_DECODE_CUS", 0);
        if (env > 0) return env;
        return (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }();
    return dev_cus;
}
extern "C" int iadr1_set_decode_cus(int n_cus) {
    IADR1_REQUIRE(n_cus >= 0 && n_cus <= 1024, "set_decode_cus: 0 (device default) .. 1024, got %d", n_cus);
    g_decode_cus = n_cus;
    return IADR1_OK;
}

extern "C" int iadr1_stream_create_cu_mask(const unsigned* cu_mask, int n_words, void** stream_out) {
    IADR1_REQUIRE(cu_mask != nullptr && n_words > 0 && stream_out != nullptr, "stream_create_cu_mask: a mask of >= 1 words and an output slot are required");
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, (const uint32_t*)cu_mask);
    if (e != hipSuccess) {
        iadr1_set_error("stream_create_cu_mask: %s", hipGetErrorString(e));
        return IADR1_ERR_LAUNCH;
    }
    *stream_out = (void*)s;
    return IADR1_OK;
}
extern "C" int iadr1_stream_destroy(void* stream) {
    IADR1_REQUIRE(stream != nullptr, "stream_destroy: null stream");
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) {
        iadr1_set_error("stream_destroy: %s", hipGetErrorString(e));
        return IADR1_ERR_LAUNCH;
    }
    return IADR1_OK;
}

extern "C" const char* iadr1_last_error(void) { return g_err; }
// 109: round 6 -- iadr1_weight_prefetch (persistent memory-side-cache prefetcher of the rollout, paced by iadr1_side_out_t.mark: three new fields at the END of the struct)
// 108: round 5 -- iadr1_gemm_tn_acc_bf16 (weight gradients from row-major dY / X: the 256 x 256 kernel with transpose reads out of LDS, no transposed copies)
// 107: round 5 -- iadr1_gemm_swiglu_rows_bf16 (row-blocked fused gate|up + SwiGLU: the policy's mlp rows of a chunk of decode steps, rebuilt on the side stream)
// 106: round 5 -- CU-masked streams, decode CU count, iadr1_attn_fwd_chunk, iadr1_wait_counter (co-scheduled rollout / teacher-forced forward); ordered two-stage
//      gradient reductions: workspace arguments on iadr1_rmsnorm_bwd / iadr1_layernorm_bwd / iadr1_colsum_acc, iadr1_rows_scatter_acc replaces iadr1_embed_bwd
// 104: round 4 -- iadr1_decode_advance gained the all_done / rotary-table arguments, the FP8-MFMA pair (iadr1_quant_rows_fp8, iadr1_gemm_nt_fp8) is gone
extern "C" int iadr1_version(void) { return 109; }

