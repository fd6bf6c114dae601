// Common device/host helpers for the iadr1 gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits; all arithmetic is fp32

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;  // 16-byte load/store unit
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the gfx950 converter (v_cvt_pk_bf16_f32: one VALU instruction per PAIR; the integer add-and-shift form costs
// ~8 per element and made the attention kernels VALU-bound: 16.6 VALU instructions per MFMA in attn_fwd)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float lo_bf(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float hi_bf(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
    return v;
}

// block-wide reductions through a small LDS scratch (>= 16 floats); all threads get the result
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    constexpr int NW = NT / WAVE;
    if constexpr (NW == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    return r;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    constexpr int NW = NT / WAVE;
    if constexpr (NW == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// ---- optional in-kernel time stamps (probe builds only: -DIADR1_STAMPS; tools/kernel_stamps.py) ------------------------
// STAMP(slot): wave 0 of every block records the 100 MHz wall clock (s_memrealtime, comparable across CUs) of that program point.
#ifdef IADR1_STAMPS
static __device__ unsigned long long g_stamps[8][4096];      // one table per translation unit (no relocatable device code in this build)
#define STAMP(slot) do { if (threadIdx.x == 0) g_stamps[slot][(blockIdx.x + blockIdx.y * gridDim.x) & 4095] = wall_clock64(); } while (0)
// probe builds only (not part of the C ABI in include/iadr1_hip.h): copies this unit's stamp table to the host and clears it
#define IADR1_STAMPS_EXPORT(unit)                                                                                   \
    extern "C" int iadr1_debug_stamps_##unit(unsigned long long* host_dst) {                                        \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                        \
        if (hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -2;         \
        static unsigned long long zeros[8][4096];                                                                   \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -3;                \
    }
#else
#define STAMP(slot) do { } while (0)
#define IADR1_STAMPS_EXPORT(unit)
#endif

// ---- host side -------------------------------------------------------------------------------
#define IADR1_OK 0
#define IADR1_ERR_ARG (-1)
#define IADR1_ERR_LAUNCH (-2)
#define IADR1_ERR_UNSUPPORTED (-3)

void iadr1_set_error(const char* fmt, ...);
int iadr1_check_launch(const char* what);

#define IADR1_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            iadr1_set_error(__VA_ARGS__);     \
            return IADR1_ERR_ARG;             \
        }                                     \
    } while (0)

// Side outputs of the decode step (rollout -> training hand-over, include/iadr1_hip.h iadr1_side_out_t): the kernels of a decode step can ALSO
// write what they compute into the row-major activation arena the policy's backward pass reads, at row  base + s * seq_stride + *step  for
// sequence s.  Decode step t processes completion token t of every sequence, which is exactly row (s, t) of the completion block of the
// shared-prefix training batch, so the teacher-forced policy forward over the completions does not have to be run again.
struct SideOut {                   // same layout as iadr1_side_out_t
    void* p0; long long ld0;
    void* p1; long long ld1;
    void* p2; long long ld2;
    const unsigned* step;          // device-resident decode step counter; nullptr = side outputs off
    long long base, seq_stride;
    unsigned* mark;                // progress word of the decode step (iadr1_rmsnorm_fwd, few-row kernel): *mark = *step * mark_mul + mark_add at kernel entry; nullptr = none
    unsigned mark_mul, mark_add;
};
// the caller's struct (host memory, may be null) -> the by-value kernel argument; argument checks in runtime.hip
int iadr1_side_arg(const void* side, SideOut* out);
// IADR1_* A/B switches of the launchers: read ONCE (`static const int x = iadr1_env_int(...)`, thread-safe static initialisation), constant afterwards
int iadr1_env_int(const char* name, int dflt);
// CU count the decode-step launchers size persistent grids for (runtime.hip: the device's, or what iadr1_set_decode_cus / IADR1_DECODE_CUS says)
int iadr1_decode_cus(void);
// read the step counter ONCE, at kernel entry (a dependent global load in an epilogue is a memory latency on the critical path of a latency-bound kernel)
__device__ __forceinline__ long long side_base(const SideOut& so) { return so.step ? so.base + (long long)*so.step : -1; }

// Decode-packed activation layout ("leading dimension 0" in the C ABI): X[M,K] stored in MFMA B-fragment order so that the
// (16 rows x 32 k) fragment a wave feeds to v_mfma_f32_16x16x32_bf16 is ONE contiguous 1 KiB load, like the packed weights:
//   Xp[m/64][k/32][(m%64)/16][lane = m%16 + 16*((k%32)/8)][k%8];   rows are padded to a multiple of 64.
__device__ __forceinline__ long long xpk_off(int m, int k, int K) {
    return (long long)(m >> 6) * K * 64 + (long long)((k >> 5) * 4 + ((m & 63) >> 4)) * 512 + ((m & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7);
}

// K cache pages (paged KV of the rollout) are stored in the MFMA A-fragment order attn_decode reads them in: inside one
// (page, kv head) block of 32 keys x D = 128, element (key r, dim d) sits at kpk_off(r, d), so fragment (ks, t) = keys
// perm_row(li) + 4t, dims ks*32 + g*8.. is ONE contiguous 1 KiB wave load (a row-major page makes it 16 half cache lines).
__device__ __forceinline__ int kpk_off(int r, int d) {
    const int li = (r >> 3) * 4 + (r & 3), t = (r >> 2) & 1;
    return ((((d >> 5) * 2 + t) * 64) + ((d >> 3) & 3) * 16 + li) * 8 + (d & 7);
}

// ---- hardware transpose read (gfx950 ds_read_b64_tr_b16) ----------------------------------------------------------------
// Measured lane map (tools/gpu_probe.py, profiles/r01_tr_read_probe.txt): inside each 16-lane group, lane i receives element (i & 3) of the
// 8-byte chunk addressed by lane 4*j + (i >> 2), j = 0..3.  With lane L pointing at tile[k0 + (L >> 2)][m0 + 4*(L & 3)] of a ROW-MAJOR [k][m]
// LDS tile, lane i gets tile[k0 + 0..3][m0 + i]: four consecutive k of column m0+i, half of an MFMA A/B fragment of the TRANSPOSED tile.
// Through the compiler builtin, not inline asm: the two 8-byte results land in adjacent registers (the MFMA operand needs four consecutive
// VGPRs; asm outputs had to be copied: 64 v_mov per 64-key tile in attn_fwd), byte offsets fold into the instruction's immediate field, and the
// compiler counts lgkmcnt itself.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;
typedef __attribute__((address_space(3))) char lds_char_t;
__device__ __forceinline__ lds_char_t* lds_ptr(const void* p) { return (lds_char_t*)p; }
// fragment of the TRANSPOSED tile: k0..k0+3 from `lo`, k0+4..k0+7 from `hi` (both are this lane's tr-read addresses, see above)
__device__ __forceinline__ bf16x8_t tr_frag_ld(lds_char_t* lo, lds_char_t* hi) {
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)lo), b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)hi);
    return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
