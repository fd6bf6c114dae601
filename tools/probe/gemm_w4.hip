// Probe (NOT product): 256x256 tile NT GEMM with FOUR waves, each owning a 128x128 wave tile (512-register mode, one wave per SIMD),
// register-double-buffered fragments, LDS-DMA double buffer, one barrier per 64-deep K tile.  Question: does cutting the LDS fragment
// traffic by a third (vs 8 waves x 128x64) lift the main-loop rate of gemm_nt_256?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) { __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)lds_wave_base, 16, 0, 0); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    const bf2_t r = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, r);
}
constexpr int T = 256, BK = 64, OP_BYTES = T * BK * 2, STAGE_BYTES = 2 * OP_BYTES;

__global__ __launch_bounds__(256, 1) void gemm_w4(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, bf16_t* __restrict__ C, int M, int N, int K,
                                                 long long lda, long long ldb, long long ldc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63, lm = l & 15, lq = l >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int tiles_m = M / T;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int m0 = tm * T, n0 = tn * T;
    // DMA: wave w moves rows [w*64, w*64+64) of each operand tile, 8 instructions of 8 rows; lane -> row (l>>3), physical chunk (l&7);
    // source chunk = physical ^ ((row>>1)&7) = (l&7) ^ (((l>>4) + 4*(j&1)) & 7)
    const bf16_t* a_src[2];
    const bf16_t* b_src[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int cs = (l & 7) ^ (((l >> 4) + 4 * par) & 7);
        a_src[par] = A + (long long)(m0 + w * 64 + (l >> 3)) * lda + cs * 8;
        b_src[par] = B + (long long)(n0 + w * 64 + (l >> 3)) * ldb + cs * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            glds16(a_src[j & 1] + (long long)(j * 8) * lda + k0, base + (w * 8 + j) * 1024);
            glds16(b_src[j & 1] + (long long)(j * 8) * ldb + k0, base + OP_BYTES + (w * 8 + j) * 1024);
        }
    };
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int phys = (kk * 4 + lq) ^ ((lm >> 1) & 7);
        a_off[kk] = (wm * 128 + lm) * 128 + phys * 16;
        b_off[kk] = OP_BYTES + (wn * 128 + lm) * 128 + phys * 16;
    }
    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[2][8], fb[2][8];
    auto readf = [&](int buf, int kk, int slot) {
        const char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa[slot][i] = *(const bf16x8_t*)(base + a_off[kk] + i * 2048);
            fb[slot][i] = *(const bf16x8_t*)(base + b_off[kk] + i * 2048);
        }
    };
    auto mma = [&](int slot) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[slot][j], fa[slot][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    const int nk = K / BK;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    readf(0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        readf(cur, 1, 1);            // k-step 1 of this tile, consumed in the second half
        mma(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt+1 has landed (issued a whole tile ago)
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage(cur, kt + 2);
        if (kt + 1 < nk) readf(cur ^ 1, 0, 0);
        mma(1);
    }
    // epilogue: lane holds C[m = i*16 + lm][n = j*16 + lq*4 .. +4)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        bf16_t* row = C + (long long)(m0 + wm * 128 + i * 16 + lm) * ldc + n0 + wn * 128 + lq * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) *(u32x2_t*)(row + j * 16) = (u32x2_t){pack2bf(acc[i][j][0], acc[i][j][1]), pack2bf(acc[i][j][2], acc[i][j][3])};
    }
}

extern "C" int run_w4(const void* A, const void* B, void* C, int M, int N, int K, hipStream_t st) {
    if (M % T || N % T || K % BK) return -1;
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute((const void*)gemm_w4, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES); done = true; }
    hipLaunchKernelGGL(gemm_w4, dim3((M / T) * (N / T)), dim3(256), 2 * STAGE_BYTES, st, (const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, (long long)K, (long long)K, (long long)N);
    return 0;
}
