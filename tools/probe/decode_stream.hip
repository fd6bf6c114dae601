// Probe (NOT product): what bounds the M=64 decode weight stream when the weights really come from HBM
// (the driver rotates through many distinct weight buffers so the 256 MB MALL cannot hold them)?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- 1. plain contiguous read, U loads in flight per lane
template <int U, bool NT>
__global__ __launch_bounds__(256) void pure_read(const u32x4_t* __restrict__ p, long long n16, uint32_t* out) {
    u32x4_t x = {0, 0, 0, 0};
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) x ^= v[u];
    }
    for (; i < n16; i += stride) x ^= p[i];
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

// ---- 2. the product's wide kernel geometry.  MODE 0: W only, 1: W + X loads, 2: + MFMA (no epilogue), 3: + LDS reduce epilogue
template <int NB, int WAVES, int DEPTH, int MODE>
__global__ __launch_bounds__(WAVES * 64) void wide_k(const uint16_t* __restrict__ Wp, const uint16_t* __restrict__ X, float* out, int N, int K) {
    extern __shared__ float red[];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, lm = l & 15, lq = l >> 4;
    const int ksteps = K >> 5;
    const long long tile_stride = (long long)ksteps * 512;
    const uint16_t* wbase = Wp + (long long)blockIdx.x * NB * tile_stride + l * 8;
    const uint16_t* xbase = X + (long long)lm * K + lq * 8;
    f32x4_t acc[4][NB];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
    u32x4_t x = {0, 0, 0, 0};
    for (int s0 = w; s0 + (DEPTH - 1) * WAVES < ksteps; s0 += WAVES * DEPTH) {
        u32x4_t wf[DEPTH][NB], xf[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long long s = s0 + d * WAVES;
#pragma unroll
            for (int j = 0; j < NB; ++j) wf[d][j] = __builtin_nontemporal_load((const u32x4_t*)(wbase + j * tile_stride + s * 512));
            if (MODE >= 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[d][i] = *(const u32x4_t*)(X + ((s * 4 + i) * 64 + l) * 8);
            } else if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[d][i] = *(const u32x4_t*)(xbase + (long long)i * 16 * K + s * 32);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (MODE == 2 || MODE == 3 || MODE >= 5) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[d][j]), __builtin_bit_cast(bf16x8_t, xf[d][i]), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) x ^= wf[d][j];
                if (MODE == 1 || MODE == 4) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) x ^= xf[d][i];
                }
            }
        }
    }
    if (MODE == 3 || MODE == 6) {
        constexpr int RLD = 33;
        float* mine = red + (size_t)w * 64 * RLD;
        for (int r = 0; r < NB / 2; ++r) {
            if (r) __syncthreads();
            for (int i = 0; i < 4; ++i) for (int jj = 0; jj < 2; ++jj) for (int e = 0; e < 4; ++e) mine[(i * 16 + lm) * RLD + jj * 16 + lq * 4 + e] = acc[i][2 * r + jj][e];
            __syncthreads();
            for (int idx = t; idx < 64 * 32; idx += WAVES * 64) {
                const int m = idx >> 5, n = idx & 31;
                float v = 0.f;
                for (int ww = 0; ww < WAVES; ++ww) v += red[((size_t)ww * 64 + m) * RLD + n];
                out[(long long)m * N + blockIdx.x * NB * 16 + r * 32 + n] = v;
            }
        }
        return;
    }
    uint32_t r;
    if (MODE == 2 || MODE == 5) { float f = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < NB; ++j) f += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; r = __float_as_uint(f); }
    else r = x[0] ^ x[1] ^ x[2] ^ x[3];
    if (r == 0x12345678u) out[blockIdx.x] = 1.f;
}

// ---- 3. X through LDS, every wave owns its own column tiles over the whole K (no cross-wave reduction):
// block = WAVES waves x TW tiles; X is staged chunk-wise (KC k per chunk, double buffered) with plain loads + ds_write.
template <int WAVES, int TW, int KC, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void ldsx_k(const uint16_t* __restrict__ Wp, const uint16_t* __restrict__ X, float* out, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // X chunk image: [KC/8 chunks of 8 k][64 rows][8] bf16 -> lane (row lm + 16 i, k-oct lq) reads 16 B at ((kc*4+lq)*64 + row)*16 B : conflict-free (consecutive rows)
    constexpr int XB = 64 * KC * 2;
    const int t = threadIdx.x, w = t >> 6, l = t & 63, lm = l & 15, lq = l >> 4;
    const int ksteps = K >> 5;
    const long long tile_stride = (long long)ksteps * 512;
    const int tile0 = (blockIdx.x * WAVES + w) * TW;
    const uint16_t* wbase = Wp + (long long)tile0 * tile_stride + l * 8;
    f32x4_t acc[4][TW];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < TW; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
    constexpr int SPC = KC / 32;             // k-steps per chunk
    const int nchunk = K / KC;
    // stage: 64 rows x KC/8 octs = 64*KC/8 16-B pieces per chunk, WAVES*64 threads
    auto stage = [&](int c, int buf) {
        for (int idx = t; idx < 64 * (KC / 8); idx += WAVES * 64) {
            const int row = idx & 63, oct = idx >> 6;
            const u32x4_t v = *(const u32x4_t*)(X + (long long)row * K + c * KC + oct * 8);
            *(u32x4_t*)(smem + buf * XB + (oct * 64 + row) * 16) = v;
        }
    };
    stage(0, 0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) stage(c + 1, buf ^ 1);
#pragma unroll
        for (int s0 = 0; s0 < SPC; s0 += DEPTH) {
            u32x4_t wf[DEPTH][TW];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int j = 0; j < TW; ++j) wf[d][j] = __builtin_nontemporal_load((const u32x4_t*)(wbase + j * tile_stride + (long long)(c * SPC + s0 + d) * 512));
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4_t xf = *(const u32x4_t*)(smem + buf * XB + (((s0 + d) * 4 + lq) * 64 + i * 16 + lm) * 16);
#pragma unroll
                    for (int j = 0; j < TW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[d][j]), __builtin_bit_cast(bf16x8_t, xf), acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // epilogue: lane holds Y[m = i*16 + lm][n = tile*16 + lq*4 + e]
    for (int j = 0; j < TW; ++j)
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 4; ++e) out[(long long)(i * 16 + lm) * N + (tile0 + j) * 16 + lq * 4 + e] = acc[i][j][e];
}

// ---- 4. persistent blocks, X register-resident: one block per CU keeps the X fragments of its waves' k-steps in VGPRs for the
// whole kernel and walks column-tile groups; the W loads of the next group are issued before the LDS reduction of the current one.
template <int WAVES, int TPI, int KSW>
__global__ __launch_bounds__(WAVES * 64) void pers_k(const uint16_t* __restrict__ Wp, const uint16_t* __restrict__ Xp, float* out, int N, int K) {
    extern __shared__ float red[];   // [WAVES][64][16*TPI + 1]
    constexpr int RLD = 16 * TPI + 1;
    const int t = threadIdx.x, w = t >> 6, l = t & 63, lm = l & 15, lq = l >> 4;
    const int ksteps = K >> 5;
    const long long tile_stride = (long long)ksteps * 512;
    u32x4_t xf[KSW][4];
#pragma unroll
    for (int j = 0; j < KSW; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[j][i] = *(const u32x4_t*)(Xp + (((long long)(w + j * WAVES) * 4 + i) * 64 + l) * 8);
    const int ngroups = N / (16 * TPI);
    u32x4_t wf[KSW][TPI];
    auto loadw = [&](int g) {
#pragma unroll
        for (int j = 0; j < KSW; ++j)
#pragma unroll
            for (int tt = 0; tt < TPI; ++tt)
                wf[j][tt] = __builtin_nontemporal_load((const u32x4_t*)(Wp + ((long long)g * TPI + tt) * tile_stride + (long long)(w + j * WAVES) * 512 + l * 8));
    };
    int g = blockIdx.x;
    if (g < ngroups) loadw(g);
    float* mine = red + (size_t)w * 64 * RLD;
    for (; g < ngroups; g += gridDim.x) {
        f32x4_t acc[4][TPI];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tt = 0; tt < TPI; ++tt) acc[i][tt] = (f32x4_t){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < KSW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int tt = 0; tt < TPI; ++tt)
                    acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[j][tt]), __builtin_bit_cast(bf16x8_t, xf[j][i]), acc[i][tt], 0, 0, 0);
        if (g + (int)gridDim.x < ngroups) loadw(g + gridDim.x);   // flies during the reduction below
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tt = 0; tt < TPI; ++tt)
#pragma unroll
                for (int e = 0; e < 4; ++e) mine[(i * 16 + lm) * RLD + tt * 16 + lq * 4 + e] = acc[i][tt][e];
        __syncthreads();
        for (int idx = t; idx < 64 * 16 * TPI; idx += WAVES * 64) {
            const int m = idx / (16 * TPI), n = idx - m * (16 * TPI);
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) v += red[((size_t)ww * 64 + m) * RLD + n];
            out[(long long)m * N + g * 16 * TPI + n] = v;
        }
        __syncthreads();
    }
}

extern "C" int run_pure(int variant, const void* p, long long bytes, uint32_t* out, hipStream_t st) {
    const long long n16 = bytes / 16;
    switch (variant) {
        case 0: hipLaunchKernelGGL((pure_read<4, true>), dim3(2048), dim3(256), 0, st, (const u32x4_t*)p, n16, out); break;
        case 1: hipLaunchKernelGGL((pure_read<8, true>), dim3(2048), dim3(256), 0, st, (const u32x4_t*)p, n16, out); break;
        case 2: hipLaunchKernelGGL((pure_read<8, false>), dim3(2048), dim3(256), 0, st, (const u32x4_t*)p, n16, out); break;
        case 3: hipLaunchKernelGGL((pure_read<8, true>), dim3(1024), dim3(256), 0, st, (const u32x4_t*)p, n16, out); break;
        case 4: hipLaunchKernelGGL((pure_read<16, true>), dim3(512), dim3(256), 0, st, (const u32x4_t*)p, n16, out); break;
        default: return -1;
    }
    return 0;
}

extern "C" int run_wide(int variant, const void* Wp, const void* X, float* out, int N, int K, hipStream_t st) {
#define LW(NB, WV, DEPTH, MODE) do { constexpr int SM = (MODE == 3 || MODE == 6) ? WV * 64 * 33 * 4 : 0; \
    if (SM > 48 * 1024) (void)hipFuncSetAttribute((const void*)wide_k<NB, WV, DEPTH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, SM); \
    hipLaunchKernelGGL((wide_k<NB, WV, DEPTH, MODE>), dim3(N / (16 * NB)), dim3(WV * 64), SM, st, (const uint16_t*)Wp, (const uint16_t*)X, out, N, K); } while (0)
#define LX(WV, TW, KC, DEPTH) do { constexpr int SM = 2 * 64 * KC * 2; \
    if (SM > 48 * 1024) (void)hipFuncSetAttribute((const void*)ldsx_k<WV, TW, KC, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, SM); \
    hipLaunchKernelGGL((ldsx_k<WV, TW, KC, DEPTH>), dim3(N / (16 * WV * TW)), dim3(WV * 64), SM, st, (const uint16_t*)Wp, (const uint16_t*)X, out, N, K); } while (0)
    switch (variant) {
        case 0: LW(8, 8, 2, 0); break;
        case 1: LW(8, 8, 2, 1); break;
        case 2: LW(8, 8, 2, 2); break;
        case 3: LW(8, 8, 2, 3); break;
        case 4: LW(4, 8, 2, 0); break;
        case 5: LW(4, 8, 2, 3); break;
        case 6: LW(4, 8, 4, 0); break;
        case 7: LW(4, 8, 4, 3); break;
        case 8: LW(2, 8, 4, 0); break;
        case 9: LW(2, 8, 4, 3); break;
        case 10: LW(4, 4, 4, 0); break;
        case 11: LW(4, 4, 4, 3); break;
        case 20: LW(8, 8, 2, 4); break;
        case 21: LW(8, 8, 2, 5); break;
        case 22: LW(8, 8, 2, 6); break;
        case 23: LW(4, 8, 2, 4); break;
        case 24: LW(4, 8, 2, 5); break;
        case 25: LW(4, 8, 2, 6); break;
        case 26: LW(4, 8, 1, 6); break;
        case 27: LW(4, 8, 4, 6); break;
        case 28: LW(2, 8, 2, 6); break;
        case 29: LW(2, 8, 4, 6); break;
        case 30: LW(4, 4, 2, 6); break;
        case 31: LW(4, 4, 4, 6); break;
        case 32: LW(8, 4, 2, 6); break;
        case 33: LW(8, 8, 1, 6); break;
#define LP(WV, TPI, KSW, GRID) do { constexpr int SM = WV * 64 * (16 * TPI + 1) * 4; \
    if (SM > 48 * 1024) (void)hipFuncSetAttribute((const void*)pers_k<WV, TPI, KSW>, hipFuncAttributeMaxDynamicSharedMemorySize, SM); \
    hipLaunchKernelGGL((pers_k<WV, TPI, KSW>), dim3(GRID), dim3(WV * 64), SM, st, (const uint16_t*)Wp, (const uint16_t*)X, out, N, K); } while (0)
        case 40: LP(8, 2, 8, 256); break;
        case 41: LP(8, 1, 8, 256); break;
        case 42: LP(8, 2, 8, 512); break;
        case 43: LP(16, 1, 4, 256); break;
        case 44: LP(16, 2, 4, 256); break;
        case 45: LP(8, 4, 8, 256); break;
        case 46: LP(8, 1, 8, 512); break;
        case 47: LP(16, 1, 4, 512); break;
        case 48: LP(4, 2, 16, 256); break;
        case 49: LP(4, 4, 16, 256); break;
        case 50: LP(4, 1, 16, 256); break;
        case 12: LX(4, 1, 256, 4); break;   // 64 cols / block
        case 13: LX(4, 1, 256, 8); break;
        case 14: LX(8, 1, 256, 8); break;   // 128 cols / block
        case 15: LX(4, 2, 256, 4); break;   // 128 cols / block, 2 tiles per wave
        case 16: LX(2, 2, 256, 4); break;   // 64 cols / block
        case 17: LX(4, 1, 512, 8); break;
        case 18: LX(2, 1, 256, 8); break;   // 32 cols / block
        default: return -1;
    }
    return 0;
}
