// Probe (NOT product): what limits the decode weight stream?  Variants of the wide skinny loop on packed weights.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// mode 0: W loads only (nontemporal), XOR-reduce so they are not dead.  mode 1: + X loads.  mode 2: + MFMA.
template <int NB, int WAVES, int MODE, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void stream_k(const uint16_t* __restrict__ Wp, const uint16_t* __restrict__ X, float* out, int N, int K) {
    const int t = threadIdx.x, w = t >> 6, l = t & 63, lm = l & 15, lq = l >> 4;
    const int ksteps = K >> 5;
    const uint16_t* wbase = Wp + l * 8;
    long long woff[NB];
    for (int j = 0; j < NB; ++j) woff[j] = ((long long)(blockIdx.x * NB + j) * ksteps) * 512;
    const uint16_t* xrow[4];
    for (int i = 0; i < 4; ++i) xrow[i] = X + (long long)(i * 16 + lm) * K + lq * 8;
    f32x4_t acc[4][NB];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4_t){0, 0, 0, 0};
    u32x4_t x = {0, 0, 0, 0};
    const int nstep = ksteps;  // 32-k steps; wave w takes steps w, w+WAVES, ...
    for (int s0 = w; s0 < nstep; s0 += WAVES * DEPTH) {
        u32x4_t wf[DEPTH][NB], xf[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int s = s0 + d * WAVES;
            const bool ok = s < nstep;
#pragma unroll
            for (int j = 0; j < NB; ++j) wf[d][j] = ok ? __builtin_nontemporal_load((const u32x4_t*)(wbase + woff[j] + (long long)s * 512)) : (u32x4_t){0, 0, 0, 0};
            if (MODE >= 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) xf[d][i] = ok ? *(const u32x4_t*)(xrow[i] + s * 32) : (u32x4_t){0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[d][j]), __builtin_bit_cast(bf16x8_t, xf[d][i]), acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < NB; ++j) x ^= wf[d][j];
                if (MODE == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) x ^= xf[d][i];
                }
            }
        }
    }
    float r = 0;
    if (MODE == 2) { for (int i = 0; i < 4; ++i) for (int j = 0; j < NB; ++j) r += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3]; }
    else r = (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
    if (r == 12345.678f) out[blockIdx.x] = r;
}

extern "C" int run_stream(int variant, const void* Wp, const void* X, float* out, int N, int K, hipStream_t st) {
#define L(NB, WV, MODE, DEPTH) hipLaunchKernelGGL((stream_k<NB, WV, MODE, DEPTH>), dim3(N / (16 * NB)), dim3(WV * 64), 0, st, (const uint16_t*)Wp, (const uint16_t*)X, out, N, K)
    switch (variant) {
        case 0: L(8, 8, 0, 1); break;   case 1: L(8, 8, 0, 2); break;   case 2: L(8, 8, 0, 4); break;
        case 3: L(8, 8, 1, 2); break;   case 4: L(8, 8, 2, 2); break;   case 5: L(4, 8, 0, 4); break;
        case 6: L(4, 16, 0, 2); break;  case 7: L(4, 16, 2, 2); break;  case 8: L(2, 16, 0, 4); break;
        case 9: L(4, 8, 2, 4); break;   case 10: L(8, 4, 0, 4); break;  case 11: L(8, 16, 0, 2); break;
        default: return -1;
    }
    return 0;
}
