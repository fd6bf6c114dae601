// FP8 GEMM on the block-scaled matrix instruction of gfx950 (BASELINE.json config 5: "fp8 MFMA weights"; opt-in, frozen-reference forward only).
//
//   C[M,N] (bf16) = ( A8[M,K] . B8[N,K]^T ) * sa[m] * sb[n] + bias[n]        A8, B8: OCP e4m3, K-contiguous; sa, sb: fp32 row scales
//
// v_mfma_scale_f32_16x16x128_f8f6f4 is the only FP8 matrix instruction of CDNA4 that runs above the bf16 rate (the un-scaled 16x16x32 fp8 form runs AT
// the bf16 rate; /opt/skills/guides/MI355X_MICROARCH.md: 4.66 PFLOP/s measured for the scaled K = 128 form, 2x bf16).  It multiplies every 32-element
// K block by an E8M0 (power of two) scale taken from a VGPR; here those block scales are all 1 (exponent byte 127) and the dynamic range is handled by ONE fp32
// scale per row of each operand (amax / 448, the row-wise form: a token row of the activations, an output row of the weights), applied to the fp32
// accumulators in the epilogue -- finer than a per-tensor scale, and the quantisation step of an element never depends on another row.
//
// Kernel: the 128 x 128 block tile / 4 waves of 64 x 64 / double-buffered LDS-DMA structure of gemm_nt_128 (gemm.hip).  A K tile is 128 fp8 = 128 bytes
// per row -- the byte geometry of gemm_nt_128's 64-bf16 K tile, so the `global_load_lds` staging and the XOR swizzle on 16-byte chunks carry over unchanged;
// a lane's operand fragment is 32 bytes (8 VGPRs): two ds_read_b128.  One MFMA per (m-tile, n-tile) and K tile does the work of four bf16 16x16x32.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BKB = 128, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BKB;             // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;      // double buffered: 64 KiB
constexpr int C_LD = BN + 8;                     // bf16 epilogue row stride (elements)
constexpr int UNIT_SCALES = 0x7F7F7F7F;          // E8M0 exponent 127 = 2^0 in every byte of the scale operand

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

struct Fp8GemmArgs {
    const uint8_t* A;
    const uint8_t* B;
    const float* sa;
    const float* sb;
    bf16_t* C;
    const bf16_t* bias;
    const void* zeros;
    int M, N, K;
    long long lda, ldb, ldc;
};

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

// Operand layout of the 16x16x128 instruction (measured: tools/probe/fp8_mfma_layout.hip, profiles/r03_fp8_mfma_layout.txt): lane l holds row (l & 15) and
// the 32 CONSECUTIVE k values [32 (l >> 4), 32 (l >> 4) + 32) -- bytes 0..31 of its 8 operand VGPRs in k order.
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_fp8_128(Fp8GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l = t & 63;

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;                       // XCD-aware bijective block -> tile map, 8-tile-high bands (as gemm_nt_128)
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int band = wg / (8 * tiles_n), in_band = wg - band * 8 * tiles_n;
    const int band_rows = min(8, tiles_m - band * 8);
    const int tm = band * 8 + in_band % band_rows, tn = in_band / band_rows;
    const int m0 = tm * BM, n0 = tn * BN;

    const uint8_t* a_src[4];
    const uint8_t* b_src[4];
    int chunk_k[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = (i * 4 + w) * 64 + l;       // linear 16-byte slot in the LDS tile image
        const int row = L >> 3, c = L & 7;
        const int cs = c ^ ((row >> 1) & 7);      // source chunk that must land in slot (row, c)
        chunk_k[i] = cs * 16;
        a_src[i] = p.A + (long long)min(m0 + row, p.M - 1) * p.lda + cs * 16;
        b_src[i] = p.B + (long long)min(n0 + row, p.N - 1) * p.ldb + cs * 16;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
        const int k0 = kt * BKB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = k0 + chunk_k[i] < p.K;
            glds16(ok ? (const void*)(a_src[i] + k0) : p.zeros, base + (i * 4 + w) * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = k0 + chunk_k[i] < p.K;
            glds16(ok ? (const void*)(b_src[i] + k0) : p.zeros, base + TILE_BYTES + (i * 4 + w) * 1024);
        }
    };

    const int wm = w >> 1, wn = w & 1;
    int a_off[2], b_off[2];      // the two 16-byte halves of the lane's 32-byte fragment
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int chunk = (l >> 4) * 2 + h;
        const int ra = wm * 64 + (l & 15), rb = wn * 64 + (l & 15);
        a_off[h] = ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4);
        b_off[h] = TILE_BYTES + rb * 128 + ((chunk ^ ((rb >> 1) & 7)) << 4);
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BKB - 1) / BKB;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        __syncthreads();  // tile kt landed (vmcnt(0) + barrier); everyone is done reading buffer cur^1
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sbase = smem + cur * STAGE_BYTES;
        v8i_t af[4], bfr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4_t lo = *(const u32x4_t*)(sbase + a_off[0] + i * 2048), hi = *(const u32x4_t*)(sbase + a_off[1] + i * 2048);
            af[i] = (v8i_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4_t lo = *(const u32x4_t*)(sbase + b_off[0] + j * 2048), hi = *(const u32x4_t*)(sbase + b_off[1] + j * 2048);
            bfr[j] = (v8i_t){(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                // swapped operands (B tile as the first operand): D[row = n][col = m] -> a lane owns 4 consecutive n of one m; formats 0 / 0 = e4m3 x e4m3
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bfr[j], af[i], acc[i][j], 0, 0, 0, UNIT_SCALES, 0, UNIT_SCALES);
    }

    // ---- epilogue: row scales, bias, bf16 through LDS as 16-byte row-contiguous stores -----------------------------------------------------------
    const int lm = l & 15, lq = l >> 4;
    __syncthreads();
    bf16_t* cs = (bf16_t*)smem;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + lq * 4;
        float bv[4], sn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int gn = min(n0 + nl + e, p.N - 1);
            bv[e] = p.bias ? bf2f(p.bias[gn]) : 0.f;
            sn[e] = p.sb[gn];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = wm * 64 + i * 16 + lm;
            const float sm = p.sa[min(m0 + ml, p.M - 1)];
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * sm * sn[e] + bv[e];
            *(u32x2_t*)(cs + ml * C_LD + nl) = (u32x2_t){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        }
    }
    __syncthreads();
    const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)p.C) & 15) == 0);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * NTHREADS + t;
        const int row = idx >> 4, ch = idx & 15;
        const int gm = m0 + row, gn = n0 + ch * 8;
        if (gm >= p.M || gn >= p.N) continue;
        bf16_t* dst = p.C + (long long)gm * p.ldc + gn;
        if (vec_ok && gn + 8 <= p.N) {
            *(u32x4_t*)dst = *(const u32x4_t*)(cs + row * C_LD + ch * 8);
        } else {
            const bf16_t* sv = cs + row * C_LD + ch * 8;
            for (int e = 0; e < 8 && gn + e < p.N; ++e) dst[e] = sv[e];
        }
    }
}

// Row-wise dynamic quantisation: scale[m] = max_k |x[m][k]| / 448 (448 = largest finite e4m3), q[m][k] = round-to-nearest-even e4m3 of x / scale (true
// division: bit-equal to torch's float8_e4m3fn cast of x / scale).  One block per row; the second pass re-reads the row from L1 / L2.
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const bf16_t* X, long long ldx, uint8_t* Q, long long ldq, float* scale, int K) {
    __shared__ float scratch[16];
    const long long m = blockIdx.x;
    const bf16_t* x = X + m * ldx;
    float amax = 0.f;
    for (int c = threadIdx.x; c < (K >> 3); c += 256) {
        const u32x4_t v = *(const u32x4_t*)(x + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(lo_bf(v[e])), fabsf(hi_bf(v[e]))));
    }
    amax = block_max<256>(amax, scratch);
    const float sc = fmaxf(amax, 1e-30f) / 448.f;
    if (threadIdx.x == 0) scale[m] = sc;
    uint8_t* q = Q + m * ldq;
    for (int c = threadIdx.x; c < (K >> 3); c += 256) {
        const u32x4_t v = *(const u32x4_t*)(x + c * 8);
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[0]) / sc, hi_bf(v[0]) / sc, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[1]) / sc, hi_bf(v[1]) / sc, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2]) / sc, hi_bf(v[2]) / sc, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[3]) / sc, hi_bf(v[3]) / sc, hi, true);
        *(u32x2_t*)(q + c * 8) = (u32x2_t){(uint32_t)lo, (uint32_t)hi};
    }
}

__device__ char g_zero16_fp8[64] __attribute__((aligned(64)));

}  // namespace

extern "C" int iadr1_quant_rows_fp8(const void* X, long long ldx, void* Q, long long ldq, float* scale, int M, int K, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && K > 0 && (K % 8) == 0 && (ldx % 8) == 0 && (ldq % 8) == 0, "quant_rows_fp8: K, ldx, ldq must be multiples of 8 (M=%d K=%d)", M, K);
    IADR1_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Q) & 7) == 0 && scale != nullptr, "quant_rows_fp8: X 16-byte, Q 8-byte aligned, scale required");
    hipLaunchKernelGGL(quant_rows_fp8_kernel, dim3(M), dim3(256), 0, stream, (const bf16_t*)X, ldx, (uint8_t*)Q, ldq, scale, K);
    return iadr1_check_launch("quant_rows_fp8");
}

extern "C" int iadr1_gemm_nt_fp8(const void* A8, const float* sa, const void* B8, const float* sb, void* C, const void* bias, int M, int N, int K, long long lda,
                                 long long ldb, long long ldc, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && N > 0 && K > 0 && sa && sb && C, "gemm_nt_fp8: empty problem / missing scales");
    IADR1_REQUIRE((K % 16) == 0 && (lda % 16) == 0 && (ldb % 16) == 0, "gemm_nt_fp8: K, lda, ldb must be multiples of 16 (16-byte chunks of e4m3); K=%d lda=%lld ldb=%lld", K, lda, ldb);
    IADR1_REQUIRE((((uintptr_t)A8) & 15) == 0 && (((uintptr_t)B8) & 15) == 0, "gemm_nt_fp8: A8 / B8 must be 16-byte aligned");
    static const void* zeros = [] {
        void* z = nullptr;
        (void)hipGetSymbolAddress(&z, HIP_SYMBOL(g_zero16_fp8));
        (void)hipFuncSetAttribute((const void*)gemm_nt_fp8_128, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        return (const void*)z;
    }();
    Fp8GemmArgs p{(const uint8_t*)A8, (const uint8_t*)B8, sa, sb, (bf16_t*)C, (const bf16_t*)bias, zeros, M, N, K, lda, ldb, ldc};
    const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_nt_fp8_128, dim3(grid), dim3(NTHREADS), SMEM_BYTES, stream, p);
    return iadr1_check_launch("gemm_nt_fp8");
}
