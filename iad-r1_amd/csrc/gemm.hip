// iadr1 GEMM family for gfx950 (CDNA4).
//
//   C[M,N] (+)= act( A[M,K] . B[N,K]^T + bias[N] )       "NT": both operands K-contiguous
//
// This is the contraction behind every Linear of the hot path (SURVEY.md section 2.3 K1,K6,K7,K8,K12,
// K14,K15; reference call sites TF:modeling_qwen2_5_vl.py:85-96,116-122,148-150,218-219,634-637,
// 552-554,1386-1387).  Forward uses W[N,K] directly; dgrad / wgrad reuse the same kernel on
// transposed copies produced by transpose.hip.
//
// Kernel gemm_nt_128: 128x128x64 block tile, 4 waves (2x2) of 64x64, v_mfma_f32_16x16x32_bf16,
// operands staged HBM->LDS with global_load_lds_dwordx4 (no VGPR round trip), 2 LDS buffers so the
// DMA of tile t+1 overlaps the MFMAs of tile t, XOR-swizzled LDS image (swizzle applied on the
// per-lane SOURCE address and on the ds_read address; the LDS destination of the DMA stays
// lane-linear as the hardware requires), XCD-aware block->tile map (8 XCDs, private L2s) with
// 8-tile-high bands so the 64 blocks resident on one XCD share A/B panels in its L2.
// MFMA operands are swapped (B-tile as the A operand) so each lane ends up with 4 consecutive
// output columns of one row: bf16 results are staged through LDS and leave as 16-byte row-contiguous
// stores; fp32 results (logit chunks, wgrad accumulation) leave as 16-byte stores/RMW per lane.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + B
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;      // double buffered: 64 KiB
constexpr int C_LD = BN + 8;                     // bf16 epilogue row stride (elements)

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    const bf16_t* bias;
    const void* zeros;  // >=16 zero bytes, source for out-of-range K chunks
    int M, N, K;
    long long lda, ldb, ldc;
    int act;  // 0 none, 1 exact GELU
    int band;  // tile rows per rasterisation band (gemm_nt_256)
    bf16_t* C2;      // OUT_SWIGLU: a[M, N/2] = silu(gate) * up (C = the gate|up matrix itself, or null when only `a` is wanted)
    long long ldc2;
    // OUT_LSE / OUT_DLOGITS (linear_logprob: the lm_head contraction whose logits never reach HBM)
    const long long* targets;   // [M] column of the row's target (< 0: ignored row)
    float* part;                // OUT_LSE: [M][nparts] pairs (max, sum of exp(x - max)) over each wave's 64-column slice
    float* tgt_logit;           // OUT_LSE: [M] the logit at the target column (rows with a target in range)
    const float* lse;           // OUT_DLOGITS: [M] log-sum-exp of the row
    const float* g;             // OUT_DLOGITS: [M] dLoss/dlogp of the row
    int nparts;
    // split-K (gemm_nt_256<OUT_F32>, iadr1_gemm_nt_splitk_acc_bf16): grid = ksplit x tiles; slice z = blockIdx.x / tiles contracts columns
    // [z * kslice, min(K, (z + 1) * kslice)) of A and B and stores its fp32 partial tile at C + z * zstride (a workspace the reduce kernel sums)
    int ksplit, kslice;
    long long zstride;
    // OUT_SWIGLU_ROWS (iadr1_gemm_swiglu_rows_bf16): GEMM row r is row (r >> rb_shift) * rb_stride + (r & (2^rb_shift - 1)) of A, C and C2 -- blocks of 2^rb_shift
    // consecutive rows rb_stride rows apart (the rows a chunk of decode steps produced in a sequence-major arena)
    int rb_shift;
    long long rb_stride;
};

enum { OUT_BF16 = 0, OUT_F32 = 1, OUT_F32_ACC = 2, OUT_SWIGLU = 3, OUT_LSE = 4, OUT_DLOGITS = 5, OUT_SWIGLU_ROWS = 6 };

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
    // LDS destination = wave-uniform base + lane*16 (hardware rule); source address is per lane.
    __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

template <int OUT>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_nt_128(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l = t & 63;

    // ---- block -> tile map (XCD-aware, bijective for any grid) --------------------------------
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int band = wg / (8 * tiles_n), in_band = wg - band * 8 * tiles_n;
    const int band_rows = min(8, tiles_m - band * 8);
    const int tm = band * 8 + in_band % band_rows, tn = in_band / band_rows;
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread DMA sources: 4 A chunks + 4 B chunks (16 B each) per K tile --------------------
    const bf16_t* a_src[4];
    const bf16_t* b_src[4];
    int chunk_k[4];  // element offset of the (swizzled) source chunk inside the K tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int L = (i * 4 + w) * 64 + l;  // linear 16-B slot in the LDS tile image
        const int row = L >> 3, c = L & 7;
        const int cs = c ^ ((row >> 1) & 7);  // source chunk that must land in slot (row, c)
        chunk_k[i] = cs * 8;
        a_src[i] = p.A + (long long)min(m0 + row, p.M - 1) * p.lda + cs * 8;
        b_src[i] = p.B + (long long)min(n0 + row, p.N - 1) * p.ldb + cs * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES;
        const int k0 = kt * BK;
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

    // ---- fragment read offsets (swizzled), wave tile 64x64 -------------------------------------------
    const int wm = w >> 1, wn = w & 1;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = kk * 4 + (l >> 4);
        const int ra = wm * 64 + (l & 15), rb = wn * 64 + (l & 15);
        a_off[kk] = ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4);
        b_off[kk] = TILE_BYTES + rb * 128 + ((chunk ^ ((rb >> 1) & 7)) << 4);
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        __syncthreads();  // tile kt landed (vmcnt(0) + barrier); everyone is done reading buffer cur^1
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sbase = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8_t*)(sbase + a_off[kk] + i * 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8_t*)(sbase + b_off[kk] + j * 2048);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    // swapped operands: D[row = n][col = m] -> lane owns 4 consecutive n of one m
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------------
    const int lm = l & 15, lq = l >> 4;
    if constexpr (OUT == OUT_BF16) {
        __syncthreads();  // all waves finished reading the operand buffers
        bf16_t* cs = (bf16_t*)smem;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = wn * 64 + j * 16 + lq * 4;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (n0 + nl + e < p.N) ? bf2f(p.bias[n0 + nl + e]) : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ml = wm * 64 + i * 16 + lm;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][e] + bv[e];
                    if (p.act == 1) v[e] = gelu_erf(v[e]);
                }
                u32x2_t pk = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                *(u32x2_t*)(cs + ml * C_LD + nl) = pk;
            }
        }
        __syncthreads();
        bf16_t* C = (bf16_t*)p.C;
        const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0);
        if (vec_ok && m0 + BM <= p.M && n0 + BN <= p.N) {   // interior tile: 8 slab reads, then 8 unpredicated 16-byte stores
            u32x4_t rv[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) rv[it] = *(const u32x4_t*)(cs + ((it * NTHREADS + t) >> 4) * C_LD + (t & 15) * 8);
#pragma unroll
            for (int it = 0; it < 8; ++it) *(u32x4_t*)(C + (long long)(m0 + ((it * NTHREADS + t) >> 4)) * p.ldc + n0 + (t & 15) * 8) = rv[it];
            return;
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = it * NTHREADS + t;
            const int row = idx >> 4, ch = idx & 15;
            const int gm = m0 + row, gn = n0 + ch * 8;
            if (gm >= p.M || gn >= p.N) continue;
            const u32x4_t v = *(const u32x4_t*)(cs + row * C_LD + ch * 8);
            bf16_t* dst = C + (long long)gm * p.ldc + gn;
            if (vec_ok && gn + 8 <= p.N) {
                *(u32x4_t*)dst = v;
            } else {
                const bf16_t* sv = cs + row * C_LD + ch * 8;  // scalar tail straight from LDS
                for (int e = 0; e < 8 && gn + e < p.N; ++e) dst[e] = sv[e];
            }
        }
    } else {
        float* C = (float*)p.C;
        const bool vec_ok = ((p.ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0);
        if (vec_ok && !p.bias && p.act == 0 && m0 + BM <= p.M && n0 + BN <= p.N) {   // interior tile, plain store / accumulate (see gemm_nt_256)
            float* base = C + (long long)(m0 + wm * 64 + lm) * p.ldc + n0 + wn * 64 + lq * 4;
#pragma unroll
            for (int i0 = 0; i0 < 4; i0 += 2) {
                f32x4_t old[2][4];
                if constexpr (OUT == OUT_F32_ACC) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) old[u][j] = *(const f32x4_t*)(base + (long long)(i0 + u) * 16 * p.ldc + j * 16);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4_t o = acc[i0 + u][j];
                        if constexpr (OUT == OUT_F32_ACC) o += old[u][j];
                        *(f32x4_t*)(base + (long long)(i0 + u) * 16 * p.ldc + j * 16) = o;
                    }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + wm * 64 + i * 16 + lm;
            if (gm >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gn = n0 + wn * 64 + j * 16 + lq * 4;
                if (gn >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][e];
                    if (p.bias && gn + e < p.N) v[e] += bf2f(p.bias[gn + e]);
                    if (p.act == 1) v[e] = gelu_erf(v[e]);
                }
                float* dst = C + (long long)gm * p.ldc + gn;
                if (vec_ok && gn + 4 <= p.N) {
                    f32x4_t o = {v[0], v[1], v[2], v[3]};
                    if constexpr (OUT == OUT_F32_ACC) {
                        const f32x4_t old = *(const f32x4_t*)dst;
                        o += old;
                    }
                    *(f32x4_t*)dst = o;
                } else {
                    for (int e = 0; e < 4 && gn + e < p.N; ++e) {
                        if constexpr (OUT == OUT_F32_ACC) dst[e] += v[e];
                        else dst[e] = v[e];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// gemm_nt_256: 256x256x64 block tile, 8 waves (2 x 4) of 128x64, deep software pipeline ("8 phases per two K
// tiles"): every K tile is consumed in 4 phases of 16 MFMAs (one 64x32 quadrant of the wave tile x K=64);
// each phase ds_reads only the operand half it newly needs (A half: 8 x b128, B half: 4 x b128), issues the
// LDS-DMA of ONE half-tile of the NEXT K tile into the other LDS buffer, and waits with a COUNTED vmcnt so two
// half-tiles stay in flight across the s_barrier -- the DMA queue is never drained inside the main loop.
// LDS: 2 buffers x (A 32 KiB + B 32 KiB) = 128 KiB, XOR-swizzled like gemm_nt_128 (source address + read side).
// Half-tiles are cut so that a phase needs the same half for every wave: A half h = rows {wm*128 + h*64 + 0..63},
// B half h = cols {wn*64 + h*32 + 0..31}.  Epilogue: each wave stages its 128x64 bf16 result in its own LDS
// slab and writes full 128-byte row segments.
// ------------------------------------------------------------------------------------------------------
constexpr int T2 = 256, NT2 = 512, HALF_BYTES = 128 * 64 * 2;          // 16 KiB per half tile
constexpr int BUF2_BYTES = 4 * HALF_BYTES;                             // A0 A1 B0 B1
constexpr int SMEM2_MAIN = 2 * BUF2_BYTES;                             // 128 KiB
constexpr int EP_LD = 64 + 8;                                          // epilogue slab row stride (elements)
constexpr int SMEM2_EPI = 8 * 128 * EP_LD * 2;                         // 147456 B
constexpr int SMEM2_BYTES = SMEM2_EPI > SMEM2_MAIN ? SMEM2_EPI : SMEM2_MAIN;

#define IADR1_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// TN form (template flag, OUT_F32 / OUT_F32_ACC only: the weight gradients  dW[M, N] (+)= A[K, M]^T . B[K, N],  A = dY and B = X as they lie in memory, the
// contraction running over their ROWS): the same pipeline and epilogue; a half-tile image is [64 contraction rows][128 columns] with 256-byte rows (chunk c of
// row r at chunk position c ^ 2 * tn_rho(r), the swizzle of the attention tiles), filled by the same 2 x 16-byte DMA per thread from row-major sources, and the
// MFMA fragments -- 8 consecutive contraction elements of one column per lane -- come out of LDS through the hardware transpose read (ds_read_b64_tr_b16, two
// per fragment: common.h tr_frag_ld).  No transposed copies of dY / X exist anywhere.
__device__ __forceinline__ int tn_rho(int r) { return (r & 3) | ((r >> 1) & 4); }
// One MFMA fragment of the TN form: contraction rows 32 kk + {0..3} and {4..7} (+ this lane's row) of the lane's column, two transpose reads at byte offsets
// kk * 8192 and + 1024.  Inline asm, not the builtin: the compiler orders every LDS-reading INTRINSIC behind all outstanding global_load_lds (it put s_waitcnt
// vmcnt(0) in front of each group of reads -- the DMA queue drained four times per K tile: 872 instead of 1230 TF/s); the kernel's own counted vmcnt / lgkmcnt
// waits and barriers are what orders these reads, exactly as for the plain ds_read_b128 of the NT form.
__device__ __forceinline__ bf16x8_t tr_frag_asm(const char* lds_addr, int kk) {
    s16x4_t a, b;
    const unsigned addr = (unsigned)(uintptr_t)lds_ptr(lds_addr);
    if (kk == 0) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a) : "v"(addr));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(b) : "v"(addr));
    } else {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(a) : "v"(addr));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(b) : "v"(addr));
    }
    return __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// M32 (round 6 experiment, probe builds with -DIADR1_PROBE_MFMA32 only; plain NT form, OUT_BF16 / OUT_F32 / OUT_F32_ACC): the same pipeline on v_mfma_f32_32x32x16_bf16 -- the wave's 128 x 64
// tile as 4 x 2 accumulators of 32 x 32, four 16-deep k-steps per K tile.  The LDS traffic is the wave TILE's and does not change (24 ds_read_b128 per K tile either
// way: a register-blocked wave reads its A and B sub-tiles once whatever the instruction shape); what changes is the MFMA issue count (32 instead of 64 per K tile)
// and the operand-register reads per FLOP.  The swizzled half images serve the 32-row fragments conflict-free as they are (rows r .. r + 31 of one 16-byte chunk
// column: the 16 lanes of every ds_read_b128 service group land in 16 different bank groups).  Results differ from the 16x16x32 form in fp32 summation order.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int OUT, bool TN = false, bool M32 = false>
__global__ __launch_bounds__(NT2, 2) void gemm_nt_256(GemmArgs p) {
    static_assert(!TN || OUT == OUT_F32 || OUT == OUT_F32_ACC, "the TN form exists for the fp32 (accumulate) outputs only");
    static_assert(!M32 || (!TN && (OUT == OUT_BF16 || OUT == OUT_F32 || OUT == OUT_F32_ACC)), "the 32x32x16 form exists for the plain NT outputs only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l = t & 63;
    const int wm = w >> 2, wn = w & 3;

    const int tiles_m = (p.M + T2 - 1) / T2, tiles_n = (p.N + T2 - 1) / T2;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    if constexpr (OUT == OUT_F32) {
        if (p.ksplit > 1) {      // block-uniform: this block's K slice and its slab of the partial-sum workspace
            const int z = bid / nwg;
            bid -= z * nwg;
            p.A += (long long)z * p.kslice * (TN ? p.lda : 1);
            p.B += (long long)z * p.kslice * (TN ? p.ldb : 1);
            p.K = min(p.kslice, p.K - z * p.kslice);
            p.C = (float*)p.C + (long long)z * p.zstride;
        }
    }
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int BR = p.band;
    const int band = wg / (BR * tiles_n), in_band = wg - band * BR * tiles_n;
    const int band_rows = min(BR, tiles_m - band * BR);
    const int tm = band * BR + in_band % band_rows, tn = in_band / band_rows;
    const int m0 = tm * T2, n0 = tn * T2;

    // ---- DMA sources: per half-tile 2 x 16 B per thread ----------------------------------------------------
    const bf16_t* src[4][2];  // [A0, A1, B0, B1][instr]
    int chunk_k[2];           // NT: element offset of the slot's chunk inside the K tile; TN: the slot's row inside the K tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int L = (i * 8 + w) * 64 + l;   // 16-B slot inside the half-tile image
        if constexpr (TN) {
            const int rl = L >> 4, cpos = L & 15;                  // row of the 64-row K tile, chunk POSITION inside its 256-byte row
            const int cl = (cpos ^ (2 * tn_rho(rl))) * 8;          // first of the 8 half-image columns this slot holds
            chunk_k[i] = rl;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int acol = m0 + (cl >> 6) * 128 + h * 64 + (cl & 63);     // A half h: output rows wm*128 + h*64 + ..  (columns of A)
                const int bcol = n0 + h * 128 + cl;      // B half h: 128 CONSECUTIVE columns of B (whole 128-byte lines per DMA row piece; wave wn owns columns
                                                         // h*128 + wn*32 + .. of the tile, see `coln` in the epilogue -- with the NT form's wn*64 + h*32 + .. a
                                                         // row piece would be 64 bytes and every line of B would be requested twice)
                src[h][i] = p.A + (long long)rl * p.lda + min(acol, p.M - 8);
                src[2 + h][i] = p.B + (long long)rl * p.ldb + min(bcol, p.N - 8);
            }
        } else {
        const int rl = L >> 3, c = L & 7;
        const int cs = c ^ ((rl >> 1) & 7);
        chunk_k[i] = cs * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int arow = (rl >> 6) * 128 + h * 64 + (rl & 63);          // A half h: rows wm*128 + h*64 + ..
            const int bcol = (rl >> 5) * 64 + h * 32 + (rl & 31);           // B half h: cols wn*64 + h*32 + ..
            if constexpr (OUT == OUT_SWIGLU_ROWS) {
                const int ar = m0 + arow;       // (M % 256 == 0: no clamp)
                src[h][i] = p.A + ((long long)(ar >> p.rb_shift) * p.rb_stride + (ar & ((1 << p.rb_shift) - 1))) * p.lda + cs * 8;
            } else {
                src[h][i] = p.A + (long long)min(m0 + arow, p.M - 1) * p.lda + cs * 8;
            }
            if constexpr (OUT == OUT_SWIGLU || OUT == OUT_SWIGLU_ROWS) {
                // fused SwiGLU: B = [gate rows 0..I) | up rows I..2I).  The block's 256 columns are 128 gate + 128 up columns of the SAME 128 outputs,
                // arranged so that B half 0 of every wave is gate and half 1 is up: a lane then holds gate (n-tiles 0,1) and up (n-tiles 2,3) of the
                // same (row, column) and the activation is computed in registers.
                src[2 + h][i] = p.B + (long long)(h * (p.N >> 1) + (n0 >> 1) + (rl >> 5) * 32 + (rl & 31)) * p.ldb + cs * 8;
            } else {
                src[2 + h][i] = p.B + (long long)min(n0 + bcol, p.N - 1) * p.ldb + cs * 8;
            }
        }
        }
    }
    auto stage_half = [&](int buf, int half /*0..3 = A0 A1 B0 B1*/, int kt) {
        char* base = smem + buf * BUF2_BYTES + half * HALF_BYTES;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = k0 + chunk_k[i] < p.K;
            if constexpr (TN) glds16(ok ? (const void*)(src[half][i] + (long long)k0 * (half < 2 ? p.lda : p.ldb)) : p.zeros, base + (i * 8 + w) * 1024);
            else glds16(ok ? (const void*)(src[half][i] + k0) : p.zeros, base + (i * 8 + w) * 1024);
        }
    };

    // ---- fragment read offsets inside a half-tile image -------------------------------------------------------
    int a_off[2], b_off[2];  // per k-step; add mi*2048 / ni*2048
    int ta_off[4], tb_off[2];  // TN: per m-tile / n-tile; add kk*8192 (+1024 for the second four contraction rows)
    if constexpr (TN) {
        // lane (lg = l & 15, g = l >> 4) addresses row 8g + (lg >> 2) [+ 32 kk, + 4] and the 8-byte half (lg & 1) of chunk  2 * tile16 + ((lg & 3) >> 1)  of the
        // half image; tn_rho of that row is ((lg >> 2) & 3) | ((g & 1) << 2) for every kk and both halves of a fragment, so the swizzle is a per-lane constant
        const int lg = l & 15, g = l >> 4;
        const int x = 2 * (((lg >> 2) & 3) | ((g & 1) << 2));
        const int rowpart = (8 * g + (lg >> 2)) * 256 + (lg & 1) * 8, cb = (lg & 3) >> 1;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) ta_off[mi] = rowpart + (((wm * 8 + mi * 2 + cb) ^ x) << 4);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) tb_off[ni] = rowpart + (((wn * 4 + ni * 2 + cb) ^ x) << 4);
    } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int chunk = kk * 4 + (l >> 4);
        const int ra = wm * 64 + (l & 15), rb = wn * 32 + (l & 15);
        a_off[kk] = ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4);
        b_off[kk] = rb * 128 + ((chunk ^ ((rb >> 1) & 7)) << 4);
    }
    }

    int a_off32[4], b_off32[4];  // M32: per 16-deep k-step; add mt*4096 (32 rows)
    if constexpr (M32) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int chunk = ks * 2 + (l >> 5);
            const int ra = wm * 64 + (l & 31), rb = wn * 32 + (l & 31);
            a_off32[ks] = ra * 128 + ((chunk ^ ((ra >> 1) & 7)) << 4);
            b_off32[ks] = rb * 128 + ((chunk ^ ((rb >> 1) & 7)) << 4);
        }
    }

    f32x4_t acc[8][4];  // [m-tile][n-tile] of the 128x64 wave tile
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    f32x16_t acc32[4][2];  // M32: [32-row m-tile][32-column n-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
    bf16x8_t af32[2][4], bf32[2][4];   // M32: current A half (2 m-tiles x 4 k-steps), BOTH B halves (n-tile x 4 k-steps)

    bf16x8_t af[4][2], bfr[2][2];  // current A half (4 m-tiles x 2 k-steps), current B half (2 n-tiles x 2 k-steps)
    bf16x8_t bfr1[TN ? 2 : 1][2];  // TN: B half 1 in registers of its own, so that B half 0 survives phases 2 - 3 and phase 4 reads nothing (transpose reads are the
                                   // form's bottleneck: 48 instead of 56 per K tile)
    auto read_a = [&](int buf, int h) {
        const char* base = smem + buf * BUF2_BYTES + h * HALF_BYTES;
        if constexpr (M32) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) af32[mt][ks] = *(const bf16x8_t*)(base + a_off32[ks] + mt * 4096);
            return;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if constexpr (TN) af[mi][kk] = tr_frag_asm(base + ta_off[mi], kk);
                else af[mi][kk] = *(const bf16x8_t*)(base + a_off[kk] + mi * 2048);
            }
    };
    auto read_b = [&](int buf, int h) {
        const char* base = smem + buf * BUF2_BYTES + (2 + h) * HALF_BYTES;
        if constexpr (M32) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bf32[h][ks] = *(const bf16x8_t*)(base + b_off32[ks]);
            return;
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                if constexpr (TN) {
                    if (h == 0) bfr[ni][kk] = tr_frag_asm(base + tb_off[ni], kk);
                    else bfr1[ni][kk] = tr_frag_asm(base + tb_off[ni], kk);
                } else {
                    bfr[ni][kk] = *(const bf16x8_t*)(base + b_off[kk] + ni * 2048);
                }
            }
    };
#define IADR1_QUAD_B(MH, NH, BF)                                                                                             \
    do {                                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                                       \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                                     \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                     \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                                     \
            acc[(MH) * 4 + mi][(NH) * 2 + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[ni][kk], af[mi][kk], acc[(MH) * 4 + mi][(NH) * 2 + ni], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                       \
    } while (0)
    // M32: one A half against BOTH B halves = four independent 32 x 32 accumulators per phase, 16 MFMAs; the same accumulator comes round every fourth MFMA
    // (128 cycles apart -- with two accumulators per phase, 64 apart, the dependent MFMAs stalled: 1125 instead of 1270 TF/s, profiles/EXPERIMENTS.md round 6)
#define IADR1_HALF32(MH)                                                                                                     \
    do {                                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                                       \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                     \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                                     \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                                     \
            acc32[(MH) * 2 + mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf32[nt][ks], af32[mt][ks], acc32[(MH) * 2 + mt][nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                       \
    } while (0)
#define IADR1_QUAD(MH, NH)                                                                                                   \
    do {                                                                                                                     \
        if constexpr (TN) { if ((NH) == 0) IADR1_QUAD_B(MH, NH, bfr); else IADR1_QUAD_B(MH, NH, bfr1); }                     \
        else IADR1_QUAD_B(MH, NH, bfr);                                                                                      \
    } while (0)

    const int nk = (p.K + BK - 1) / BK;
    // Schedule (u = K tile, buffer u&1).  A half-tile region is re-filled ONE phase after the phase that read it;
    // that is safe because every phase retires its ds_reads (lgkmcnt(0)) BEFORE its barrier:
    //   phase 1: read A0,B0(u)  | DMA B0(u+1) -> other buffer   | barrier | quadrant (0,0)
    //   phase 2: read B1(u)     | DMA A0(u+2) -> this buffer    | barrier | quadrant (0,1)
    //   phase 3: read A1(u)     | DMA B1(u+2) -> this buffer    | barrier | quadrant (1,1)
    //   phase 4: read B0(u)     | DMA A1(u+2) -> this buffer    | vmcnt(6): all of tile u+1 landed, three
    //                                                              half-tiles of u+2 stay in flight | barrier | (1,0)
    // prologue: tile 0 complete + A0,B1,A1 of tile 1 in flight.
    stage_half(0, 0, 0);
    stage_half(0, 2, 0);
    stage_half(0, 3, 0);
    stage_half(0, 1, 0);
    if (nk > 1) {
        stage_half(1, 0, 1);
        stage_half(1, M32 ? 2 : 3, 1);
        stage_half(1, M32 ? 3 : 1, 1);
        IADR1_VMCNT(6);
    } else {
        IADR1_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();

#define IADR1_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    // Role split: the two waves of every SIMD (wm = 0 / 1) run half a phase apart -- while one issues its MFMAs the
    // other does its ds_reads / DMA issue -- by giving the wm=1 half ONE extra barrier here (and the wm=0 half one at
    // the end).  Each phase therefore has two barriers: [reads, DMA, lgkmcnt(0)] | barrier | [16 MFMA] | barrier.
    // The region re-fill / landing rules above still hold: a reader always passes one more barrier than the wait or
    // the last read it depends on.
    if (wm == 1) __builtin_amdgcn_s_barrier();
    if constexpr (M32) {
        // Two phases per K tile (same role split, same rules: a region is re-filled one phase after the phase that read it, every phase retires its ds_reads before
        // its barrier):
        //   phase A: read A0,B0,B1(u) | DMA A1(u+1) -> other buffer            | barrier | 16 MFMA: A0 x (B0, B1) | barrier
        //   phase B: read A1(u)       | DMA A0,B0,B1(u+2) -> this buffer, vmcnt(6): all of tile u+1 landed, three half-tiles of u+2 in flight | barrier | 16 MFMA: A1 x (B0, B1) | barrier
        // prologue: tile 0 complete + A0,B0,B1 of tile 1 in flight.  20 ds_read_b128 per K tile (B0 is not read twice), 4 barriers instead of 8.
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1, nxt = cur ^ 1;
            const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
            read_a(cur, 0);
            read_b(cur, 0);
            read_b(cur, 1);
            if (more1) stage_half(nxt, 1, kt + 1);
            IADR1_LGKM0();
            __builtin_amdgcn_s_barrier();
            IADR1_HALF32(0);
            __builtin_amdgcn_s_barrier();

            read_a(cur, 1);
            if (more2) { stage_half(cur, 0, kt + 2); stage_half(cur, 2, kt + 2); stage_half(cur, 3, kt + 2); IADR1_VMCNT(6); } else if (more1) { IADR1_VMCNT(0); }
            IADR1_LGKM0();
            __builtin_amdgcn_s_barrier();
            IADR1_HALF32(1);
            __builtin_amdgcn_s_barrier();
        }
    } else
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1, nxt = cur ^ 1;
        const bool more1 = kt + 1 < nk, more2 = kt + 2 < nk;
        read_a(cur, 0);
        read_b(cur, 0);
        if (more1) stage_half(nxt, 2, kt + 1);
        IADR1_LGKM0();
        __builtin_amdgcn_s_barrier();
        IADR1_QUAD(0, 0);
        __builtin_amdgcn_s_barrier();

        read_b(cur, 1);
        if (more2) stage_half(cur, 0, kt + 2);
        IADR1_LGKM0();
        __builtin_amdgcn_s_barrier();
        IADR1_QUAD(0, 1);
        __builtin_amdgcn_s_barrier();

        read_a(cur, 1);
        if (more2) stage_half(cur, 3, kt + 2);
        IADR1_LGKM0();
        __builtin_amdgcn_s_barrier();
        IADR1_QUAD(1, 1);
        __builtin_amdgcn_s_barrier();

        if constexpr (!TN) read_b(cur, 0);
        if (more2) { stage_half(cur, 1, kt + 2); IADR1_VMCNT(6); } else if (more1) { IADR1_VMCNT(0); }
        IADR1_LGKM0();
        __builtin_amdgcn_s_barrier();
        IADR1_QUAD(1, 0);
        __builtin_amdgcn_s_barrier();
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef IADR1_LGKM0
#undef IADR1_QUAD
#undef IADR1_HALF32
#undef IADR1_QUAD_B

    // ---- epilogue ------------------------------------------------------------------------------------------------
    if constexpr (M32) {
        // swapped-operand 32 x 32 accumulator: lane (lr = l & 31, lh = l >> 5) holds, of m-tile mt / n-tile nt, row mt*32 + lr and the four consecutive columns
        // nt*32 + g*8 + lh*4 + e of accumulator elements g*4 + e (g = 0..3)
        const int lr = l & 31, lh = l >> 5;
        if constexpr (OUT == OUT_BF16) {
            __syncthreads();  // every wave is done with the operand buffers
            bf16_t* slab = (bf16_t*)smem + (size_t)w * 128 * EP_LD;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = nt * 32 + g * 8 + lh * 4;
                    const int gn = n0 + wn * 64 + nl;
                    float bv[4] = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) bv[e] = (gn + e < p.N) ? bf2f(p.bias[gn + e]) : 0.f;
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc32[mt][nt][g * 4 + e] + bv[e];
                            if (p.act == 1) v[e] = gelu_erf(v[e]);
                        }
                        *(u32x2_t*)(slab + (mt * 32 + lr) * EP_LD + nl) = (u32x2_t){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                    }
                }
            bf16_t* C = (bf16_t*)p.C;
            const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0);
            if (vec_ok && m0 + wm * 128 + 128 <= p.M && n0 + wn * 64 + 64 <= p.N) {
                bf16_t* dst0 = C + (long long)(m0 + wm * 128 + (l >> 3)) * p.ldc + n0 + wn * 64 + (l & 7) * 8;
                const bf16_t* src0 = slab + (l >> 3) * EP_LD + (l & 7) * 8;
                u32x4_t rv[16];
#pragma unroll
                for (int it = 0; it < 16; ++it) rv[it] = *(const u32x4_t*)(src0 + it * 8 * EP_LD);
#pragma unroll
                for (int it = 0; it < 16; ++it) *(u32x4_t*)(dst0 + (long long)it * 8 * p.ldc) = rv[it];
                return;
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = it * 8 + (l >> 3), ch = l & 7;
                const int gm = m0 + wm * 128 + row, gn = n0 + wn * 64 + ch * 8;
                if (gm >= p.M || gn >= p.N) continue;
                bf16_t* dst = C + (long long)gm * p.ldc + gn;
                if (vec_ok && gn + 8 <= p.N) {
                    *(u32x4_t*)dst = *(const u32x4_t*)(slab + row * EP_LD + ch * 8);
                } else {
                    const bf16_t* sv = slab + row * EP_LD + ch * 8;
                    for (int e = 0; e < 8 && gn + e < p.N; ++e) dst[e] = sv[e];
                }
            }
        } else {
            float* C = (float*)p.C;
            const bool vec_ok = ((p.ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int gm = m0 + wm * 128 + mt * 32 + lr;
                if (gm >= p.M) continue;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int gn = n0 + wn * 64 + nt * 32 + g * 8 + lh * 4;
                        if (gn >= p.N) continue;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = acc32[mt][nt][g * 4 + e];
                            if (p.bias && gn + e < p.N) v[e] += bf2f(p.bias[gn + e]);
                            if (p.act == 1) v[e] = gelu_erf(v[e]);
                        }
                        float* dst = C + (long long)gm * p.ldc + gn;
                        if (vec_ok && gn + 4 <= p.N) {
                            f32x4_t o = {v[0], v[1], v[2], v[3]};
                            if constexpr (OUT == OUT_F32_ACC) o += *(const f32x4_t*)dst;
                            *(f32x4_t*)dst = o;
                        } else {
                            for (int e = 0; e < 4 && gn + e < p.N; ++e) {
                                if constexpr (OUT == OUT_F32_ACC) dst[e] += v[e];
                                else dst[e] = v[e];
                            }
                        }
                    }
            }
        }
        return;
    }
    const int lm = l & 15, lq = l >> 4;
    if constexpr (OUT == OUT_SWIGLU || OUT == OUT_SWIGLU_ROWS) {
        // interior tiles only (the launcher guarantees M % 256 == 0, N % 256 == 0, 16-byte aligned outputs)
        __syncthreads();  // every wave is done with the operand buffers
        bf16_t* slab = (bf16_t*)smem + (size_t)w * 128 * EP_LD;
        const int I = p.N >> 1;
        const long long row0 = m0 + wm * 128;
        const int ca = (n0 >> 1) + wn * 32;           // first of this wave's 32 output columns
        // row of the output matrices that GEMM row r lands in (the identity but for the row-blocked form)
        auto orow = [&](long long r) -> long long {
            if constexpr (OUT == OUT_SWIGLU_ROWS) return (r >> p.rb_shift) * p.rb_stride + (r & ((1 << p.rb_shift) - 1));
            else return r;
        };
        if (p.C) {   // the gate|up matrix itself (backward needs it): slab columns 0..31 = gate, 32..63 = up
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    *(u32x2_t*)(slab + (i * 16 + lm) * EP_LD + j * 16 + lq * 4) = (u32x2_t){pack2bf(acc[i][j][0], acc[i][j][1]), pack2bf(acc[i][j][2], acc[i][j][3])};
            bf16_t* C = (bf16_t*)p.C;
            const int ch = l & 7;
            bf16_t* dstc = C + (ch < 4 ? ca + ch * 8 : I + ca + (ch - 4) * 8);
            bf16_t* dst0 = dstc + (row0 + (l >> 3)) * p.ldc;
            const bf16_t* src0 = slab + (l >> 3) * EP_LD + ch * 8;
            u32x4_t rv[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) rv[it] = *(const u32x4_t*)(src0 + it * 8 * EP_LD);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                if constexpr (OUT == OUT_SWIGLU_ROWS) *(u32x4_t*)(dstc + orow(row0 + (l >> 3) + it * 8) * p.ldc) = rv[it];
                else *(u32x4_t*)(dst0 + (long long)it * 8 * p.ldc) = rv[it];
            }
        }
        // a = bf16(silu(bf16 gate)) * bf16 up, the arithmetic of swiglu_fwd_kernel on the rounded gate|up values
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float av[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gq = bf2f(f2bf(acc[i][j][e])), uq = bf2f(f2bf(acc[i][j + 2][e]));
                    av[e] = bf2f(f2bf(gq / (1.f + __expf(-gq)))) * uq;
                }
                *(u32x2_t*)(slab + (i * 16 + lm) * EP_LD + j * 16 + lq * 4) = (u32x2_t){pack2bf(av[0], av[1]), pack2bf(av[2], av[3])};
            }
        {
            bf16_t* dstc = p.C2 + ca + (l & 3) * 8;      // 4 lanes per 64-byte row segment, 16 rows per instruction
            bf16_t* dst0 = dstc + (row0 + (l >> 2)) * p.ldc2;
            const bf16_t* src0 = slab + (l >> 2) * EP_LD + (l & 3) * 8;
            u32x4_t rv[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) rv[it] = *(const u32x4_t*)(src0 + it * 16 * EP_LD);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if constexpr (OUT == OUT_SWIGLU_ROWS) *(u32x4_t*)(dstc + orow(row0 + (l >> 2) + it * 16) * p.ldc2) = rv[it];
                else *(u32x4_t*)(dst0 + (long long)it * 16 * p.ldc2) = rv[it];
            }
        }
    } else if constexpr (OUT == OUT_LSE) {
        // linear_logprob forward: per row and 64-column wave slice the pair (max, sum exp(x - max)) and the logit at the row's target column; the
        // logits themselves are never stored.  Lane (lm, lq) holds columns j*16 + lq*4 + e of rows i*16 + lm: in-lane over (j, e), then over lq.
        const int cbase = n0 + wn * 64;
        const bool full = cbase + 64 <= p.N;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = m0 + wm * 128 + i * 16 + lm;
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (full || cbase + j * 16 + lq * 4 + e < p.N) mx = fmaxf(mx, acc[i][j][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (full || cbase + j * 16 + lq * 4 + e < p.N) sm += __expf(acc[i][j][e] - mx);
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            if (row < p.M) {
                if (lq == 0) *(f32x2_t*)(p.part + ((long long)row * p.nparts + tn * 4 + wn) * 2) = (f32x2_t){mx, sm};
                const long long tg = p.targets[row];
                if (tg >= cbase && tg < cbase + 64) {
                    const int d = (int)(tg - cbase) - lq * 4;     // this lane holds d = j*16 + e, e in 0..3
                    float tv = 0.f;
                    bool hit = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (d == j * 16 + e) { tv = acc[i][j][e]; hit = true; }
                    if (hit) p.tgt_logit[row] = tv;
                }
            }
        }
    } else if constexpr (OUT == OUT_BF16 || OUT == OUT_DLOGITS) {
        __syncthreads();  // every wave is done with the operand buffers
        bf16_t* slab = (bf16_t*)smem + (size_t)w * 128 * EP_LD;
        if constexpr (OUT == OUT_DLOGITS) {
            // linear_logprob backward: the recomputed logits leave as dlogits = g * (onehot(target) - exp(x - lse)) in bf16 (iadr1_dlogits_rows's arithmetic)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = min(m0 + wm * 128 + i * 16 + lm, p.M - 1);
                const float rl = p.lse[row], rg = p.g[row];
                const long long tg = p.targets[row];
                const int d = (tg >= 0 && tg < p.N) ? (int)tg - (n0 + wn * 64) - lq * 4 : -1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = -rg * __expf(acc[i][j][e] - rl) + (d == j * 16 + e ? rg : 0.f);
                    *(u32x2_t*)(slab + (i * 16 + lm) * EP_LD + j * 16 + lq * 4) = (u32x2_t){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (OUT == OUT_DLOGITS) break;
            const int nl = j * 16 + lq * 4;
            const int gn = n0 + wn * 64 + nl;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (gn + e < p.N) ? bf2f(p.bias[gn + e]) : 0.f;
            }
            if (p.act == 1) {       // one uniform branch per column group, not one per element
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(acc[i][j][e] + bv[e]);
                    *(u32x2_t*)(slab + (i * 16 + lm) * EP_LD + nl) = (u32x2_t){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + bv[e];
                    *(u32x2_t*)(slab + (i * 16 + lm) * EP_LD + nl) = (u32x2_t){pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
                }
            }
        }
        // same wave wrote and reads its slab: only its own LDS ops need to retire (compiler inserts lgkmcnt)
        bf16_t* C = (bf16_t*)p.C;
        const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0);
        if (vec_ok && m0 + wm * 128 + 128 <= p.M && n0 + wn * 64 + 64 <= p.N) {
            // interior sub-tile (all of them on the hot shapes): 16 slab reads, then 16 row-contiguous 16-byte stores, no per-store predicate
            // (a predicated store is a branch; 16 of them serialise the slab reads behind the stores)
            bf16_t* dst0 = C + (long long)(m0 + wm * 128 + (l >> 3)) * p.ldc + n0 + wn * 64 + (l & 7) * 8;
            const bf16_t* src0 = slab + (l >> 3) * EP_LD + (l & 7) * 8;
            u32x4_t rv[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) rv[it] = *(const u32x4_t*)(src0 + it * 8 * EP_LD);
#pragma unroll
            for (int it = 0; it < 16; ++it) *(u32x4_t*)(dst0 + (long long)it * 8 * p.ldc) = rv[it];
            return;
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 8 + (l >> 3), ch = l & 7;
            const int gm = m0 + wm * 128 + row, gn = n0 + wn * 64 + ch * 8;
            if (gm >= p.M || gn >= p.N) continue;
            bf16_t* dst = C + (long long)gm * p.ldc + gn;
            if (vec_ok && gn + 8 <= p.N) {
                *(u32x4_t*)dst = *(const u32x4_t*)(slab + row * EP_LD + ch * 8);
            } else {
                const bf16_t* sv = slab + row * EP_LD + ch * 8;
                for (int e = 0; e < 8 && gn + e < p.N; ++e) dst[e] = sv[e];
            }
        }
    } else {
        float* C = (float*)p.C;
        const bool vec_ok = ((p.ldc & 3) == 0) && ((((uintptr_t)C) & 15) == 0);
        // first output column of n-tile j of this lane (TN: the wave's 64 columns are two 32-column runs 128 apart, see the B half images)
        auto coln = [&](int j) -> int {
            if constexpr (TN) return n0 + (j >> 1) * 128 + wn * 32 + (j & 1) * 16 + lq * 4;
            else return n0 + wn * 64 + j * 16 + lq * 4;
        };
        if (vec_ok && !p.bias && p.act == 0 && m0 + wm * 128 + 128 <= p.M && (TN ? n0 + T2 <= p.N : n0 + wn * 64 + 64 <= p.N)) {
            // interior sub-tile, plain store / accumulate (every wgrad and the lm_head): no predicates, and for the accumulate form the 8 loads
            // of two row groups are in flight together instead of one load -> add -> store round trip per 16 bytes
            float* base = C + (long long)(m0 + wm * 128 + lm) * p.ldc;
#pragma unroll
            for (int i0 = 0; i0 < 8; i0 += 2) {
                f32x4_t old[2][4];
                if constexpr (OUT == OUT_F32_ACC) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) old[u][j] = *(const f32x4_t*)(base + (long long)(i0 + u) * 16 * p.ldc + coln(j));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4_t o = acc[i0 + u][j];
                        if constexpr (OUT == OUT_F32_ACC) o += old[u][j];
                        *(f32x4_t*)(base + (long long)(i0 + u) * 16 * p.ldc + coln(j)) = o;
                    }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gm = m0 + wm * 128 + i * 16 + lm;
            if (gm >= p.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gn = coln(j);
                if (gn >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][e];
                    if (p.bias && gn + e < p.N) v[e] += bf2f(p.bias[gn + e]);
                    if (p.act == 1) v[e] = gelu_erf(v[e]);
                }
                float* dst = C + (long long)gm * p.ldc + gn;
                if (vec_ok && gn + 4 <= p.N) {
                    f32x4_t o = {v[0], v[1], v[2], v[3]};
                    if constexpr (OUT == OUT_F32_ACC) o += *(const f32x4_t*)dst;
                    *(f32x4_t*)dst = o;
                } else {
                    for (int e = 0; e < 4 && gn + e < p.N; ++e) {
                        if constexpr (OUT == OUT_F32_ACC) dst[e] += v[e];
                        else dst[e] = v[e];
                    }
                }
            }
        }
    }
}

// linear_logprob, second launch: one wave per row merges the row's (max, sum) pairs in a fixed order -> lse, logp = logit[target] - lse (0 for ignored rows).
__global__ __launch_bounds__(256) void lse_combine_kernel(const float* part, const float* tgt_logit, const long long* targets, float* logp, float* lse_out, int R, int V,
                                                          int nparts) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
    if (row >= R) return;
    const f32x2_t* pp = (const f32x2_t*)part + (long long)row * nparts;
    float m = -INFINITY, s = 0.f;
    for (int i = l; i < nparts; i += 64) {
        const f32x2_t v = pp[i];
        if (v[0] > m) { s *= __expf(m - v[0]); m = v[0]; }
        if (v[0] != -INFINITY) s += v[1] * __expf(v[0] - m);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(m, o), s2 = __shfl_xor(s, o);
        const float mm = fmaxf(m, m2);
        s = (m == -INFINITY ? 0.f : s * __expf(m - mm)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mm));
        m = mm;
    }
    if (l == 0) {
        const float lse = m + logf(s);
        if (lse_out) lse_out[row] = lse;
        const long long tg = targets[row];
        logp[row] = (tg >= 0 && tg < V) ? tgt_logit[row] - lse : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------
// Skinny GEMM for the rollout decode step:  Y[M (64 per grid.y), N] = X[M,K] . W[N,K]^T (+ bias).
// HBM-bound weight stream (6.17 GB of bf16 weights per decode step for 3B), and
// for the small projections latency-bound: the whole K extent is therefore spread over MANY waves --
// a block is WAVES (8/16) waves that share 16*NB output columns and take the 64-wide K slabs round-robin
// (optionally also split over grid.z), every wave keeps U slabs in flight, W fragments go straight
// HBM->VGPR with non-temporal 16-byte loads, X fragments come from L2.  The per-wave partial 64 x 16NB tiles
// are combined through LDS and leave as bf16 (+bias), fp32 logits, or -- when K is also split over grid.z --
// as fp32 partial slabs [z][M][N] that the fused residual+RMSNorm kernel sums (no atomics, no zero-init).
// ------------------------------------------------------------------------------------------------------
struct SkinnyArgs {
    const bf16_t* X;
    const bf16_t* W;
    void* Y;
    const bf16_t* bias;
    int M, N, K;
    long long ldx, ldw, ldy;
    int out_mode;  // 0 bf16 (+bias), 1 fp32, 2 fp32 partial slabs [gridDim.z][M][ldy], 3 fused SwiGLU (wide kernel, gate/up tile pairs),
                   // 4 fused q|k|v epilogue of the decode step (narrow kernel): bias + rotary + K/V cache append
    // out_mode 4 only: weights packed by iadr1_pack_qkv_rope_bf16 (rotary partners d, d+64 share a 16-column tile)
    const float* rope_cos;     // [M][D/2]
    const float* rope_sin;
    const long long* slot;     // [M] page*32 + offset of the new token, < 0: no cache write
    bf16_t* kcache;
    bf16_t* vcache;
    int Hq, Hkv;
    SideOut so;                // out_mode 4: p0 = roped q|k|v rows; out_mode 3 (persistent kernel): p0 = gate|up rows, p1 = SwiGLU rows (common.h)
    const float* wscale;       // non-null: W is FP8 (OCP e4m3) decode-packed (iadr1_pack_weight_fp8), wscale[n] = dequantisation scale of output row n
    int xcd_order;             // persistent kernel with side outputs: XCD-aware order of the tile groups
};

// FUSEV (out_mode 4, NB == 1 only): the blocks of the K heads also compute the V tile of the same head and 16-dim slice from the X fragments they have already
// loaded, and the grid holds the q and k tiles only.  For the 7B-class widths (28 q + 4 k + 4 v heads = 288 tiles on 256 CUs) that is ONE round of blocks
// instead of two: the second round cost 8 us of a 20 us launch (tools/narrow_ab.py), while a second 115 KB weight tile next to 458 KB of X costs a block ~25 %.
// MG = 16-row groups of X the block multiplies (4 = the 64-row decode tile).  Rollouts of <= 16 / <= 32 sequences (the reference launch scripts' B = 1 x G = 4,
// the evaluation harness) run MG = 1 / 2: a quarter / half of the X fragment loads (each 16-column block re-reads ALL X rows from L2: 41 MB per q|k|v launch at 64 rows
// against 10.5 MB of weights), of the MFMAs and of the cross-wave reduction.  Row groups are independent: a row's bits do not depend on MG.
template <int NB, int WAVES, bool FUSEV = false, int MG = 4>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(SkinnyArgs p) {
    constexpr int U = 2, BNC = 16 * NB, RLD = BNC + 1;
    static_assert(!FUSEV || (NB == 1 && MG == 4), "the fused V tile exists for the 16-column q|k|v kernel at the full 64-row tile");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;  // [WAVES][64][RLD] (+ a second one for the fused V tile)
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int lm = l & 15, lq = l >> 4;
    const int n0 = blockIdx.x * BNC;
    const int m_base = blockIdx.y * 64;
    const int nslab = p.K >> 6;  // full 64-wide slabs; a trailing 32-wide half slab (K % 64 == 32) is handled after the loops
    const int per_z = (nslab + gridDim.z - 1) / gridDim.z;
    const int s_begin = blockIdx.z * per_z, s_end = min(nslab, s_begin + per_z);
    STAMP(0);

    f32x4_t acc[MG][NB];
#pragma unroll
    for (int i = 0; i < MG; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // fused V tile of a K-head block (block-uniform): tile index of the same (kv head, 16-dim slice) among the V tiles
    const bool with_v = FUSEV && (int)(blockIdx.x >> 3) >= p.Hq;
    const int vtile = FUSEV ? (int)blockIdx.x + p.Hkv * 8 : 0;
    f32x4_t accv[FUSEV ? 4 : 1];
#pragma unroll
    for (int i = 0; i < (FUSEV ? 4 : 1); ++i) accv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // a thread's epilogue column is the same in every round (WAVES*64 is a multiple of BNC): fetch its bias now, not in the tail
    static_assert((WAVES * 64) % BNC == 0, "epilogue column must be loop invariant");
    const float bias_pre = ((p.out_mode == 0 || p.out_mode == 4) && p.bias) ? bf2f(p.bias[min(n0 + (t % BNC), p.N - 1)]) : 0.f;
    const long long side0 = (NB == 1 && p.out_mode == 4) ? side_base(p.so) : -1;
    // out_mode 4 (one output per thread when WAVES*64 == 64*16): rotary factors and the cache slot are fetched up front as well
    float pre_cos = 1.f, pre_sin = 0.f;
    long long pre_slot = -1;
    if (NB == 1 && p.out_mode == 4) {
        const int gm0 = min(m_base + (t >> 4), p.M - 1), dd0 = (blockIdx.x & 7) * 8 + (t & 7);
        pre_slot = p.slot[gm0];
        if ((int)(blockIdx.x >> 3) < p.Hq + p.Hkv) { pre_cos = p.rope_cos[(long long)gm0 * 64 + dd0]; pre_sin = p.rope_sin[(long long)gm0 * 64 + dd0]; }
    }
    // packed weights: fragment (n-tile, 32-k step) is 1 KiB contiguous in lane order -> one fully coalesced load
    const int ksteps = p.K >> 5;
    const bf16_t* wrow[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) wrow[j] = p.W + ((long long)min(blockIdx.x * NB + j, (p.N >> 4) - 1) * ksteps) * 512 + l * 8;
    const bf16_t* wrow_v = p.W + ((long long)min(vtile, (p.N >> 4) - 1) * ksteps) * 512 + l * 8;
    const bool xpk = p.ldx == 0;  // decode-packed X (see the wide kernel)
    const long long xstep = xpk ? 2048 : 32;
    const bf16_t* xrow[MG];
#pragma unroll
    for (int i = 0; i < MG; ++i)
        xrow[i] = xpk ? p.X + (long long)blockIdx.y * p.K * 64 + i * 512 + l * 8 : p.X + (long long)min(m_base + i * 16 + lm, p.M - 1) * p.ldx + lq * 8;

    // unpredicated loads (see the wide kernel): full trips of U slabs, then single-slab tail trips
    auto trip = [&](int sb, auto u_tag, auto v_tag) {
        constexpr int UU = decltype(u_tag)::value;
        constexpr bool WV = decltype(v_tag)::value;      // this block also walks its V tile (block-uniform: chosen once, outside the loops)
        bf16x8_t wf[UU][2][NB], xf[UU][2][MG], wv[FUSEV ? UU : 1][2];
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int k = (sb + u * WAVES) * 64 + kk * 32;
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    wf[u][kk][j] = __builtin_bit_cast(bf16x8_t, __builtin_nontemporal_load((const u32x4_t*)(wrow[j] + (long long)(k >> 5) * 512)));
                if constexpr (WV) wv[u][kk] = __builtin_bit_cast(bf16x8_t, __builtin_nontemporal_load((const u32x4_t*)(wrow_v + (long long)(k >> 5) * 512)));
#pragma unroll
                for (int i = 0; i < MG; ++i) xf[u][kk][i] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(xrow[i] + (long long)(k >> 5) * xstep));
            }
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MG; ++i) {
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u][kk][j], xf[u][kk][i], acc[i][j], 0, 0, 0);
                    if constexpr (WV) accv[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[u][kk], xf[u][kk][i], accv[i], 0, 0, 0);
                }
    };
    int sb = s_begin + w;
    if (FUSEV && with_v) {
        for (; sb + (U - 1) * WAVES < s_end; sb += WAVES * U) trip(sb, std::integral_constant<int, U>{}, std::integral_constant<bool, FUSEV>{});
        for (; sb < s_end; sb += WAVES) trip(sb, std::integral_constant<int, 1>{}, std::integral_constant<bool, FUSEV>{});
    } else {
        for (; sb + (U - 1) * WAVES < s_end; sb += WAVES * U) trip(sb, std::integral_constant<int, U>{}, std::false_type{});
        for (; sb < s_end; sb += WAVES) trip(sb, std::integral_constant<int, 1>{}, std::false_type{});
    }
    if ((p.K & 32) && w == 0 && blockIdx.z == gridDim.z - 1) {  // wave-uniform: the odd 32-wide tail of K
        const int k = nslab * 64;
        bf16x8_t wt[NB], xt[MG];
#pragma unroll
        for (int j = 0; j < NB; ++j) wt[j] = __builtin_bit_cast(bf16x8_t, __builtin_nontemporal_load((const u32x4_t*)(wrow[j] + (long long)(k >> 5) * 512)));
#pragma unroll
        for (int i = 0; i < MG; ++i) xt[i] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(xrow[i] + (long long)(k >> 5) * xstep));
#pragma unroll
        for (int i = 0; i < MG; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[j], xt[i], acc[i][j], 0, 0, 0);
    }
    // lane owns rows n = j*16 + lq*4 + e of column m = i*16 + lm  (swapped-operand C layout)
    STAMP(1);
    float* mine = red + (size_t)w * 64 * RLD;
#pragma unroll
    for (int i = 0; i < MG; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) mine[(i * 16 + lm) * RLD + j * 16 + lq * 4 + e] = acc[i][j][e];
    float* redv = red + (size_t)WAVES * 64 * RLD;
    if constexpr (FUSEV) {
        if (with_v) {
            float* minev = redv + (size_t)w * 64 * RLD;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) minev[(i * 16 + lm) * RLD + lq * 4 + e] = accv[i][e];
        }
    }
    __syncthreads();
    STAMP(2);
    if (NB == 1 && p.out_mode == 4) {
        // Decode-step q|k|v epilogue (replaces the separate rope + kv-store launch): this block owns one 16-column tile = 8
        // rotary pairs (d, d+64) of one head (q/k heads) or 16 plain dims (v heads).  Same rounding points as the unfused path:
        // bf16(x.W + b), rotary in fp32 on those bf16 values (TF:153-171), bf16 result.
        constexpr int D = 128, HALF = 64;
        const int head = blockIdx.x >> 3, j = blockIdx.x & 7;
        for (int idx = t; idx < MG * 16 * 16; idx += WAVES * 64) {
            const int m = idx >> 4, n = idx & 15, gm = m_base + m;
            if (gm >= p.M) continue;
            float vs = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) vs += red[((size_t)ww * 64 + m) * RLD + n];
            vs = bf2f(f2bf(vs + bias_pre));
            // the rotary partner (column n ^ 8 of the same row) is finished by lane ^ 8 of this wave: take its value instead of summing it again
            const float vp = __shfl_xor(vs, 8, WAVE);
            const bool first = idx == t && WAVES * 64 == 64 * 16;   // the prefetched values belong to this (m, n)
            const long long sl = first ? pre_slot : p.slot[gm];
            const long long page = sl >> 5;
            const int off = (int)(sl & 31);
            if (head < p.Hq + p.Hkv) {
                const int dd = j * 8 + (n & 7);
                const float cc = first ? pre_cos : p.rope_cos[(long long)gm * HALF + dd], ss = first ? pre_sin : p.rope_sin[(long long)gm * HALF + dd];
                const float a = n < 8 ? vs : vp, b = n < 8 ? vp : vs;
                const float r = n < 8 ? a * cc - b * ss : b * cc + a * ss;
                const int dim = n < 8 ? dd : HALF + dd;
                if (head < p.Hq) ((bf16_t*)p.Y)[(long long)gm * p.ldy + head * D + dim] = f2bf(r);
                else if (sl >= 0) p.kcache[(page * p.Hkv + (head - p.Hq)) * 32 * D + kpk_off(off, dim)] = f2bf(r);
                if (side0 >= 0) ((bf16_t*)p.so.p0)[(side0 + (long long)gm * p.so.seq_stride) * p.so.ld0 + head * D + dim] = f2bf(r);   // training layout: [q heads | k heads | v heads]
            } else {
                if (sl >= 0) p.vcache[(page * p.Hkv + (head - p.Hq - p.Hkv)) * (long long)D * 32 + (j * 16 + n) * 32 + off] = f2bf(vs);
                if (side0 >= 0) ((bf16_t*)p.so.p0)[(side0 + (long long)gm * p.so.seq_stride) * p.so.ld0 + head * D + j * 16 + n] = f2bf(vs);
            }
            if constexpr (FUSEV) {
                if (with_v) {      // the V tile of the same kv head and 16-dim slice: bias, cache append, training row -- what the V head's own block does above
                    float vv = 0.f;
#pragma unroll
                    for (int ww = 0; ww < WAVES; ++ww) vv += redv[((size_t)ww * 64 + m) * RLD + n];
                    vv = bf2f(f2bf(vv + (p.bias ? bf2f(p.bias[vtile * 16 + n]) : 0.f)));
                    const int hv = head + p.Hkv;
                    if (sl >= 0) p.vcache[(page * p.Hkv + (hv - p.Hq - p.Hkv)) * (long long)D * 32 + (j * 16 + n) * 32 + off] = f2bf(vv);
                    if (side0 >= 0) ((bf16_t*)p.so.p0)[(side0 + (long long)gm * p.so.seq_stride) * p.so.ld0 + hv * D + j * 16 + n] = f2bf(vv);
                }
            }
        }
        STAMP(3);
        return;
    }
    for (int idx = t; idx < MG * 16 * BNC; idx += WAVES * 64) {
        const int m = idx / BNC, n = idx - m * BNC;
        const int gm = m_base + m, gn = n0 + n;
        if (gm >= p.M || gn >= p.N) continue;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) v += red[((size_t)ww * 64 + m) * RLD + n];
        if (p.out_mode == 0) {
            v += bias_pre;
            ((bf16_t*)p.Y)[(long long)gm * p.ldy + gn] = f2bf(v);
        } else if (p.out_mode == 1) {
            ((float*)p.Y)[(long long)gm * p.ldy + gn] = v;
        } else {
            ((float*)p.Y)[((long long)blockIdx.z * p.M + gm) * p.ldy + gn] = v;
        }
    }
}

// Wide variant (MLP up/down projections, lm_head): one wave owns 16*NB (64/128) columns for its K slabs, so an
// X fragment fetched from L2 feeds NB MFMA column tiles (vector-memory traffic per weight byte drops from 3-5x
// to 1.5-2x: the X re-read through L2, not HBM, limited the narrow kernel), and the 32-deep half-slabs are
// software pipelined: the loads of step j+1 are in flight while the MFMAs of step j run.  K may also be split
// over grid.z (narrow-N, long-K down projection) -> fp32 partial slabs for the fused residual+RMSNorm.
// eight FP8 (OCP e4m3) weights -> a bf16 MFMA fragment: v_cvt_scalef32_pk_bf16_fp8, one instruction per pair (exact: every e4m3 value is a bf16 value)
__device__ __forceinline__ bf16x8_t fp8x8_to_bf16(uint32_t lo, uint32_t hi) {
    const bf16x2_t a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2_t c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    return (bf16x8_t){a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}

// FP8 = true: the weights are FP8 decode-packed -- Wp8[n/16][k/64][lane][16 B], bytes 0-7 = the lane's 8 weights of k-step 2s, bytes 8-15 = of k-step 2s+1
// (iadr1_pack_weight_fp8) -- so ONE 16-byte load feeds two 32-deep MFMA steps and the weight stream is half as long; the fragments are widened to bf16
// in registers and the output columns are multiplied by the per-row dequantisation scale in the epilogue.  A "step" of the loops is then 64 deep.
template <int NB, int WAVES, bool FP8 = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_wide_kernel(SkinnyArgs p) {
    constexpr int RC = 32, RLD = RC + 1;  // columns per reduction round
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;  // [WAVES][64][RLD]
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const int lm = l & 15, lq = l >> 4;
    const int n0 = blockIdx.x * 16 * NB;
    const int m_base = blockIdx.y * 64;
    const int nstep = FP8 ? p.K >> 6 : p.K >> 5;
    const int per_z = (nstep + gridDim.z - 1) / gridDim.z;
    const int st_begin = blockIdx.z * per_z, st_end = min(nstep, st_begin + per_z);

    f32x4_t acc[4][NB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // addressing kept in a few registers: tile j of this block starts j*tile_stride after tile 0 (wide kernel requires
    // N % (16*NB) == 0), X row group i is i*16 rows further down
    const int ksteps = p.K >> 5;
    const long long tile_stride = (long long)ksteps * 512;      // elements of one 16-row weight tile (= bytes in the FP8 form)
    const bf16_t* wbase = p.W + (long long)blockIdx.x * NB * tile_stride + l * 8;
    const char* wbase8 = (const char*)p.W + (long long)blockIdx.x * NB * tile_stride + l * 16;
    // X row-major: lane (lm, lq) reads 16 B of row m_base + 16 i + lm (16 cache lines per wave-load).  X decode-packed
    // (ldx == 0, common.h xpk_off): fragment (k-step, row group) is 1 KiB contiguous in lane order, rows padded to 64.
    const bool xpk = p.ldx == 0;
    const int xr0 = min(m_base + lm, p.M - 1);
    const bf16_t* xbase = xpk ? p.X + (long long)blockIdx.y * p.K * 64 + l * 8 : p.X + (long long)xr0 * p.ldx + lq * 8;
    const long long xgroup = xpk ? 512 : 16 * p.ldx;
    const long long xstep = xpk ? 2048 : 32;
    const int xgroups_ok = xpk ? 4 : (p.M - m_base - lm + 15) / 16;  // row groups i < xgroups_ok are real rows for this lane

    // K is walked in 32-deep steps; wave w takes steps s_begin*2 + w, + WAVES, ... and keeps DEPTH steps in flight
    // (all their loads are issued before the first MFMA of the trip).  Measured against slab-wise software pipelining
    // (tools/stream_probe.py): 4.2 vs 3.0 TB/s on the gate|up stream.
    // No load in these loops is predicated: a per-load `if (ok)` makes hipcc branch around every load and wait for
    // each one (24 branches + waits per trip measured: 43 us instead of 21 us on the gate|up stream).  K % 32 == 0 is
    // guaranteed by the packed layout; the ragged end of a wave's step list runs in the DEPTH=1 tail loop.
    constexpr int DEPTH = 2;
    auto trip = [&](int s0, auto depth_tag) {
        constexpr int DD = decltype(depth_tag)::value;
        if constexpr (FP8) {
            u32x4_t wq[DD][NB];
            bf16x8_t xq[DD][2][4];
#pragma unroll
            for (int d = 0; d < DD; ++d) {
                const long long st = s0 + d * WAVES;
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) wq[d][jj] = __builtin_nontemporal_load((const u32x4_t*)(wbase8 + jj * tile_stride + st * 1024));
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        xq[d][h][i] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(xbase + (i < xgroups_ok ? i * xgroup : 0) + (2 * st + h) * xstep));
            }
#pragma unroll
            for (int d = 0; d < DD; ++d)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int jj = 0; jj < NB; ++jj) {
                        const bf16x8_t wfr = fp8x8_to_bf16(wq[d][jj][2 * h], wq[d][jj][2 * h + 1]);
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xq[d][h][i], acc[i][jj], 0, 0, 0);
                    }
            return;
        }
        bf16x8_t wf[DD][NB], xf[DD][4];
#pragma unroll
        for (int d = 0; d < DD; ++d) {
            const long long st = s0 + d * WAVES;
#pragma unroll
            for (int jj = 0; jj < NB; ++jj)
                wf[d][jj] = __builtin_bit_cast(bf16x8_t, __builtin_nontemporal_load((const u32x4_t*)(wbase + jj * tile_stride + st * 512)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
                xf[d][i] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(xbase + (i < xgroups_ok ? i * xgroup : 0) + st * xstep));
        }
#pragma unroll
        for (int d = 0; d < DD; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[d][jj], xf[d][i], acc[i][jj], 0, 0, 0);
    };
    int s0 = st_begin + w;
    for (; s0 + (DEPTH - 1) * WAVES < st_end; s0 += WAVES * DEPTH) trip(s0, std::integral_constant<int, DEPTH>{});
    for (; s0 < st_end; s0 += WAVES) trip(s0, std::integral_constant<int, 1>{});
    float* mine = red + (size_t)w * 64 * RLD;
#pragma unroll
    for (int r = 0; r < NB / 2; ++r) {
        if (r) __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int e = 0; e < 4; ++e) mine[(i * 16 + lm) * RLD + jj * 16 + lq * 4 + e] = acc[i][2 * r + jj][e];
        __syncthreads();
        if (p.out_mode == 3) {
            // tiles (2r, 2r+1) of this round are the GATE and UP projections of the same 16 output columns
            // (iadr1_pack_gateup_bf16 interleaves them): a = silu(bf16(gate)) * bf16(up), TF::95-96,552-554
            for (int idx = t; idx < 64 * 16; idx += WAVES * 64) {
                const int m = idx >> 4, n = idx & 15;
                const int gm = m_base + m, gn = (n0 >> 1) + r * 16 + n;
                if (gm >= p.M || gn >= (p.N >> 1)) continue;
                float g = 0.f, u = 0.f;
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) {
                    g += red[((size_t)ww * 64 + m) * RLD + n];
                    u += red[((size_t)ww * 64 + m) * RLD + 16 + n];
                }
                if constexpr (FP8) { g *= p.wscale[gn]; u *= p.wscale[(p.N >> 1) + gn]; }
                g = bf2f(f2bf(g));
                u = bf2f(f2bf(u));
                const float sg = bf2f(f2bf(g / (1.f + __expf(-g))));
                ((bf16_t*)p.Y)[p.ldy ? (long long)gm * p.ldy + gn : xpk_off(gm, gn, p.N >> 1)] = f2bf(sg * u);
            }
            continue;
        }
        for (int idx = t; idx < 64 * RC; idx += WAVES * 64) {
            const int m = idx / RC, n = idx - m * RC;
            const int gm = m_base + m, gn = n0 + r * RC + n;
            if (gm >= p.M || gn >= p.N) continue;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) v += red[((size_t)ww * 64 + m) * RLD + n];
            if constexpr (FP8) v *= p.wscale[gn];
            if (p.out_mode == 0) {
                if (p.bias) v += bf2f(p.bias[gn]);
                ((bf16_t*)p.Y)[(long long)gm * p.ldy + gn] = f2bf(v);
            } else if (p.out_mode == 1) {
                ((float*)p.Y)[(long long)gm * p.ldy + gn] = v;
            } else {
                ((float*)p.Y)[((long long)blockIdx.z * p.M + gm) * p.ldy + gn] = v;
            }
        }
    }
}

// Persistent variant for the big decode streams (gate|up, down, lm_head): ONE block per CU stays resident, keeps the X fragments of
// its waves' k-steps in VGPRs for the whole launch (X is read once per CU instead of once per 64-column block: the X re-read was
// 3 us of the 24.6 us gate|up call) and walks groups of two column tiles; the W loads of the next group are issued BEFORE the LDS
// reduction of the current one, so the weight stream does not stop during epilogues (with one-shot blocks every block of the grid
// streams, then every block reduces).  tools/decode_stream.py on weights that really come from HBM: 19.8 vs 24.9 us on the 90 MB
// gate|up stream (4.55 TB/s; a plain read of the same bytes runs at 5.4).  grid.x = nz * bps blocks: slice z = blockIdx.x / bps of K
// (split-K partial slabs, out_mode 2), block b = blockIdx.x % bps takes tile groups b, b + bps, ...
// FP8 = true: the weights are the FP8 decode pack (see gemm_skinny_wide_kernel): a lane's 16-byte load holds its fragments of k-steps 2s and 2s + 1, so a
// wave owns the DOUBLE steps w + j * WAVES (j < KSW / 2) and keeps the X fragments of both halves; the fragments are widened to bf16 in registers and the
// output columns multiplied by the per-row scale in the epilogue.  Half the weight bytes per tile, the same MFMA count.
template <int WAVES, int KSW, bool FP8 = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_pers_kernel(SkinnyArgs p) {
    constexpr int TPI = 2, RLD = 16 * TPI + 1;
    static_assert(!FP8 || KSW % 2 == 0, "FP8 packs pair the k-steps");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;  // [WAVES][64][RLD]
    const int t = threadIdx.x, w = t >> 6, l = t & 63, lm = l & 15, lq = l >> 4;
    const int b = blockIdx.x, bps = gridDim.x;
    const int m_base = blockIdx.y * 64;
    const long long sb = p.out_mode == 3 ? side_base(p.so) : -1;
    STAMP(4);
    const long long tile_stride = (long long)WAVES * KSW * 512;   // K == 32 * WAVES * KSW exactly (host checks); elements of a bf16 tile = BYTES of an FP8 tile
    // X fragments of this wave's k-steps w + j*WAVES (FP8: 2 (w + (j/2) WAVES) + (j & 1)): loaded once, resident for the whole launch
    const bool xpk = p.ldx == 0;
    const bf16_t* xbase = xpk ? p.X + (long long)blockIdx.y * p.K * 64 + l * 8 : p.X + (long long)min(m_base + lm, p.M - 1) * p.ldx + lq * 8;
    const long long xgroup = xpk ? 512 : 16 * p.ldx, xstep = xpk ? 2048 : 32;
    const int xgroups_ok = xpk ? 4 : (p.M - m_base - lm + 15) / 16;
    const bf16_t* wlane = p.W + (long long)w * 512 + l * 8;
    const char* wlane8 = (const char*)p.W + (long long)w * 1024 + l * 16;
    const int ngroups = p.N / (16 * TPI);
    bf16x8_t wf[FP8 ? 1 : KSW][TPI];
    u32x4_t wq[FP8 ? KSW / 2 : 1][TPI];
    auto loadw = [&](int g) {
        if constexpr (FP8) {
            const char* wb = wlane8 + (long long)g * TPI * tile_stride;
#pragma unroll
            for (int j = 0; j < KSW / 2; ++j)
#pragma unroll
                for (int tt = 0; tt < TPI; ++tt) wq[j][tt] = __builtin_nontemporal_load((const u32x4_t*)(wb + tt * tile_stride + j * (WAVES * 1024)));
        } else {
            const bf16_t* wb = wlane + (long long)g * TPI * tile_stride;
#pragma unroll
            for (int j = 0; j < KSW; ++j)
#pragma unroll
                for (int tt = 0; tt < TPI; ++tt) wf[j][tt] = __builtin_bit_cast(bf16x8_t, __builtin_nontemporal_load((const u32x4_t*)(wb + tt * tile_stride + j * (WAVES * 512))));
        }
    };
    // Order of the tile groups.  Plain: block b takes b, b + bps, ...  With side outputs (rows of the row-major training arena: 32 bytes of a row per
    // group and array) the four groups that share a 128-byte line of a row go, in the SAME iteration, to four blocks of ONE XCD (blockIdx.x % 8 is
    // the XCD a block lands on), so that the partial-line stores merge in that XCD's L2 instead of leaving four L2s as four masked writes:
    // XCD x owns the groups g with (g / 4) % 8 == x; its bps / 8 blocks take them round-robin.
    const bool xcd_order = sb >= 0 && (bps & 7) == 0 && p.xcd_order;
    const int gq0 = xcd_order ? (b >> 3) : b, gstep = xcd_order ? (bps >> 3) : bps;
    auto group_of = [&](int q) { return xcd_order ? (((q >> 2) * 8 + (b & 7)) * 4 + (q & 3)) : q; };
    int gq = gq0;
    int g = group_of(gq);
    // the first group's weights are requested BEFORE the X fragments: they come from HBM (the longer latency) and do not depend on anything, the
    // X fragments are 256 KB per CU out of L2 -- measured with in-kernel stamps (profiles/r02_decode_stamps.txt): 6.4 us from entry to the first
    // group's MFMAs with X first
    if (g < ngroups) loadw(g);
    bf16x8_t xf[KSW][4];
#pragma unroll
    for (int j = 0; j < KSW; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            xf[j][i] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(xbase + (i < xgroups_ok ? i * xgroup : 0) + (long long)(FP8 ? 2 * (w + (j >> 1) * WAVES) + (j & 1) : w + j * WAVES) * xstep));
    float* mine = red + (size_t)w * 64 * RLD;
    for (; g < ngroups; gq += gstep, g = group_of(gq)) {
        const int g_next = group_of(gq + gstep);
        f32x4_t acc[4][TPI];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tt = 0; tt < TPI; ++tt) acc[i][tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if constexpr (FP8) {
#pragma unroll
            for (int j = 0; j < KSW; ++j)
#pragma unroll
                for (int tt = 0; tt < TPI; ++tt) {
                    const bf16x8_t wfr = fp8x8_to_bf16(wq[j >> 1][tt][2 * (j & 1)], wq[j >> 1][tt][2 * (j & 1) + 1]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xf[j][i], acc[i][tt], 0, 0, 0);
                }
        } else {
            // (round 6) the weight registers of k-step j are re-loaded for the NEXT group as soon as this group's MFMAs of k-step j have been issued -- no second
            // register set, and the stream does not pause for the MFMA phase (before: the whole next group was requested after the last MFMA).  Same arithmetic, same
            // bits; gate|up 21.2 -> 20.6 us, lm_head 113.0 -> 111.7 us on rotating weights, decode step -0.01 ... -0.03 ms (profiles/r06_skinny_reissue_ab.txt)
            if (g_next < ngroups) {
                const bf16_t* wbn = wlane + (long long)g_next * TPI * tile_stride;
#pragma unroll
                for (int j = 0; j < KSW; ++j) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int tt = 0; tt < TPI; ++tt) acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][tt], xf[j][i], acc[i][tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < TPI; ++tt) wf[j][tt] = __builtin_bit_cast(bf16x8_t, __builtin_nontemporal_load((const u32x4_t*)(wbn + tt * tile_stride + j * (WAVES * 512))));
                }
            } else {
#pragma unroll
            for (int j = 0; j < KSW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int tt = 0; tt < TPI; ++tt) acc[i][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][tt], xf[j][i], acc[i][tt], 0, 0, 0);
            }
        }
        if (gq == gq0) STAMP(5);
        if (FP8 && g_next < ngroups) loadw(g_next);     // (FP8 pack: the whole next group at once, in flight during the reduction below)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tt = 0; tt < TPI; ++tt)
#pragma unroll
                for (int e = 0; e < 4; ++e) mine[(i * 16 + lm) * RLD + tt * 16 + lq * 4 + e] = acc[i][tt][e];
        __syncthreads();
        const int n0 = g * 16 * TPI;
        if (p.out_mode == 3) {
            // tiles (2g, 2g+1) are the GATE and UP projections of the same 16 output columns (iadr1_pack_gateup_bf16)
            for (int idx = t; idx < 64 * 16; idx += WAVES * 64) {
                const int m = idx >> 4, n = idx & 15, gm = m_base + m, gn = g * 16 + n;
                if (gm >= p.M) continue;
                float gsum = 0.f, usum = 0.f;
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) {
                    gsum += red[((size_t)ww * 64 + m) * RLD + n];
                    usum += red[((size_t)ww * 64 + m) * RLD + 16 + n];
                }
                if constexpr (FP8) { gsum *= p.wscale[gn]; usum *= p.wscale[(p.N >> 1) + gn]; }
                gsum = bf2f(f2bf(gsum));
                usum = bf2f(f2bf(usum));
                const float sg = bf2f(f2bf(gsum / (1.f + __expf(-gsum))));
                ((bf16_t*)p.Y)[p.ldy ? (long long)gm * p.ldy + gn : xpk_off(gm, gn, p.N >> 1)] = f2bf(sg * usum);
                if (sb >= 0) {      // the same values in the training layout: gate|up row [gate | up] and the activation row
                    const long long r = sb + (long long)gm * p.so.seq_stride;
                    bf16_t* gu = (bf16_t*)p.so.p0 + r * p.so.ld0;
                    gu[gn] = f2bf(gsum);
                    gu[(p.N >> 1) + gn] = f2bf(usum);
                    ((bf16_t*)p.so.p1)[r * p.so.ld1 + gn] = f2bf(sg * usum);
                }
            }
        } else {
            for (int idx = t; idx < 64 * 16 * TPI; idx += WAVES * 64) {
                const int m = idx / (16 * TPI), n = idx - m * (16 * TPI), gm = m_base + m, gn = n0 + n;
                if (gm >= p.M) continue;
                float v = 0.f;
#pragma unroll
                for (int ww = 0; ww < WAVES; ++ww) v += red[((size_t)ww * 64 + m) * RLD + n];
                if constexpr (FP8) v *= p.wscale[gn];
                if (p.out_mode == 0) {
                    if (p.bias) v += bf2f(p.bias[gn]);
                    ((bf16_t*)p.Y)[(long long)gm * p.ldy + gn] = f2bf(v);
                } else if (p.out_mode == 1) {
                    ((float*)p.Y)[(long long)gm * p.ldy + gn] = v;
                } else {
                    ((float*)p.Y)[(long long)gm * p.ldy + gn] = v;   // out_mode 2 with a single K slice
                }
            }
        }
        __syncthreads();
    }
    STAMP(6);
}

// Split-K form of the persistent kernel for the long-K narrow-N down projection (out_mode 2, fp32 partial slabs): grid.x = nz * bps,
// slice z = blockIdx.x / bps owns k-steps [z*per_z, ...), its bps blocks walk 16-column tiles b, b + bps, ... (one tile per iteration so
// that a block has >= 4 iterations to pipeline: with two-tile groups it had 2 and ran slower than the one-shot kernel).  A wave's steps
// past the end of the slice get a zero X fragment and a clamped (valid, redundant) W address: no predicated loads in the loop.
// FP8 = true: K is walked in 64-deep DOUBLE steps of the FP8 decode pack (KSW of them per wave), see gemm_skinny_pers_kernel.
template <int WAVES, int KSW, bool FP8 = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_pers_split_kernel(SkinnyArgs p, int nz, int bps) {
    constexpr int RLD = 16 + 1, XS = FP8 ? 2 : 1;      // X fragments per step of the weight walk
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;  // [WAVES][64][RLD]
    const int t = threadIdx.x, w = t >> 6, l = t & 63, lm = l & 15, lq = l >> 4;
    const int z = blockIdx.x / bps, b = blockIdx.x - z * bps;
    const int m_base = blockIdx.y * 64;
    const int ksteps = FP8 ? p.K >> 6 : p.K >> 5;        // steps of the weight walk: 32 deep (bf16), 64 deep (FP8: one 16-byte load = two MFMA k-steps)
    const int per_z = (ksteps + nz - 1) / nz;
    const int st_begin = z * per_z, st_end = min(ksteps, st_begin + per_z);
    const long long tile_stride = (long long)ksteps * (FP8 ? 1024 : 512);      // elements of a bf16 tile / bytes of an FP8 tile
    const bool xpk = p.ldx == 0;
    const bf16_t* xbase = xpk ? p.X + (long long)blockIdx.y * p.K * 64 + l * 8 : p.X + (long long)min(m_base + lm, p.M - 1) * p.ldx + lq * 8;
    const long long xgroup = xpk ? 512 : 16 * p.ldx, xstep = xpk ? 2048 : 32;
    const int xgroups_ok = xpk ? 4 : (p.M - m_base - lm + 15) / 16;
    bf16x8_t xf[KSW * XS][4];
    int woff[KSW];
#pragma unroll
    for (int j = 0; j < KSW; ++j) woff[j] = min(st_begin + w + j * WAVES, st_end - 1) * (FP8 ? 1024 : 512) + l * (FP8 ? 16 : 8);
    const int ntiles = p.N >> 4;
    u32x4_t wf[KSW];
    auto loadw = [&](int g) {
        if constexpr (FP8) {
            const char* wb = (const char*)p.W + (long long)g * tile_stride;
#pragma unroll
            for (int j = 0; j < KSW; ++j) wf[j] = __builtin_nontemporal_load((const u32x4_t*)(wb + woff[j]));
        } else {
            const bf16_t* wb = p.W + (long long)g * tile_stride;
#pragma unroll
            for (int j = 0; j < KSW; ++j) wf[j] = __builtin_nontemporal_load((const u32x4_t*)(wb + woff[j]));
        }
    };
    int g = b;
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        const int st = st_begin + w + j * WAVES;
        const bool ok = st < st_end;
#pragma unroll
        for (int h = 0; h < XS; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4_t v = {0, 0, 0, 0};
                if (ok) v = *(const u32x4_t*)(xbase + (i < xgroups_ok ? i * xgroup : 0) + (long long)(st * XS + h) * xstep);
                xf[j * XS + h][i] = __builtin_bit_cast(bf16x8_t, v);
            }
    }
    if (g < ntiles) loadw(g);          // after the X fragments here: measured 12.2 vs 12.8 us with the weights first (the opposite of the un-split kernel)
    float* mine = red + (size_t)w * 64 * RLD;
    for (; g < ntiles; g += bps) {
        f32x4_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            if constexpr (FP8) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bf16x8_t wfr = fp8x8_to_bf16(wf[j][2 * h], wf[j][2 * h + 1]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xf[j * 2 + h][i], acc[i], 0, 0, 0);
                }
            } else {
                const bf16x8_t wfr = __builtin_bit_cast(bf16x8_t, wf[j]);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr, xf[j][i], acc[i], 0, 0, 0);
            }
        }
        if (g + bps < ntiles) loadw(g + bps);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) mine[(i * 16 + lm) * RLD + lq * 4 + e] = acc[i][e];
        __syncthreads();
        for (int idx = t; idx < 64 * 16; idx += WAVES * 64) {
            const int m = idx >> 4, n = idx & 15, gm = m_base + m;
            if (gm >= p.M) continue;
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) v += red[((size_t)ww * 64 + m) * RLD + n];
            if constexpr (FP8) v *= p.wscale[g * 16 + n];          // the per-row dequantisation scale distributes over the K slices
            ((float*)p.Y)[((long long)z * p.M + gm) * p.ldy + g * 16 + n] = v;
        }
        __syncthreads();
    }
}

// Decode-packing of a fused gate|up matrix W[2I, K] for the SwiGLU-fused skinny GEMM: packed 16-row tile 2q holds
// gate rows [16q, 16q+16), tile 2q+1 the matching up rows [I+16q, ...), so one block owns both halves of its columns.
__global__ __launch_bounds__(256) void pack_gateup_kernel(const bf16_t* W, long long ldw, bf16_t* Wp, int I, int K) {
    const int ksteps = K >> 5;
    const long long total = (long long)(2 * I >> 4) * ksteps * 64;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long long ts = i >> 6;
        const int ks = (int)(ts % ksteps);
        const long long tile = ts / ksteps;
        const long long n = (tile & 1 ? I : 0) + (tile >> 1) * 16 + (lane & 15);
        const int k = ks * 32 + (lane >> 4) * 8;
        *(u32x4_t*)(Wp + i * 8) = *(const u32x4_t*)(W + n * ldw + k);
    }
}

// Repack W[N,K] (row-major) into MFMA-fragment order for the decode stream:
//   Wp[n/16][k/32][lane = (n%16) + 16*((k%32)/8)][k%8]
__global__ __launch_bounds__(256) void pack_weight_kernel(const bf16_t* W, long long ldw, bf16_t* Wp, int N, int K) {
    const int ksteps = K >> 5;
    const long long total = (long long)(N >> 4) * ksteps * 64;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long long ts = i >> 6;
        const int ks = (int)(ts % ksteps);
        const long long tile = ts / ksteps;
        const long long n = tile * 16 + (lane & 15);
        const int k = ks * 32 + (lane >> 4) * 8;
        *(u32x4_t*)(Wp + i * 8) = *(const u32x4_t*)(W + n * ldw + k);
    }
}

// Decode-packing of the fused q|k|v matrix for the out_mode-4 epilogue: like pack_weight_kernel, but inside every q and k head
// the 128 rows are dealt to the 8 column tiles as [8j, 8j+8) ++ [64+8j, 64+8j+8) so a tile holds complete rotary pairs; v heads keep
// their natural order.  The bias vector is permuted the same way.
__global__ __launch_bounds__(256) void pack_qkv_rope_kernel(const bf16_t* W, long long ldw, const bf16_t* bias, bf16_t* Wp, bf16_t* bias_p, int n_rope_heads, int N, int K) {
    const int ksteps = K >> 5;
    const long long total = (long long)(N >> 4) * ksteps * 64;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long long ts = i >> 6;
        const int ks = (int)(ts % ksteps);
        const int tile = (int)(ts / ksteps), lm = lane & 15;
        const int head = tile >> 3, j = tile & 7;
        const long long n = head < n_rope_heads ? (long long)head * 128 + (lm < 8 ? 8 * j + lm : 64 + 8 * j + lm - 8) : (long long)tile * 16 + lm;
        const int k = ks * 32 + (lane >> 4) * 8;
        *(u32x4_t*)(Wp + i * 8) = *(const u32x4_t*)(W + n * ldw + k);
        if (ks == 0 && lane < 16 && bias) bias_p[tile * 16 + lm] = bias[n];
    }
}

// FP8 decode pack, pass 1: dequantisation scale of every output row, scale[n] = max_k |w[n][k]| / 448 (448 = largest finite OCP e4m3 value)
__global__ __launch_bounds__(256) void fp8_row_scale_kernel(const bf16_t* W, long long ldw, float* scale, int K) {
    __shared__ float scratch[16];
    const long long n = blockIdx.x;
    float amax = 0.f;
    for (int c = threadIdx.x; c < (K >> 3); c += 256) {
        const u32x4_t v = *(const u32x4_t*)(W + n * ldw + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(lo_bf(v[e])), fabsf(hi_bf(v[e]))));
    }
    amax = block_max<256>(amax, scratch);
    if (threadIdx.x == 0) scale[n] = fmaxf(amax, 1e-30f) / 448.f;
}
// pass 2: Wp8[n/16][k/64][lane][16 bytes] (see gemm_skinny_wide_kernel<.., FP8>); I > 0: gate|up matrix, 16-row tiles of gate and up interleaved like
// pack_gateup_kernel.  Round-to-nearest-even through v_cvt_pk_fp8_f32; |w / scale| <= 448 by construction.
__global__ __launch_bounds__(256) void pack_fp8_kernel(const bf16_t* W, long long ldw, const float* scale, uint32_t* Wp8, int N, int K, int I) {
    const int dsteps = K >> 6;
    const long long total = (long long)(N >> 4) * dsteps * 64;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int lane = (int)(i & 63);
        const long long ts = i >> 6;
        const int sd = (int)(ts % dsteps);
        const long long tile = ts / dsteps;
        const long long n = I > 0 ? ((tile & 1 ? I : 0) + (tile >> 1) * 16 + (lane & 15)) : tile * 16 + (lane & 15);
        const float sc = scale[n];     // true divisions below (not a reciprocal multiply): the quantised values are then the round-to-nearest e4m3 of w / scale exactly
        u32x4_t o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = sd * 64 + h * 32 + (lane >> 4) * 8;
            const u32x4_t v = *(const u32x4_t*)(W + n * ldw + k);
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[0]) / sc, hi_bf(v[0]) / sc, lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[1]) / sc, hi_bf(v[1]) / sc, lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[2]) / sc, hi_bf(v[2]) / sc, hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(lo_bf(v[3]) / sc, hi_bf(v[3]) / sc, hi, true);
            o[2 * h] = (uint32_t)lo;
            o[2 * h + 1] = (uint32_t)hi;
        }
        *(u32x4_t*)(Wp8 + i * 4) = o;
    }
}

// X[M,K] row-major -> decode-packed (common.h xpk_off); pad rows (M..roundup64) are zero-filled
__global__ __launch_bounds__(256) void pack_act_kernel(const bf16_t* X, long long ldx, bf16_t* Xp, int M, int K) {
    const int Mp = (M + 63) & ~63;
    const long long total = (long long)Mp * (K >> 3);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / (K >> 3)), c = (int)(i % (K >> 3));
        const u32x4_t v = m < M ? *(const u32x4_t*)(X + (long long)m * ldx + c * 8) : (u32x4_t){0, 0, 0, 0};
        *(u32x4_t*)(Xp + xpk_off(m, c * 8, K)) = v;
    }
}

// C[m][n] += sum_z ws[z][m][n], z in slice order (fixed: the sum does not depend on which block finished first); 4 floats per thread
__global__ __launch_bounds__(256) void splitk_reduce_acc_kernel(const float* ws, float* C, long long ldc, int M, int N, int ks, long long zstride) {
    const int nq = N >> 2;
    const long long total = (long long)M * nq;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / nq), c = (int)(i - (long long)m * nq) * 4;
        const float* src = ws + (long long)m * N + c;
        f32x4_t a = *(const f32x4_t*)src;
        for (int z = 1; z < ks; ++z) a += *(const f32x4_t*)(src + z * zstride);
        float* dst = C + (long long)m * ldc + c;
        *(f32x4_t*)dst = *(const f32x4_t*)dst + a;
    }
}

__device__ char g_zero16[64] __attribute__((aligned(64)));

}  // namespace

static const void* zeros_ptr() {
    static void* z = nullptr;
    if (!z) (void)hipGetSymbolAddress(&z, HIP_SYMBOL(g_zero16));
    return z;
}

extern "C" int iadr1_gemm_nt_bf16(const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                                  long long lda, long long ldb, long long ldc, int out_mode, int act, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
    IADR1_REQUIRE((K % 8) == 0 && (lda % 8) == 0 && (ldb % 8) == 0, "gemm_nt: K, lda, ldb must be multiples of 8 (16-byte chunks); K=%d lda=%lld ldb=%lld", K, lda, ldb);
    IADR1_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0, "gemm_nt: A/B must be 16-byte aligned");
    IADR1_REQUIRE(out_mode >= 0 && out_mode <= 2, "gemm_nt: bad out_mode %d", out_mode);
    constexpr int band_rows = 4;      // tile rows per rasterisation band: 1 / 2 / 4 / 8 measured within 4 % on the hot shapes, 4 best (profiles/r05_gemm_band.txt)
    GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, zeros_ptr(), M, N, K, lda, ldb, ldc, act, band_rows, nullptr, 0};
    static const int force_tile = iadr1_env_int("IADR1_GEMM_TILE", 0);
    static const bool attr_done = [] {
        (void)hipFuncSetAttribute((const void*)gemm_nt_128<OUT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_128<OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_128<OUT_F32_ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_F32_ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
#ifdef IADR1_PROBE_MFMA32
        (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_BF16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_F32, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_F32_ACC, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
#endif
        return true;
    }();
    (void)attr_done;
    // 256^2 deep-pipeline kernel when the grid fills the chip with big tiles; 128^2 kernel for small / ragged problems
    const long long tiles256 = (long long)((M + T2 - 1) / T2) * ((N + T2 - 1) / T2);
    const bool big = force_tile == 256 || (force_tile != 128 && M >= 512 && N >= 512 && tiles256 >= 192);
#ifdef IADR1_PROBE_MFMA32      // probe builds only (tools/build_variant.py mfma32 -DIADR1_PROBE_MFMA32; tools/gemm_mfma32_ab.py): the 256^2 kernel on v_mfma_f32_32x32x16_bf16.
    if (big) {                  // Measured round 6: 9 % SLOWER than the 16x16x32 form on every hot shape (profiles/r06_gemm_mfma32_ab.txt) -- not in the product build.
        const int grid = (int)tiles256;
        if (out_mode == 0) hipLaunchKernelGGL((gemm_nt_256<OUT_BF16, false, true>), dim3(grid), dim3(NT2), SMEM2_BYTES, stream, p);
        else if (out_mode == 1) hipLaunchKernelGGL((gemm_nt_256<OUT_F32, false, true>), dim3(grid), dim3(NT2), SMEM2_BYTES, stream, p);
        else hipLaunchKernelGGL((gemm_nt_256<OUT_F32_ACC, false, true>), dim3(grid), dim3(NT2), SMEM2_BYTES, stream, p);
        return iadr1_check_launch("gemm_nt_bf16");
    }
#endif
    if (big) {
        const int grid = (int)tiles256;
        if (out_mode == 0) hipLaunchKernelGGL(gemm_nt_256<OUT_BF16>, dim3(grid), dim3(NT2), SMEM2_BYTES, stream, p);
        else if (out_mode == 1) hipLaunchKernelGGL(gemm_nt_256<OUT_F32>, dim3(grid), dim3(NT2), SMEM2_BYTES, stream, p);
        else hipLaunchKernelGGL(gemm_nt_256<OUT_F32_ACC>, dim3(grid), dim3(NT2), SMEM2_BYTES, stream, p);
        return iadr1_check_launch("gemm_nt_bf16");
    }
    const int grid = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (out_mode == 0) hipLaunchKernelGGL(gemm_nt_128<OUT_BF16>, dim3(grid), dim3(NTHREADS), SMEM_BYTES, stream, p);
    else if (out_mode == 1) hipLaunchKernelGGL(gemm_nt_128<OUT_F32>, dim3(grid), dim3(NTHREADS), SMEM_BYTES, stream, p);
    else hipLaunchKernelGGL(gemm_nt_128<OUT_F32_ACC>, dim3(grid), dim3(NTHREADS), SMEM_BYTES, stream, p);
    return iadr1_check_launch("gemm_nt_bf16");
}

// Split-K form of the accumulate mode for contractions with FEW output tiles and a long K -- the weight gradients of the narrow projections
// (dW_o [2048 x 2048], dW_qkv [2560 x 2048] over K = 20480 token rows: 64 / 80 tiles of 256^2 on 256 CUs ran at 310 / 612 TFLOP/s), the down projection's
// (344 tiles: 1.3 rounds) and the vision tower's: ksplit K slices per tile fill the chip, fp32 partial tiles go to a workspace, a second launch adds them
// to C in slice order (no atomics: bit-reproducible, and equal to the un-split kernel up to fp32 summation order).
extern "C" long long iadr1_gemm_nt_splitk_workspace_bytes(int M, int N, int ksplit) {
    return (M <= 0 || N <= 0 || ksplit <= 1) ? 0 : (long long)ksplit * M * N * 4;
}

extern "C" int iadr1_gemm_nt_splitk_acc_bf16(const void* A, const void* B, float* C, void* workspace, int M, int N, int K, long long lda, long long ldb, long long ldc,
                                             int ksplit, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && N > 0 && K > 0 && ksplit >= 2, "gemm_nt_splitk: empty problem / ksplit < 2 (M=%d N=%d K=%d ksplit=%d)", M, N, K, ksplit);
    IADR1_REQUIRE((K % 8) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (N % 4) == 0 && (ldc % 4) == 0, "gemm_nt_splitk: K, lda, ldb multiples of 8, N, ldc multiples of 4");
    IADR1_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0 && (((uintptr_t)C) & 15) == 0 && workspace && (((uintptr_t)workspace) & 15) == 0,
                  "gemm_nt_splitk: A, B, C and the workspace must be 16-byte aligned");
    const int kslice = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;      // whole 64-deep K tiles per slice
    IADR1_REQUIRE((long long)(ksplit - 1) * kslice < K, "gemm_nt_splitk: ksplit %d leaves an empty slice for K = %d", ksplit, K);
    constexpr int band_rows = 4;      // tile rows per rasterisation band: 1 / 2 / 4 / 8 measured within 4 % on the hot shapes, 4 best (profiles/r05_gemm_band.txt)
    GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, workspace, nullptr, zeros_ptr(), M, N, K, lda, ldb, (long long)N, 0, band_rows, nullptr, 0};
    p.ksplit = ksplit; p.kslice = kslice; p.zstride = (long long)M * N;
    static const bool attr_done = [] { (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES); return true; }();
    (void)attr_done;
    const int tiles = ((M + T2 - 1) / T2) * ((N + T2 - 1) / T2);
    hipLaunchKernelGGL(gemm_nt_256<OUT_F32>, dim3(tiles * ksplit), dim3(NT2), SMEM2_BYTES, stream, p);
    if (int rc = iadr1_check_launch("gemm_nt_splitk (partials)")) return rc;
    long long blocks = ((long long)M * (N / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_acc_kernel, dim3((int)blocks), dim3(256), 0, stream, (const float*)workspace, C, ldc, M, N, ksplit, (long long)M * N);
    return iadr1_check_launch("gemm_nt_splitk (reduce)");
}

// Weight gradients without transposed copies:  C[M, N] (fp32) += A[K, M]^T . B[K, N]  with A (= dY) and B (= X) row-major as the backward holds them, the contraction
// over their K rows (gemm_nt_256<.., TN = true>).  ksplit >= 2: the split form above (K slices of whole 64-row tiles, fp32 partial tiles in `workspace` --
// iadr1_gemm_nt_splitk_workspace_bytes(M, N, ksplit) -- added to C in slice order); ksplit <= 1: one accumulating launch.  Bit-equal to iadr1_gemm_nt_bf16 /
// iadr1_gemm_nt_splitk_acc_bf16 on transposed copies (same tiles, same order of the contraction).
extern "C" int iadr1_gemm_tn_acc_bf16(const void* A, const void* B, float* C, void* workspace, int M, int N, int K, long long lda, long long ldb, long long ldc,
                                      int ksplit, hipStream_t stream) {
    IADR1_REQUIRE(M >= 256 && N >= 256 && K > 0, "gemm_tn: needs M >= 256, N >= 256, K > 0 (M=%d N=%d K=%d)", M, N, K);
    IADR1_REQUIRE((M % 8) == 0 && (N % 8) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (N % 4) == 0 && (ldc % 4) == 0, "gemm_tn: M, N, lda, ldb multiples of 8, ldc a multiple of 4");
    IADR1_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)B) & 15) == 0 && (((uintptr_t)C) & 15) == 0, "gemm_tn: A, B, C must be 16-byte aligned");
    constexpr int band_rows = 4;      // tile rows per rasterisation band: 1 / 2 / 4 / 8 measured within 4 % on the hot shapes, 4 best (profiles/r05_gemm_band.txt)
    static const bool attr_done = [] {
        (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_F32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_F32_ACC, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
        return true;
    }();
    (void)attr_done;
    const int tiles = ((M + T2 - 1) / T2) * ((N + T2 - 1) / T2);
    if (ksplit <= 1) {
        GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, C, nullptr, zeros_ptr(), M, N, K, lda, ldb, ldc, 0, band_rows, nullptr, 0};
        hipLaunchKernelGGL((gemm_nt_256<OUT_F32_ACC, true>), dim3(tiles), dim3(NT2), SMEM2_BYTES, stream, p);
        return iadr1_check_launch("gemm_tn_acc_bf16");
    }
    IADR1_REQUIRE(workspace && (((uintptr_t)workspace) & 15) == 0, "gemm_tn: the split form needs a 16-byte aligned workspace");
    const int kslice = ((K + ksplit - 1) / ksplit + BK - 1) / BK * BK;      // whole 64-row K tiles per slice
    IADR1_REQUIRE((long long)(ksplit - 1) * kslice < K, "gemm_tn: ksplit %d leaves an empty slice for K = %d", ksplit, K);
    GemmArgs p{(const bf16_t*)A, (const bf16_t*)B, workspace, nullptr, zeros_ptr(), M, N, K, lda, ldb, (long long)N, 0, band_rows, nullptr, 0};
    p.ksplit = ksplit; p.kslice = kslice; p.zstride = (long long)M * N;
    hipLaunchKernelGGL((gemm_nt_256<OUT_F32, true>), dim3(tiles * ksplit), dim3(NT2), SMEM2_BYTES, stream, p);
    if (int rc = iadr1_check_launch("gemm_tn (partials)")) return rc;
    long long blocks = ((long long)M * (N / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_acc_kernel, dim3((int)blocks), dim3(256), 0, stream, (const float*)workspace, C, ldc, M, N, ksplit, (long long)M * N);
    return iadr1_check_launch("gemm_tn (reduce)");
}

// linear_logprob: logp[r] = log_softmax(H[r] . W^T)[targets[r]] and lse[r] without the [M, V] logits ever reaching HBM (gemm_nt_256 with the OUT_LSE
// epilogue + lse_combine_kernel), and its backward's first half: dlogits (bf16) from the recomputed logits (OUT_DLOGITS epilogue).
extern "C" long long iadr1_linear_logprob_workspace_bytes(int M, int V) {
    if (M <= 0 || V <= 0) return 0;
    const long long nparts = (long long)((V + T2 - 1) / T2) * 4;
    return (long long)M * nparts * 8 + (long long)M * 4;
}

static int linear_logprob_args(const char* what, const void* H, const void* W, int M, int V, int K, long long ldh, long long ldw) {
    IADR1_REQUIRE(M > 0 && V > 0 && K > 0, "%s: empty problem M=%d V=%d K=%d", what, M, V, K);
    IADR1_REQUIRE((K % 8) == 0 && (ldh % 8) == 0 && (ldw % 8) == 0, "%s: K, ldh, ldw must be multiples of 8 (16-byte chunks); K=%d ldh=%lld ldw=%lld", what, K, ldh, ldw);
    IADR1_REQUIRE((((uintptr_t)H) & 15) == 0 && (((uintptr_t)W) & 15) == 0, "%s: H/W must be 16-byte aligned", what);
    return 0;
}

extern "C" int iadr1_linear_logprob_fwd(const void* H, const void* W, const long long* targets, float* logp, float* lse, void* workspace, int M, int V, int K,
                                        long long ldh, long long ldw, hipStream_t stream) {
    if (int rc = linear_logprob_args("linear_logprob_fwd", H, W, M, V, K, ldh, ldw)) return rc;
    IADR1_REQUIRE(targets && logp && workspace && (((uintptr_t)workspace) & 7) == 0, "linear_logprob_fwd: targets, logp and an 8-byte aligned workspace are required");
    constexpr int band_rows = 4;      // tile rows per rasterisation band: 1 / 2 / 4 / 8 measured within 4 % on the hot shapes, 4 best (profiles/r05_gemm_band.txt)
    const int tiles_n = (V + T2 - 1) / T2, nparts = tiles_n * 4;
    float* part = (float*)workspace;
    float* tgt_logit = part + (long long)M * nparts * 2;
    GemmArgs p{(const bf16_t*)H, (const bf16_t*)W, nullptr, nullptr, zeros_ptr(), M, V, K, ldh, ldw, 0, 0, band_rows, nullptr, 0, targets, part, tgt_logit, nullptr, nullptr, nparts};
    static const bool attr_done = [] { (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_LSE>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES); return true; }();
    (void)attr_done;
    hipLaunchKernelGGL(gemm_nt_256<OUT_LSE>, dim3(((M + T2 - 1) / T2) * tiles_n), dim3(NT2), SMEM2_BYTES, stream, p);
    if (int rc = iadr1_check_launch("linear_logprob_fwd")) return rc;
    hipLaunchKernelGGL(lse_combine_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, part, tgt_logit, targets, logp, lse, M, V, nparts);
    return iadr1_check_launch("linear_logprob_fwd (combine)");
}

extern "C" int iadr1_linear_logprob_dlogits(const void* H, const void* W, const long long* targets, const float* lse, const float* g, void* dl, long long ldd, int M, int V,
                                            int K, long long ldh, long long ldw, hipStream_t stream) {
    if (int rc = linear_logprob_args("linear_logprob_dlogits", H, W, M, V, K, ldh, ldw)) return rc;
    IADR1_REQUIRE(targets && lse && g && dl, "linear_logprob_dlogits: targets, lse, g and dl are required");
    constexpr int band_rows = 4;      // tile rows per rasterisation band: 1 / 2 / 4 / 8 measured within 4 % on the hot shapes, 4 best (profiles/r05_gemm_band.txt)
    GemmArgs p{(const bf16_t*)H, (const bf16_t*)W, dl, nullptr, zeros_ptr(), M, V, K, ldh, ldw, ldd, 0, band_rows, nullptr, 0, targets, nullptr, nullptr, lse, g, 0};
    static const bool attr_done = [] { (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_DLOGITS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES); return true; }();
    (void)attr_done;
    hipLaunchKernelGGL(gemm_nt_256<OUT_DLOGITS>, dim3(((M + T2 - 1) / T2) * ((V + T2 - 1) / T2)), dim3(NT2), SMEM2_BYTES, stream, p);
    return iadr1_check_launch("linear_logprob_dlogits");
}

// gate|up projection with the SwiGLU fused into the epilogue of gemm_nt_256 (training / prefill shapes): GU[M, 2I] = A . W^T (stored when GU != null:
// the backward pass needs it) and Aout[M, I] = bf16(silu(GU[:, :I])) * GU[:, I:], bit-identical to iadr1_gemm_nt_bf16 + iadr1_swiglu_fwd.
extern "C" int iadr1_gemm_swiglu_bf16(const void* A, const void* W, void* GU, void* Aout, int M, int I, int K, long long lda, long long ldw,
                                      long long ldgu, long long ldaout, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && I > 0 && K > 0 && Aout != nullptr, "gemm_swiglu: empty problem");
    IADR1_REQUIRE((M % 256) == 0 && (I % 128) == 0 && (K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (ldaout % 8) == 0 && (GU == nullptr || (ldgu % 8) == 0),
                  "gemm_swiglu: needs M %% 256 == 0, I %% 128 == 0 and 16-byte row strides (M=%d I=%d K=%d); use gemm_nt + swiglu_fwd otherwise", M, I, K);
    IADR1_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && (((uintptr_t)GU) & 15) == 0 && (((uintptr_t)Aout) & 15) == 0, "gemm_swiglu: operands must be 16-byte aligned");
    constexpr int band_rows = 4;      // tile rows per rasterisation band: 1 / 2 / 4 / 8 measured within 4 % on the hot shapes, 4 best (profiles/r05_gemm_band.txt)
    GemmArgs p{(const bf16_t*)A, (const bf16_t*)W, GU, nullptr, zeros_ptr(), M, 2 * I, K, lda, ldw, ldgu, 0, band_rows, (bf16_t*)Aout, ldaout};
    static const bool attr_done = [] { (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_SWIGLU>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES); return true; }();
    (void)attr_done;
    hipLaunchKernelGGL(gemm_nt_256<OUT_SWIGLU>, dim3((M / 256) * (I / 128)), dim3(NT2), SMEM2_BYTES, stream, p);
    return iadr1_check_launch("gemm_swiglu_bf16");
}

// The same contraction over ROW BLOCKS of a larger matrix: GEMM row r (0 <= r < M) is row (r / block) * block_stride + r % block of A, GU and Aout (block a power
// of two >= 16, all three matrices addressed from the given base pointers).  Per row the arithmetic is iadr1_gemm_swiglu_bf16's, bit for bit.
extern "C" int iadr1_gemm_swiglu_rows_bf16(const void* A, const void* W, void* GU, void* Aout, int M, int I, int K, long long lda, long long ldw,
                                           long long ldgu, long long ldaout, int block, long long block_stride, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && I > 0 && K > 0 && Aout != nullptr, "gemm_swiglu_rows: empty problem");
    IADR1_REQUIRE((M % 256) == 0 && (I % 128) == 0 && (K % 8) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && (ldaout % 8) == 0 && (GU == nullptr || (ldgu % 8) == 0),
                  "gemm_swiglu_rows: needs M %% 256 == 0, I %% 128 == 0 and 16-byte row strides (M=%d I=%d K=%d)", M, I, K);
    IADR1_REQUIRE(block >= 16 && (block & (block - 1)) == 0 && (M % block) == 0 && block_stride >= block, "gemm_swiglu_rows: block must be a power of two >= 16 dividing M, block_stride >= block (block=%d)", block);
    IADR1_REQUIRE((((uintptr_t)A) & 15) == 0 && (((uintptr_t)W) & 15) == 0 && (((uintptr_t)GU) & 15) == 0 && (((uintptr_t)Aout) & 15) == 0, "gemm_swiglu_rows: operands must be 16-byte aligned");
    constexpr int band_rows = 4;      // tile rows per rasterisation band: 1 / 2 / 4 / 8 measured within 4 % on the hot shapes, 4 best (profiles/r05_gemm_band.txt)
    GemmArgs p{(const bf16_t*)A, (const bf16_t*)W, GU, nullptr, zeros_ptr(), M, 2 * I, K, lda, ldw, ldgu, 0, band_rows, (bf16_t*)Aout, ldaout};
    p.rb_shift = __builtin_ctz((unsigned)block);
    p.rb_stride = block_stride;
    static const bool attr_done = [] { (void)hipFuncSetAttribute((const void*)gemm_nt_256<OUT_SWIGLU_ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES); return true; }();
    (void)attr_done;
    hipLaunchKernelGGL(gemm_nt_256<OUT_SWIGLU_ROWS>, dim3((M / 256) * (I / 128)), dim3(NT2), SMEM2_BYTES, stream, p);
    return iadr1_check_launch("gemm_swiglu_rows_bf16");
}

extern "C" int iadr1_gemm_skinny_bf16(const void* X, const void* W, void* Y, const void* bias, int M, int N, int K, long long ldx,
                                      long long ldw, long long ldy, int out_mode, int ksplit, const void* side, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_skinny: empty problem");
    IADR1_REQUIRE((K % 32) == 0 && (N % 16) == 0 && (ldx % 8) == 0, "gemm_skinny: packed weights need K %% 32 == 0 and N %% 16 == 0 (K=%d N=%d)", K, N);
    IADR1_REQUIRE(ldy != 0 || (out_mode == 3 && (N % 64) == 0), "gemm_skinny: a decode-packed output (ldy == 0) exists for the fused-SwiGLU mode only");
    IADR1_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)W) & 15) == 0, "gemm_skinny: X/W must be 16-byte aligned");
    (void)ldw;
    IADR1_REQUIRE(out_mode >= 0 && out_mode <= 3 && ksplit >= 1 && (ksplit == 1 || out_mode == 2), "gemm_skinny: out_mode 0-3; ksplit > 1 needs out_mode 2 (partial slabs)");
    IADR1_REQUIRE(out_mode != 3 || (N % 128) == 0, "gemm_skinny: fused SwiGLU needs N (= 2*I) to be a multiple of 128, got %d", N);
    SkinnyArgs p{};
    p.X = (const bf16_t*)X; p.W = (const bf16_t*)W; p.Y = Y; p.bias = (const bf16_t*)bias; p.M = M; p.N = N; p.K = K;
    p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.out_mode = out_mode;
    p.xcd_order = 1;      // (decode step with side outputs 2.886 -> 2.834 ms: profiles/EXPERIMENTS.md round 2)
    const int mz = (M + 63) / 64;
    // dynamic LDS: the cross-wave reduction buffer [WAVES][64][RLD]
    constexpr int SM1 = 16 * 64 * 17 * 4, SM2 = 8 * 64 * 33 * 4, SMW = 8 * 64 * 33 * 4, SMP = SMW, SMS = 8 * 64 * 17 * 4;
    // launcher configuration, fixed at first use: A/B switches, the CU count, LDS opt-ins of every kernel this entry point can launch
    static const int pers = iadr1_env_int("IADR1_SKINNY_PERS", 1);
    static const int dev_cus_ = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, SM1);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<1, 16, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SM1);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<1, 16, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SM1);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SM2);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_wide_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SMW);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_wide_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SMW);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_kernel<8, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, SMP);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_kernel<8, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, SMP);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_kernel<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, SMP);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_split_kernel<8, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, SMS);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_split_kernel<8, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, SMS);
        return (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }();
    (void)dev_cus_;
    const int ncu = iadr1_decode_cus();      // the CUs the decode stream owns (runtime.hip): persistent grids are one block per such CU
    // wide kernels: NB=8 (128 columns / block) for the big-N streams (gate|up, lm_head); NB=4 with K split over
    // grid.z for the long-K narrow-N down projection; the narrow kernel for the small projections (latency-bound)
    // persistent X-resident kernel for the big un-split streams (gate|up, lm_head): K = 32 * 8 waves * KSW exactly, >= 2 tile groups per CU.
    // Not for the split-K down projection (2 groups per block: two exposed memory latencies, 18.9 vs 14.2 us measured).
    {
        {   // split-K slabs (down projection): <= 6 k-steps per wave in a slice, >= 4 tiles per block
            const int kst = K >> 5, per_z = (kst + ksplit - 1) / ksplit, bps = ksplit > 0 ? ncu / ksplit : 0;
            if (pers && ksplit > 1 && out_mode == 2 && per_z <= 80 && bps >= 1 && (N >> 4) >= 4 * bps && (kst % ksplit == 0 || (ksplit - 1) * per_z < kst)) {
                IADR1_REQUIRE(side == nullptr, "gemm_skinny: no side outputs in the split-K slab form");
                // <= 48 k-steps per slice: 6 resident X steps per wave (3B widths: 344 / 8 = 43); <= 80: 10 (7B widths: 592 / 8 = 74; 160 VGPRs of X fragments)
                if (per_z <= 48) hipLaunchKernelGGL((gemm_skinny_pers_split_kernel<8, 6>), dim3(bps * ksplit, mz, 1), dim3(512), SMS, stream, p, ksplit, bps);
                else hipLaunchKernelGGL((gemm_skinny_pers_split_kernel<8, 10>), dim3(bps * ksplit, mz, 1), dim3(512), SMS, stream, p, ksplit, bps);
                return iadr1_check_launch("gemm_skinny_bf16");
            }
        }
        const int ksw = K / 256;
        const bool pers_ok = pers && ksplit == 1 && out_mode <= 3 && (N % 32) == 0 && (K % 256) == 0 && (ksw == 8 || ksw == 6 || ksw == 4) && (N / 32) >= 2 * ncu;
        if (int e = iadr1_side_arg(side, &p.so)) return e;
        IADR1_REQUIRE(!p.so.step || (out_mode == 3 && pers_ok),
                      "gemm_skinny: side outputs exist for the fused-SwiGLU projection in the persistent kernel only (mode %d N=%d K=%d)", out_mode, N, K);
        if (pers_ok) {
            const dim3 grid(ncu, mz, 1), block(512);
            if (ksw == 8) hipLaunchKernelGGL((gemm_skinny_pers_kernel<8, 8>), grid, block, SMP, stream, p);
            else if (ksw == 6) hipLaunchKernelGGL((gemm_skinny_pers_kernel<8, 6>), grid, block, SMP, stream, p);
            else hipLaunchKernelGGL((gemm_skinny_pers_kernel<8, 4>), grid, block, SMP, stream, p);
            return iadr1_check_launch("gemm_skinny_bf16");
        }
    }
    // with decode-packed X the X fragments are cheap coalesced L2 reads, and 64-column blocks (two co-resident per CU, 344
    // blocks on the 3B gate|up) beat 128-column ones: 24.6 vs 33.7 us on the 90 MB gate|up stream (tools/decode_stream.py)
    const bool big = (N >= 8192 && ksplit == 1) || out_mode == 3;
    const int nb = ldx == 0 ? 4 : 8;      // 64-column blocks with packed X, 128-column ones with row-major X
    if (big && nb == 4 && (N % 64) == 0) hipLaunchKernelGGL((gemm_skinny_wide_kernel<4, 8>), dim3(N / 64, mz, 1), dim3(512), SMW, stream, p);
    else if (big && (N % 128) == 0) hipLaunchKernelGGL((gemm_skinny_wide_kernel<8, 8>), dim3((N + 127) / 128, mz, 1), dim3(512), SMW, stream, p);
    else if (ksplit > 1 && (N % 64) == 0 && (N / 64) * ksplit >= 192) hipLaunchKernelGGL((gemm_skinny_wide_kernel<4, 8>), dim3((N + 63) / 64, mz, ksplit), dim3(512), SMW, stream, p);
    else if (N >= 8192) hipLaunchKernelGGL((gemm_skinny_kernel<2, 8>), dim3((N + 31) / 32, mz, ksplit), dim3(512), SM2, stream, p);
    else if (M <= 16) hipLaunchKernelGGL((gemm_skinny_kernel<1, 16, false, 1>), dim3((N + 15) / 16, mz, ksplit), dim3(1024), SM1, stream, p);      // (see MG at the kernel)
    else if (M <= 32) hipLaunchKernelGGL((gemm_skinny_kernel<1, 16, false, 2>), dim3((N + 15) / 16, mz, ksplit), dim3(1024), SM1, stream, p);
    else hipLaunchKernelGGL((gemm_skinny_kernel<1, 16>), dim3((N + 15) / 16, mz, ksplit), dim3(1024), SM1, stream, p);
    return iadr1_check_launch("gemm_skinny_bf16");
}

extern "C" int iadr1_pack_weight_bf16(const void* W, long long ldw, void* Wp, int N, int K, hipStream_t stream) {
    IADR1_REQUIRE(N > 0 && K > 0 && (N % 16) == 0 && (K % 32) == 0 && (ldw % 8) == 0, "pack_weight: need N %% 16 == 0, K %% 32 == 0 (N=%d K=%d)", N, K);
    long long blocks = ((long long)N * K / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)W, ldw, (bf16_t*)Wp, N, K);
    return iadr1_check_launch("pack_weight_bf16");
}

extern "C" int iadr1_gemm_qkv_rope_kv_bf16(const void* X, const void* Wp, const void* bias_p, void* q_out, const float* rope_cos, const float* rope_sin,
                                           const long long* slot, void* kcache, void* vcache, int M, int Hq, int Hkv, int D, int K, long long ldx,
                                           long long ldq, const void* side, hipStream_t stream) {
    IADR1_REQUIRE(D == 128, "gemm_qkv_rope_kv: head dim %d not built (128 is)", D);
    IADR1_REQUIRE(M > 0 && Hq > 0 && Hkv > 0 && (K % 32) == 0 && (ldx % 8) == 0, "gemm_qkv_rope_kv: need K %% 32 == 0 (K=%d)", K);
    IADR1_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Wp) & 15) == 0, "gemm_qkv_rope_kv: X/W must be 16-byte aligned");
    SkinnyArgs p{};
    p.X = (const bf16_t*)X; p.W = (const bf16_t*)Wp; p.Y = q_out; p.bias = (const bf16_t*)bias_p; p.M = M; p.N = (Hq + 2 * Hkv) * D; p.K = K;
    p.ldx = ldx; p.ldw = K; p.ldy = ldq; p.out_mode = 4;
    p.rope_cos = rope_cos; p.rope_sin = rope_sin; p.slot = slot; p.kcache = (bf16_t*)kcache; p.vcache = (bf16_t*)vcache; p.Hq = Hq; p.Hkv = Hkv;
    if (int e = iadr1_side_arg(side, &p.so)) return e;
    constexpr int SM1 = 16 * 64 * 17 * 4;
    static const int dev_cus_ = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, SM1);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<1, 16, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, SM1);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<1, 16, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, SM1);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<1, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SM1);
        return (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }();
    (void)dev_cus_;
    const int ncu = iadr1_decode_cus();      // the CUs the decode stream owns (runtime.hip): persistent grids are one block per such CU
    // more tiles than CUs, but the q and k tiles alone fit: the K-head blocks take their V tile along (one round of blocks instead of two)
    const int tiles = p.N / 16, rope_tiles = (Hq + Hkv) * (D / 16);
    if (tiles > ncu && rope_tiles <= ncu && (K % 64) == 0) {
        hipLaunchKernelGGL((gemm_skinny_kernel<1, 16, true>), dim3(rope_tiles, (M + 63) / 64, 1), dim3(1024), 2 * SM1, stream, p);
        return iadr1_check_launch("gemm_qkv_rope_kv_bf16");
    }
    if (M <= 16) hipLaunchKernelGGL((gemm_skinny_kernel<1, 16, false, 1>), dim3(p.N / 16, 1, 1), dim3(1024), SM1, stream, p);
    else if (M <= 32) hipLaunchKernelGGL((gemm_skinny_kernel<1, 16, false, 2>), dim3(p.N / 16, 1, 1), dim3(1024), SM1, stream, p);
    else hipLaunchKernelGGL((gemm_skinny_kernel<1, 16>), dim3(p.N / 16, (M + 63) / 64, 1), dim3(1024), SM1, stream, p);
    return iadr1_check_launch("gemm_qkv_rope_kv_bf16");
}

extern "C" int iadr1_pack_qkv_rope_bf16(const void* W, long long ldw, const void* bias, void* Wp, void* bias_p, int Hq, int Hkv, int D, int K, hipStream_t stream) {
    IADR1_REQUIRE(D == 128 && K > 0 && (K % 32) == 0 && (ldw % 8) == 0, "pack_qkv_rope: need D == 128, K %% 32 == 0 (D=%d K=%d)", D, K);
    const int N = (Hq + 2 * Hkv) * D;
    long long blocks = ((long long)N * K / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_qkv_rope_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)W, ldw, (const bf16_t*)bias, (bf16_t*)Wp, (bf16_t*)bias_p, Hq + Hkv, N, K);
    return iadr1_check_launch("pack_qkv_rope_bf16");
}

extern "C" int iadr1_pack_act_bf16(const void* X, long long ldx, void* Xp, int M, int K, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && K > 0 && (K % 32) == 0 && (ldx % 8) == 0, "pack_act: need K %% 32 == 0 (K=%d)", K);
    long long blocks = ((long long)((M + 63) & ~63) * K / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_act_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)X, ldx, (bf16_t*)Xp, M, K);
    return iadr1_check_launch("pack_act_bf16");
}

extern "C" int iadr1_pack_gateup_bf16(const void* W, long long ldw, void* Wp, int I, int K, hipStream_t stream) {
    IADR1_REQUIRE(I > 0 && K > 0 && (I % 64) == 0 && (K % 32) == 0 && (ldw % 8) == 0, "pack_gateup: need I %% 64 == 0, K %% 32 == 0 (I=%d K=%d)", I, K);
    long long blocks = ((long long)2 * I * K / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_gateup_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)W, ldw, (bf16_t*)Wp, I, K);
    return iadr1_check_launch("pack_gateup_bf16");
}

// Decode-time skinny GEMM on FP8 weights (include/iadr1_hip.h): the wide kernel, 64 output columns per block, K slices over grid.z.
extern "C" int iadr1_gemm_skinny_fp8w(const void* X, const void* Wp8, const float* wscale, void* Y, const void* bias, int M, int N, int K, long long ldx, long long ldy,
                                      int out_mode, int ksplit, hipStream_t stream) {
    IADR1_REQUIRE(M > 0 && N > 0 && K > 0 && wscale != nullptr, "gemm_skinny_fp8w: empty problem / no scales");
    IADR1_REQUIRE((K % 64) == 0 && (N % 64) == 0 && (ldx % 8) == 0, "gemm_skinny_fp8w: FP8-packed weights need K %% 64 == 0 and N %% 64 == 0 (K=%d N=%d)", K, N);
    IADR1_REQUIRE(ldy != 0 || (out_mode == 3 && (N % 64) == 0), "gemm_skinny_fp8w: a decode-packed output (ldy == 0) exists for the fused-SwiGLU mode only");
    IADR1_REQUIRE((((uintptr_t)X) & 15) == 0 && (((uintptr_t)Wp8) & 15) == 0, "gemm_skinny_fp8w: X/W must be 16-byte aligned");
    IADR1_REQUIRE(out_mode >= 0 && out_mode <= 3 && ksplit >= 1 && (ksplit == 1 || out_mode == 2), "gemm_skinny_fp8w: out_mode 0-3; ksplit > 1 needs out_mode 2 (partial slabs)");
    IADR1_REQUIRE(out_mode != 3 || (N % 128) == 0, "gemm_skinny_fp8w: fused SwiGLU needs N (= 2*I) to be a multiple of 128, got %d", N);
    SkinnyArgs p{};
    p.X = (const bf16_t*)X; p.W = (const bf16_t*)Wp8; p.wscale = wscale; p.Y = Y; p.bias = (const bf16_t*)bias; p.M = M; p.N = N; p.K = K;
    p.ldx = ldx; p.ldw = K; p.ldy = ldy; p.out_mode = out_mode;
    constexpr int SMW = 8 * 64 * 33 * 4, SMS = 8 * 64 * 17 * 4;
    static const int pers = iadr1_env_int("IADR1_SKINNY_PERS", 1);
    static const int dev_cus_ = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_wide_kernel<4, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMW);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_kernel<8, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMW);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_kernel<8, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMW);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_kernel<8, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMW);
        (void)hipFuncSetAttribute((const void*)gemm_skinny_pers_split_kernel<8, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SMS);
        return (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }();
    (void)dev_cus_;
    const int ncu = iadr1_decode_cus();      // the CUs the decode stream owns (runtime.hip): persistent grids are one block per such CU
    const int mz = (M + 63) / 64;
    // the persistent X-resident forms (same shape rules as the bf16 launcher): split-K slabs for the long-K down projection, one block per CU for the
    // big un-split streams (gate|up, lm_head).  The one-shot wide kernel takes everything else (K = 3584 of the 7B widths: the X fragments of a wave's
    // k-steps no longer fit in registers).
    {
        const int dst = K >> 6, per_z = (dst + ksplit - 1) / ksplit, bps = ncu / ksplit;
        if (pers && ksplit > 1 && out_mode == 2 && per_z <= 24 && bps >= 1 && (N >> 4) >= 4 * bps && (dst % ksplit == 0 || (ksplit - 1) * per_z < dst)) {
            hipLaunchKernelGGL((gemm_skinny_pers_split_kernel<8, 3, true>), dim3(bps * ksplit, mz, 1), dim3(512), SMS, stream, p, ksplit, bps);
            return iadr1_check_launch("gemm_skinny_fp8w");
        }
        const int ksw = K / 256;
        if (pers && ksplit == 1 && (N % 32) == 0 && (K % 256) == 0 && (ksw == 8 || ksw == 6 || ksw == 4) && (N / 32) >= 2 * ncu) {
            const dim3 grid(ncu, mz, 1), block(512);
            if (ksw == 8) hipLaunchKernelGGL((gemm_skinny_pers_kernel<8, 8, true>), grid, block, SMW, stream, p);
            else if (ksw == 6) hipLaunchKernelGGL((gemm_skinny_pers_kernel<8, 6, true>), grid, block, SMW, stream, p);
            else hipLaunchKernelGGL((gemm_skinny_pers_kernel<8, 4, true>), grid, block, SMW, stream, p);
            return iadr1_check_launch("gemm_skinny_fp8w");
        }
    }
    hipLaunchKernelGGL((gemm_skinny_wide_kernel<4, 8, true>), dim3(N / 64, mz, ksplit), dim3(512), SMW, stream, p);
    return iadr1_check_launch("gemm_skinny_fp8w");
}

extern "C" int iadr1_pack_weight_fp8(const void* W, long long ldw, void* Wp8, float* scale, int N, int K, int gateup_I, hipStream_t stream) {
    IADR1_REQUIRE(N > 0 && K > 0 && (N % 16) == 0 && (K % 64) == 0 && (ldw % 8) == 0, "pack_weight_fp8: need N %% 16 == 0, K %% 64 == 0 (N=%d K=%d)", N, K);
    IADR1_REQUIRE(gateup_I == 0 || (2 * gateup_I == N && (gateup_I % 64) == 0), "pack_weight_fp8: gateup_I must be N / 2 and a multiple of 64");
    IADR1_REQUIRE(scale != nullptr && (((uintptr_t)Wp8) & 15) == 0, "pack_weight_fp8: scale buffer / 16-byte aligned destination required");
    hipLaunchKernelGGL(fp8_row_scale_kernel, dim3(N), dim3(256), 0, stream, (const bf16_t*)W, ldw, scale, K);
    long long blocks = ((long long)N * K / 16 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_fp8_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)W, ldw, scale, (uint32_t*)Wp8, N, K, gateup_I);
    return iadr1_check_launch("pack_weight_fp8");
}

IADR1_STAMPS_EXPORT(gemm)
