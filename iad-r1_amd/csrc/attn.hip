// Flash-style attention for gfx950 (reference arithmetic
// TF:modeling_qwen2_5_vl.py:186-208 (softmax in fp32 over QK^T*d^-0.5 + mask, then PV), ViT segments
// :225-291 (non-causal inside cu_seqlens windows), decoder :641-689 (causal + left-pad key mask, GQA)).
//
// One family of kernels for both towers, parameterised by head dim D (128 text, 80 vision -> the QK^T
// contraction is zero-padded to 96) and by an explicit segment list (start,end) in the flat token axis:
// ViT windows / images are segments, decoder sequences are segments that skip their left padding.
//
// Dataflow (all MFMA 16x16x32 bf16, fp32 accumulate).  Everything is computed TRANSPOSED so that no
// cross-lane transpose of the probability tile is ever needed:
//   S^T[key,q] = K . Q^T          A = K rows from LDS (row-major tile), B = Q fragment held in VGPRs
//   P^T = softmax columns         a lane owns ONE q column: max/sum = in-lane + 2 shuffles (xor 16, 32)
//   O^T[d,q]  += V^T . P^T        A = V^T rows from LDS (tile stored transposed), B = P^T straight from
//                                 the S^T accumulators.  The K rows of S^T tiles are permuted
//                                 (row i of tile 2j <-> key 8*(i/4)+i%4, tile 2j+1 <-> +4) so that the
//                                 accumulator registers of a lane are 8 CONSECUTIVE keys = one B fragment.
// Backward = the same trick twice: bwd_dq owns q columns (dQ^T += K^T . dS^T), bwd_dkdv owns key
// columns (dV^T += dO^T . P, dK^T += Q^T . dS) and loops the GQA group so dK/dV need no atomics.
#include "common.h"
#include <stdlib.h>

namespace {

struct AttnArgs {
    const bf16_t* q;
    const bf16_t* k;
    const bf16_t* v;
    bf16_t* o;
    float* lse;             // [Hq, T] natural-log logsumexp of the scaled scores
    const int* seg_start;   // [nseg] first token (flat index) of each segment
    const int* seg_end;     // [nseg] one past the last token
    // Shared-prefix attention (the G completions of one prompt attend to the prompt's keys, which are computed ONCE):
    // NULL, or [nseg][4] = {prefix_start, prefix_len, child_first, child_count}.  A segment's keys are the token range
    // [prefix_start, +prefix_len) (all visible) followed by its own tokens (causal); segments [child_first, +child_count)
    // are the ones that use THIS segment as their prefix (their queries feed this segment's dK/dV).
    const int* seg_prefix;
    // Chunked teacher-forced forward (iadr1_attn_fwd_chunk): NULL, or [nseg][4] = {q_first, q_count, blk_log2, blk_stride}.  Only the query rows
    // [q_first, +q_count) of the segment are computed (its earlier rows were computed by earlier calls; keys are all rows [0, seg_end - seg_start)), and logical
    // row j of the segment lives at flat row  seg_start + (j >> blk_log2) * blk_stride + (j & (2^blk_log2 - 1))  -- the time-blocked completion layout in which
    // the rows that ALL sequences produce during the same 2^blk_log2 decode steps are contiguous (blk_log2 = 31: the plain contiguous segment).
    const int* seg_view;
    long long ldq, ldk, ldv, ldo;  // token row strides (elements); head h lives at column h*D
    int T, Hq, Hkv;
    int causal;
    float scale;
    // backward only
    const bf16_t* dout;
    const float* delta;     // [Hq, T] rowsum(dO * O)
    bf16_t* dq;
    bf16_t* dk;
    bf16_t* dv;
    long long lddo, lddq, lddk, lddv;
    // dK/dV with the q heads of a GQA group split over `hs` blocks (balances the long query loops of shared-prefix segments):
    // fp32 partials dkv_ws[hs][T][Hkv][2][D], summed in a fixed order by attn_dkdv_reduce_kernel (deterministic, no atomics)
    float* dkv_ws;
    int hs;
    int seg_off;            // first segment of this launch (the launchers split the segment list into a long-segment head and a short-segment tail)
};

template <int D>
struct Cfg {
    static constexpr int DQK = (D + 31) / 32 * 32;  // contraction length of QK^T, zero padded
    static constexpr int KS = DQK / 32;             // MFMA k-steps over d
    static constexpr int DT = D / 16;               // 16-wide output tiles over d
    static constexpr int LD = 128;                  // 256-byte tile rows = exactly one LDS bank row, 16 chunks of 16 B (swizzled, see tile_off)
};

// LDS image of a staged [rows][d] tile.  Row r is 256 bytes; its 16-byte chunk c sits at chunk position  c ^ 2*rho(r),  rho(r) = (r & 3) | ((r >> 1) & 4).
// Both read patterns of these kernels are then bank-conflict free (measured before, on 272-byte padded rows: SQ_LDS_BANK_CONFLICT = 39-43 % of
// SQ_LDS_IDX_ACTIVE in all three kernels, profiles/r02_mfma_busy.json):
//  * fragment reads (ds_read_b128; lane (li, g) reads chunk 4*ks + g of row perm_row(t, li)): the hardware serves lanes {li 0-3, 12-15 at g} with
//    {li 4-11 at g^1} together; their rows are {0-3, 24-27} and {8-11, 16-19} (+4 for odd t, +32 per tile pair), rho enumerates each set 0..7, so
//    the first set lands on the even chunk positions (relative to 4*ks + g) and the second, one chunk further, on the odd ones;
//  * transpose reads (ds_read_b64_tr_b16; a 32-lane half reads 8-byte halves of chunks 2*dt, 2*dt+1 of rows (li>>2) + 8*(g&1) [+4]): the rows differ
//    in bits {0, 1, 3}, exactly the bits of rho, so the eight rows use eight different chunk pairs = all 64 banks once.
// (the padded 272-byte-row form this replaced -- conflicts 0.40 -> 0.00, 1-5 % on the kernels -- is in profiles/EXPERIMENTS.md rounds 1-2; its compile-time switch was removed in round 6)
__device__ __forceinline__ int swz_rho(int r) { return (r & 3) | ((r >> 1) & 4); }
__device__ __forceinline__ int tile_off(int r, int c, int /*LD*/) { return r * 256 + ((c ^ (2 * swz_rho(r))) << 4); }    // byte offset of chunk c of row r

__device__ __forceinline__ bf16x8_t ld_frag_g(const bf16_t* p, bool ok) {
    u32x4_t v = {0, 0, 0, 0};
    if (ok) v = *(const u32x4_t*)p;
    return __builtin_bit_cast(bf16x8_t, v);
}
__device__ __forceinline__ bf16x8_t ld_frag_s(const bf16_t* p) { return *(const bf16x8_t*)p; }
// fragment (row, 16-byte chunk) of a staged tile
template <int LD>
__device__ __forceinline__ bf16x8_t ld_frag_t(const bf16_t* tile, int row, int chunk) { return *(const bf16x8_t*)((const char*)tile + tile_off(row, chunk, LD)); }

// v_exp_f32 directly: exp2f() wraps it in a denormal-range rescale (~5 extra VALU instructions per call); softmax terms that small are zero anyway
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ bf16x8_t pack_frag(const f32x4_t a, const f32x4_t b) {
    u32x4_t v = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    return __builtin_bit_cast(bf16x8_t, v);
}

// key (or q) row that MFMA-tile row i of 16-row tile `t` stands for, inside a 64/32-row block
__device__ __forceinline__ int perm_row(int t, int i) { return (t >> 1) * 32 + (i >> 2) * 8 + (i & 3) + 4 * (t & 1); }

// Stage `rows` token rows x D columns (global, row stride ld) into LDS row-major [rows][LD], zero filling
// rows >= nvalid and columns >= D.
template <int D, int ROWS>
__device__ __forceinline__ void stage_rows(bf16_t* dst, const bf16_t* src, long long ld, int nvalid) {
    constexpr int CPR = Cfg<D>::DQK / 8;
    for (int idx = threadIdx.x; idx < ROWS * CPR; idx += 256) {
        const int r = idx / CPR, c = idx - r * CPR;
        u32x4_t v = {0, 0, 0, 0};
        if (r < nvalid && c * 8 < D) v = *(const u32x4_t*)(src + (long long)r * ld + c * 8);
        *(u32x4_t*)((char*)dst + tile_off(r, c, Cfg<D>::LD)) = v;
    }
}
// Split staging (issue-early / write-late): the global loads of the NEXT tile are issued into registers before the
// MFMAs of the current tile and written to LDS after the barrier that ends it, so their latency hides under compute.
template <int D, int ROWS, int NT = 256>
struct TileRegs {
    static constexpr int CPR = Cfg<D>::DQK / 8;
    static constexpr int N = (ROWS * CPR + NT - 1) / NT;
    u32x4_t v[N];
    __device__ __forceinline__ void load(const bf16_t* src, long long ld, int nvalid) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = i * NT + threadIdx.x;
            const int r = idx / CPR, c = idx - r * CPR;
            v[i] = (u32x4_t){0, 0, 0, 0};
            if (idx < ROWS * CPR && r < nvalid && c * 8 < D) v[i] = *(const u32x4_t*)(src + (long long)r * ld + c * 8);
        }
    }
    // rows j0 .. j0 + ROWS of a segment whose logical row j lives at  seg_base + ((j >> blk) * bstride + (j & (2^blk - 1))) * ld  (AttnArgs::seg_view)
    __device__ __forceinline__ void load_blk(const bf16_t* seg_base, long long ld, int nvalid, int j0, int blk, int bstride) {
        const int bmask = (int)((1u << blk) - 1u);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = i * NT + threadIdx.x;
            const int r = idx / CPR, c = idx - r * CPR;
            const int j = j0 + r;
            v[i] = (u32x4_t){0, 0, 0, 0};
            if (idx < ROWS * CPR && r < nvalid && c * 8 < D) v[i] = *(const u32x4_t*)(seg_base + (long long)((j >> blk) * bstride + (j & bmask)) * ld + c * 8);
        }
    }
    __device__ __forceinline__ void store(bf16_t* dst) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = i * NT + threadIdx.x;
            const int r = idx / CPR, c = idx - r * CPR;
            if (idx < ROWS * CPR) *(u32x4_t*)((char*)dst + tile_off(r, c, Cfg<D>::LD)) = v[i];
        }
    }
};

// Same tile stored TRANSPOSED: dst[d][r], row stride LDT (elements), d in [0, DQK)
template <int D, int ROWS, int LDT>
__device__ __forceinline__ void stage_rows_t(bf16_t* dst, const bf16_t* src, long long ld, int nvalid) {
    constexpr int CPR = Cfg<D>::DQK / 8;
    for (int idx = threadIdx.x; idx < ROWS * CPR; idx += 256) {
        // lanes walk rows fastest inside groups of 16 so that a wave's LDS writes hit 16 consecutive columns
        const int r = (idx & 15) + (idx / (16 * CPR)) * 16, c = (idx >> 4) % CPR;
        u32x4_t v = {0, 0, 0, 0};
        if (r < nvalid && c * 8 < D) v = *(const u32x4_t*)(src + (long long)r * ld + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            dst[(c * 8 + 2 * e) * LDT + r] = (bf16_t)(v[e] & 0xffffu);
            dst[(c * 8 + 2 * e + 1) * LDT + r] = (bf16_t)(v[e] >> 16);
        }
    }
}

constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// hardware transpose reads: tr_frag_ld (common.h)
// lane-constant part of a tr-read address inside a row-major [rows][LD] tile: row (li>>2) + 8*g, column 4*(li&3)
__device__ __forceinline__ uint32_t tr_lane_off(int li, int g, int LD) { return (uint32_t)(((g * 8 + (li >> 2)) * LD + 4 * (li & 3)) * 2); }
// byte offset of d-tile dt (16 columns = chunks 2*dt, 2*dt+1) relative to tr_lane_off, for THIS lane's rows (swizzled image: the chunk pair moves with rho of the row)
__device__ __forceinline__ uint32_t tr_dt_off(int dt, int li, int g) {
    return (uint32_t)((dt ^ ((li >> 2) | ((g & 1) << 2))) << 5);
}

// =====================================================================================================
// forward
// =====================================================================================================
template <int D, int R>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
    using C = Cfg<D>;
    constexpr int BM = 64 * R, BN = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = (bf16_t*)smem;          // [BN][LD]
    bf16_t* Vs = Ks + BN * C::LD;        // [BN][LD] row-major; consumed through transpose reads

    const int seg = blockIdx.y + p.seg_off, head = blockIdx.z, kvh = head / (p.Hq / p.Hkv);
    const int s0 = p.seg_start[seg], slen = p.seg_end[seg] - s0;
    // segment view (chunked forward): query sub-range + blocked row map; the plain call is {0, slen, 31, 0}
    const int* sv = p.seg_view ? p.seg_view + seg * 4 : nullptr;
    const int qfirst = sv ? sv[0] : 0, qend = sv ? min(slen, sv[0] + sv[1]) : slen;
    const int blk = sv ? sv[2] : 31, bstride = sv ? sv[3] : 0, bmask = (int)((1u << blk) - 1u);
    auto row_of = [&](int j) { return s0 + (j >> blk) * bstride + (j & bmask); };
    const int ntile = (qend - qfirst + BM - 1) / BM;
    const int qt = (int)gridDim.x - 1 - (int)blockIdx.x;  // heaviest (last) causal tiles first
    if (qt >= ntile) return;
    const int q0 = qfirst + qt * BM;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, g = l >> 4;

    bf16x8_t qf[R][C::KS];
    int qrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        qrow[r] = q0 + (w * R + r) * 16 + li;
        const bf16_t* src = p.q + (long long)row_of(qrow[r]) * p.ldq + head * D;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int d0 = ks * 32 + g * 8;
            qf[r][ks] = ld_frag_g(src + d0, qrow[r] < qend && d0 < D);
        }
    }
    float m[R], lsum[R];
    f32x4_t acc[C::DT][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        m[r] = -INFINITY;
        lsum[r] = 0.f;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) acc[dt][r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const float c = p.scale * LOG2E;
    const int ps0 = p.seg_prefix ? p.seg_prefix[seg * 4] : 0, plen = p.seg_prefix ? p.seg_prefix[seg * 4 + 1] : 0;

    TileRegs<D, BN> kreg, vreg;
    // One continuous tile stream over the two key ranges -- the shared prefix (every key visible) then the segment's own tokens
    // (causal) -- so the register prefetch of the next tile also runs across the range boundary (a restart there costs a full
    // global-load latency per block, as much as several tiles of work).
    const int np_tiles = (plen + BN - 1) / BN;
    const int own_end = p.causal ? min(slen, q0 + BM) : slen;
    const int nt = np_tiles + (own_end + BN - 1) / BN;
    auto tile_load = [&](int t) {
        const bool pre = t < np_tiles;
        const int kv0 = (pre ? t : t - np_tiles) * BN;
        const int valid = (pre ? plen : slen) - kv0;
        // (the shared prefix is always a plain contiguous range; own rows go through the segment's row map)
        kreg.load_blk(p.k + (long long)(pre ? ps0 : s0) * p.ldk + kvh * D, p.ldk, valid, kv0, pre ? 31 : blk, pre ? 0 : bstride);
        vreg.load_blk(p.v + (long long)(pre ? ps0 : s0) * p.ldv + kvh * D, p.ldv, valid, kv0, pre ? 31 : blk, pre ? 0 : bstride);
    };
    if (nt > 0) tile_load(0);
    for (int t = 0; t < nt; ++t) {
        const bool pre = t < np_tiles;
        const int kv0 = (pre ? t : t - np_tiles) * BN;
        const int klen = pre ? plen : slen;
        const bool kcausal = !pre && p.causal;
        __syncthreads();           // every wave is done reading the previous tile
        kreg.store(Ks);
        vreg.store(Vs);
        __syncthreads();
        if (t + 1 < nt) tile_load(t + 1);   // next tile's loads fly while this tile is computed

        f32x4_t s[4][R];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < R; ++r) s[kt][r] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            bf16x8_t kf[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) kf[kt] = ld_frag_t<C::LD>(Ks, perm_row(kt, li), ks * 4 + g);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < R; ++r) s[kt][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[r][ks], s[kt][r], 0, 0, 0);
        }
        // lane (g, e) of tile kt holds key kv0 + (kt>>1)*32 + g*8 + (kt&1)*4 + e for q column qrow[r]
        const bool need_mask = (kv0 + BN > klen) || (kcausal && kv0 + BN > q0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = s[kt][r][e];
                    if (need_mask) {
                        const int key = kv0 + (kt >> 1) * 32 + g * 8 + (kt & 1) * 4 + e;
                        if (key >= klen || (kcausal && key > qrow[r])) t = -INFINITY;
                    }
                    s[kt][r][e] = t;
                    mx = fmaxf(mx, t);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
            mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
            const float mn = fmaxf(m[r], mx * c);     // the running max lives in scaled (log2) units; the scale rides in the fma below
            const float alpha = (mn == -INFINITY) ? 1.f : fast_exp2(m[r] - mn);
            const float mref = (mn == -INFINITY) ? 0.f : mn;
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = fast_exp2(__builtin_fmaf(s[kt][r][e], c, -mref));
                    s[kt][r][e] = pv;
                    ps += pv;
                }
            lsum[r] = lsum[r] * alpha + ps;
            m[r] = mn;
            // after the first tiles the running max rarely moves: skip the rescale of the 32 accumulator registers (they live in AGPRs, so each
            // multiply is a read + multiply + write) when no lane of the wave needs it -- multiplying by 1 is the identity, the result is unchanged
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) acc[dt][r] *= alpha;
            }
        }
        bf16x8_t pf[R][2];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < 2; ++j) pf[r][j] = pack_frag(s[2 * j][r], s[2 * j + 1][r]);
        {
            // O^T[d,q] += V^T . P^T with V^T fragments produced by transpose reads of the row-major V tile;
            // reads of d-tile dt+1 are in flight while the MFMAs of d-tile dt run
            lds_char_t* vbase = lds_ptr(Vs) + tr_lane_off(li, g, C::LD);
            // the reads of d-tile dt+1 are issued before the MFMAs of d-tile dt
            bf16x8_t vf[2][2];
            auto fetch = [&](bf16x8_t (&dst)[2], int dt) {
#pragma unroll
                for (int j = 0; j < 2; ++j) dst[j] = tr_frag_ld(vbase + (j * 32) * C::LD * 2 + tr_dt_off(dt, li, g), vbase + (j * 32 + 4) * C::LD * 2 + tr_dt_off(dt, li, g));
            };
            fetch(vf[0], 0);
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                if (dt + 1 < C::DT) fetch(vf[(dt + 1) & 1], dt + 1);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[dt][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt & 1][j], pf[r][j], acc[dt][r], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float lt = lsum[r];
        lt += __shfl_xor(lt, 16, WAVE);
        lt += __shfl_xor(lt, 32, WAVE);
        if (qrow[r] >= qend) continue;
        const float inv = lt > 0.f ? 1.f / lt : 0.f;
        const int orow = row_of(qrow[r]);
        bf16_t* dst = p.o + (long long)orow * p.ldo + head * D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const f32x4_t a = acc[dt][r];
            *(u32x2_t*)(dst + dt * 16 + g * 4) = (u32x2_t){pack2bf(a[0] * inv, a[1] * inv), pack2bf(a[2] * inv, a[3] * inv)};
        }
        if (g == 0 && p.lse) p.lse[(long long)head * p.T + orow] = (lt > 0.f) ? (m[r] + log2f(lt)) * LN2 : -INFINITY;
    }
}

// =====================================================================================================
// backward, part 0: delta[h][t] = sum_d dO[t,h,d] * O[t,h,d]
// =====================================================================================================
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* o, long long ldo, const bf16_t* dout, long long lddo, float* delta,
                                                         int T, int Hq) {
    // 16 lanes per (token, head): each lane covers D/16 elements
    const long long item = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (item >= (long long)T * Hq) return;
    const long long t = item / Hq;
    const int h = (int)(item % Hq);
    float s = 0.f;
    for (int d = sub * 8; d < D; d += 128) {
        const u32x4_t a = *(const u32x4_t*)(o + t * ldo + h * D + d), b = *(const u32x4_t*)(dout + t * lddo + h * D + d);
#pragma unroll
        for (int e = 0; e < 4; ++e) s += lo_bf(a[e]) * lo_bf(b[e]) + hi_bf(a[e]) * hi_bf(b[e]);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
    if (sub == 0) delta[(long long)h * T + t] = s;
}

// =====================================================================================================
// backward, part 1: dQ.  Block = 64 q rows of one (segment, q head); wave = 16 q columns.
// =====================================================================================================
template <int D, int R>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs p) {
    // R = 16-row q groups per wave.  The kernel reads every staged K / V fragment from LDS once per q group it feeds: with R = 1 a block pulls
    // 64 KB of fragments through the LDS port per 32-key tile against 24 MFMAs per wave (LDS-bound ~2.7x with two blocks per CU); R = 2 halves
    // the LDS bytes per MFMA.
    using C = Cfg<D>;
    constexpr int BM = 64 * R, KB = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Ks = (bf16_t*)smem;       // [KB][LD]  (also read transposed for dQ^T += K^T . dS^T)
    bf16_t* Vs = Ks + KB * C::LD;     // [KB][LD]

    const int seg = blockIdx.y + p.seg_off, head = blockIdx.z, kvh = head / (p.Hq / p.Hkv);
    const int s0 = p.seg_start[seg], slen = p.seg_end[seg] - s0;
    const int ntile = (slen + BM - 1) / BM;
    const int qt = (int)gridDim.x - 1 - (int)blockIdx.x;
    if (qt >= ntile) return;
    const int q0 = qt * BM;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, g = l >> 4;
    int qrow[R];
    bool qok[R];
    bf16x8_t qf[R][C::KS], dof[R][C::KS];
    float lse2[R], dl[R];
    const float c = p.scale * LOG2E;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        qrow[r] = q0 + (w * R + r) * 16 + li;
        qok[r] = qrow[r] < slen;
        const bf16_t* qs = p.q + (long long)(s0 + qrow[r]) * p.ldq + head * D;
        const bf16_t* ds = p.dout + (long long)(s0 + qrow[r]) * p.lddo + head * D;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int d0 = ks * 32 + g * 8;
            qf[r][ks] = ld_frag_g(qs + d0, qok[r] && d0 < D);
            dof[r][ks] = ld_frag_g(ds + d0, qok[r] && d0 < D);
        }
        lse2[r] = qok[r] ? p.lse[(long long)head * p.T + s0 + qrow[r]] * LOG2E : 0.f;
        dl[r] = qok[r] ? p.delta[(long long)head * p.T + s0 + qrow[r]] : 0.f;
    }

    f32x4_t acc[R][C::DT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) acc[r][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int ps0 = p.seg_prefix ? p.seg_prefix[seg * 4] : 0, plen = p.seg_prefix ? p.seg_prefix[seg * 4 + 1] : 0;
    TileRegs<D, KB> kreg, vreg;
    // one continuous tile stream over [shared prefix (all visible)] ++ [own tokens (causal)], see attn_fwd_kernel
    const int np_tiles = (plen + KB - 1) / KB;
    const int own_end = p.causal ? min(slen, q0 + BM) : slen;
    const int nt = np_tiles + (own_end + KB - 1) / KB;
    auto tile_load = [&](int t) {
        const bool pre = t < np_tiles;
        const int kv0 = (pre ? t : t - np_tiles) * KB;
        const int row = (pre ? ps0 : s0) + kv0, valid = (pre ? plen : slen) - kv0;
        kreg.load(p.k + (long long)row * p.ldk + kvh * D, p.ldk, valid);
        vreg.load(p.v + (long long)row * p.ldv + kvh * D, p.ldv, valid);
    };
    if (nt > 0) tile_load(0);
    for (int t = 0; t < nt; ++t) {
        const bool pre = t < np_tiles;
        const int kv0 = (pre ? t : t - np_tiles) * KB;
        const int klen = pre ? plen : slen;
        const bool kcausal = !pre && p.causal;
        __syncthreads();
        kreg.store(Ks);
        vreg.store(Vs);
        __syncthreads();
        if (t + 1 < nt) tile_load(t + 1);

        f32x4_t st[R][2], dpt[R][2];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { st[r][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dpt[r][kt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int row = perm_row(kt, li);
                const bf16x8_t kfr = ld_frag_t<C::LD>(Ks, row, ks * 4 + g), vfr = ld_frag_t<C::LD>(Vs, row, ks * 4 + g);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    st[r][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[r][ks], st[r][kt], 0, 0, 0);
                    dpt[r][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, dof[r][ks], dpt[r][kt], 0, 0, 0);
                }
            }
        bf16x8_t dsf[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            f32x4_t ds[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kv0 + g * 8 + kt * 4 + e;
                    const bool ok = qok[r] && key < klen && !(kcausal && key > qrow[r]);
                    const float pv = ok ? fast_exp2(__builtin_fmaf(st[r][kt][e], c, -lse2[r])) : 0.f;
                    ds[kt][e] = pv * (dpt[r][kt][e] - dl[r]);
                }
            dsf[r] = pack_frag(ds[0], ds[1]);
        }
        {
            lds_char_t* kbase = lds_ptr(Ks) + tr_lane_off(li, g, C::LD);
            bf16x8_t kf[2];
            kf[0] = tr_frag_ld(kbase + tr_dt_off(0, li, g), kbase + 4 * C::LD * 2 + tr_dt_off(0, li, g));
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                if (dt + 1 < C::DT) kf[(dt + 1) & 1] = tr_frag_ld(kbase + tr_dt_off(dt + 1, li, g), kbase + 4 * C::LD * 2 + tr_dt_off(dt + 1, li, g));
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[dt & 1], dsf[r], acc[r][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!qok[r]) continue;
        bf16_t* dst = p.dq + (long long)(s0 + qrow[r]) * p.lddq + head * D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const f32x4_t a = acc[r][dt] * p.scale;
            *(u32x2_t*)(dst + dt * 16 + g * 4) = (u32x2_t){pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
        }
    }
}

// =====================================================================================================
// backward, part 2: dK, dV.  Block = 64 keys of one (segment, kv head); wave = 16 key columns; loops
// over the q heads of the GQA group and over 32-row q blocks.
// =====================================================================================================
template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_bwd_dkdv_kernel(AttnArgs p) {
    // WAVES x 16 key columns per block.  8 waves (128 keys): every staged q block feeds twice as many key columns, the block count halves, and two
    // waves per SIMD overlap each other's LDS-read / MFMA chains (one 4-wave block per CU ran its phases back to back: 2.2 us per 32-row q block
    // for 0.27 us of MFMA work).
    using C = Cfg<D>;
    constexpr int BN = 16 * WAVES, QB = 32, NT = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* Qs = (bf16_t*)smem;        // [QB][LD]  (also read transposed: dK^T += Q^T . dS)
    bf16_t* dOs = Qs + QB * C::LD;     // [QB][LD]  (also read transposed: dV^T += dO^T . P)
    float* lse_s = (float*)(dOs + QB * C::LD);    // [QB]
    float* del_s = lse_s + QB;                    // [QB]

    // grid = (key tile, kv head x head split, segment): the segment is the SLOWEST dimension of the dispatch order.  Shared-prefix segments (the prompts,
    // first in the segment list) carry ~15x the iterations of a completion segment (their query loop runs over all G children); dispatched in
    // (tile, segment, head) order the eight head slices of the prompts were spread over the whole launch and the last ones started 515 us into
    // an 837 us kernel although every one of them fits on the chip at once (stamps: tools/attn_stamps.py).
    const int hs = p.hs, kvh = blockIdx.y / hs, part = blockIdx.y % hs;
    const int seg = blockIdx.z + p.seg_off, group = p.Hq / p.Hkv, gph = group / hs;   // gph q heads per block
    const int s0 = p.seg_start[seg], slen = p.seg_end[seg] - s0;
    const int ntile = (slen + BN - 1) / BN;
    const int kt0 = blockIdx.x;  // causal: low kv tiles see the most q rows and are scheduled first
    if (kt0 >= ntile) return;
    const int kv0 = kt0 * BN;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, g = l >> 4;
    const int key = kv0 + w * 16 + li;
    const bool kok = key < slen;

    bf16x8_t kf[C::KS], vf[C::KS];
    {
        const bf16_t* ks_ = p.k + (long long)(s0 + key) * p.ldk + kvh * D;
        const bf16_t* vs_ = p.v + (long long)(s0 + key) * p.ldv + kvh * D;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) {
            const int d0 = ks * 32 + g * 8;
            kf[ks] = ld_frag_g(ks_ + d0, kok && d0 < D);
            vf[ks] = ld_frag_g(vs_ + d0, kok && d0 < D);
        }
    }
    const float c = p.scale * LOG2E;
    f32x4_t dkacc[C::DT], dvacc[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) { dkacc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dvacc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

    // q sources of this key block, per q head of the GQA group: the segment's own rows (causal: from q_begin on), then -- when
    // this segment is the shared prefix of others -- every row of each child segment (no mask: a child sees the whole prefix).
    const int q_begin = p.causal ? (kv0 / QB) * QB : 0;
    const int nq_own = (slen - q_begin + QB - 1) / QB;
    const int child_first = p.seg_prefix ? p.seg_prefix[seg * 4 + 2] : 0, child_count = p.seg_prefix ? p.seg_prefix[seg * 4 + 3] : 0;
    int nq = nq_own;
    for (int cidx = 0; cidx < child_count; ++cidx) nq += (p.seg_end[child_first + cidx] - p.seg_start[child_first + cidx] + QB - 1) / QB;
    const int nit = gph * nq;                              // flattened (head, source, q block) iteration space
    // TWO q blocks are in flight: the loads of block it+2 are issued while block it is computed (one block ahead left every iteration waiting for
    // its loads: 2.16 us per iteration against 0.27 us of MFMA work, tools/attn_stamps.py).  Register sets A / B alternate; the loop is unrolled by two
    // so that every access to them is static.
    TileRegs<D, QB, NT> qregA, doregA, qregB, doregB;
    float lse_rA = 0.f, del_rA = 0.f, lse_rB = 0.f, del_rB = 0.f;
    // cursor of the NEXT block to prefetch
    int c_head = part * gph, c_src = nq_own > 0 ? -1 : 0, c_qb = 0;
    int rowsA = 0, relA = 0, rowsB = 0, relB = 0;
    auto prefetch = [&](TileRegs<D, QB, NT>& qreg, TileRegs<D, QB, NT>& doreg, float& lse_r, float& del_r, int& o_rows, int& o_rel) {
        int base, rows, rel;
        if (c_src < 0) {
            const int q0 = q_begin + c_qb * QB;
            base = s0 + q0; rows = slen - q0; rel = q0;
            if (++c_qb >= nq_own) { c_qb = 0; c_src = 0; }
        } else {
            const int cs = p.seg_start[child_first + c_src], clen = p.seg_end[child_first + c_src] - cs;
            base = cs + c_qb * QB; rows = clen - c_qb * QB; rel = 1 << 30;
            if (++c_qb >= (clen + QB - 1) / QB) { c_qb = 0; ++c_src; }
        }
        const int head = kvh * group + c_head;
        if (c_src >= child_count) { ++c_head; c_src = nq_own > 0 ? -1 : 0; c_qb = 0; }
        qreg.load(p.q + (long long)base * p.ldq + head * D, p.ldq, rows);
        doreg.load(p.dout + (long long)base * p.lddo + head * D, p.lddo, rows);
        if (threadIdx.x < QB) {
            const bool okr = (int)threadIdx.x < rows;
            lse_r = okr ? p.lse[(long long)head * p.T + base + threadIdx.x] * LOG2E : 0.f;
            del_r = okr ? p.delta[(long long)head * p.T + base + threadIdx.x] : 0.f;
        }
        o_rows = rows; o_rel = rel;
    };
#ifdef IADR1_STAMPS
    long long ph[4] = {0, 0, 0, 0}, tph = 0;
#define PH_BEGIN() tph = wall_clock64()
#define PH_END(k) do { const long long now_ = wall_clock64(); ph[k] += now_ - tph; tph = now_; } while (0)
#else
#define PH_BEGIN() do { } while (0)
#define PH_END(k) do { } while (0)
#endif
    auto compute = [&](const TileRegs<D, QB, NT>& qreg, const TileRegs<D, QB, NT>& doreg, float lse_r, float del_r, int cur_rows, int cur_rel) {
        PH_BEGIN();
        __syncthreads();
        qreg.store(Qs);
        doreg.store(dOs);
        if (threadIdx.x < QB) { lse_s[threadIdx.x] = lse_r; del_s[threadIdx.x] = del_r; }
        __syncthreads();
        PH_END(0);
    };
    auto mma = [&](int cur_rows, int cur_rel) {
        // S[q, key] and dP[q, key]: A = Q / dO rows (permuted so a lane's registers are 8 consecutive q), B = K / V fragments
        f32x4_t s[2], dp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { s[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dp[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = perm_row(t, li);
                s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_t<C::LD>(Qs, row, ks * 4 + g), kf[ks], s[t], 0, 0, 0);
                dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag_t<C::LD>(dOs, row, ks * 4 + g), vf[ks], dp[t], 0, 0, 0);
            }
#ifdef IADR1_STAMPS
        asm volatile("" :: "v"(s[0][0]), "v"(dp[1][3]));
        PH_END(1);
#endif
        // lane (g, e) of tile t holds q = q0 + g*8 + t*4 + e for key column `key`
        f32x4_t pv[2], ds[2];
        // the q block is fully visible to every key of this block (all rows valid, no causal cut): the common case -- every block of a child
        // segment and every own block below the diagonal -- needs no per-element predicate.  Block-uniform, so no divergence.
        const bool all_visible = cur_rows >= QB && (!p.causal || cur_rel >= kv0 + BN - 1);
        f32x4_t l4[2], d4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { l4[t] = *(const f32x4_t*)(lse_s + g * 8 + t * 4); d4[t] = *(const f32x4_t*)(del_s + g * 8 + t * 4); }
        if (all_visible) {
            const float km = kok ? 1.f : 0.f;     // key columns past the segment end contribute nothing
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pp = km * fast_exp2(__builtin_fmaf(s[t][e], c, -l4[t][e]));
                    pv[t][e] = pp;
                    ds[t][e] = pp * (dp[t][e] - d4[t][e]);
                }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qi = g * 8 + t * 4 + e;
                    const bool ok = kok && qi < cur_rows && !(p.causal && key > cur_rel + qi);
                    const float pp = ok ? fast_exp2(__builtin_fmaf(s[t][e], c, -l4[t][e])) : 0.f;
                    pv[t][e] = pp;
                    ds[t][e] = pp * (dp[t][e] - d4[t][e]);
                }
        }
        const bf16x8_t pf = pack_frag(pv[0], pv[1]), dsf = pack_frag(ds[0], ds[1]);
#ifdef IADR1_STAMPS
        asm volatile("" :: "v"(pf), "v"(dsf));
        PH_END(2);
#endif
        const uint32_t off = tr_lane_off(li, g, C::LD);
        lds_char_t* qbase = lds_ptr(Qs) + off;
        lds_char_t* dobase = lds_ptr(dOs) + off;
        bf16x8_t dof[2], qf2[2];
        dof[0] = tr_frag_ld(dobase + tr_dt_off(0, li, g), dobase + 4 * C::LD * 2 + tr_dt_off(0, li, g));
        qf2[0] = tr_frag_ld(qbase + tr_dt_off(0, li, g), qbase + 4 * C::LD * 2 + tr_dt_off(0, li, g));
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            if (dt + 1 < C::DT) {
                dof[(dt + 1) & 1] = tr_frag_ld(dobase + tr_dt_off(dt + 1, li, g), dobase + 4 * C::LD * 2 + tr_dt_off(dt + 1, li, g));
                qf2[(dt + 1) & 1] = tr_frag_ld(qbase + tr_dt_off(dt + 1, li, g), qbase + 4 * C::LD * 2 + tr_dt_off(dt + 1, li, g));
            }
            dvacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[dt & 1], pf, dvacc[dt], 0, 0, 0);
            dkacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf2[dt & 1], dsf, dkacc[dt], 0, 0, 0);
        }
#ifdef IADR1_STAMPS
        asm volatile("" :: "v"(dvacc[C::DT - 1][0]), "v"(dkacc[C::DT - 1][3]));
        PH_END(3);
#endif
    };
    if (nit > 0) prefetch(qregA, doregA, lse_rA, del_rA, rowsA, relA);
    if (nit > 1) prefetch(qregB, doregB, lse_rB, del_rB, rowsB, relB);
#ifdef IADR1_STAMPS
    if (threadIdx.x == 0) g_stamps[6][(blockIdx.x + (blockIdx.y + blockIdx.z * gridDim.y) * gridDim.x) & 4095] = (unsigned long long)nit;
    if (threadIdx.x == 0) g_stamps[4][(blockIdx.x + (blockIdx.y + blockIdx.z * gridDim.y) * gridDim.x) & 4095] = wall_clock64();
#endif
    for (int it = 0; it < nit; it += 2) {
        {
            const int cr = rowsA, cl = relA;
            compute(qregA, doregA, lse_rA, del_rA, cr, cl);          // block `it` -> LDS (its registers are free again)
            if (it + 2 < nit) prefetch(qregA, doregA, lse_rA, del_rA, rowsA, relA);
            mma(cr, cl);
        }
        if (it + 1 < nit) {
            const int cr = rowsB, cl = relB;
            compute(qregB, doregB, lse_rB, del_rB, cr, cl);
            if (it + 3 < nit) prefetch(qregB, doregB, lse_rB, del_rB, rowsB, relB);
            mma(cr, cl);
        }
    }
#ifdef IADR1_STAMPS
    if (threadIdx.x == 0) {
        const int sl_ = (blockIdx.x + (blockIdx.y + blockIdx.z * gridDim.y) * gridDim.x) & 4095;
        g_stamps[5][sl_] = wall_clock64();
        g_stamps[0][sl_] = ph[0]; g_stamps[1][sl_] = ph[1]; g_stamps[2][sl_] = ph[2]; g_stamps[3][sl_] = ph[3];
    }
#endif
    if (!kok) return;
    if (hs > 1) {
        float* wd = p.dkv_ws + ((((long long)part * p.T + s0 + key) * p.Hkv + kvh) * 2) * D;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            *(f32x4_t*)(wd + dt * 16 + g * 4) = dkacc[dt] * p.scale;
            *(f32x4_t*)(wd + D + dt * 16 + g * 4) = dvacc[dt];
        }
        return;
    }
    bf16_t* dkd = p.dk + (long long)(s0 + key) * p.lddk + kvh * D;
    bf16_t* dvd = p.dv + (long long)(s0 + key) * p.lddv + kvh * D;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) {
        const f32x4_t a = dkacc[dt] * p.scale, b = dvacc[dt];
        *(u32x2_t*)(dkd + dt * 16 + g * 4) = (u32x2_t){pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
        *(u32x2_t*)(dvd + dt * 16 + g * 4) = (u32x2_t){pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    }
}

// sum of the `hs` head-split partials of dK/dV (fixed order) -> bf16; one thread per 4 consecutive d of one (token, kv head, k|v)
template <int D>
__global__ __launch_bounds__(256) void attn_dkdv_reduce_kernel(AttnArgs p) {
    const int seg = blockIdx.y + p.seg_off;
    const int s0 = p.seg_start[seg], slen = p.seg_end[seg] - s0;
    const int per_tok = p.Hkv * 2 * (D / 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)slen * per_tok; i += (long long)gridDim.x * 256) {
        const int tok = (int)(i / per_tok), r = (int)(i % per_tok);
        const int kvh = r / (2 * (D / 4)), which = (r / (D / 4)) & 1, d = (r % (D / 4)) * 4;
        const long long off = ((((long long)(s0 + tok)) * p.Hkv + kvh) * 2 + which) * D + d;
        f32x4_t a = *(const f32x4_t*)(p.dkv_ws + off);
        for (int h = 1; h < p.hs; ++h) a += *(const f32x4_t*)(p.dkv_ws + (long long)h * p.T * p.Hkv * 2 * D + off);
        bf16_t* dst = which ? p.dv + (long long)(s0 + tok) * p.lddv + kvh * D + d : p.dk + (long long)(s0 + tok) * p.lddk + kvh * D + d;
        *(u32x2_t*)dst = (u32x2_t){pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
    }
}

// =====================================================================================================
// Paged-KV decode attention (rollout).  One query token per sequence.
//   K cache page: [Hkv][PAGE=32][D] row-major;  V cache page: [Hkv][D][PAGE] (keys contiguous) so that both
//   MFMA A operands are 16-byte contiguous global loads; a page is exactly one 32-key MFMA k-step.
//   Block = (sequence, kv head); the `group` q heads of that kv head are the MFMA N dimension (<=16).
//   4 waves take pages round-robin, each keeps an online-softmax state; combined through LDS at the end.
//   Sequences of one GRPO group share their prompt pages through the block table (prefill once per prompt).
// =====================================================================================================
// K pages are stored in the MFMA A-fragment order attn_decode reads them in: kpk_off (common.h)

struct DecodeArgs {
    const bf16_t* q;          // [B, Hq*D] (row stride ldq)
    const bf16_t* kcache;     // [npages][Hkv][32 x D in kpk_off order]
    const bf16_t* vcache;     // [npages][Hkv][D][32]
    const int* block_table;   // [B][max_pages]
    const int* ctx_len;       // [B] number of valid keys (including the current token)
    bf16_t* o;                // [B, Hq*D]
    long long ldq, ldo;
    int B, Hq, Hkv, max_pages;
    float scale;
    int gseq;                 // > 1: sequences [g * gseq, (g + 1) * gseq) share their prompt pages (group rollout) -> XCD-aware block order, see the kernel
};

template <int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void attn_decode_kernel(DecodeArgs p, SideOut so) {
    constexpr int KS = D / 32, DT = D / 16, PAGE = 32;
    extern __shared__ float dsm[];
    const int GS = (p.Hq / p.Hkv <= 8) ? 9 : 17;      // q-head columns kept per d (+1 pad): only `group` of the 16 MFMA columns are real
    float* red_m = dsm;                                // [WAVES][16]
    float* red_l = dsm + WAVES * 16;                   // [WAVES][16]
    float* red_o = dsm + 2 * WAVES * 16;               // [WAVES][D][GS]
    // Block -> (sequence, kv head).  Plain: grid (B, Hkv).  Group rollout (gseq > 1, 1-D grid): workgroup ids go round-robin over the 8 XCDs, each with its own
    // L2, so the plain order scatters the gseq sequences that share a prompt's K/V pages over all XCDs and every copy is an L2 miss.  Here XCD x takes the prompt
    // groups x, x + 8, ... and ALL blocks of a group (its sequences x kv heads) run on that one XCD: the pages come from HBM once and from L2 gseq - 1 times.
    int b = blockIdx.x, kvh = blockIdx.y;
    if (p.gseq > 1) {
        const int id = blockIdx.x, npg = p.gseq * p.Hkv, k = id >> 3;
        const int gi = (id & 7) + 8 * (k / npg), mem = k - (k / npg) * npg;
        if (gi * p.gseq >= p.B) return;
        kvh = mem / p.gseq;
        b = gi * p.gseq + (mem - kvh * p.gseq);
    }
    const int group = p.Hq / p.Hkv;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, g = l >> 4;
    const long long sb = side_base(so);
    STAMP(0);
    const int n = p.ctx_len[b];
    const int npage = (n + PAGE - 1) / PAGE;

    bf16x8_t qf[KS];
    {
        const bool ok = li < group;
        const bf16_t* src = p.q + (long long)b * p.ldq + (kvh * group + li) * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = ld_frag_g(src + ks * 32 + g * 8, ok);
    }
    float m = -INFINITY, lsum = 0.f;
    f32x4_t acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const float c = p.scale * LOG2E;

    for (int pg = w; pg < npage; pg += WAVES) {
        const int phys = p.block_table[(long long)b * p.max_pages + pg];
        const bf16_t* kp = p.kcache + ((long long)phys * p.Hkv + kvh) * PAGE * D;
        const bf16_t* vp = p.vcache + ((long long)phys * p.Hkv + kvh) * D * PAGE;
        // all K and V fragments of the page are requested up front: one memory latency per page, not two
        bf16x8_t kfr[KS][2], vfr[DT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) kfr[ks][t] = ld_frag_g(kp + ((ks * 2 + t) * 64 + l) * 8, true);  // keys perm_row(li) + 4t (kpk_off)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vfr[dt] = ld_frag_g(vp + (dt * 16 + li) * PAGE + g * 8, true);
        f32x4_t s[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[ks][t], qf[ks], s[t], 0, 0, 0);
        float mx = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = pg * PAGE + g * 8 + t * 4 + e;
                const float tv = key < n ? s[t][e] * c : -INFINITY;
                s[t][e] = tv;
                mx = fmaxf(mx, tv);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
        mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
        const float mn = fmaxf(m, mx);
        const float alpha = (mn == -INFINITY) ? 1.f : fast_exp2(m - mn);
        const float mref = (mn == -INFINITY) ? 0.f : mn;
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pv = fast_exp2(s[t][e] - mref);
                s[t][e] = pv;
                ps += pv;
            }
        lsum = lsum * alpha + ps;
        m = mn;
        const bf16x8_t pf = pack_frag(s[0], s[1]);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            acc[dt] *= alpha;
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[dt], pf, acc[dt], 0, 0, 0);
        }
    }
    STAMP(1);
    lsum += __shfl_xor(lsum, 16, WAVE);
    lsum += __shfl_xor(lsum, 32, WAVE);
    if (g == 0) { red_m[w * 16 + li] = m; red_l[w * 16 + li] = lsum; }
    if (li < GS - 1) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red_o[(w * D + dt * 16 + g * 4 + e) * GS + li] = acc[dt][e];
    }
    __syncthreads();
    STAMP(2);
    // combine the 4 partial states: thread -> (q head j, d)
    for (int idx = threadIdx.x; idx < group * D; idx += WAVES * 64) {
        const int j = idx / D, d = idx - j * D;
        float M = red_m[j];
#pragma unroll
        for (int ww = 1; ww < WAVES; ++ww) M = fmaxf(M, red_m[ww * 16 + j]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int ww = 0; ww < WAVES; ++ww) {
            const float f = (red_m[ww * 16 + j] == -INFINITY) ? 0.f : fast_exp2(red_m[ww * 16 + j] - M);
            num += f * red_o[(ww * D + d) * GS + j];
            den += f * red_l[ww * 16 + j];
        }
        const bf16_t ov = f2bf(den > 0.f ? num / den : 0.f);
        p.o[p.ldo ? (long long)b * p.ldo + (kvh * group + j) * D + d : xpk_off(b, (kvh * group + j) * D + d, p.Hq * D)] = ov;
        if (sb >= 0) {      // the row the attention backward of the policy reads, and its log-sum-exp (natural log, as attn_fwd stores it)
            const long long r = sb + (long long)b * so.seq_stride;
            ((bf16_t*)so.p0)[r * so.ld0 + (kvh * group + j) * D + d] = ov;
            if (d == 0) ((float*)so.p1)[(long long)(kvh * group + j) * so.ld1 + r] = den > 0.f ? (M + log2f(den)) * LN2 : -INFINITY;
        }
    }
    STAMP(3);
}

// Group-shared form of the decode attention for LONG shared prompts (LLaVA families: 3000-4000 prompt tokens; MHA decoders): the G sequences of a prompt group
// share their full prompt pages through the block table, so ONE block per (group, kv head) reads those pages once for all G * (Hq / Hkv) query rows (up to NT
// MFMA column tiles of 16) instead of every (sequence, kv head) block reading its own copy -- G times less K/V traffic for the prompt part, which is nearly all
// of it (15 GB of 29 GB per decode step at LLaVA-OneVision-7B shapes).  Phase 1: the 8 waves walk the shared pages; their partial softmax states are combined
// through LDS into one state per query row.  Phase 2: wave s takes sequence s of the group: its private pages (prompt remainder + completion), then the merge
// with the shared state of its rows and the output / side-output stores.  No cross-block exchange; grid = (groups, Hkv).
struct DecodeGroupArgs {
    DecodeArgs a;
    const int* shared_pages;  // [B / G]: leading block-table entries (full prompt pages) shared by the G sequences of a group
    int G;
    // Few (group, kv head) pairs and very long prompts (LLaVA-OneVision-7B: 8 x 4 blocks for 3900 shared tokens): the shared pages are split over `chunks` blocks
    // (grid.z) in a first launch (mode 1) that leaves one partial state per (group, kv head, chunk, row) in `ws`; a second launch (mode 2, grid (groups, Hkv))
    // merges them and runs phase 2.  mode 0: both phases in one launch.
    int mode, chunks;
    float* ws;                // [groups][Hkv][chunks][RMAX][D + 2]: o[D], m, l per row
};

template <int D, int NT>
__global__ __launch_bounds__(512) void attn_decode_group_kernel(DecodeGroupArgs gp, SideOut so) {
    constexpr int KS = D / 32, DT = D / 16, PAGE = 32, WAVES = 8, RMAX = 16 * NT, GS = 17;
    const DecodeArgs& p = gp.a;
    extern __shared__ float dsm[];
    float* red_m = dsm;                                 // [WAVES][16]
    float* red_l = red_m + WAVES * 16;                  // [WAVES][16]
    float* red_o = red_l + WAVES * 16;                  // [WAVES][D][GS]
    float* sh_m = red_o + WAVES * D * GS;               // [RMAX] state of the shared part per query row (log2 domain, like m / l below)
    float* sh_l = sh_m + RMAX;
    float* sh_o = sh_l + RMAX;                          // [RMAX][D]
    const int gi = blockIdx.x, kvh = blockIdx.y, group = p.Hq / p.Hkv, G = gp.G;
    const int rows = G * group;                         // <= RMAX (host checks)
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, g = l >> 4;
    const long long sb = side_base(so);
    const int b0 = gi * G;
    const int ns = gp.shared_pages[gi];
    const float c = p.scale * LOG2E;

    // ---- phase 1: shared pages, all query rows of the group ----------------------------------------------------------------------------------
    // Q fragments of the NT column tiles live in LDS in fragment order (one conflict-free 16-byte read per fragment and page): 64 VGPRs less than keeping
    // them in registers, which with the 128 accumulator registers of NT = 4 spilled to scratch
    bf16_t* qs = (bf16_t*)(sh_o + RMAX * D);            // [NT][KS][64 lanes][8]
    for (int t = w; t < NT; t += WAVES) {
        const int r = t * 16 + li, sq = r / group, j = r - sq * group;
        const bool ok = r < rows;
        const bf16_t* src = p.q + (long long)(b0 + (ok ? sq : 0)) * p.ldq + (kvh * group + (ok ? j : 0)) * D;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) *(bf16x8_t*)(qs + ((t * KS + ks) * 64 + l) * 8) = ld_frag_g(src + ks * 32 + g * 8, ok);
    }
    __syncthreads();
    float m[NT], lsum[NT];
    f32x4_t acc[NT][DT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        m[t] = -INFINITY;
        lsum[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[t][dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    const int per = gp.mode == 1 ? (ns + gp.chunks - 1) / gp.chunks : ns;           // pages of this block's chunk (mode 1: blockIdx.z)
    const int pg0 = gp.mode == 1 ? (int)blockIdx.z * per : 0, pg1 = gp.mode == 2 ? 0 : min(ns, pg0 + per);
    float* wsb = gp.ws + (((long long)gi * p.Hkv + kvh) * gp.chunks) * RMAX * (D + 2);
    for (int pg = pg0 + w; pg < pg1; pg += WAVES) {
        asm volatile("" ::: "memory");      // keeps the (loop-invariant) Q fragment reads inside the loop: hoisted, they are 64 registers again
        const int phys = p.block_table[(long long)b0 * p.max_pages + pg];
        const bf16_t* kp = p.kcache + ((long long)phys * p.Hkv + kvh) * PAGE * D;
        const bf16_t* vp = p.vcache + ((long long)phys * p.Hkv + kvh) * D * PAGE;
        bf16x8_t kfr[KS][2], vfr[DT];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) kfr[ks][t2] = ld_frag_g(kp + ((ks * 2 + t2) * 64 + l) * 8, true);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vfr[dt] = ld_frag_g(vp + (dt * 16 + li) * PAGE + g * 8, true);
#pragma unroll
        for (int t = 0; t < NT; ++t) {      // every key of a shared page is a prompt token of every sequence of the group: no masking
            f32x4_t sc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
                    sc[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[ks][t2], *(const bf16x8_t*)(qs + ((t * KS + ks) * 64 + l) * 8), sc[t2], 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[t2][e] *= c; mx = fmaxf(mx, sc[t2][e]); }
            mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
            mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
            const float mn = fmaxf(m[t], mx);
            const float alpha = fast_exp2(m[t] - mn);           // m = -inf on the first page: exp2(-inf) = 0
            float ps = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[t2][e] = fast_exp2(sc[t2][e] - mn); ps += sc[t2][e]; }
            lsum[t] = lsum[t] * alpha + ps;
            m[t] = mn;
            const bf16x8_t pf = pack_frag(sc[0], sc[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                acc[t][dt] *= alpha;
                acc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[dt], pf, acc[t][dt], 0, 0, 0);
            }
        }
    }
    // combine the 8 waves' states, one column tile at a time, into the per-row shared state (LDS; mode 1: this chunk's slot of the workspace)
#pragma unroll
    for (int t = 0; t < NT && gp.mode != 2; ++t) {
        float ls = lsum[t];
        ls += __shfl_xor(ls, 16, WAVE);
        ls += __shfl_xor(ls, 32, WAVE);
        if (g == 0) { red_m[w * 16 + li] = m[t]; red_l[w * 16 + li] = ls; }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) red_o[(w * D + dt * 16 + g * 4 + e) * GS + li] = acc[t][dt][e];
        __syncthreads();
        for (int idx = threadIdx.x; idx < 16 * D; idx += WAVES * 64) {
            const int j = idx / D, d = idx - j * D;
            float M = red_m[j];
#pragma unroll
            for (int ww = 1; ww < WAVES; ++ww) M = fmaxf(M, red_m[ww * 16 + j]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int ww = 0; ww < WAVES; ++ww) {
                const float f = (red_m[ww * 16 + j] == -INFINITY) ? 0.f : fast_exp2(red_m[ww * 16 + j] - M);
                num += f * red_o[(ww * D + d) * GS + j];
                den += f * red_l[ww * 16 + j];
            }
            if (gp.mode == 1) {
                float* dst = wsb + ((long long)blockIdx.z * RMAX + t * 16 + j) * (D + 2);
                dst[d] = num;
                if (d == 0) { dst[D] = M; dst[D + 1] = den; }
            } else {
                sh_o[(t * 16 + j) * D + d] = num;
                if (d == 0) { sh_m[t * 16 + j] = M; sh_l[t * 16 + j] = den; }
            }
        }
        __syncthreads();
    }
    if (gp.mode == 1) return;
    if (gp.mode == 2) {      // merge the chunks' partial states of the first launch (fixed order: reproducible)
        for (int idx = threadIdx.x; idx < rows * D; idx += WAVES * 64) {
            const int r = idx / D, d = idx - r * D;
            float M = -INFINITY;
            for (int cc = 0; cc < gp.chunks; ++cc) M = fmaxf(M, wsb[((long long)cc * RMAX + r) * (D + 2) + D]);
            float num = 0.f, den = 0.f;
            for (int cc = 0; cc < gp.chunks; ++cc) {
                const float* src = wsb + ((long long)cc * RMAX + r) * (D + 2);
                const float f = (src[D] == -INFINITY) ? 0.f : fast_exp2(src[D] - M);
                num += f * src[d];
                den += f * src[D + 1];
            }
            sh_o[r * D + d] = num;
            if (d == 0) { sh_m[r] = M; sh_l[r] = den; }
        }
        __syncthreads();
    }

    // ---- phase 2: wave s <- sequence s of the group: private pages, merge with the shared state, stores ----------------------------------------
    for (int sq = w; sq < G; sq += WAVES) {
        const int b = b0 + sq;
        const int n = p.ctx_len[b];
        const int npage = (n + PAGE - 1) / PAGE;
        bf16x8_t q1[KS];
        {
            const bool ok = li < group;
            const bf16_t* src = p.q + (long long)b * p.ldq + (kvh * group + (ok ? li : 0)) * D;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) q1[ks] = ld_frag_g(src + ks * 32 + g * 8, ok);
        }
        float m1 = -INFINITY, l1 = 0.f;
        f32x4_t a1[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) a1[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (int pg = ns; pg < npage; ++pg) {
            const int phys = p.block_table[(long long)b * p.max_pages + pg];
            const bf16_t* kp = p.kcache + ((long long)phys * p.Hkv + kvh) * PAGE * D;
            const bf16_t* vp = p.vcache + ((long long)phys * p.Hkv + kvh) * D * PAGE;
            bf16x8_t kfr[KS][2], vfr[DT];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) kfr[ks][t2] = ld_frag_g(kp + ((ks * 2 + t2) * 64 + l) * 8, true);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vfr[dt] = ld_frag_g(vp + (dt * 16 + li) * PAGE + g * 8, true);
            f32x4_t sc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) sc[t2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[ks][t2], q1[ks], sc[t2], 0, 0, 0);
            float mx = -INFINITY;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = pg * PAGE + g * 8 + t2 * 4 + e;
                    const float tv = key < n ? sc[t2][e] * c : -INFINITY;
                    sc[t2][e] = tv;
                    mx = fmaxf(mx, tv);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
            mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
            const float mn = fmaxf(m1, mx);
            const float alpha = (mn == -INFINITY) ? 1.f : fast_exp2(m1 - mn);
            const float mref = (mn == -INFINITY) ? 0.f : mn;
            float ps = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int e = 0; e < 4; ++e) { sc[t2][e] = fast_exp2(sc[t2][e] - mref); ps += sc[t2][e]; }
            l1 = l1 * alpha + ps;
            m1 = mn;
            const bf16x8_t pf = pack_frag(sc[0], sc[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                a1[dt] *= alpha;
                a1[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[dt], pf, a1[dt], 0, 0, 0);
            }
        }
        l1 += __shfl_xor(l1, 16, WAVE);
        l1 += __shfl_xor(l1, 32, WAVE);
        if (li < group) {       // lane (li = head j of the group, g): dims dt * 16 + g * 4 + e
            const int r = sq * group + li;
            const float ms = sh_m[r], lsh = sh_l[r];
            const float M = fmaxf(ms, m1);
            const float fs = (ms == -INFINITY) ? 0.f : fast_exp2(ms - M), f1 = (m1 == -INFINITY) ? 0.f : fast_exp2(m1 - M);
            const float den = fs * lsh + f1 * l1;
            const float inv = den > 0.f ? 1.f / den : 0.f;
            const int col0 = (kvh * group + li) * D;
            const long long rr = sb >= 0 ? sb + (long long)b * so.seq_stride : 0;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + g * 4;
                const f32x4_t so4 = *(const f32x4_t*)(sh_o + (long long)r * D + d);
                const u32x2_t ov = {pack2bf((fs * so4[0] + f1 * a1[dt][0]) * inv, (fs * so4[1] + f1 * a1[dt][1]) * inv),
                                    pack2bf((fs * so4[2] + f1 * a1[dt][2]) * inv, (fs * so4[3] + f1 * a1[dt][3]) * inv)};
                *(u32x2_t*)(p.o + (p.ldo ? (long long)b * p.ldo + col0 + d : xpk_off(b, col0 + d, p.Hq * D))) = ov;
                if (sb >= 0) *(u32x2_t*)((bf16_t*)so.p0 + rr * so.ld0 + col0 + d) = ov;
            }
            if (sb >= 0 && g == 0) ((float*)so.p1)[(long long)(kvh * group + li) * so.ld1 + rr] = den > 0.f ? (M + log2f(den)) * LN2 : -INFINITY;
        }
    }
}

// Write K/V rows of `T` tokens into the paged cache (prefill: many tokens; decode: one per sequence).
// slot[t] = physical page * 32 + offset, or < 0 to skip (padding).
template <int D>
__global__ __launch_bounds__(256) void kv_store_kernel(const bf16_t* k, long long ldk, const bf16_t* v, long long ldv, const long long* slot,
                                                       bf16_t* kcache, bf16_t* vcache, int T, int Hkv) {
    constexpr int CPR = D / 8;
    const long long total = (long long)T * Hkv * CPR;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % CPR);
        const long long th = i / CPR;
        const int h = (int)(th % Hkv);
        const long long t = th / Hkv;
        const long long sl = slot[t];
        if (sl < 0) continue;
        const long long page = sl >> 5;
        const int off = (int)(sl & 31);
        const u32x4_t kv = *(const u32x4_t*)(k + t * ldk + h * D + c * 8);
        *(u32x4_t*)(kcache + (page * Hkv + h) * 32 * D + kpk_off(off, c * 8)) = kv;
        const u32x4_t vv = *(const u32x4_t*)(v + t * ldv + h * D + c * 8);
        bf16_t* vd = vcache + (page * Hkv + h) * (long long)D * 32 + off;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            vd[(c * 8 + 2 * e) * 32] = (bf16_t)(vv[e] & 0xffffu);
            vd[(c * 8 + 2 * e + 1) * 32] = (bf16_t)(vv[e] >> 16);
        }
    }
}

// Decode-step fusion: rotary on the q and k heads of each new token (in place for q; k goes straight to its
// cache row) + v into the transposed cache page.  One launch instead of rope + kv_store.
template <int D>
__global__ __launch_bounds__(256) void rope_kv_store_kernel(bf16_t* qkv, long long ld, const float* cs, const float* sn, const long long* slot,
                                                            bf16_t* kcache, bf16_t* vcache, int T, int Hq, int Hkv) {
    constexpr int HALF = D / 2, RC = HALF / 4, VC = D / 8;   // rope items (4-element pairs) and v items (8 elements) per head
    const int per_tok = (Hq + Hkv) * RC + Hkv * VC;
    const long long total = (long long)T * per_tok;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long t = i / per_tok;
        int r = (int)(i - t * per_tok);
        const long long sl = slot[t];
        const long long page = sl >> 5;
        const int off = (int)(sl & 31);
        bf16_t* row = qkv + t * ld;
        if (r < (Hq + Hkv) * RC) {
            const int h = r / RC, c = r - h * RC;
            bf16_t* p = row + h * D + c * 4;
            const u32x2_t lo = *(const u32x2_t*)p, hi = *(const u32x2_t*)(p + HALF);
            const f32x4_t cc = *(const f32x4_t*)(cs + t * HALF + c * 4), ss = *(const f32x4_t*)(sn + t * HALF + c * 4);
            const float a[4] = {lo_bf(lo[0]), hi_bf(lo[0]), lo_bf(lo[1]), hi_bf(lo[1])};
            const float b[4] = {lo_bf(hi[0]), hi_bf(hi[0]), lo_bf(hi[1]), hi_bf(hi[1])};
            float oa[4], ob[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { oa[e] = a[e] * cc[e] - b[e] * ss[e]; ob[e] = b[e] * cc[e] + a[e] * ss[e]; }
            const u32x2_t va = {pack2bf(oa[0], oa[1]), pack2bf(oa[2], oa[3])}, vb = {pack2bf(ob[0], ob[1]), pack2bf(ob[2], ob[3])};
            if (h < Hq) {
                *(u32x2_t*)p = va;
                *(u32x2_t*)(p + HALF) = vb;
            } else if (sl >= 0) {
                bf16_t* kd = kcache + (page * Hkv + (h - Hq)) * 32 * D;
                *(u32x2_t*)(kd + kpk_off(off, c * 4)) = va;
                *(u32x2_t*)(kd + kpk_off(off, c * 4 + HALF)) = vb;
            }
        } else if (sl >= 0) {
            r -= (Hq + Hkv) * RC;
            const int h = r / VC, c = r - h * VC;
            const u32x4_t vv = *(const u32x4_t*)(row + (Hq + Hkv + h) * D + c * 8);
            bf16_t* vd = vcache + (page * Hkv + h) * (long long)D * 32 + off;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                vd[(c * 8 + 2 * e) * 32] = (bf16_t)(vv[e] & 0xffffu);
                vd[(c * 8 + 2 * e + 1) * 32] = (bf16_t)(vv[e] >> 16);
            }
        }
    }
}

template <typename K>
void set_smem(K kern, int bytes) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }

}  // namespace

static int check_common(int T, int Hq, int Hkv, int D, long long ldq, long long ldk, long long ldv, long long ldo) {
    IADR1_REQUIRE(D == 128 || D == 80, "attention: head dim %d not built (128 and 80 are)", D);
    IADR1_REQUIRE(T > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attention: bad head counts Hq=%d Hkv=%d", Hq, Hkv);
    IADR1_REQUIRE((ldq % 8) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (ldo % 8) == 0, "attention: row strides must be multiples of 8 elements");
    return IADR1_OK;
}

// Launch ranges of a segment list.  nseg_head > 0: the first nseg_head segments may be up to max_seqlen long, the remaining ones at most
// max_seqlen_tail (shared-prefix batches: 8 prompts of 512 tokens in front of 64 completions of 256).  One grid sized for the longest segment
// dispatched thousands of empty blocks for the short ones, and workgroup dispatch (~90 ns each, measured with in-kernel stamps) was the time of
// the short-segment part of the backward kernels.
struct SegRange { int off, n, maxlen; };
static int seg_ranges(int nseg, int max_seqlen, int nseg_head, int max_seqlen_tail, SegRange out[2]) {
    if (nseg_head > 0 && nseg_head < nseg && max_seqlen_tail > 0 && max_seqlen_tail <= max_seqlen) {
        out[0] = SegRange{0, nseg_head, max_seqlen};
        out[1] = SegRange{nseg_head, nseg - nseg_head, max_seqlen_tail};
        return 2;
    }
    out[0] = SegRange{0, nseg, max_seqlen};
    return 1;
}

extern "C" int iadr1_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* seg_start, const int* seg_end,
                              const int* seg_prefix, int nseg, int max_seqlen, int nseg_head, int max_seqlen_tail, int T, int Hq, int Hkv, int D, long long ldq, long long ldk, long long ldv,
                              long long ldo, int causal, float scale, hipStream_t stream) {
    if (int e = check_common(T, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return e;
    IADR1_REQUIRE(nseg > 0 && max_seqlen > 0, "attn_fwd: empty segment list");
    AttnArgs p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
    p.seg_start = seg_start; p.seg_end = seg_end; p.seg_prefix = seg_prefix; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.T = T; p.Hq = Hq; p.Hkv = Hkv; p.causal = causal; p.scale = scale;
    constexpr int force_r = 1;      // query tiles per block (2 measured slower: profiles/EXPERIMENTS.md rounds 1-2)
    SegRange rg[2];
    const int nr = seg_ranges(nseg, max_seqlen, nseg_head, max_seqlen_tail, rg);
    for (int ri = 0; ri < nr; ++ri) {
        p.seg_off = rg[ri].off;
        const int ml = rg[ri].maxlen, ns = rg[ri].n;
        // 64-row q tiles (R=1, ~180 VGPR, 2 blocks/CU) measured 1.4x faster than 128-row tiles (R=2, 1 block/CU): occupancy wins
        const bool small = ml <= 64 || force_r == 1;
#define LAUNCH_FWD(DD, RR)                                                                                           \
    do {                                                                                                             \
        const int smem = (2 * 64 * Cfg<DD>::LD) * 2;                                                                 \
        set_smem(attn_fwd_kernel<DD, RR>, smem);                                                                     \
        const int bm = 64 * RR;                                                                                      \
        hipLaunchKernelGGL((attn_fwd_kernel<DD, RR>), dim3((ml + bm - 1) / bm, ns, Hq), dim3(256), smem, stream, p); \
    } while (0)
        if (D == 128) { if (small) LAUNCH_FWD(128, 1); else LAUNCH_FWD(128, 2); }
        else { if (small) LAUNCH_FWD(80, 1); else LAUNCH_FWD(80, 2); }
#undef LAUNCH_FWD
    }
    return iadr1_check_launch("attn_fwd");
}

// Chunked teacher-forced forward (include/iadr1_hip.h): the query rows seg_view[i] = {q_first, q_count, ..} of every segment against all its keys so far.
extern "C" int iadr1_attn_fwd_chunk(const void* q, const void* k, const void* v, void* o, float* lse, const int* seg_start, const int* seg_end,
                                    const int* seg_prefix, const int* seg_view, int nseg, int max_q_count, int T, int Hq, int Hkv, int D, long long ldq,
                                    long long ldk, long long ldv, long long ldo, float scale, hipStream_t stream) {
    if (int e = check_common(T, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return e;
    IADR1_REQUIRE(nseg > 0 && max_q_count > 0 && seg_view != nullptr, "attn_fwd_chunk: empty segment list / no segment views");
    AttnArgs p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
    p.seg_start = seg_start; p.seg_end = seg_end; p.seg_prefix = seg_prefix; p.seg_view = seg_view; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.T = T; p.Hq = Hq; p.Hkv = Hkv; p.causal = 1; p.scale = scale;
    const int smem = (2 * 64 * Cfg<128>::LD) * 2;
    if (D == 128) {
        set_smem(attn_fwd_kernel<128, 1>, smem);
        hipLaunchKernelGGL((attn_fwd_kernel<128, 1>), dim3((max_q_count + 63) / 64, nseg, Hq), dim3(256), smem, stream, p);
    } else {
        set_smem(attn_fwd_kernel<80, 1>, smem);
        hipLaunchKernelGGL((attn_fwd_kernel<80, 1>), dim3((max_q_count + 63) / 64, nseg, Hq), dim3(256), smem, stream, p);
    }
    return iadr1_check_launch("attn_fwd_chunk");
}

extern "C" int iadr1_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse, float* delta,
                              void* dq, void* dk, void* dv, float* dkv_ws, int head_splits, const int* seg_start, const int* seg_end, const int* seg_prefix, int nseg, int max_seqlen,
                              int nseg_head, int max_seqlen_tail, int T,
                              int Hq, int Hkv, int D, long long ldq, long long ldk, long long ldv, long long ldo, long long lddo,
                              long long lddq, long long lddk, long long lddv, int causal, float scale, hipStream_t stream) {
    if (int e = check_common(T, Hq, Hkv, D, ldq, ldk, ldv, ldo)) return e;
    IADR1_REQUIRE((lddo % 8) == 0 && (lddq % 8) == 0 && (lddk % 8) == 0 && (lddv % 8) == 0, "attn_bwd: gradient row strides must be multiples of 8");
    IADR1_REQUIRE(nseg > 0 && max_seqlen > 0, "attn_bwd: empty segment list");
    AttnArgs p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.lse = (float*)lse;
    p.seg_start = seg_start; p.seg_end = seg_end; p.seg_prefix = seg_prefix; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.T = T; p.Hq = Hq; p.Hkv = Hkv; p.causal = causal; p.scale = scale;
    p.dout = (const bf16_t*)dout; p.delta = delta; p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
    const int hs_all = (dkv_ws && head_splits > 1) ? head_splits : 1;
    IADR1_REQUIRE((Hq / Hkv) % hs_all == 0, "attn_bwd: head_splits=%d must divide the GQA group %d", hs_all, Hq / Hkv);
    p.dkv_ws = dkv_ws;
    const long long items = (long long)T * Hq * 16;
    constexpr int dq_r = 2, kv_waves = 8;      // the measured choices (8-wave dK/dV, two query tiles per dQ block: profiles/EXPERIMENTS.md rounds 1-2)
    SegRange rg[2];
    const int nr = seg_ranges(nseg, max_seqlen, nseg_head, max_seqlen_tail, rg);
#define LAUNCH_BWD(DD)                                                                                                                   \
    do {                                                                                                                                 \
        hipLaunchKernelGGL(attn_delta_kernel<DD>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)o, ldo,    \
                           (const bf16_t*)dout, lddo, delta, T, Hq);                                                                     \
        for (int ri = 0; ri < nr; ++ri) {                                                                                                \
            p.seg_off = rg[ri].off;                                                                                                      \
            const int ml = rg[ri].maxlen, ns = rg[ri].n;                                                                                 \
            /* the head split of dK/dV balances the long query loops of shared-prefix segments: only the head range needs it (a short  */ \
            /* segment split 4 ways is 4x the blocks for 9 iterations each)                                                            */ \
            p.hs = (nr == 2 && ri == 1) ? 1 : hs_all;                                                                                    \
            const int smem_dq = (2 * 32 * Cfg<DD>::LD) * 2;                                                                              \
            if (dq_r == 2 && ml > 64) {                                                                                                  \
                set_smem(attn_bwd_dq_kernel<DD, 2>, smem_dq);                                                                            \
                hipLaunchKernelGGL((attn_bwd_dq_kernel<DD, 2>), dim3((ml + 127) / 128, ns, Hq), dim3(256), smem_dq, stream, p);          \
            } else {                                                                                                                     \
                set_smem(attn_bwd_dq_kernel<DD, 1>, smem_dq);                                                                            \
                hipLaunchKernelGGL((attn_bwd_dq_kernel<DD, 1>), dim3((ml + 63) / 64, ns, Hq), dim3(256), smem_dq, stream, p);            \
            }                                                                                                                            \
            const int smem_kv = (2 * 32 * Cfg<DD>::LD) * 2 + 64 * 4;                                                                     \
            if (kv_waves == 8 && ml > 64) {                                                                                              \
                set_smem(attn_bwd_dkdv_kernel<DD, 8>, smem_kv);                                                                          \
                hipLaunchKernelGGL((attn_bwd_dkdv_kernel<DD, 8>), dim3((ml + 127) / 128, Hkv * p.hs, ns), dim3(512), smem_kv, stream, p); \
            } else {                                                                                                                     \
                set_smem(attn_bwd_dkdv_kernel<DD, 4>, smem_kv);                                                                          \
                hipLaunchKernelGGL((attn_bwd_dkdv_kernel<DD, 4>), dim3((ml + 63) / 64, Hkv * p.hs, ns), dim3(256), smem_kv, stream, p);  \
            }                                                                                                                            \
            if (p.hs > 1) hipLaunchKernelGGL(attn_dkdv_reduce_kernel<DD>, dim3((ml * Hkv * 2 * (DD / 4) + 255) / 256, ns), dim3(256), 0, stream, p); \
        }                                                                                                                                \
    } while (0)
    if (D == 128) LAUNCH_BWD(128); else LAUNCH_BWD(80);
#undef LAUNCH_BWD
    return iadr1_check_launch("attn_bwd");
}

extern "C" int iadr1_attn_decode(const void* q, const void* kcache, const void* vcache, const int* block_table, const int* ctx_len, void* o,
                                 int B, int Hq, int Hkv, int D, int max_pages, long long ldq, long long ldo, float scale, int seqs_per_group, const void* side, hipStream_t stream) {
    IADR1_REQUIRE(D == 128, "attn_decode: head dim %d not built (128 is)", D);
    IADR1_REQUIRE(B > 0 && Hq % Hkv == 0 && Hq / Hkv <= 16, "attn_decode: GQA group must be <= 16");
    IADR1_REQUIRE((ldq % 8) == 0, "attn_decode: ldq must be a multiple of 8");
    IADR1_REQUIRE(seqs_per_group >= 0 && (seqs_per_group <= 1 || B % seqs_per_group == 0), "attn_decode: B must be a multiple of seqs_per_group");
    const int gseq = seqs_per_group > 1 ? seqs_per_group : 0;      // the blocks of a prompt group on one XCD: its L2 dedupes the shared prompt pages
    DecodeArgs p{(const bf16_t*)q, (const bf16_t*)kcache, (const bf16_t*)vcache, block_table, ctx_len, (bf16_t*)o, ldq, ldo, B, Hq, Hkv, max_pages, scale, gseq};
    const dim3 grid = gseq ? dim3(8 * ((B / gseq + 7) / 8) * gseq * Hkv) : dim3(B, Hkv);
    // 16 waves per (sequence, kv head) block: a wave then walks ~1.5 pages instead of ~3 at ctx ~ 640 (the kernel is a chain of dependent
    // page loads on only B*Hkv = 128 CUs)
    constexpr int waves = 16;
    const int gs = (Hq / Hkv <= 8) ? 9 : 17;
    SideOut so;
    if (int e = iadr1_side_arg(side, &so)) return e;
    if (waves == 16) {
        const int smem = (2 * 16 * 16 + 16 * 128 * gs) * 4;
        set_smem(attn_decode_kernel<128, 16>, smem);
        hipLaunchKernelGGL((attn_decode_kernel<128, 16>), grid, dim3(1024), smem, stream, p, so);
    } else {
        const int smem = (2 * 8 * 16 + 8 * 128 * gs) * 4;
        set_smem(attn_decode_kernel<128, 8>, smem);
        hipLaunchKernelGGL((attn_decode_kernel<128, 8>), grid, dim3(512), smem, stream, p, so);
    }
    return iadr1_check_launch("attn_decode");
}

extern "C" int iadr1_attn_decode_group(const void* q, const void* kcache, const void* vcache, const int* block_table, const int* ctx_len, const int* shared_pages, void* o,
                                       int B, int G, int Hq, int Hkv, int D, int max_pages, long long ldq, long long ldo, float scale, int chunks, float* ws,
                                       const void* side, hipStream_t stream) {
    IADR1_REQUIRE(D == 128, "attn_decode_group: head dim %d not built (128 is)", D);
    IADR1_REQUIRE(B > 0 && G >= 1 && (B % G) == 0 && Hq % Hkv == 0 && Hq / Hkv <= 16 && G * (Hq / Hkv) <= 64 && shared_pages != nullptr,
                  "attn_decode_group: B must be a multiple of the group size and a group may hold at most 64 query rows per kv head (G=%d, heads per kv head=%d)", G, Hkv ? Hq / Hkv : 0);
    IADR1_REQUIRE((ldq % 8) == 0 && (ldo % 4) == 0, "attn_decode_group: ldq must be a multiple of 8, ldo of 4");
    IADR1_REQUIRE(chunks >= 1 && chunks <= 64 && (chunks == 1 || ws != nullptr), "attn_decode_group: 1 <= chunks <= 64, and chunks > 1 needs the workspace");
    DecodeGroupArgs p{{(const bf16_t*)q, (const bf16_t*)kcache, (const bf16_t*)vcache, block_table, ctx_len, (bf16_t*)o, ldq, ldo, B, Hq, Hkv, max_pages, scale, 0}, shared_pages, G, 0, chunks, ws};
    SideOut so, none{};
    if (int e = iadr1_side_arg(side, &so)) return e;
    const int rows = G * (Hq / Hkv), nt = rows <= 16 ? 1 : (rows <= 32 ? 2 : 4);
    const int smem = (2 * 8 * 16 + 8 * 128 * 17 + 2 * 16 * nt + 16 * nt * 128) * 4 + nt * 4 * 64 * 8 * 2;       // reduction scratch, shared-part state, Q fragments
    const dim3 block(512);
    auto launch = [&](dim3 grid, const SideOut& s_) {
        if (nt == 1) { set_smem(attn_decode_group_kernel<128, 1>, smem); hipLaunchKernelGGL((attn_decode_group_kernel<128, 1>), grid, block, smem, stream, p, s_); }
        else if (nt == 2) { set_smem(attn_decode_group_kernel<128, 2>, smem); hipLaunchKernelGGL((attn_decode_group_kernel<128, 2>), grid, block, smem, stream, p, s_); }
        else { set_smem(attn_decode_group_kernel<128, 4>, smem); hipLaunchKernelGGL((attn_decode_group_kernel<128, 4>), grid, block, smem, stream, p, s_); }
    };
    if (chunks == 1) {
        launch(dim3(B / G, Hkv), so);
    } else {
        p.mode = 1;
        launch(dim3(B / G, Hkv, chunks), none);
        p.mode = 2;
        launch(dim3(B / G, Hkv), so);
    }
    return iadr1_check_launch("attn_decode_group");
}

extern "C" int iadr1_kv_store(const void* k, long long ldk, const void* v, long long ldv, const long long* slot, void* kcache, void* vcache,
                              int T, int Hkv, int D, hipStream_t stream) {
    IADR1_REQUIRE(D == 128, "kv_store: head dim %d not built (128 is)", D);
    IADR1_REQUIRE(T > 0 && (ldk % 8) == 0 && (ldv % 8) == 0, "kv_store: strides must be multiples of 8");
    long long blocks = ((long long)T * Hkv * 16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(kv_store_kernel<128>, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)k, ldk, (const bf16_t*)v, ldv, slot, (bf16_t*)kcache, (bf16_t*)vcache, T, Hkv);
    return iadr1_check_launch("kv_store");
}

extern "C" int iadr1_rope_kv_store(void* qkv, long long ld, const float* cos_t, const float* sin_t, const long long* slot, void* kcache,
                                   void* vcache, int T, int Hq, int Hkv, int D, hipStream_t stream) {
    IADR1_REQUIRE(D == 128, "rope_kv_store: head dim %d not built (128 is)", D);
    IADR1_REQUIRE(T > 0 && (ld % 8) == 0, "rope_kv_store: ld must be a multiple of 8");
    const long long total = (long long)T * ((Hq + Hkv) * 16 + Hkv * 16);
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rope_kv_store_kernel<128>, dim3((int)blocks), dim3(256), 0, stream, (bf16_t*)qkv, ld, cos_t, sin_t, slot, (bf16_t*)kcache, (bf16_t*)vcache, T, Hq, Hkv);
    return iadr1_check_launch("rope_kv_store");
}

IADR1_STAMPS_EXPORT(attn)
#ifdef IADR1_STAMPS
// probe builds only: blocks per CU the runtime admits for the attention kernels at their launch configuration
extern "C" int iadr1_debug_attn_occupancy(int* out6) {
    const int smem_kv = (2 * 32 * Cfg<128>::LD) * 2 + 64 * 4, smem_dq = (2 * 32 * Cfg<128>::LD) * 2, smem_f = (2 * 64 * Cfg<128>::LD) * 2;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&out6[0], attn_bwd_dkdv_kernel<128, 4>, 256, smem_kv);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&out6[1], attn_bwd_dkdv_kernel<128, 8>, 512, smem_kv);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&out6[2], attn_bwd_dq_kernel<128, 1>, 256, smem_dq);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&out6[3], attn_bwd_dq_kernel<128, 2>, 256, smem_dq);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&out6[4], attn_fwd_kernel<128, 1>, 256, smem_f);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&out6[5], attn_fwd_kernel<128, 2>, 256, smem_f);
    return 0;
}
#endif
