/* iadr1_hip.h -- C ABI of libiadr1_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the IAD-R1
 * post-training hot path (PA-SFT / SC-GRPO on Qwen2.5-VL).
 *
 * The reference (Yanhui-Lee/IAD-R1) is pure Python and has no FFI of its own;
 * each entry point below replaces the third-party kernel the reference reaches at the cited call site
 * (TF: = transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py as installed, 5.15.0 line numbers;
 * REF: = /root/reference).  INTEGRATION.md shows the ctypes / AttentionInterface stubs a maintainer
 * of the reference would add to bind them.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch types.  All pointers are DEVICE pointers unless noted.
 *   - the caller owns every buffer; nothing here allocates, frees or synchronises.
 *   - asynchronous and stream-ordered on `stream` (a hipStream_t, passed as void*); legal inside
 *     hipGraph capture; re-entrant: no mutable global state (the IADR1_* A/B switches
 *     of the launchers are read once, at first use, into constants).
 *   - bf16 tensors are raw uint16 bit patterns; arithmetic is fp32; row strides (`ld*`) are in elements.
 *   - return 0 on success, negative on error; iadr1_last_error() returns the thread-local message.
 */
#ifndef IADR1_HIP_H
#define IADR1_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef void* iadr1_stream_t; /* hipStream_t */

int iadr1_version(void);
const char* iadr1_last_error(void);
/* Co-scheduling of the rollout and the teacher-forced training forward (round 5).  The reference runs its rollout engine and its training ranks on
 * DIFFERENT GPUs at the same time (REF:scripts/train/SC_GRPO/SC_GRPO_Qwen_Instruct_2_5_VL_3B.sh:40-42: 3 training ranks + one vLLM GPU,
 * REF:train/stage_rl/trainer/sc_grpo_trainer.py:637-683 || :737-743); here one MI355X is split by CU masks instead: the latency-bound decode
 * replay keeps most CUs, the MFMA-bound forward over the tokens already sampled runs on the rest, each on its own HIP stream.
 *   iadr1_stream_create_cu_mask: hipExtStreamCreateWithCUMask.  `cu_mask` is a HOST array of `n_words` 32-bit words; bit i enables CU i in the
 *     driver's numbering (round-robin over the 8 XCDs: bits [8k, 8k+8) are one CU of every XCD).  `stream_out` (HOST) receives the hipStream_t.
 *   iadr1_stream_destroy: hipStreamDestroy of such a stream.
 *   iadr1_set_decode_cus: the number of CUs the decode-step launchers (iadr1_gemm_skinny_bf16, iadr1_gemm_qkv_rope_kv_bf16, iadr1_gemm_skinny_fp8w)
 *     size their persistent one-block-per-CU grids for; 0 = the device's CU count.  THREAD-LOCAL launcher configuration (like iadr1_last_error): the thread
 *     that captures or launches a decode step sets it right before and resets it to 0 right after; other threads and other engines never see it. */
int iadr1_stream_create_cu_mask(const unsigned* cu_mask, int n_words, void** stream_out);
int iadr1_stream_destroy(void* stream);
int iadr1_set_decode_cus(int n_cus);
/* Rollout -> training hand-over.  The four decode-step entry points that take a trailing `side` argument can ALSO write what they compute into
 * row-major training buffers, at row  base + s * seq_stride + *step  for sequence s (`step`: device-resident decode step counter).  `side` is a
 * HOST pointer to this struct, read during the call (NULL: no side outputs); the struct holds DEVICE pointers:
 *   iadr1_rmsnorm_fwd (T <= 256)        p0 = residual stream rows [.., ld0] (bf16), p1 = normalised rows [.., ld1], p2 = rstd (fp32, one per row)
 *   iadr1_gemm_qkv_rope_kv_bf16         p0 = roped q|k|v rows [.., ld0]
 *   iadr1_attn_decode                   p0 = attention output rows [.., ld0], p1 = log-sum-exp fp32 [Hq][ld1]
 *   iadr1_gemm_skinny_bf16 out_mode 3   p0 = gate|up rows [.., ld0], p1 = SwiGLU rows [.., ld1]   (persistent kernel shapes only)
 * Decode step t processes completion token t of every sequence = row (s, t) of the completion block of the shared-prefix training batch, so the
 * policy's teacher-forced forward over the completions (REF:505-513 on the policy model) need not be run again before backward.  Unused p*: NULL. */
typedef struct iadr1_side_out {
    void* p0; long long ld0;
    void* p1; long long ld1;
    void* p2; long long ld2;
    const unsigned* step;
    long long base, seq_stride;
    /* progress mark (iadr1_rmsnorm_fwd's few-row kernel only; NULL: none): block 0 stores *step * mark_mul + mark_add into *mark when it starts -- the rollout
     * sets it on the first kernel of every decoder layer (mark_mul = layers, mark_add = layer) so that iadr1_weight_prefetch can pace itself by the decode step */
    unsigned* mark;
    unsigned mark_mul, mark_add;
} iadr1_side_out_t;

/* ---- dense contractions ----------------------------------------------------------------------------
 * C[M,N] (+)= act(A[M,K] . B[N,K]^T + bias[N]).  out_mode: 0 = bf16 store, 1 = fp32 store, 2 = fp32
 * accumulate (C += ...).  act: 0 none, 1 exact GELU.  K, lda, ldb multiples of 8; A/B 16-byte aligned.
 * Replaces every nn.Linear / the stride==kernel Conv3d of the path: TF:85-96 (ViT MLP), :116-122 (patch
 * embed), :148-150 (merger), :218-219 (ViT qkv/proj), :552-554 (decoder MLP), :634-637 (q/k/v/o),
 * :1386-1387 (lm_head), and their autograd backward (dgrad / wgrad run on transposed operands). */
int iadr1_gemm_nt_bf16(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, long long lda,
                       long long ldb, long long ldc, int out_mode, int act, iadr1_stream_t stream);
/* C[M,N] (fp32) += A[M,K] . B[N,K]^T for contractions with few output tiles and a long K (the weight gradients dW = dY^T . X of the narrow projections: 64-80 tiles
 * of 256 x 256 on 256 CUs): `ksplit` K slices per tile, fp32 partial tiles in `workspace` (iadr1_gemm_nt_splitk_workspace_bytes, 16-byte aligned), summed into C
 * in slice order by a second launch -- bit-reproducible, no atomics.  Same operand rules as iadr1_gemm_nt_bf16; N, ldc multiples of 4.  Replaces the autograd
 * weight-gradient GEMMs of TF:634-637 (q/k/v/o), :552-554 (MLP), :85-96,218-219 (ViT). */
long long iadr1_gemm_nt_splitk_workspace_bytes(int M, int N, int ksplit);
int iadr1_gemm_nt_splitk_acc_bf16(const void* A, const void* B, float* C, void* workspace, int M, int N, int K, long long lda, long long ldb,
                                  long long ldc, int ksplit, iadr1_stream_t stream);
/* Weight gradients WITHOUT transposed copies:  C[M,N] (fp32) += A[K,M]^T . B[K,N]  with A (= dY) and B (= X) row-major as the backward holds them, the contraction over
 * their K rows (the 256 x 256 kernel with hardware transpose reads out of LDS, ds_read_b64_tr_b16).  ksplit >= 2: K slices of whole 64-row tiles, fp32 partial tiles in
 * `workspace` (iadr1_gemm_nt_splitk_workspace_bytes(M, N, ksplit)), added to C in slice order; ksplit <= 1: one accumulating launch (workspace unused).  M, N >= 256 and
 * multiples of 8.  Bit-equal to iadr1_gemm_nt_bf16 (out_mode 2) / iadr1_gemm_nt_splitk_acc_bf16 on transposed copies.  Replaces autograd's weight-gradient matmuls of every
 * Linear on the path (TF:552-554, 660-700 backward; REF:train/stage_rl/trainer/sc_grpo_trainer.py:794 `loss.backward()` via accelerate). */
int iadr1_gemm_tn_acc_bf16(const void* A, const void* B, float* C, void* workspace, int M, int N, int K, long long lda, long long ldb, long long ldc,
                           int ksplit, iadr1_stream_t stream);
/* gate|up projection + SwiGLU in one launch (training / prefill shapes): GU[M, 2I] = A[M,K] . W[2I,K]^T is stored when GU != NULL (the backward
 * pass reads it), Aout[M, I] = bf16(silu(gate)) * up with gate = GU[:, :I], up = GU[:, I:], computed in the GEMM epilogue from the rounded gate|up
 * values: bit-identical to iadr1_gemm_nt_bf16 followed by iadr1_swiglu_fwd.  M %% 256 == 0, I %% 128 == 0.  Replaces TF:552-554
 * (down_proj(act_fn(gate_proj(x)) * up_proj(x))) up to the down projection. */
int iadr1_gemm_swiglu_bf16(const void* A, const void* W, void* GU, void* Aout, int M, int I, int K, long long lda, long long ldw,
                           long long ldgu, long long ldaout, iadr1_stream_t stream);
/* The same contraction over ROW BLOCKS of larger matrices: GEMM row r (0 <= r < M) is row (r / block) * block_stride + r % block of A, GU and Aout, counted from
 * the given base pointers (block: a power of two >= 16 dividing M; M %% 256 == 0).  Per row the arithmetic is iadr1_gemm_swiglu_bf16's, bit for bit.  Used by the
 * co-scheduled pass (iad-r1_amd/overlap.py) to rebuild the POLICY's gate|up and SwiGLU rows of a chunk of decode steps -- `block` steps of every sequence,
 * sequences `block_stride` rows apart in the sequence-major training arena -- on the side stream, so that the decode step's gate|up kernel need not store them
 * (the same mlp rows of REF:train/stage_rl/trainer/sc_grpo_trainer.py:722-735's policy forward, TF:552-554). */
int iadr1_gemm_swiglu_rows_bf16(const void* A, const void* W, void* GU, void* Aout, int M, int I, int K, long long lda, long long ldw,
                                long long ldgu, long long ldaout, int block, long long block_stride, iadr1_stream_t stream);
/* Decode-time skinny GEMM: Y[M,N] = X[M,K] . W[N,K]^T (M small, 64 rows per pass); HBM-bound weight stream,
 * K spread over 8-16 waves per block (+ optional grid split `ksplit`), no atomics.  out_mode 0: bf16 + bias;
 * 1: fp32 (logits); 2: fp32 partial slabs Y[ksplit][M][ldy] summed by iadr1_rmsnorm_fwd (x32 path); 3: fused
 * SwiGLU over a gate|up matrix packed with iadr1_pack_gateup_bf16 (N = 2*I rows in, Y is [M, I]).
 * ldx == 0: X is DECODE-PACKED (MFMA B-fragment order, rows padded to 64: Xp[m/64][k/32][(m%64)/16][m%16 + 16*((k%32)/8)][k%8],
 * see iadr1_pack_act_bf16) -- every X fragment load is then 1 KiB contiguous like the weights; iadr1_rmsnorm_fwd (ldy == 0),
 * iadr1_attn_decode (ldo == 0) and out_mode 3 here (ldy == 0) emit that layout directly, so the decode step never repacks.
 * Replaces the same Linears inside vLLM's decode step (REF:train/stage_rl/trainer/sc_grpo_trainer.py:667). */
int iadr1_gemm_skinny_bf16(const void* X, const void* W, void* Y, const void* bias, int M, int N, int K, long long ldx,
                           long long ldw, long long ldy, int out_mode, int ksplit, const iadr1_side_out_t* side, iadr1_stream_t stream);
/* W[N,K] row-major -> decode-packed MFMA-fragment order Wp[N/16][K/32][64 lanes][8] (what iadr1_gemm_skinny_bf16 reads:
 * every wave-level load of the weight stream is then 1 KiB contiguous).  N % 16 == 0, K % 32 == 0. */
int iadr1_pack_weight_bf16(const void* W, long long ldw, void* Wp, int N, int K, iadr1_stream_t stream);
/* gate|up matrix W[2I,K] -> decode-packed with gate/up 16-row tiles interleaved, for out_mode 3 (fused SwiGLU) of
 * iadr1_gemm_skinny_bf16: Y[M, I] = silu(X.Wgate^T) * (X.Wup^T).  I % 64 == 0. */
int iadr1_pack_gateup_bf16(const void* W, long long ldw, void* Wp, int I, int K, iadr1_stream_t stream);
/* Decode-step fusion of the q|k|v projection with iadr1_rope_kv_store: Y = X.Wqkv^T + b, rotary on the q and k heads (fp32 on the
 * bf16-rounded projections, TF:153-171), q heads -> q_out[M, >= Hq*D] (row stride ldq), new K / V rows -> the paged cache at slot[m]
 * (< 0: skipped).  Wp / bias_p come from iadr1_pack_qkv_rope_bf16: decode-packed with the rotary partners (d, d+64) of every q / k
 * head dealt into the same 16-column tile.  ldx == 0: X decode-packed.  One launch instead of two per layer of the rollout. */
int iadr1_gemm_qkv_rope_kv_bf16(const void* X, const void* Wp, const void* bias_p, void* q_out, const float* rope_cos,
                                const float* rope_sin, const long long* slot, void* kcache, void* vcache, int M, int Hq, int Hkv,
                                int D, int K, long long ldx, long long ldq, const iadr1_side_out_t* side, iadr1_stream_t stream);
int iadr1_pack_qkv_rope_bf16(const void* W, long long ldw, const void* bias, void* Wp, void* bias_p, int Hq, int Hkv, int D, int K,
                             iadr1_stream_t stream);
/* FP8 weights for the decode stream (BASELINE config 5: "fp8 weights"; opt-in, the rollout only -- the training passes and the log-probs of the loss stay
 * bf16).  iadr1_pack_weight_fp8: W[N,K] bf16 -> OCP e4m3 with ONE fp32 scale per output row (scale[n] = max_k |w[n][k]| / 448, round to nearest even),
 * decode-packed so that a lane's 16-byte load holds its 8 weights of TWO consecutive 32-deep k-steps: Wp8[N/16][K/64][64 lanes][16 B].  gateup_I > 0:
 * W is a gate|up matrix [2I, K], tiles interleaved as by iadr1_pack_gateup_bf16.  K % 64 == 0.
 * iadr1_gemm_skinny_fp8w: Y = (X . dequant(Wp8)^T) with the out_modes 0-3 of iadr1_gemm_skinny_bf16 (the fragments are widened to bf16 in registers --
 * exact, every e4m3 value is a bf16 value -- and fed to the bf16 MFMA; columns are multiplied by wscale in the epilogue).  N % 64 == 0.
 * Stated tolerance of the format: |w - dequant(quant(w))| <= 2^-4 |w| + scale * 2^-10 per weight (3 mantissa bits); the kernel adds nothing to it
 * (tests/test_hip_kernels.py::test_fp8_weight_gemm compares against the dequantised weights at the bf16 GEMM's own tolerance). */
int iadr1_pack_weight_fp8(const void* W, long long ldw, void* Wp8, float* scale, int N, int K, int gateup_I, iadr1_stream_t stream);
int iadr1_gemm_skinny_fp8w(const void* X, const void* Wp8, const float* wscale, void* Y, const void* bias, int M, int N, int K, long long ldx,
                           long long ldy, int out_mode, int ksplit, iadr1_stream_t stream);
/* X[M,K] row-major -> decode-packed activations Xp (buffer of roundup(M,64)*K elements; pad rows zeroed).  K % 32 == 0. */
int iadr1_pack_act_bf16(const void* X, long long ldx, void* Xp, int M, int K, iadr1_stream_t stream);
int iadr1_transpose_bf16(const void* in, long long ldi, void* out, long long ldo, int R, int C, iadr1_stream_t stream);

/* ---- RMSNorm (TF:65-79) --------------------------------------------------------------------------------
 * y = w * bf16((x [+ res]) * rsqrt(mean((x+res)^2) + eps)).  Exactly one of x (bf16) / x32 is given; x32 =
 * `nsplit` fp32 partial slabs [nsplit][T][ldx] from the split-K skinny GEMM (summed here, optional bias xbias).
 * res_out receives x+res (new residual stream), rstd the per-row statistic for the backward.  Any output
 * pointer may be NULL. */
int iadr1_rmsnorm_fwd(const void* x, const float* x32, int nsplit, const void* xbias, const void* res, void* res_out,
                      const void* w, void* y, float* rstd, int T, int H, long long ldx, long long ldr, long long ldy,
                      float eps, const iadr1_side_out_t* side, iadr1_stream_t stream);
/* dx = dres + d rmsnorm / dx ; dw (fp32, may be NULL) += sum_t dy * x * rstd.  The sum over token rows is two-stage and ORDERED (round 5): every block writes its
 * partial gain gradient to `workspace` (iadr1_rmsnorm_bwd_workspace_bytes, fp32, needed when dw != NULL), a second launch adds the partials in block order -- no float
 * atomics anywhere in the backward path, two runs of the same step give the same bits. */
long long iadr1_rmsnorm_bwd_workspace_bytes(int T, int H);
int iadr1_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, const void* dres, void* dx,
                      float* dw, float* workspace, int T, int H, long long ld, iadr1_stream_t stream);

/* ---- LayerNorm with bias (Qwen2-VL vision tower: nn.LayerNorm(eps=1e-6), TF:models/qwen2_vl/modeling_qwen2_vl.py:281,428-429)
 * y = ((x [+ res]) - mean) * rstd * w + b; backward returns dx (+ dres) and accumulates dw, db (fp32). */
int iadr1_layernorm_fwd(const void* x, const void* res, void* res_out, const void* w, const void* b, void* y, float* mean,
                        float* rstd, int T, int H, long long ldx, long long ldr, long long ldy, float eps, iadr1_stream_t stream);
long long iadr1_layernorm_bwd_workspace_bytes(int T, int H);
int iadr1_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, const void* dres,
                        void* dx, float* dw, float* db, float* workspace, int T, int H, long long ld, iadr1_stream_t stream);

/* ---- rotary embeddings: vision 2-D rotary TF:153-171 and decoder M-RoPE TF:557-599 -----------------------
 * In place on `nheads` consecutive heads of width D per token row; cos/sin fp32 [T, D/2] (the M-RoPE
 * t/h/w section select is folded into the table by the host).  backward != 0 applies the transpose. */
int iadr1_rope_inplace(void* x, long long ld, const float* cos_t, const float* sin_t, int T, int nheads, int D,
                       int backward, iadr1_stream_t stream);

/* ---- activations ---------------------------------------------------------------------------------------
 * SwiGLU TF:95-96,552-554 over gu = [gate | up] (width 2*I); exact GELU of the patch merger TF:143. */
int iadr1_swiglu_fwd(const void* gu, long long ldg, void* a, long long lda, int T, int I, iadr1_stream_t stream);
int iadr1_swiglu_bwd(const void* da, long long lda, const void* gu, long long ldg, void* dgu, long long ldd, int T, int I,
                     iadr1_stream_t stream);
int iadr1_gelu_fwd(const void* z, void* a, long long n, iadr1_stream_t stream);
int iadr1_gelu_bwd(const void* da, const void* z, void* dz, long long n, iadr1_stream_t stream);
/* GELU, tanh approximation: SigLIP vision MLP of the LLaVA-OneVision branch (transformers/models/siglip/modeling_siglip.py:310-322) */
int iadr1_gelu_tanh_fwd(const void* z, void* a, long long n, iadr1_stream_t stream);
int iadr1_gelu_tanh_bwd(const void* da, const void* z, void* dz, long long n, iadr1_stream_t stream);
/* QuickGELU x*sigmoid(1.702x): Qwen2-VL vision MLP (TF:models/qwen2_vl/modeling_qwen2_vl.py:293-301) */
int iadr1_quick_gelu_fwd(const void* z, void* a, long long n, iadr1_stream_t stream);
int iadr1_quick_gelu_bwd(const void* da, const void* z, void* dz, long long n, iadr1_stream_t stream);
/* out[n] (fp32) += sum_t dy[t][n]   (bias gradients); partial sums of <= 64 row groups in `workspace` (iadr1_colsum_workspace_bytes), added in group order: no atomics */
long long iadr1_colsum_workspace_bytes(int T, int N);
int iadr1_colsum_acc(const void* dy, long long ld, float* out, float* workspace, int T, int N, iadr1_stream_t stream);

/* ---- embedding lookup + image-feature scatter (TF:1204-1215 masked_scatter) -----------------------------------
 * out[t] = img_index[t] >= 0 ? img[img_index[t]] : E[ids[t]].
 * Backward = iadr1_rows_scatter_acc: dst[rows[u]] (fp32 [., H]) += sum_{k in [ptr[u], ptr[u+1])} src[idx[k]] (src bf16 [T, H]) for the U DISTINCT destination rows `rows`
 * (embedding rows by token id / image-embedding rows), the token rows of each listed by a host-built CSR; one block per destination row adds them in list order:
 * single writer, fixed order, no atomics (round 5: replaces iadr1_embed_bwd's float atomics). */
int iadr1_embed_fwd(const long long* ids, const int* img_index, const void* E, const void* img, void* out, int T, int H,
                    iadr1_stream_t stream);
int iadr1_rows_scatter_acc(const void* src, const long long* rows, const int* ptr, const int* idx, float* dst, int U, int H, iadr1_stream_t stream);

/* out[t] (bf16 [T,H]) = sum_k weights[k] * src[idx[k]], k in [ptr[t], ptr[t+1]) (fp32 sum, one rounding; an empty list gives zeros; weights NULL = 1).
 * Two uses: (1) the gradient of the rows the lm_head consumed (REF:...sc_grpo_trainer.py:505-513 only reads P-1 .. S-2 of each row; PA-SFT the
 * supervised positions) scattered back onto the token rows of the decoder output, several selected rows may share one token row; (2) the
 * LLaVA-OneVision feature packing (transformers/models/llava_onevision/modeling_llava_onevision.py:280-348: crop grid, unpad, bilinear shrink,
 * image_newline per row) and its transpose, as a sparse linear map built on the host.  ptr: [T+1], idx / weights: [ptr[T]], src rows contiguous. */
int iadr1_rows_gather_sum(const void* src, const int* ptr, const int* idx, const float* weights, void* out, int T, int H, iadr1_stream_t stream);

/* ---- casts ---------------------------------------------------------------------------------------------- */
int iadr1_cast_f32_to_bf16(const float* in, long long ldi, void* out, long long ldo, int R, int C, int Cpad,
                           iadr1_stream_t stream);
int iadr1_cast_bf16_to_f32(const void* in, float* out, long long n, iadr1_stream_t stream);
int iadr1_f32_bias_to_bf16(float* in_zeroed_after, const void* bias, void* out, long long R, int C, iadr1_stream_t stream);

/* ---- attention -------------------------------------------------------------------------------------------
 * Flash-style varlen attention over explicit segments [seg_start[i], seg_end[i]) of the flat token axis;
 * D in {128, 80}; GQA via Hq/Hkv.  Replaces flash-attn / eager attention at TF:186-208,225-291 (ViT,
 * non-causal windows) and TF:641-689 (decoder, causal, left padding = segments that start late).
 * lse: [Hq, T] fp32.  Backward also needs a delta scratch [Hq, T] fp32.
 * seg_prefix: NULL, or [nseg][4] int32 {prefix_start, prefix_len, child_first, child_count} for SHARED-PREFIX attention: the
 * keys of segment i are tokens [prefix_start, +prefix_len) (all visible) followed by its own tokens (causal).  Used for
 * the SC-GRPO policy / reference passes: the G completions of a prompt (REF:...sc_grpo_trainer.py:705-746 runs the prompt G
 * times inside [B*G, P+C] rows) attend to ONE copy of the prompt's keys/values -- same math, the prompt tokens go through
 * every layer once per group instead of G times.  Segments [child_first, +child_count) are those whose prefix is segment i
 * (their queries contribute to segment i's dK/dV); segments must be non-empty; max_seqlen covers own lengths only.
 * dkv_ws / head_splits (backward): NULL / 1, or an fp32 scratch of head_splits*T*Hkv*2*D floats: the q heads of each GQA group
 * are then split over head_splits dK/dV blocks (the query loop of a shared prefix block is G+1 segments long) and the partials
 * are summed in a fixed order -- deterministic, no atomics.
 * nseg_head / max_seqlen_tail: 0 / 0, or a hint about the segment lengths: the first nseg_head segments are at most max_seqlen long, the
 * others at most max_seqlen_tail (<= max_seqlen).  The two ranges are then launched with their own grids (no empty blocks for the short
 * segments; with shared prefixes the head split applies to the head range, which must hold every segment that has children). */
int iadr1_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* seg_start,
                   const int* seg_end, const int* seg_prefix, int nseg, int max_seqlen, int nseg_head, int max_seqlen_tail, int T, int Hq, int Hkv, int D, long long ldq,
                   long long ldk, long long ldv, long long ldo, int causal, float scale, iadr1_stream_t stream);
/* Chunked teacher-forced forward (round 5: the frozen reference's pass over the completion tokens runs UNDER the rollout, a few decode steps' worth of rows
 * at a time, REF:train/stage_rl/trainer/sc_grpo_trainer.py:737-743 next to :637-683).  Same kernel and arithmetic as iadr1_attn_fwd (causal), plus per segment
 * seg_view[i] = {q_first, q_count, blk_log2, blk_stride}: only the query rows [q_first, +q_count) of segment i are computed -- against its prefix and ALL its
 * own rows [0, seg_end - seg_start) written so far (causal) -- and logical row j of the segment lives at flat row
 *     seg_start + (j >> blk_log2) * blk_stride + (j & (2^blk_log2 - 1))
 * (time-blocked completion rows: the rows all sequences produce in the same 2^blk_log2 decode steps are contiguous, so every token-wise kernel and GEMM of a
 * chunk runs on one contiguous row range; blk_log2 = 31, blk_stride = 0: a plain contiguous segment).  A query row's result is bit-identical to the one
 * iadr1_attn_fwd computes for it over the whole sequence: same 64-key tiles in the same order.  max_q_count >= every q_count.  Prefix ranges are plain. */
int iadr1_attn_fwd_chunk(const void* q, const void* k, const void* v, void* o, float* lse, const int* seg_start, const int* seg_end,
                         const int* seg_prefix, const int* seg_view, int nseg, int max_q_count, int T, int Hq, int Hkv, int D, long long ldq,
                         long long ldk, long long ldv, long long ldo, float scale, iadr1_stream_t stream);
int iadr1_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   float* delta, void* dq, void* dk, void* dv, float* dkv_ws, int head_splits, const int* seg_start,
                   const int* seg_end, const int* seg_prefix, int nseg, int max_seqlen, int nseg_head, int max_seqlen_tail, int T, int Hq, int Hkv, int D, long long ldq, long long ldk, long long ldv,
                   long long ldo, long long lddo, long long lddq, long long lddk, long long lddv, int causal,
                   float scale, iadr1_stream_t stream);
/* Paged-KV decode attention + cache writes for the group rollout (vLLM's role at REF:...sc_grpo_trainer.py:
 * 343-358,667).  Pages hold 32 keys: K page [Hkv][32 x D in the MFMA-fragment order
 * attn_decode reads, private to these three entry points], V page [Hkv][D][32].  slot = page*32 + offset. */
int iadr1_attn_decode(const void* q, const void* kcache, const void* vcache, const int* block_table, const int* ctx_len,
                      void* o, int B, int Hq, int Hkv, int D, int max_pages, long long ldq, long long ldo, float scale, int seqs_per_group,
                      const iadr1_side_out_t* side, iadr1_stream_t stream);
/* seqs_per_group > 1 (a hint, results do not depend on it): sequences [g*n, (g+1)*n) share their prompt pages through the block table (group rollout); the blocks of
 * one prompt group are then placed on ONE XCD so that its L2 serves all but the first reader of a shared page. */
/* The same attention for prompt GROUPS with long shared prompts (LLaVA families, MHA decoders): sequences [g*G, (g+1)*G) share their first shared_pages[g] block-table
 * entries (the full prompt pages, as the rollout lays them out); one block per (group, kv head) reads those pages ONCE for all G * Hq/Hkv (<= 64) query rows, then
 * every sequence's private pages, and merges.  Same results as iadr1_attn_decode up to fp32 summation order; G times less K/V traffic for the shared part.
 * chunks > 1 (few groups x kv heads, very long prompts): the shared pages are split over `chunks` blocks in a first launch whose partial states go through
 * `ws` (fp32 [B/G][Hkv][chunks][64 or 32 or 16 rows][D + 2], rows = the tile-rounded G * Hq/Hkv) to a second launch that merges them in chunk order. */
int iadr1_attn_decode_group(const void* q, const void* kcache, const void* vcache, const int* block_table, const int* ctx_len, const int* shared_pages,
                            void* o, int B, int G, int Hq, int Hkv, int D, int max_pages, long long ldq, long long ldo, float scale, int chunks, float* ws,
                            const iadr1_side_out_t* side, iadr1_stream_t stream);
int iadr1_kv_store(const void* k, long long ldk, const void* v, long long ldv, const long long* slot, void* kcache,
                   void* vcache, int T, int Hkv, int D, iadr1_stream_t stream);
/* decode-step fusion of iadr1_rope_inplace (q,k heads) + iadr1_kv_store for one new token per sequence */
int iadr1_rope_kv_store(void* qkv, long long ld, const float* cos_t, const float* sin_t, const long long* slot, void* kcache,
                        void* vcache, int T, int Hq, int Hkv, int D, iadr1_stream_t stream);

/* ---- log-probs / losses -----------------------------------------------------------------------------------
 * logprob_rows: REF:train/stage_rl/trainer/sc_grpo_trainer.py:510-513 (log_softmax + gather, no temperature)
 * and PA-SFT cross entropy TF:loss/loss_utils.py:32-71 (logp = -CE; target < 0 is ignored).
 * dlogits_rows: g[row] * (onehot - softmax) in bf16.   grpo_loss: REF:...sc_grpo_trainer.py:746,796-798,816. */
int iadr1_logprob_rows(const float* logits, long long ld, const long long* targets, float* logp, float* lse, int R, int V,
                       iadr1_stream_t stream);
int iadr1_dlogits_rows(const float* logits, long long ld, const long long* targets, const float* lse, const float* g,
                       void* dl, long long ldd, int R, int V, iadr1_stream_t stream);
/* linear_logprob -- SURVEY section 8(b).5 "fused K15+K16, returns per-token logp and lse" (and linear_ce: logp = -CE, target < 0 ignored):
 *   logp[r] = log_softmax(H[r,:K] . W[V,K]^T)[targets[r]],  lse[r] = logsumexp of that row    (REF:505-513 lm_head + log_softmax + gather; PA-SFT CE TF:loss/loss_utils.py:32-71)
 * without the [M, V] logits reaching HBM: the GEMM's epilogue reduces every 64-column slice to (max, sum exp) and picks the target logit, a second launch
 * merges the slices in a fixed order (deterministic).  `workspace`: iadr1_linear_logprob_workspace_bytes(M, V) bytes, caller-owned, 8-byte aligned.
 * iadr1_linear_logprob_dlogits: the backward's first half -- the logits are recomputed and leave as dl[M, V] = g[r] * (onehot(target) - exp(x - lse[r])) in bf16
 * (the arithmetic of iadr1_dlogits_rows); dH = dl . W and dW += dl^T . H are iadr1_gemm_nt_bf16 calls.  K, ldh, ldw multiples of 8, ldd multiple of 8. */
long long iadr1_linear_logprob_workspace_bytes(int M, int V);
int iadr1_linear_logprob_fwd(const void* H, const void* W, const long long* targets, float* logp, float* lse, void* workspace, int M, int V, int K,
                             long long ldh, long long ldw, iadr1_stream_t stream);
int iadr1_linear_logprob_dlogits(const void* H, const void* W, const long long* targets, const float* lse, const float* g, void* dl, long long ldd,
                                 int M, int V, int K, long long ldh, long long ldw, iadr1_stream_t stream);
/* n_total_rows: number of sequences the batch mean runs over (>= N when the step is micro-batched) */
int iadr1_grpo_loss(const float* logp, const float* ref_logp, const float* adv, const int* mask, float beta, int n_total_rows,
                    float* dlogp, float* kl, float* row_loss, float* row_kl, int N, int C, iadr1_stream_t stream);

/* ---- optimizer (HF Trainer default AdamW + clip_grad_norm_, TF:trainer.py:1168) ------------------------------- */
/* out[0] = sum(g^2), deterministic (fixed-order two-stage reduction; scratch = 2048 floats) so that data-parallel
 * replicas derive a bit-identical clip coefficient */
int iadr1_sumsq(const float* g, long long n, float* partials2048, float* out, iadr1_stream_t stream);
int iadr1_adamw_flat(float* master, float* m, float* v, float* grad_zeroed_after, void* param_bf16, long long n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                     const float* norm2, float max_norm, iadr1_stream_t stream);

/* ---- sampler (vLLM SamplingParams(temperature, top_p, top_k), REF:...sc_grpo_trainer.py:353-358) -------------- */
long long iadr1_sample_workspace_bytes(int B);
int iadr1_sample_topk_topp(const float* logits, long long ld, long long* out, void* workspace, int B, int V, float temperature,
                           int top_k, float top_p, int suppress_token, unsigned long long seed, unsigned step,
                           const unsigned* step_ptr, const unsigned long long* seed_ptr, iadr1_stream_t stream);
/* step_ptr / seed_ptr: optional device-resident step counter / seed that replace the scalar arguments (a captured hipGraph freezes
 * kernel arguments: the rollout replays one graph for every decode step of every optimizer step). */

/* ---- device-resident rollout bookkeeping (one decode step = a fixed, graph-replayable launch sequence) ------
 * rope_table: cos/sin [B, half] for the current text positions (TF:1165-1176: pos = kv_len + rope_delta).
 * decode_advance: append sampled token (EOS -> finished, then pad; REF:...sc_grpo_trainer.py:680-683,722-726),
 * bump pos / ctx_len / cache slot / step.  B <= 256 (one block). */
int iadr1_rope_table(const int* pos, const float* inv_freq, float* cos_t, float* sin_t, int B, int half,
                     iadr1_stream_t stream);
int iadr1_decode_advance(const long long* sampled, long long* cur_tok, long long* out_tokens, int C, int* pos, int* ctx_len,
                         long long* slot, const int* block_table, int max_pages, int* finished, unsigned* step, int eos,
                         int pad, int B, int* all_done, const float* inv_freq, float* cos_t, float* sin_t, int half,
                         iadr1_stream_t stream);
/* wait_counter: stream-ordered wait until the device counter `counter` (the decode step counter iadr1_decode_advance bumps, on another stream) has reached
 * `target`: one wave polls it with agent-scope loads.  `timed_out` (optional, device int) is set to 1 when `timeout_ms` (<= 60000) elapsed first -- the wait then
 * returns anyway, a counter that never arrives cannot hang the queue.  The chunked reference pass (iadr1_attn_fwd_chunk) gates each chunk on the decode
 * step that produced its last target token with this instead of an event recorded between two hipGraph launches (0.06 ms per decode step cheaper). */
int iadr1_wait_counter(const unsigned* counter, unsigned target, int timeout_ms, int* timed_out, iadr1_stream_t stream);
/* weight_prefetch: ONE persistent launch for a whole group rollout (REF:train/stage_rl/trainer/sc_grpo_trainer.py:637-683) that pulls the decode step's weights
 * into the 256 MB memory-side cache just ahead of the launches that stream them.  Launch it on a CU-masked stream of its own (iadr1_stream_create_cu_mask), with
 * `n_blocks` = the CUs that stream owns.  `segs`: DEVICE table [n_units][n_seg][2] of (address, bytes; bytes %% 16 == 0) -- unit u = the weight segments of decoder
 * layer u in the order the step reads them.  `mark`: the decode step's progress word (iadr1_side_out_t.mark: step * n_units + layer, stored by the first kernel of
 * every layer).  On seeing mark m in [first_mark, last_mark] a block reads its share of unit (m + lead) %% n_units; it never runs further ahead than that, drops a
 * unit the decode step has passed, returns when the word exceeds last_mark, when `*stop` (optional DEVICE word) reaches `epoch` (the rollout stores its sequence number there
 * behind its last replay) or after `timeout_ms` (<= 60000).  Enqueue it WITHOUT a stream dependency: the word must read 0 when the launch starts (the rollout zeroes it behind
 * its last replay).  `nt`: 1 = non-temporal loads.  `status` (optional DEVICE int[4]): timed-out flag, units read, units dropped as stale, MiB read (the last three from block 0). */
int iadr1_weight_prefetch(const long long* segs, int n_units, int n_seg, const unsigned* mark, unsigned first_mark, unsigned last_mark, const unsigned* stop,
                          unsigned epoch, int lead, int nt, int n_blocks, int timeout_ms, int* status, iadr1_stream_t stream);
/* all_done (optional): 1 when every sequence has finished after this token -- the host polls it through pinned memory without draining the
 * queue (the reference's vLLM stops a request at EOS, REF:343-358).  inv_freq (optional, with cos_t / sin_t [B, half]): the rotary table
 * of the NEXT step's positions is written here, bit-identical to iadr1_rope_table on the bumped positions (one launch less per decode step). */

#ifdef __cplusplus
}
#endif
#endif
