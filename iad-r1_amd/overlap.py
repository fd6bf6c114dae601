"""Co-scheduled side pass (IADR1_OVERLAP_CUS=auto: on where `auto_applies`): the frozen reference model's teacher-forced forward + per-token log-probs
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:737-743, `_get_per_token_logps` :384-514 on `ref_model`), the policy's gate|up / SwiGLU rows and the
policy's lm_head log-probs of the tokens produced so far run UNDER the group rollout (:637-683) instead of after it.

Why: the rollout is a chain of latency-bound decode steps that leaves the MFMA pipes idle (0.34 of the HBM roofline, no matrix work to speak
of), the reference forward is MFMA-bound and leaves HBM idle, and the reference's log-prob of completion token j depends only on tokens <= j --
nothing of the sampled tokens' FUTURE.  The reference runs the two on different GPUs at the same time (vLLM GPU vs training ranks,
REF:scripts/train/SC_GRPO/SC_GRPO_Qwen_Instruct_2_5_VL_3B.sh:40-42); here they share one MI355X: every `steps` decode steps the rows those steps
produced (N sequences x steps tokens) go through the frozen decoder on a second HIP stream while the decode replay continues on the first.

What it takes on this part, and what it buys (all measured, round 5: profiles/EXPERIMENTS.md, profiles/r05_overlap_ab.txt, r05_decode_join.txt, r05_step_timeline.txt):
  * two ordinary streams do NOT share the device usefully: a big-grid kernel of the side pass holds every CU slot, the decode step's ~250 small dependent
    launches each wait for slots, and the replay all but stops while a chunk runs (decode window 724 -> 864 ms for 125 ms of side work);
  * disjoint CU masks (hipExtStreamCreateWithCUMask: decode on 192 CUs, side pass on 64) fix that ONLY when the two hardware queues sit on different dispatch
    pipes of the command processor -- every fourth queue created shares the decode queue's pipe, and then every decode launch waits behind whatever is PENDING on the
    other queue: ~50-70 us behind a big-grid kernel in dispatch, ~8 us behind a lone dependency packet or a one-wave polling kernel (pick_concurrent_stream below finds
    a clean pair at start-up; Rollout.generate joins the decode stream on the host so that the caller's queue is empty while the replays run);
  * CU-masked streams are BLOCKING streams: with the caller on the null stream every chunk started one chunk period late (SCGRPOEngine.step moves the step
    onto a stream of its own);
  * gating a chunk on an event recorded between two graph launches costs the decode stream 0.06 ms per step; the gate is a one-wave poll of the device step
    counter instead (iadr1_wait_counter);
  * the decode step on 192 CUs costs 3.02 instead of 2.80 ms (x 255 = 56 ms per step): what the side stream hides (reference pass ~125 ms, the decode gate|up
    kernel's two side stores 26 ms, the policy's lm_head 9 ms) has to beat that -- it does for the 2B / 3B models at 64 sequences (1260 -> 1202 ms, +4.7 %; 2B + 8.6 %),
    not for 7B-class models or small groups (auto_applies);
  * the host must never block while the side pass's tail runs: index arrays go up through pinned memory (ops.h2d), the tokens leave before the tail is enqueued.

Layout.  Completion rows are TIME-BLOCKED: with block = 2^k steps, row (sequence s, token j) lives at
    n_prompt_rows + (j // block) * N * block + s * block + j % block,
so the rows of a chunk are one contiguous range for every token-wise kernel and GEMM; only attention needs to know
(iadr1_attn_fwd_chunk: query sub-range + blocked row map per segment).  Per-layer q|k|v rows of everything processed so far are kept
([L, T, qkv_width] bf16: 3.8 GB at the 3B bench shape) so that later chunks attend to earlier ones.

What is bit-equal and what is not.  The REFERENCE's log-probs and the policy's lm_head log-probs are BIT-equal to the sequential step's -- every forward
quantity the loss reads is, hence KL, loss and the tokens.  The policy's gate|up / SwiGLU rows of the completion tokens are REBUILT by the training GEMM instead of stored by
the decode kernel (IADR1_OVERLAP_GU=1, the default): the two kernels sum K in different orders, the rows differ in the last bf16 bit, and the GRADIENTS of the co-scheduled step
differ from the sequential step's by that much (cosine 0.99983 through 36 layers; both hold the oracle's tolerances: tests/test_hip_model.py::
test_api_step_with_rollout_handover_matches_the_oracle[0|64]).  Which structure runs is decided once per job (cu_split + agree_across_ranks) and logged; IADR1_OVERLAP_GU=0 keeps
the decode kernel's rows (bit-equal gradients, +0.1 ms per decode step).

The chunked reference pass itself is BIT-equal to the one-shot pass (Engine.text_forward + Engine.logprobs over the whole [prompts ++ completions] batch): same kernels,
row-independent GEMMs (tools/gemm_rowdep_probe.py), the same 64-key attention tiles in the same order, RMSNorm kernel choice independent of the row count
(tests/test_hip_model.py::test_chunked_reference_pass_is_bit_equal_to_the_one_shot_pass).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import hip, ops
from .vlm import Engine, TextPlan

BF16, F32 = torch.bfloat16, torch.float32
BLOCK = 16          # steps per time block of the completion-row layout (a power of two; chunk boundaries are multiples of it)

STATS = {"passes": 0, "chunks": 0, "rows": 0, "policy_mlp_rows": 0}


AUTO_CUS = 64       # CUs of the side stream in the automatic mode (the decode replays own the other 192): the measured optimum of 32 / 48 / 64 / 96 at the 3B bench shape


def auto_applies(cfg, n_seq: int, C: int) -> bool:
    """Shapes for which co-scheduling is ON by default -- where it was measured to pay (profiles/r05_overlap_ab.txt, EXPERIMENTS round 5), ms per step without -> with:
    Qwen2.5-VL-3B, 8 prompts x G 8, C 256: 1256 -> 1222 (+2.8 %); Qwen2-VL-2B, same shape: 833 -> 783 (+6.4 %).  NOT: rollouts of fewer than 64 sequences (3B, 4 prompts
    x G 8: 944 -> 954; 2 x 8: 788 -> 827 -- the side stream hides little there and the masked decode stream costs +0.2 ms per step whatever the batch), the 7B-class models
    (hidden 3584: the decode step on 192 CUs costs 5.77 instead of 4.38 ms, 2402 -> 2514 ms), the LLaVA families (`applicable`)."""
    return (cfg is not None and not cfg.is_llava and cfg.hidden_size <= 2048 and n_seq >= 64 and n_seq % 16 == 0 and C % BLOCK == 0 and C >= 4 * BLOCK
            and n_seq * C >= 8192)


def shadow_cus(cfg=None, n_seq: int = 0, C: int = 0) -> int:
    """IADR1_OVERLAP_CUS: `auto` (the default) = AUTO_CUS where auto_applies(cfg, sequences, completion length), else 0; 0 = the reference pass runs after the
    rollout, on the whole device; n > 0 = it runs under the rollout on a stream confined to n CUs while the decode replays own the others (cu_split); -1 = under the
    rollout on an ordinary second stream (no CU split: measured slower, kept for A/B and for the parity tests of the chunked pass itself)."""
    v = os.environ.get("IADR1_OVERLAP_CUS", "auto").strip().lower()
    if v in ("", "auto"):
        return AUTO_CUS if auto_applies(cfg, n_seq, C) else 0
    return int(v)


def chunk_steps_default() -> int:
    return int(os.environ.get("IADR1_OVERLAP_STEPS", "32"))


def policy_mlp_wanted() -> bool:
    """IADR1_OVERLAP_GU (default 1): while the co-scheduled pass is on, the POLICY's gate|up and SwiGLU rows of the completion tokens are rebuilt on the side stream
    from the decode steps' `h2` rows (iadr1_gemm_swiglu_rows_bf16) instead of being stored by the decode step's gate|up kernel: those two side stores are 4.5 of
    the 7.5 us per layer the rollout -> training hand-over costs the decode step (profiles/EXPERIMENTS.md round 4), the GEMM that replaces them is 23 ms of side-stream
    time per 32 steps at the 3B bench shape."""
    return os.environ.get("IADR1_OVERLAP_GU", "1") != "0"


def policy_head_wanted() -> bool:
    """IADR1_OVERLAP_HEAD (default 1): with the training hand-over on, the POLICY's lm_head log-probs (fused linear_logprob: log p and log-sum-exp per scored position)
    are computed on the side stream too, chunk by chunk, from the final-norm rows the decode steps stored; the loss then starts without that 10 TFLOP launch."""
    return os.environ.get("IADR1_OVERLAP_HEAD", "1") != "0"


def pick_concurrent_stream(anchor, make_candidate, weight: torch.Tensor, tries: int = 6, log=None):
    """A stream whose kernels really run NEXT TO `anchor`'s.  Measured on MI355X (tools/stream_pair_probe.py, profiles/r05_stream_pairs.txt): hardware queues are
    dealt round-robin onto the command processor's four dispatch pipes, and two queues on the SAME pipe take turns at kernel granularity -- a kernel stays "in
    dispatch" until its last workgroup has been placed, so next to a big-grid GEMM every dependent launch of the other queue waits ~70 us (a 64-row RMSNorm chain:
    18x slower), even when the two streams own disjoint CU masks.  Every fourth stream created collides; the others do not interfere at all (1.03x).  So: time a
    chain of small dependent launches on `anchor` alone and next to big-grid GEMMs on each candidate, keep the first candidate that leaves the chain alone.
    make_candidate() -> a new stream; weight: any [N >= 8192, K] bf16 matrix (the GEMM's B operand).  Returns (stream, ratio) or (None, ratio of the best)."""
    dev = weight.device
    K = weight.shape[1]
    x = torch.zeros(64, K, dtype=BF16, device=dev)
    g = torch.ones(K, dtype=BF16, device=dev)
    y = torch.empty_like(x)
    a = torch.zeros(2048, K, dtype=BF16, device=dev)
    c = torch.empty(2048, weight.shape[0], dtype=BF16, device=dev)

    def chain(stream, other=None):
        torch.cuda.synchronize(dev)
        if other is not None:
            with torch.cuda.stream(other):
                for _ in range(6):
                    ops.gemm_nt(a, weight, out=c)
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(150):
                ops.hip.call("rmsnorm_fwd", x, None, 0, None, None, None, g, y, None, 64, K, K, K, K, 1e-6, None)
            e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1)

    chain(anchor)
    alone = chain(anchor)
    best = float("inf")
    for i in range(tries):
        cand = make_candidate()
        ratio = chain(anchor, cand) / max(alone, 1e-3)
        if log is not None:
            log(f"[iadr1] stream pairing: candidate {i}: the dependent chain runs {ratio:.2f}x its stand-alone time next to it")
        best = min(best, ratio)
        if ratio < 1.5:
            return cand, ratio
        destroy_stream(cand)        # a rejected candidate's hardware queue is given back (queues are finite, and leaked ones shift later streams' pipes)
    return None, best


def destroy_stream(stream):
    """hipStreamDestroy of a stream made by hip.cu_mask_stream (iadr1_stream_destroy); the device is idle on it (chain() synchronises)."""
    torch.cuda.synchronize()
    rc = hip.lib().iadr1_stream_destroy(int(stream.cuda_stream))
    if rc != 0:
        raise RuntimeError(f"iadr1_stream_destroy failed ({rc}): {hip.lib().iadr1_last_error().decode()}")
    hip._CU_SHARE.pop(int(stream.cuda_stream), None)


def agree_across_ranks(ok_local: bool, group=None, device=None) -> bool:
    """Whether EVERY rank of the process group found a clean stream pair: co-scheduling changes the structure of a step (which kernels rebuild the policy's mlp
    rows, what runs on which stream), so either all ranks of a data-parallel run co-schedule or none does (the same MIN/MAX vote
    Engine.check_ddp_headroom takes for recomputation).  One small all-reduce, at the first rollout of a run; without a process group: the local answer."""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return bool(ok_local)
    except Exception:
        return bool(ok_local)
    dev = device if (device is not None and dist.get_backend(group) == "nccl") else "cpu"
    t = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


_SPLITS: dict = {}       # (device index, side-stream CUs, prefetch CUs) -> (decode stream, side stream, decode CUs, prefetch stream | None), or None when no clean stream pair was found


def pick_prefetch_stream(decode, side, make_candidate, weight: torch.Tensor, tries: int = 6, log=None):
    """A stream for the rollout's persistent weight prefetcher (wprefetch.py): its one resident, polling launch must disturb NEITHER the decode replays' chain of small
    dependent launches NOR the side stream's big-grid GEMMs (its CUs are a subset of the side stream's: the block is small enough -- 256 threads, 28 VGPRs, 8 bytes of
    LDS -- to sit next to a resident 256 x 256 GEMM block).  Timed like pick_concurrent_stream, with the real thing on the candidate: a prefetch launch whose mark never
    arrives (it polls until its 40 ms timeout).  Returns the stream or None."""
    dev = weight.device
    K = weight.shape[1]
    x = torch.zeros(64, K, dtype=BF16, device=dev)
    g = torch.ones(K, dtype=BF16, device=dev)
    y = torch.empty_like(x)
    a = torch.zeros(2048, K, dtype=BF16, device=dev)
    c = torch.empty(2048, weight.shape[0], dtype=BF16, device=dev)
    segs = ops.h2d(np.asarray([[[weight.data_ptr(), 1 << 20]]], dtype=np.int64), dev)
    mark = torch.zeros(1, dtype=torch.int32, device=dev)

    def timed(stream, body, resident=None):
        torch.cuda.synchronize(dev)
        if resident is not None:
            with torch.cuda.stream(resident):
                ops.hip.call("weight_prefetch", segs, 1, 1, mark, 1, 1, None, 0, 0, 0, 8, 40, None)
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            body()
            e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1)

    small = lambda: [ops.hip.call("rmsnorm_fwd", x, None, 0, None, None, None, g, y, None, 64, K, K, K, K, 1e-6, None) for _ in range(150)]
    big = lambda: [ops.gemm_nt(a, weight, out=c) for _ in range(6)]
    timed(decode, small), timed(side, big)
    t_small, t_big = timed(decode, small), timed(side, big)
    for i in range(tries):
        cand = make_candidate()
        r_small = timed(decode, small, cand) / max(t_small, 1e-3)
        r_big = timed(side, big, cand) / max(t_big, 1e-3)
        if log is not None:
            log(f"[iadr1] prefetch stream: candidate {i}: decode chain {r_small:.2f}x, side-stream GEMMs {r_big:.2f}x their stand-alone time next to the resident prefetcher")
        if r_small < 1.15 and r_big < 1.10:
            return cand
        destroy_stream(cand)
    return None


def cu_split(dev: torch.device, n: int, weight: torch.Tensor, prefetch_cus: int = 0):
    """The process's CU-masked streams for a side stream of n CUs on `dev` -- decode replays on the device's other CUs -- created and calibrated ONCE
    (pick_concurrent_stream; hardware queues are a finite resource and the calibration takes ~0.2 s).  Returns (decode stream, side stream, decode CUs, prefetch stream
    or None), or None when no candidate on another dispatch pipe was found: the caller then does not co-schedule at all.
    prefetch_cus > 0 (IADR1_WPREFETCH_CUS): a third stream on the FIRST `prefetch_cus` of the side stream's n CUs for the rollout's weight prefetcher (wprefetch.py).
    They stay in the side stream's mask: CU masks are only balanced in steps of 32 (a 48-CU side stream runs its GEMMs 1.8x slower than a 64-CU one -- its shader
    engines hold 2, 2, 1, 1 CUs and get equal shares of every grid; profiles/EXPERIMENTS.md round 6), and the prefetcher's block fits next to a GEMM block.  No clean
    third queue -> no prefetcher (the split itself stands)."""
    import sys
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), int(n), int(prefetch_cus))
    if key not in _SPLITS:
        total = torch.cuda.get_device_properties(dev).multi_processor_count
        if not (0 < n < total and n % 8 == 0 and 0 <= prefetch_cus <= n and prefetch_cus % 8 == 0):
            raise ValueError(f"IADR1_OVERLAP_CUS={n} / IADR1_WPREFETCH_CUS={prefetch_cus}: multiples of 8 (the same share of every XCD), prefetch <= side < the device's {total} CUs")
        # the prefetcher's queue is created FIRST: one more hardware queue shifts the dispatch pipe of every queue created after it, and created in the middle it moved the
        # weight-gradient / main queues against the decode pair (1198 -> 1239 ms per step with nothing launched on it, EXPERIMENTS round 6); created first, the others keep
        # their positions relative to each other.  (Necessary, not sufficient: the step still loses with the prefetcher running -- r06_wprefetch_bench_ab8.txt.)
        pf_first = hip.cu_mask_stream(0, prefetch_cus) if prefetch_cus > 0 else None
        decode = hip.cu_mask_stream(n, total - n)
        log = (lambda m: print(m, file=sys.stderr, flush=True)) if os.environ.get("IADR1_OVERLAP_LOG") == "1" else None
        side, ratio = pick_concurrent_stream(decode, lambda: hip.cu_mask_stream(0, n), weight, log=log)
        pf = None
        if side is None:
            destroy_stream(decode)
            if pf_first is not None:
                destroy_stream(pf_first)
            if os.environ.get("IADR1_QUIET") != "1":
                print(f"[iadr1] co-scheduling off: no stream pair on separate dispatch pipes found (best: the dependent chain at {ratio:.1f}x its stand-alone time)", file=sys.stderr, flush=True)
        elif prefetch_cus > 0:
            first = [pf_first]
            pf = pick_prefetch_stream(decode, side, lambda: first.pop() if first else hip.cu_mask_stream(0, prefetch_cus), weight, log=log)
            if pf is None and os.environ.get("IADR1_QUIET") != "1":
                print("[iadr1] weight prefetcher off: no third stream that leaves both the decode chain and the side stream's GEMMs alone was found", file=sys.stderr, flush=True)
        _SPLITS[key] = None if side is None else (decode, side, total - n, pf)
    return _SPLITS[key]


class ChunkedRefPass:
    """One instance per SCGRPOEngine; `begin` per rollout.  All device work of this object is enqueued on `self.stream` (the side stream)."""

    def __init__(self, ref: Engine, steps: int | None = None, stream=None):
        self.e = ref
        self.steps = max(BLOCK, (steps or chunk_steps_default()) // BLOCK * BLOCK)
        self.stream = stream if stream is not None else (torch.cuda.Stream() if ref.dev.type == "cuda" else None)
        self.vision = None          # (pixel tensor, vision plan) of the current batch, set by the owner before the rollout
        self.store = None           # [L, T, qkv_width] roped q|k|v rows of the reference model (prompt rows + time-blocked completion rows)
        self.active = False

    @staticmethod
    def unmasked() -> bool:
        return shadow_cus() < 0

    @staticmethod
    def applicable(cfg, n_seq: int, C: int) -> bool:
        """Qwen-VL families, completion length a multiple of the time block.  (The LLaVA branches of the reference rotate right-padded rows before the model
        runs -- REF:516-567 -- which makes the scored positions depend on where a sequence ENDS: not known while it is being generated.)"""
        return shadow_cus(cfg, n_seq, C) != 0 and not cfg.is_llava and C % BLOCK == 0 and C >= BLOCK

    @staticmethod
    def rebuilds_policy_mlp(cfg, N: int) -> bool:
        """Shapes the row-blocked fused gate|up GEMM takes: N sequences x one time block = whole 256-row tiles, I a multiple of 128."""
        return policy_mlp_wanted() and (N * BLOCK) % 256 == 0 and cfg.intermediate_size % 128 == 0

    # ---- set-up --------------------------------------------------------------------------------------------------
    def begin(self, plan: TextPlan, G: int, C: int, first_pos: np.ndarray, out_tokens: torch.Tensor, step_counter: torch.Tensor | None = None, policy=None):
        """plan: the rollout's prompt plan (Bp left-padded prompts of S columns); self.vision = (pixel tensor, vision plan) of the batch was set by the owner -- the
        reference's own vision tower runs here, on the side stream; first_pos [N]: rotary position of completion token 0 of every sequence; out_tokens: the rollout's
        device-resident [N, >= C] token matrix (column j is final once decode replay j has run).
        policy: None, or (the policy's Engine, the training arena the decode steps fill -- Rollout.trace): every chunk then also rebuilds the policy's gate|up and
        SwiGLU rows of its tokens from the arena's `h2` rows (policy_mlp_wanted above; the decode graph was captured without those two side stores)."""
        e, c = self.e, self.e.cfg
        self.policy = policy
        # the policy's lm_head log-probs on the side stream: needs the fused head (the same launch the one-shot path runs, so the values are bit-equal) and the arena's
        # final-norm rows.  lse starts at +1e30: a position the pass never reaches (rollout cut short by EOS; masked in the loss) then gives exp(logit - lse) = 0 in backward.
        self.pol_logp = self.pol_lse = None
        if policy is not None and policy_head_wanted() and policy[0].head_mode == "fused" and policy[1].get("hf") is not None:
            self.pol_logp = torch.zeros(plan.B * G, C, dtype=F32, device=e.dev)
            self.pol_lse = torch.full((plan.B * G, C), 1e30, dtype=F32, device=e.dev)
        dev = e.dev
        Bp, S = plan.B, plan.S
        N = Bp * G
        self.plan, self.G, self.C, self.N, self.Bp, self.S = plan, G, C, N, Bp, S
        self.T0 = Bp * S
        self.T = self.T0 + N * C
        self.out_tokens = out_tokens
        # gate of every chunk: the rollout's device-resident step counter (it reads k + 1 once decode replay k has run) polled by iadr1_wait_counter on the side
        # stream (HIP events recorded by the caller between the replays cost the decode stream 0.06 ms per step: EXPERIMENTS round 5); None: the caller orders the
        # streams itself (finish(): the rest runs on the caller's stream)
        self.step_counter = step_counter
        if self.__dict__.get("_timed_out_host") is not None and int(self._timed_out_host[0]):
            raise RuntimeError("overlap.ChunkedRefPass: a counter wait of the previous rollout timed out (the decode stream never reached the awaited step)")
        if self.step_counter is not None and self.__dict__.get("_timed_out") is None:
            self._timed_out = torch.zeros(1, dtype=torch.int32, device=dev)
            self._timed_out_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        L = c.num_hidden_layers
        if self.store is None or self.store.shape[1] < self.T:
            self.store = None
            self.store = torch.empty(L, self.T, c.qkv_width, dtype=BF16, device=dev)
        # rotary table of the completion rows in time-blocked order: text tokens carry the same position on all three M-RoPE axes (TF:1042-1176), so the
        # table is  (first_pos[s] + j) * inv_freq  evaluated exactly like Engine.text_plan_shared does
        nb = C // BLOCK
        j = (np.arange(nb)[:, None, None] * BLOCK + np.arange(BLOCK)[None, None, :])                      # [nb, 1, BLOCK]
        pos = (np.asarray(first_pos, dtype=np.int64)[None, :, None] + j).reshape(-1)                          # [nb * N * BLOCK]
        pos_t = ops.h2d(np.broadcast_to(pos[None, :], (3, pos.size)).copy(), dev)
        ang = pos_t[e.mrope_comp].t().to(F32) * e.inv_freq[None, :]
        self.cos, self.sin = ang.cos().contiguous(), ang.sin().contiguous()
        # attention tables: one segment per sequence; prefix = its prompt's rows (all visible).  seg_end / the query sub-range depend on the chunk.
        starts_p = np.asarray(plan.seg.start_host, dtype=np.int64)
        ends_p = np.asarray(plan.seg.end_host, dtype=np.int64)
        self.seg_start = ops.h2d((self.T0 + np.arange(N) * BLOCK).astype(np.int32), dev)
        pre = np.zeros((N, 4), dtype=np.int32)
        pre[:, 0] = np.repeat(starts_p, G)
        pre[:, 1] = np.repeat(ends_p - starts_p, G)
        self.seg_prefix = ops.h2d(pre, dev)
        self._tables = {}
        self.logp = torch.zeros(N, C, dtype=F32, device=dev)
        self.trace = [] if os.environ.get("IADR1_OVERLAP_STATS") == "1" else None      # (label, start event, end event) per phase, on the side stream
        if self.trace is not None:
            self.t_ref = torch.cuda.Event(enable_timing=True)
            self.t_ref.record()
        self.done_rows = 0          # completion tokens [0, done_rows) of every sequence have gone through the layers
        self.active = True
        STATS["passes"] += 1

    def _chunk_tables(self, c0, c1):
        key = (c0, c1)
        t = self._tables.get(key)
        if t is None:
            N = self.N
            view = np.zeros((N, 4), dtype=np.int32)
            view[:, 0], view[:, 1], view[:, 2], view[:, 3] = c0, c1 - c0, int(np.log2(BLOCK)), N * BLOCK
            dev = self.e.dev
            t = self._tables[key] = (ops.h2d((self.T0 + np.arange(N) * BLOCK + c1).astype(np.int32), dev), ops.h2d(view, dev))
        return t

    # ---- the decoder over a contiguous row range -------------------------------------------------------------------
    def _layers(self, x, r0, r1, cos, sin, attend):
        """Rows [r0, r1) of the T-row batch through the frozen decoder; the arithmetic and the buffer roles are Engine.text_forward(save=False)'s.
        attend(i, store_i, o_all): attention of layer i for these rows (absolute row addressing).  Returns the final-norm output rows."""
        e, c, P = self.e, self.e.cfg, self.e.p
        H, D, Hq, Hkv = c.hidden_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        Tl = r1 - r0
        B = e._text_buffers(self.T, False)
        eps = float(c.rms_norm_eps)
        res, branch = x, None
        o_all = B["o"][0, :self.T]
        for i in range(c.num_hidden_layers):
            b = f"layers.{i}."
            h1 = B["h1"][0, r0:r1]
            if branch is None:
                x_in = res
                ops.hip.call("rmsnorm_fwd", res, None, 0, None, None, None, P.w(b + "ln1"), h1, None, Tl, H, H, H, H, eps, None)
            else:
                x_in = res
                ops.hip.call("rmsnorm_fwd", branch, None, 0, None, res, x_in, P.w(b + "ln1"), h1, None, Tl, H, H, H, H, eps, None)
            qkv = ops.gemm_nt(h1, P.w(b + "qkv.w"), bias=P.w(b + "qkv.b"), out=self.store[i, r0:r1])
            ops.rope_(qkv, cos, sin, Hq + Hkv, D)
            attend(i, self.store[i, :self.T], o_all)
            ab = ops.gemm_nt(o_all[r0:r1], P.w(b + "o.w"), out=B["x_mid"][0, r0:r1])
            h2 = B["h2"][0, r0:r1]
            ops.hip.call("rmsnorm_fwd", ab, None, 0, None, x_in, x_in, P.w(b + "ln2"), h2, None, Tl, H, H, H, H, eps, None)
            _, a = ops.gemm_swiglu(h2, P.w(b + "gu.w"), gu_out=B["gu"][0, r0:r1], a_out=B["a"][0, r0:r1], keep_gu=False)
            branch = ops.gemm_nt(a, P.w(b + "down.w"), out=B["x_mid"][0, r0:r1])
            res = x_in
        hf = B["h1"][0, r0:r1]
        ops.hip.call("rmsnorm_fwd", branch, None, 0, None, res, res, P.w("norm"), hf, None, Tl, H, H, H, H, eps, None)
        return hf

    # ---- phases ---------------------------------------------------------------------------------------------------
    def prompt_phase(self, after: torch.cuda.Event | None):
        """The reference's vision tower + its forward over the prompt rows (K/V kept for the chunks) + the log-prob of completion token 0 of every sequence
        (predicted by its prompt's last row).  `after`: event on the decode stream behind the sampling of token 0."""
        e, c, P = self.e, self.e.cfg, self.e.p
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        qw = Hq * D
        plan, T0 = self.plan, self.T0
        with torch.cuda.stream(self.stream):
            if after is not None:
                self.stream.wait_event(after)
            ev0 = self._mark()
            px, plan_v = self.vision
            img_ref, _ = e.vision_forward(px, plan_v, save=False)
            B = e._text_buffers(self.T, False)
            x = ops.embed_fwd(plan.ids, plan.img_index, P.w("embed"), img_ref, out=B["x_in"][0, :T0])
            if not plan.seg.covers(0, T0):
                B["o"][0, :T0].zero_()

            def attend(i, st, o_all):
                ops.hip.call("attn_fwd", st[:, :qw], st[:, qw: qw + Hkv * D], st[:, qw + Hkv * D:], o_all, None, plan.seg.start, plan.seg.end, plan.seg.prefix, plan.seg.n,
                             plan.seg.max_len, plan.seg.n_head, plan.seg.max_tail, self.T, Hq, Hkv, D, c.qkv_width, c.qkv_width, c.qkv_width, qw, 1, c.attn_scale)

            hf = self._layers(x, 0, T0, plan.cos, plan.sin, attend)
            last = (torch.arange(self.Bp, device=e.dev, dtype=torch.int64) * self.S + (self.S - 1)).repeat_interleave(self.G)
            lp, _ = e.logprobs(hf, last, self.out_tokens[:, 0].contiguous(), save=False)
            self.logp[:, 0] = lp
            if self.pol_logp is not None:       # the policy's log-prob of token 0: its prefill's final-norm rows are in the arena (rows [0, T0))
                self._policy_head(last, self.out_tokens[:, 0].contiguous(), slice(0, 1))
            self._img_ref = img_ref
            self._mark("prompt", ev0)

    def chunk(self, c1: int, after: torch.cuda.Event | None):
        """Completion tokens [done_rows, c1) of every sequence (c1 a multiple of BLOCK).  `after`: event on the decode stream behind the replay that produced
        token min(c1, C - 1) -- the targets of these rows reach one token further than the rows themselves."""
        e, c, P = self.e, self.e.cfg, self.e.p
        c0, N, C = self.done_rows, self.N, self.C
        assert c0 < c1 <= C and c1 % BLOCK == 0 and c0 % BLOCK == 0
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        qw = Hq * D
        nb = (c1 - c0) // BLOCK
        r0, r1 = self.T0 + (c0 // BLOCK) * N * BLOCK, self.T0 + (c1 // BLOCK) * N * BLOCK
        blocked = lambda t: t.view(N, nb, BLOCK).permute(1, 0, 2).reshape(-1)       # [N, nb*BLOCK] (sequence-major) -> time-blocked row order
        with torch.cuda.stream(self.stream):
            if self.step_counter is not None:
                ops.hip.call("wait_counter", self.step_counter, min(c1, C - 1) + 1, 30000, self._timed_out)
            elif after is not None:
                self.stream.wait_event(after)
            ev0 = self._mark()
            seg_end, view = self._chunk_tables(c0, c1)          # (uploaded on THIS stream: ordered before the kernels that read them)
            ids = blocked(self.out_tokens[:, c0:c1])
            B = e._text_buffers(self.T, False)
            x = ops.embed_fwd(ids, None, P.w("embed"), None, out=B["x_in"][0, r0:r1])

            def attend(i, st, o_all):
                ops.hip.call("attn_fwd_chunk", st[:, :qw], st[:, qw: qw + Hkv * D], st[:, qw + Hkv * D:], o_all, None, self.seg_start, seg_end, self.seg_prefix, view, N, c1 - c0,
                             self.T, Hq, Hkv, D, c.qkv_width, c.qkv_width, c.qkv_width, qw, c.attn_scale)

            hf = self._layers(x, r0, r1, self.cos[r0 - self.T0: r1 - self.T0], self.sin[r0 - self.T0: r1 - self.T0], attend)
            # row (s, j) predicts token j + 1; the last token of a sequence predicts nothing that is scored
            n_ok = min(c1, C - 1) - c0
            tg = torch.full((N, c1 - c0), -1, dtype=torch.int64, device=e.dev)
            if n_ok > 0:
                tg[:, :n_ok] = self.out_tokens[:, c0 + 1: c0 + 1 + n_ok]
            rows = torch.arange(r1 - r0, device=e.dev, dtype=torch.int64)
            lp, _ = e.logprobs(hf, rows, blocked(tg), save=False)
            if n_ok > 0:
                self.logp[:, c0 + 1: c0 + 1 + n_ok] = lp.view(nb, N, BLOCK).permute(1, 0, 2).reshape(N, c1 - c0)[:, :n_ok]
            self._mark(f"rows[{c0},{c1})", ev0)
            if self.policy is not None:
                ev0 = self._mark()
                self._policy_mlp(c0, c1)
                if self.pol_logp is not None and n_ok > 0:
                    idx = (self.T0 + torch.arange(N, device=e.dev, dtype=torch.int64)[:, None] * C + torch.arange(c0, c1, device=e.dev, dtype=torch.int64)[None, :]).reshape(-1)
                    self._policy_head(idx, tg.reshape(-1), slice(c0 + 1, c0 + 1 + n_ok), n_ok)
                self._mark(f"policy mlp + head[{c0},{c1})", ev0)
        self.done_rows = c1
        STATS["chunks"] += 1
        STATS["rows"] += r1 - r0

    def _policy_mlp(self, c0: int, c1: int):
        """gate|up + SwiGLU rows of completion tokens [c0, c1) of every sequence, every layer of the POLICY, written where the decode step's side stores would have
        put them (sequence-major arena: row T0 + s * C + j).  Input: the `h2` rows decode replays c0 + 1 .. c1 stored (replay C never runs: the last token's rows are
        zero, finite, and outside every loss term -- as in the stored form).  Row blocks are powers of two: a 48-step chunk is a 32- and a 16-step launch."""
        pe, tr = self.policy
        P, L = pe.p, pe.cfg.num_hidden_layers
        j = c0
        while j < c1:
            blk = 1 << ((c1 - j).bit_length() - 1)
            r = self.T0 + j
            for i in range(L):
                ops.gemm_swiglu_rows(tr["h2"][i][r:], P.w(f"layers.{i}.gu.w"), tr["gu"][i][r:], tr["a"][i][r:], self.N, blk, self.C)
            STATS["policy_mlp_rows"] += self.N * blk
            j += blk

    def _policy_head(self, rows: torch.Tensor, targets: torch.Tensor, cols: slice, n_ok: int | None = None):
        """log p / log-sum-exp of the policy's lm_head at arena rows `rows` ([N x k], sequence-major; targets < 0: not scored) -> columns `cols` of pol_logp / pol_lse
        (the first n_ok of the k positions per sequence).  Engine.logprobs' fused launch on other rows: per row the same bits."""
        pe, tr = self.policy
        W = pe.p.w(pe.p.lm_head_name())
        hsel = ops.embed_fwd(rows, None, tr["hf"], None)
        R = rows.numel()
        need = ops.linear_logprob_ws_bytes(R, W.shape[0])
        ws = self.__dict__.get("_head_ws")
        if ws is None or ws.numel() < need:
            ws = self._head_ws = torch.empty(need, dtype=torch.uint8, device=self.e.dev)
        lp, ls = ops.linear_logprob(hsel, W, targets, ws=ws)
        k = R // self.N
        n_ok = k if n_ok is None else n_ok
        self.pol_logp[:, cols] = lp.view(self.N, k)[:, :n_ok]
        self.pol_lse[:, cols] = ls.view(self.N, k)[:, :n_ok]

    def boundaries(self, C: int) -> set:
        """Decode steps after which a chunk is handed to the side stream: every `steps`, and once more one block before the end, so that what is left when the
        rollout ends (it runs on the caller's stream, on the whole device: finish) is one block of rows."""
        b = set(range(self.steps, C, self.steps))
        tail = os.environ.get("IADR1_OVERLAP_TAIL", "auto")
        if C - BLOCK > 0 and (tail == "block" or (tail == "auto" and self.policy is None)):
            b.add(C - BLOCK)
        return b

    def finish(self, n_tokens: int, after: torch.cuda.Event | None) -> torch.Tensor:
        """The rollout has ended with `n_tokens` columns of out_tokens produced (== C unless every sequence hit EOS earlier): run what is left (columns past the
        end hold pad tokens and are masked by the caller's completion mask).  Returns the [N, C] reference log-probs; the CALLER's stream must wait for
        `self.stream` before reading them (`join`)."""
        need = min(self.C, (int(n_tokens) + BLOCK - 1) // BLOCK * BLOCK)
        # the decode replays are over: the rest runs on the CALLER's stream -- every CU instead of the side stream's share -- behind what the side stream still holds
        side, cur = self.stream, torch.cuda.current_stream()
        cur.wait_stream(side)
        if after is not None:
            cur.wait_event(after)
        self.stream, gate = cur, self.step_counter
        self.step_counter = None
        try:
            while self.done_rows < need:
                self.chunk(min(need, self.done_rows + self.steps), None)
        finally:
            self.stream, self.step_counter = side, gate
        if self.step_counter is not None:
            self._timed_out_host.copy_(self._timed_out, non_blocking=True)      # read by check_timed_out(), before the optimizer step that would apply this pass's numbers
            self._timed_out_event = torch.cuda.Event()
            self._timed_out_event.record()
        self.active = False
        return self.logp

    def check_timed_out(self):
        """Raise if a counter wait of the last pass timed out (the chunk then ran on rows that were not final: its log-probs and the rebuilt policy rows are wrong).
        Called by the owner after the backward has been enqueued and BEFORE the optimizer step: the flag's copy was enqueued behind the rollout, ~300 ms of
        GPU time before the point where the host asks, so the wait is over when it is called."""
        ev = self.__dict__.get("_timed_out_event")
        if ev is None:
            return
        ev.synchronize()
        self._timed_out_event = None
        if int(self._timed_out_host[0]):
            self._timed_out.zero_()
            self._timed_out_host.zero_()
            raise RuntimeError("overlap.ChunkedRefPass: a counter wait of this step's co-scheduled pass timed out (the decode stream never reached the awaited step); "
                               "the step's gradients were NOT applied")

    def _mark(self, label=None, start=None):
        if self.trace is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if label is not None:
            self.trace.append((label, start, ev))
        return ev

    def report(self) -> list:
        """IADR1_OVERLAP_STATS=1: [(phase, start ms, end ms)] on the side stream, relative to begin(); synchronises."""
        if not self.trace:
            return []
        torch.cuda.synchronize(self.e.dev)
        return [(lb, round(self.t_ref.elapsed_time(a), 1), round(self.t_ref.elapsed_time(b), 1)) for lb, a, b in self.trace]

    def join(self):
        """Nothing is left on the side stream after finish() (which ran on the caller's stream behind it); kept for callers on OTHER streams."""
        torch.cuda.current_stream().wait_stream(self.stream)
