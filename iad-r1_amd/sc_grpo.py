"""SC-GRPO micro-step on the HIP engine: the arithmetic of `SCGRPOTrainer.compute_loss`
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:586-819) + backward + optimizer, re-scheduled for
one MI355X per rank:

  rollout (in-process, shared weights, hipGraph decode)  ->  rewards (CPU plugin functions)  ->
  advantages (per-prompt groups, rank-local)  ->  frozen-ref log-probs  ->  policy log-probs + backward  ->
  bucketed gradient all-reduce over RCCL (overlapped with the tail of backward)  ->  fused AdamW.

Ordering: sequences are laid out prompt-major (p0 x G, p1 x G, ...) everywhere, i.e. the interleaved order the
reference's reward / advantage code assumes (REF:754,775-792); its tensor tiling (REF:625-628) coincides with
this at per_device_train_batch_size=1, the only setting its scripts use.
"""
from __future__ import annotations

import os
import sys

from dataclasses import dataclass

import numpy as np
import torch

from . import hip, ops
from .overlap import ChunkedRefPass
from .params import ParamStore, VLMConfig
from .rollout import Rollout
from .vlm import Engine

BF16, F32 = torch.bfloat16, torch.float32


@dataclass
class GRPOArgs:
    """The subset of trl.GRPOConfig / TrainingArguments the reference's SC-GRPO path actually reads
    (SURVEY.md section 2.1 #12 and section 5), with the same defaults."""
    num_generations: int = 8
    max_prompt_length: int | None = 512
    max_completion_length: int = 256
    beta: float = 0.04
    temperature: float = 0.9
    top_k: int = 50            # hard-coded in the reference (REF:353-358)
    top_p: float = 0.9
    learning_rate: float = 1e-6
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    gradient_accumulation_steps: int = 1
    micro_batch_seqs: int = 16  # sequences per forward/backward pass (activation memory knob; results do not depend on it)
    seed: int = 42
    suppress_eos: bool = False  # benchmark mode: fixed-length completions
    use_hip_graph: bool = True
    # compute every prompt once per group in the reference / policy passes (shared-prefix attention; same math as the
    # reference's G-fold repeated rows).  Off: IADR1_SHARE_PREFIX=0 or share_prefix=False.
    share_prefix: bool = os.environ.get("IADR1_SHARE_PREFIX", "1") != "0"
    # let the rollout's prefill double as the prompt part of the policy's training forward (needs share_prefix and one whole-batch micro-batch)
    reuse_prefill: bool = os.environ.get("IADR1_REUSE_PREFILL", "1") != "0"
    # the rollout's decode steps also write the completion rows of the policy's activation arena, so the policy forward over the completions is not
    # run before backward (needs reuse_prefill's conditions and the fused decode kernels)
    reuse_decode: bool = os.environ.get("IADR1_REUSE_DECODE", "1") != "0"
    # LLaVA branches only: reproduce the reference's `_ensure_left_padding_data` (REF:502-504,516-567) -- a row whose completion ended early (right
    # padding, no left padding) is rotated before the model runs while the log-probs are still sliced / masked at the un-rotated columns, so its
    # loss terms are the log-probs of the tokens `C - len` positions EARLIER (the end of the prompt, then the start of the completion).  False
    # scores the completion tokens themselves, as the Qwen branches do.
    llava_rotate_right_padded_rows: bool = True
    # `--gradient_checkpointing` (REF scripts/train/SC_GRPO/*.sh:56): "off" | "auto" | "on" -- vlm.Engine.recompute_wanted.  With recomputation the policy's
    # forward is run after the rollout (the rollout's own activations are not kept either)
    recompute: str = "off"


def eos_completion_mask(completion_ids: np.ndarray, eos_token_id: int) -> np.ndarray:
    """REF:722-726 -- keep tokens up to and including the first EOS."""
    is_eos = completion_ids == eos_token_id
    n, c = is_eos.shape
    eos_idx = np.where(is_eos.any(1), is_eos.argmax(1), c)
    return (np.arange(c)[None, :] <= eos_idx[:, None]).astype(np.int32)


def pad(rows, padding_value=0, padding_side: str = "right", pad_to_multiple_of: int | None = None) -> np.ndarray:
    """trl `pad` (REF trl/trl/trainer/utils.py:418-478): stack arrays of possibly different shape into one, every dimension grown to the
    largest (dimension 0 rounded up to `pad_to_multiple_of`), the data of each row at the start ("right" padding) or the end ("left") of
    dimension 0, `padding_value` elsewhere."""
    arrs = [np.asarray(r) for r in rows]
    if not arrs:
        return np.zeros((0, 0), dtype=np.int64)
    shape = list(np.max([a.shape for a in arrs], 0))
    if pad_to_multiple_of is not None:
        shape[0] += (-shape[0]) % pad_to_multiple_of
    out = np.full((len(arrs), *shape), padding_value, dtype=arrs[0].dtype)
    for i, a in enumerate(arrs):
        if padding_side == "left":
            first = slice(shape[0] - a.shape[0], shape[0])
        elif padding_side == "right":
            first = slice(0, a.shape[0])
        else:
            raise ValueError("padding_side must be 'left' or 'right'")
        out[(i, first, *[slice(0, n) for n in a.shape[1:]])] = a
    return out


def right_pad(rows, pad_value: int) -> np.ndarray:
    """The call at REF:682: completion id lists, right-padded with the pad token."""
    return pad([np.asarray(r, dtype=np.int64) for r in rows], pad_value, "right").astype(np.int64).reshape(len(rows), -1)


def right_padding_shift(row_ids: np.ndarray, pad_token_id: int) -> int:
    """The trigger of the reference's `_ensure_left_padding_data` (REF:534-541), on the ids alone: the first pad token of the [prompt | completion] row
    starts an all-pad suffix -> that many columns are rotated to the front (returned); otherwise (no pad, left padding, a pad inside the text) the row is
    left alone (0).  Derived from the ids like the reference, not from the EOS mask: a supplied completion that is shorter than C without an EOS is
    rotated too, and with pad == EOS the EOS itself moves into the padding."""
    pads = np.flatnonzero(row_ids == pad_token_id)
    if pads.size == 0:
        return 0
    first = int(pads[0])
    return int(len(row_ids) - first) if bool((row_ids[first:] == pad_token_id).all()) else 0


def group_advantages(rewards: torch.Tensor, G: int):
    """REF:787-793: group mean, UNBIASED std, eps 1e-4."""
    r = rewards.view(-1, G)
    mean = r.mean(1).repeat_interleave(G)
    std = r.std(1).repeat_interleave(G) if G > 1 else torch.zeros_like(mean)
    return (rewards - mean) / (std + 1e-4), std


class GradReducer:
    """Data-parallel gradient exchange: the ONLY collective of the data path.  The flat fp32 gradient buffer is
    all-reduced in contiguous buckets of at most `bucket_bytes` on the wire (256 MB: large enough to run at link rate, small enough that the
    first bucket leaves while backward is still producing the next) on a side stream; decoder-layer ranges are launched as soon as the last
    micro-batch's backward has left that layer, so the exchange overlaps the rest of backward.

    Wire type: bf16 by default -- what the reference moves (DeepSpeed ZeRO-3 reduce-scatters the bf16 gradients of a bf16 model,
    scripts/train/zero3.json + `--bf16`) and half the bytes over xGMI (7.5 GB instead of 15 GB for the 3B model): each bucket is cast into a bf16
    staging buffer, summed there by RCCL, and cast back into the fp32 gradient buffer, all stream-ordered on the side stream.
    IADR1_REDUCE_DTYPE=fp32 (or wire="fp32") sums the fp32 buffer in place.  Ring / tree all-reduce leaves bit-identical sums on every rank
    either way, so the replicas' parameters stay bit-identical.  The bucket sequence depends on the parameter layout only, never on a rank's
    data.  Works on CPU / gloo too (tests)."""

    RING = 3        # bf16 staging buffers of one bucket each: one being cast into, one on the wire, one being cast back

    def __init__(self, store: ParamStore, group=None, bucket_bytes: int = 256 << 20, wire: str | None = None):
        import torch.distributed as dist
        self.dist, self.store, self.group = dist, store, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.cuda = store.grad.is_cuda
        # IADR1_FORCE_REDUCE=1: run the exchange even in a one-rank group (exercises the side-stream / bucket / RCCL path on a single GPU)
        self.active = self.world > 1 or (bool(os.environ.get("IADR1_FORCE_REDUCE")) and dist.is_available() and dist.is_initialized())
        self.stream = torch.cuda.Stream() if (self.cuda and self.active) else None
        self.wire = (wire or os.environ.get("IADR1_REDUCE_DTYPE", "bf16")).lower()
        if self.wire not in ("bf16", "fp32"):
            raise ValueError("IADR1_REDUCE_DTYPE must be bf16 or fp32")
        # collective per bucket: "all_reduce" (the library picks ring / tree), or "rs_ag": reduce_scatter_tensor + all_gather_into_tensor -- the two halves of an
        # all-reduce issued explicitly, which on the fully connected xGMI mesh of one node can run as direct exchanges (every rank sends 1/N of the bucket to every
        # peer at once: SURVEY section 5 estimates ~12 ms against ~86 ms for a ring over 7.5 GB).  Bit-identical sums on every rank either way (each shard is reduced
        # on ONE rank, then copied).  Needs a backend with reduce_scatter_tensor (RCCL); on others (gloo, the CPU tests) the request falls back to all_reduce.
        self.algo = os.environ.get("IADR1_REDUCE_ALGO", "all_reduce").lower()
        if self.algo not in ("all_reduce", "rs_ag"):
            raise ValueError("IADR1_REDUCE_ALGO must be all_reduce or rs_ag")
        if self.algo == "rs_ag" and not (self.active and self.cuda and dist.get_backend(group) == "nccl"):
            self.algo = "all_reduce"
        pad = 8 * self.world                    # rs_ag: equal 16-byte aligned shards
        self.bucket_elems = max(1, min(int(bucket_bytes) // (2 if self.wire == "bf16" else 4), store.n_total))
        if self.algo == "rs_ag":
            self.bucket_elems = (self.bucket_elems + pad - 1) // pad * pad
        # staging: a RING of bucket-sized bf16 buffers (<= 768 MB), not a second copy of the model (16.6 GB at 7B, which the 7B / LLaVA-OneVision
        # configurations do not have to spare next to 245 GB of training state).  A buffer is reused only after its previous bucket has been cast
        # back into the fp32 gradient buffer -- stream order on the side stream guarantees it (the cast-back is enqueued before the next cast-in)
        self.ring = [torch.empty(self.bucket_elems, dtype=BF16, device=store.grad.device) for _ in range(self.RING)] if (self.active and self.wire == "bf16") else None
        if self.active and self.algo == "rs_ag" and self.ring is None:        # fp32 wire: the padded bucket is staged too (the gradient buffer's tail is not a multiple of the shard)
            self.ring = [torch.empty(self.bucket_elems, dtype=F32, device=store.grad.device) for _ in range(self.RING)]
        self.pending = [None] * self.RING
        self.slot = 0
        self.bytes_on_wire = 0          # per step, for the bench line
        self.n_buckets = 0
        c = store.cfg
        s = store.slots
        self.layer_range = []
        for i in range(c.num_hidden_layers):
            lo = s[f"layers.{i}.qkv.w"].offset
            hi = s[f"layers.{i + 1}.qkv.w"].offset if i + 1 < c.num_hidden_layers else (s["lm_head"].offset if "lm_head" in s else store.n_decay)
            self.layer_range.append((lo, hi))
        self.first_layer_off = s["layers.0.qkv.w"].offset
        self.reset()

    def exposed_ms(self) -> float | None:
        """Milliseconds of the LAST finish() that the compute stream waited for the exchange (synchronises on the second event); None without an exchange."""
        ev = self.__dict__.get("_exposed_events")
        if ev is None:
            return None
        ev[1].synchronize()
        return float(ev[0].elapsed_time(ev[1]))

    def staging_bytes(self) -> int:
        """Device memory the exchange adds to the training state."""
        return 0 if self.ring is None else sum(r.numel() * r.element_size() for r in self.ring)

    def reset(self):
        self.done = []

    def _complete(self, k):
        """Bucket in ring slot k: wait for its collective (stream-ordered on RCCL: the side stream waits, the host does not), cast the sum back."""
        p = self.pending[k]
        if p is None:
            return
        work, st, lo, hi = p
        work.wait()
        g = self.store.grad[lo:hi]
        if st.dtype == F32:
            g.copy_(st[: hi - lo])
        elif self.cuda:
            hip.call("cast_bf16_to_f32", st, g, hi - lo)
        else:
            g.copy_(st[: hi - lo])
        self.pending[k] = None

    def _bucket(self, lo, hi):
        g = self.store.grad[lo:hi]
        if self.ring is None:
            self.dist.all_reduce(g, group=self.group)
            return
        k = self.slot
        self.slot = (k + 1) % self.RING
        self._complete(k)                  # the slot's previous bucket (RING buckets ago) leaves the buffer first
        n = hi - lo
        st = self.ring[k][:n]
        if st.dtype == F32:
            st.copy_(g)
        elif self.cuda:   # fp32 -> bf16 (round to nearest even) and back through the library's cast kernels
            hip.call("cast_f32_to_bf16", g, n, st, n, 1, n, n)
        else:
            st.copy_(g)
        if self.algo == "rs_ag":
            # the bucket padded to world equal shards (the pad is zeroed: it is summed and gathered like data, nobody reads it back)
            npad = (n + 8 * self.world - 1) // (8 * self.world) * (8 * self.world)
            full = self.ring[k][:npad]
            if npad > n:
                full[n:].zero_()
            shard = full.view(self.world, -1)[self.dist.get_rank(self.group)]
            self.dist.reduce_scatter_tensor(shard, full, group=self.group, async_op=True)       # in place: rank r's shard of `full` receives the sum (NCCL's in-place form)
            work = self.dist.all_gather_into_tensor(full, shard, group=self.group, async_op=True)   # same communicator, same stream: ordered behind the reduce-scatter
            self.pending[k] = (work, st, lo, hi)
            return
        self.pending[k] = (self.dist.all_reduce(st, group=self.group, async_op=True), st, lo, hi)

    def _drain(self):
        for i in range(self.RING):         # oldest first
            self._complete((self.slot + i) % self.RING)

    def _reduce(self, lo, hi, drain=False):
        if not self.active or (hi <= lo and not drain):
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _nullcontext()
        with ctx:
            for b0 in range(lo, hi, self.bucket_elems):
                b1 = min(hi, b0 + self.bucket_elems)
                self._bucket(b0, b1)      # on RCCL the collective is enqueued behind this stream's work: no host block
                self.bytes_on_wire += (b1 - b0) * (2 if self.wire == "bf16" else 4)
                self.n_buckets += 1
            if drain:
                self._drain()
        if hi > lo:
            self.done.append((lo, hi))

    def layer_ready(self, i):
        self._reduce(*self.layer_range[i])

    def all_layers_ready(self):
        """The decoder-layer ranges in backward order -- for a last micro-batch that ran no backward on this rank (nothing supervised in it): the
        bucket sequence must be the same on every rank whatever the rank-local data was."""
        sent = set(self.done)
        for i in reversed(range(len(self.layer_range))):
            if self.layer_range[i] not in sent:
                self.layer_ready(i)

    def finish(self):
        """Reduce whatever has not been sent yet (vision tower, embedding, lm_head, norm gains / biases) and join.
        `exposed_ms()` afterwards: GPU time the compute stream spent waiting here for the exchange -- the part of the collective that did NOT hide under backward
        (two events on the compute stream: at entry = end of backward in stream order, and behind the join)."""
        ev = None
        if self.stream is not None and self.active:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        self.all_layers_ready()
        covered = sorted(self.done)
        cur = 0
        for lo, hi in covered + [(self.store.n_total, self.store.n_total)]:
            if lo > cur:
                self._reduce(cur, lo)
            cur = max(cur, hi)
        self._reduce(0, 0, drain=True)      # the buckets still in the ring: wait + cast back
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        if ev is not None:
            ev[1].record()
            self._exposed_events = ev
        self.last_bytes_on_wire, self.last_n_buckets = self.bytes_on_wire, self.n_buckets
        self.bytes_on_wire = self.n_buckets = 0
        self.reset()


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class SCGRPOEngine:
    def __init__(self, cfg: VLMConfig, policy: ParamStore, ref: ParamStore, args: GRPOArgs, group=None):
        self.cfg, self.args = cfg, args
        self.pol, self.ref = Engine(policy), Engine(ref)
        self.dev = policy.device
        self.reducer = GradReducer(policy, group)
        self.pol.resident_extra = ref.resident_bytes() + self.reducer.staging_bytes()       # static inputs of Engine.recompute_wanted("auto")
        self.opt_step = 0
        self.accum = 0
        self._rollout = None
        self.shadow_logps, self.shadow_policy_head, self.last_step_shadowed = None, None, False
        self._shadow = None         # overlap.ChunkedRefPass: the frozen reference's pass over the sampled tokens runs under the rollout (step())
        self.norm2 = torch.zeros(1, dtype=F32, device=self.dev)
        self.norm_scratch = torch.zeros(2048, dtype=F32, device=self.dev)

    # ---- vision: once per unique image ---------------------------------------------------------------------
    def _vision(self, batch, want_policy_ctx: bool):
        return self.pol.vision_inputs(batch)

    def _per_row_images(self, batch, grids, rows):
        ipp = batch.get("images_per_prompt") or [1] * len(batch["input_ids"])
        gpr, off, k = [], [], 0
        for n in ipp:
            gpr.append(grids[k: k + n])
            off.append([int(rows[j]) for j in range(k, k + n)])
            k += n
        return gpr, off

    # ---- rollout -------------------------------------------------------------------------------------------------
    def vision_policy(self, batch, save: bool):
        """Policy vision tower once per step: its output feeds both the rollout prefill and the training forward."""
        grids, plan_v, px, rows = self._vision(batch, save)
        img, vctx = self.pol.vision_forward(px, plan_v, save=save)
        return {"grids": grids, "plan": plan_v, "px": px, "rows": rows, "img": img, "ctx": vctx}

    def rollout(self, batch, vis=None, greedy=False, train_carry=None, shadow_ref: bool = False) -> np.ndarray:
        """Completion ids [Bp*G, <=C] (numpy, right-padded with pad after EOS like REF:680-683).
        shadow_ref: also run the frozen reference's teacher-forced pass over the tokens as they are produced, on a second stream under the decode replays
        (overlap.ChunkedRefPass); afterwards self.shadow_logps is the [N, C] reference log-prob tensor (None when the pass does not apply) and the consumer
        must `self._shadow.join()` before reading it."""
        a = self.args
        ids, mask = np.asarray(batch["input_ids"]), np.asarray(batch["attention_mask"])
        if a.max_prompt_length is not None:
            ids, mask = ids[:, -a.max_prompt_length:], mask[:, -a.max_prompt_length:]
        vis = vis or self.vision_policy(batch, save=False)
        grids, rows, img_pol = vis["grids"], vis["rows"], vis["img"]
        gpr, off = self._per_row_images(batch, grids, rows)
        plan = self.pol.text_plan(ids, mask, gpr, off)
        Bp = ids.shape[0]
        N = Bp * a.num_generations
        # the KV pool, the block table and the captured graph are sized for a prompt length: a longer padded prompt than any seen so far rebuilds them (grow only;
        # prompt lengths vary from batch to batch with the image sizes of the any-resolution families)
        if self._rollout is None or self._rollout.N != N or self._rollout.max_new < a.max_completion_length or self._rollout.max_prompt < ids.shape[1]:
            grow = max(ids.shape[1], self._rollout.max_prompt if self._rollout is not None else 0)
            self._rollout = None        # release the old pool before the new one is allocated
            split = self._cu_split(N)      # (the launcher configuration that goes with it is set by Rollout.generate for the duration of each call)
            self._rollout = Rollout(self.pol, N, grow, a.max_completion_length, max_prompts=Bp, use_graph=a.use_hip_graph, **split)
            if split and self.__dict__.get("_wprefetch_stream") is not None:
                # opt-in (IADR1_WPREFETCH_CUS): the NEXT layer's narrow projections (q|k|v, o: 18 MiB at the 3B widths) pulled into the memory-side cache while a layer runs
                from .wprefetch import WeightPrefetcher, wanted_cus
                self._rollout.wprefetch = WeightPrefetcher(self.pol, self._wprefetch_stream, wanted_cus(), what=(), next_what=("qkv", "o"), lead=0)
        c, st = self.cfg, self.pol.p
        # decode steps that also fill the training arena (no policy forward over the completions afterwards): needs the fused decode kernels that carry
        # the side outputs (q|k|v + rotary + cache append, persistent fused-SwiGLU gate|up GEMM)
        # (not with the opt-in FP8 weight stream: the policy's training activations must come from its bf16 weights, so its forward runs after the rollout)
        trace = (train_carry is not None and a.reuse_decode and st.qkv_rope_packed and not st.decode_fp8 and self._rollout_fuses_swiglu(N))
        shadow = None
        self.shadow_logps, self.shadow_policy_head = None, None
        if shadow_ref and self.dev.type == "cuda" and ChunkedRefPass.applicable(c, N, a.max_completion_length) and (self._rollout.decode_stream is not None or ChunkedRefPass.unmasked()):
            if self._shadow is None or self._shadow.stream is not (self.__dict__.get("_shadow_stream") or self._shadow.stream):
                self._shadow = ChunkedRefPass(self.ref, stream=self.__dict__.get("_shadow_stream"))
            shadow = self._shadow
            shadow.vision = (vis["px"], vis["plan"])
        if os.environ.get("IADR1_QUIET") != "1" and self.__dict__.get("_path_logged") != (shadow is not None):
            self._path_logged = shadow is not None      # once per change: WHICH step structure runs (the two differ in the last bf16 bit of the policy's mlp rows)
            if shadow is not None:
                rb = trace and shadow.rebuilds_policy_mlp(c, N)
                print(f"[iadr1] co-scheduled step: the frozen reference's pass{', the policy mlp rows (REBUILT by the training GEMM)' if rb else ''} and the lm_head log-probs run on "
                      f"{torch.cuda.get_device_properties(self.dev).multi_processor_count - self._rollout.decode_cus if self._rollout.decode_cus else 'all'} CUs under the rollout (IADR1_OVERLAP_CUS=0: sequential step; IADR1_OVERLAP_GU=0: rows stored by the decode step)",
                      file=sys.stderr, flush=True)
            elif shadow_ref:
                print("[iadr1] sequential step: the frozen reference's pass runs after the rollout, on the whole device", file=sys.stderr, flush=True)
        toks = self._rollout.generate(plan, img_pol, a.num_generations, a.max_completion_length, temperature=0.0 if greedy else a.temperature,
                                      top_k=a.top_k, top_p=a.top_p, seed=a.seed + 1000003 * self.opt_step + 7919 * self.accum, suppress_eos=a.suppress_eos,
                                      train_carry=train_carry, train_trace=trace, shadow=shadow)
        if shadow is not None:
            self.shadow_logps = shadow.logp
            self.shadow_policy_head = (shadow.pol_logp, shadow.pol_lse) if shadow.pol_logp is not None else None
        if trace:
            train_carry["traced"] = True
        return self._rollout.tokens_host()      # (the host copy left before the side pass's tail was enqueued: 1230.0 -> 1223.4 ms, EXPERIMENTS round 5)

    def _cu_split(self, N: int) -> dict:
        """Co-scheduling with a CU split (IADR1_OVERLAP_CUS = auto for this shape, or n > 0): the shadow pass is confined to n CUs and the decode replays to the
        others -- two CU-masked streams (include/iadr1_hip.h iadr1_stream_create_cu_mask) on different dispatch pipes (overlap.cu_split, one pair per process); the
        decode launchers size their persistent grids for the smaller device (iadr1_set_decode_cus).  Masked streams are BLOCKING streams: `step` moves off the null
        stream while they are in use."""
        from .overlap import cu_split, shadow_cus
        n = shadow_cus(self.cfg, N, self.args.max_completion_length)
        self._shadow_stream = None
        if n <= 0 or self.dev.type != "cuda" or not ChunkedRefPass.applicable(self.cfg, N, self.args.max_completion_length):
            return {}
        from .wprefetch import wanted_cus
        sp = cu_split(self.dev, n, self.ref.p.w("layers.0.gu.w"), prefetch_cus=wanted_cus())
        # one decision for the whole data-parallel job, taken once (the first rollout of a run, the same call on every rank: N and the completion length are the
        # launch configuration's, identical across ranks): a rank without a clean stream pair switches co-scheduling off for all of them
        agreed = self.__dict__.setdefault("_co_sched_agreed", {})
        if n not in agreed:
            from .overlap import agree_across_ranks
            agreed[n] = agree_across_ranks(sp is not None, self.reducer.group, self.dev)
            if sp is not None and not agreed[n] and os.environ.get("IADR1_QUIET") != "1":
                print("[iadr1] co-scheduling off on this rank too: another rank of the process group found no clean stream pair", file=sys.stderr, flush=True)
        if sp is None or not agreed[n]:       # no candidate on another dispatch pipe (here or on some rank): no co-scheduling at all (the one-shot reference pass after the rollout)
            if sp is None and os.environ.get("IADR1_OVERLAP_CUS", "auto").strip().lower() not in ("", "auto") and os.environ.get("IADR1_OVERLAP_STRICT", "1") != "0":
                # a FORCED split that cannot be had is an error, not a silent change of the step's structure (the rebuilt policy rows differ from the stored ones in the
                # last bf16 bit: ADVICE r5 -- a run that asked for one path must not get the other)
                raise RuntimeError(f"IADR1_OVERLAP_CUS={n}: no stream pair on separate dispatch pipes was found; unset it (auto: sequential fallback) or set IADR1_OVERLAP_STRICT=0")
            return {}
        self._shadow_stream = sp[1]
        self._wprefetch_stream = sp[3]
        return {"decode_cus": sp[2], "decode_stream": sp[0]}

    def _rollout_fuses_swiglu(self, N) -> bool:
        """True when the decode gate|up projection runs on the persistent fused-SwiGLU kernel (the one with side outputs): include/iadr1_hip.h."""
        c = self.cfg
        ncu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        K, N2 = c.hidden_size, 2 * c.intermediate_size
        return (self._rollout.fuse_swiglu and os.environ.get("IADR1_SKINNY_PERS", "1") != "0" and K % 256 == 0 and K // 256 in (4, 6, 8)
                and N2 % 128 == 0 and N2 // 32 >= 2 * ncu and N <= 256)

    # ---- loss + gradients for given completions ------------------------------------------------------------------
    def loss_and_grads(self, batch, completions, rewards_per_func, backward: bool = True, last_micro_step: bool = True, vis=None, train_carry=None, defer_metrics: bool = False,
                       ref_logps=None, policy_head=None):
        """completions: list of Bp*G id lists (prompt-major) or an [N,C] array already padded;
        rewards_per_func: [N, n_funcs] float tensor/array.  Accumulates gradients into policy.grad.
        ref_logps: [N, C] reference log-probs already computed for exactly these completions by the rollout's shadow pass (rollout(shadow_ref=True)): the
        reference forward is then not run here; the shadow stream is joined right before the loss kernel reads them."""
        a, c = self.args, self.cfg
        G = a.num_generations
        ids_p, mask_p = np.asarray(batch["input_ids"]), np.asarray(batch["attention_mask"])
        if a.max_prompt_length is not None:
            ids_p, mask_p = ids_p[:, -a.max_prompt_length:], mask_p[:, -a.max_prompt_length:]
        Bp, P = ids_p.shape
        N = Bp * G
        comp = completions if isinstance(completions, np.ndarray) else right_pad(completions, c.pad_token_id)
        assert comp.shape[0] == N
        C = comp.shape[1]
        cmask = eos_completion_mask(comp, c.eos_token_id)
        ids = np.concatenate([np.repeat(ids_p, G, 0), comp], 1)
        mask = np.concatenate([np.repeat(mask_p, G, 0), cmask.astype(mask_p.dtype)], 1)
        S = P + C
        cmask_d = ops.h2d(cmask, self.dev)
        state = {}

        def advantages():
            """Rewards may be handed over as a callable: it is evaluated here, after the reference and policy forwards of the first
            micro-batch have been enqueued, so the host-side reward functions (regex, ~9 ms for 64 samples) run while the GPU works."""
            if not state:
                rpf = rewards_per_func() if callable(rewards_per_func) else rewards_per_func
                rpf = torch.as_tensor(np.asarray(rpf), dtype=F32)
                adv, std = group_advantages(rpf.sum(1), G)
                # pinned + non-blocking: a pageable host-to-device copy synchronises the stream, i.e. would make the host wait here for the reference and policy
                # forwards it has just enqueued (and leave the GPU idle for the ~1 ms the host then needs to launch the loss kernel)
                state.update(rpf=rpf, rewards=rpf.sum(1), adv=adv, std=std, adv_d=ops.h2d(adv, self.dev))
            return state

        if vis is None or (backward and vis["ctx"] is None):
            vis = self.vision_policy(batch, save=backward)
        grids, plan_v, px, rows, img_pol, vctx = vis["grids"], vis["plan"], vis["px"], vis["rows"], vis["img"], vis["ctx"]
        have_ref = ref_logps is not None and tuple(ref_logps.shape) == (N, C)
        img_ref = None if have_ref else self.ref.vision_forward(px, plan_v, save=False)[0]
        gpr, off = self._per_row_images(batch, grids, rows)
        dimg32 = torch.zeros(img_pol.shape, dtype=F32, device=self.dev) if backward else None

        logp_all = torch.empty(N, C, dtype=F32, device=self.dev)
        ref_all = torch.empty(N, C, dtype=F32, device=self.dev)
        kl_all = torch.empty(N, C, dtype=F32, device=self.dev)
        row_loss = torch.empty(N, dtype=F32, device=self.dev)
        row_kl = torch.empty(N, dtype=F32, device=self.dev)
        mb = max(1, min(a.micro_batch_seqs, N))
        # shared-prefix layout (vlm.Engine.text_plan_shared): every prompt runs through the layers once per group instead of
        # G times -- needs micro-batches made of whole groups.  IADR1_SHARE_PREFIX=0 keeps the reference's [n, P+C] rows.
        share = a.share_prefix and G > 1 and mb % G == 0
        starts = list(range(0, N, mb))
        col = np.arange(C)
        for si, r0 in enumerate(starts):
            r1 = min(N, r0 + mb)
            n = r1 - r0
            if share:
                b0, b1 = r0 // G, r1 // G
                plan = self.pol.text_plan_shared(ids_p[b0:b1], mask_p[b0:b1], comp[r0:r1], cmask[r0:r1], G, gpr[b0:b1], off[b0:b1])
                sel = self.pol.shared_logit_rows(b1 - b0, P, n, C, G)[0]
            else:
                rows_b = [r // G for r in range(r0, r1)]
                plan = self.pol.text_plan(ids[r0:r1], mask[r0:r1], [gpr[b] for b in rows_b], [off[b] for b in rows_b])
                sel = (np.arange(n)[:, None] * S + (P - 1) + col[None, :]).reshape(-1)        # logits rows P-1 .. S-2
            tgt = ids[r0:r1, P:].astype(np.int64).copy()
            if c.is_llava and a.llava_rotate_right_padded_rows:
                sel = sel.reshape(n, C).copy()
                for r in range(r0, r1):
                    pl = right_padding_shift(ids[r], c.pad_token_id)
                    if pl == 0:
                        continue                       # no right padding / left padding present: the reference leaves the row alone
                    ln = int(cmask[r].sum())           # the loss mask stays the un-rotated completion mask (REF:722-728)
                    if P - pl - 1 < 0:
                        # the window would start inside the rotated row's padding: the reference then scores pad tokens predicted from masked positions
                        # (its values depend on how the attention backend treats fully masked queries) -- not a defined target
                        raise ValueError(f"llava left-padding fix-up: row {r} ends in {pl} pad columns but its prompt has only {P} positions; "
                                         "the reference's rotated window would read masked positions (prompt shorter than the padding of an early-ended completion)")
                    # the row is shifted right by pl columns: window column t holds original token P - pl + t
                    for t_ in range(ln):
                        qi = P - pl + t_               # predicted token (original column), read from the hidden state of column qi - 1
                        hi = qi - 1
                        if share:
                            sel[r - r0, t_] = (r // G - b0) * P + hi if hi < P else (b1 - b0) * P + (r - r0) * C + (hi - P)
                        else:
                            sel[r - r0, t_] = (r - r0) * S + hi
                        tgt[r - r0, t_] = ids[r, qi]
                sel = sel.reshape(-1)
            rows_d = ops.h2d(sel.astype(np.int64), self.dev)
            tgt_d = ops.h2d(tgt.reshape(-1), self.dev)
            if not have_ref:
                hf, _ = self.ref.text_forward(plan, img_ref, save=False)
                rl, _ = self.ref.logprobs(hf, rows_d, tgt_d, save=False)
                del hf
            if train_carry is not None and "x0" in train_carry and share and backward and n == N and C == a.max_completion_length and vis is not None and vis["ctx"] is not None:
                # the rollout's prefill already ran (and saved) the prompt rows of this batch: only the completion rows go through the layers
                T_all = plan.ids.numel()
                train_carry["full_plan"] = plan
                if train_carry.get("traced"):      # ... and its decode steps wrote the completion rows: nothing to run at all
                    hf, ctx = self.pol.text_context_from_trace(plan.tail, ((b1 - b0) * P, T_all, T_all), train_carry)
                else:
                    hf, ctx = self.pol.text_forward(plan.tail, None, save=True, rows=((b1 - b0) * P, T_all, T_all), carry=train_carry)
            else:
                hf, ctx = self.pol.text_forward(plan, img_pol, save=backward, recompute=backward and self.pol.recompute_wanted(plan.ids.numel(), a.recompute))
            pre = None
            if have_ref and policy_head is not None and share and n == N and train_carry is not None and train_carry.get("traced") and not (c.is_llava and a.llava_rotate_right_padded_rows):
                pre = (policy_head[0].reshape(-1), policy_head[1].reshape(-1))       # the shadow pass scored the policy's rows too (same rows, same targets, same launch)
            lp, lctx = self.pol.logprobs(hf, rows_d, tgt_d, save=backward, rows_host=sel, precomputed=pre)
            if pre is not None and si == 0:
                self._shadow.join()
            adv_d = advantages()["adv_d"]
            if have_ref:
                if si == 0:
                    self._shadow.join()          # the tail chunk of the shadow pass ran next to the policy's lm_head; everything after this reads its log-probs
                rl = ref_logps[r0:r1].contiguous()
            dlogp, kl, rloss, rkl = ops.grpo_loss(lp.view(n, C), rl.view(n, C), adv_d[r0:r1].contiguous(), cmask_d[r0:r1].contiguous(), a.beta, n_total_rows=N)
            logp_all[r0:r1], ref_all[r0:r1], kl_all[r0:r1] = lp.view(n, C), rl.view(n, C), kl
            row_loss[r0:r1], row_kl[r0:r1] = rloss, rkl
            if backward:
                dhf = self.pol.logprobs_backward(dlogp.view(-1), lctx)
                del hf, lctx
                hook = self.reducer.layer_ready if (last_micro_step and si == len(starts) - 1) else None
                self.pol.text_backward(dhf, ctx, dimg32, layer_done=hook)
                del ctx, dhf
        if backward:
            dimg = ops.f32_bias_to_bf16(dimg32, None)
            self.pol.vision_backward(dimg, vctx)
            self.accum += 1
        st = advantages()
        metrics = {"completion_length": float(cmask.sum(1).mean()), "reward": float(st["rewards"].mean()), "reward_std": float(st["std"].mean())}
        dev_means = torch.stack([row_loss.mean(), row_kl.mean()])       # one device -> host read for both

        def finalize():
            lk = dev_means.tolist()
            metrics.update(loss=lk[0], kl=lk[1])
            return metrics
        if defer_metrics:      # the caller enqueues more work (the optimizer) before it reads the two device-side means: no idle GPU while the host comes back from the sync
            metrics["finalize"] = finalize
        else:
            finalize()
        return {"metrics": metrics, "logps": logp_all, "ref_logps": ref_all, "kl": kl_all, "advantages": st["adv"], "completion_mask": cmask,
                "rewards_per_func": st["rpf"], "ids": ids, "mask": mask}

    # ---- optimizer -------------------------------------------------------------------------------------------------
    def optimizer_step(self):
        a, st = self.args, self.pol.p
        if self._shadow is not None:
            self._shadow.check_timed_out()      # a co-scheduled pass that ran on unfinished rows must not reach the weights
        self.reducer.finish()
        self.opt_step += 1
        scale = 1.0 / (self.reducer.world * max(1, self.accum))
        hip.call("sumsq", st.grad, st.n_total, self.norm_scratch, self.norm2)
        self.grad_scale = scale          # grad_norm() = sqrt(norm2) * scale: the norm of the averaged gradient, before clipping (HF's `grad_norm` log key)
        for lo, hi, wd in ((0, st.n_decay, a.weight_decay), (st.n_decay, st.n_total, 0.0)):
            if hi > lo:
                hip.call("adamw_flat", st.master[lo:hi], st.m[lo:hi], st.v[lo:hi], st.grad[lo:hi], st.flat[lo:hi], hi - lo, a.learning_rate,
                         a.adam_beta1, a.adam_beta2, a.adam_epsilon, wd, self.opt_step, scale, self.norm2, a.max_grad_norm)
        st.refresh_shadows()
        self.accum = 0
        if self.pol.check_ddp_headroom(a.recompute, group=self.reducer.group):
            # the rollout's captured graph and its training-arena views keep the freed arena's blocks alive: drop them too, THEN hand the memory back to the device
            self._rollout = None
            torch.cuda.empty_cache()

    def grad_norm(self) -> float:
        """Global L2 norm of the last optimizer step's (averaged, unclipped) gradient; synchronises the device."""
        return float(self.norm2.sqrt().item()) * getattr(self, "grad_scale", 1.0)

    # ---- the whole micro-step ----------------------------------------------------------------------------------------
    def step(self, *args, **kw):
        """`_step` on a stream of the engine's own when the caller is on the NULL stream and the CU split is in use: the two CU-masked streams are blocking
        streams (hipExtStreamCreateWithCUMask has no non-blocking form), i.e. every operation on the null stream waits for both of them and holds back what they
        enqueue afterwards -- measured: each chunk of the shadow pass then started exactly one chunk period late.  The caller's stream is ordered before and
        after the step as usual."""
        if self.dev.type == "cuda" and torch.cuda.current_stream(self.dev).cuda_stream == 0 and self._splits_cus(args[0] if args else kw.get("batch")):
            if "_main_stream" not in self.__dict__:
                self._main_stream = torch.cuda.Stream(self.dev)
            outer = torch.cuda.current_stream(self.dev)
            self._main_stream.wait_stream(outer)
            with torch.cuda.stream(self._main_stream):
                out = self._step(*args, **kw)
            outer.wait_stream(self._main_stream)
            return out
        return self._step(*args, **kw)

    def _splits_cus(self, batch) -> bool:
        from .overlap import shadow_cus
        try:
            n_seq = len(batch["input_ids"]) * self.args.num_generations
        except Exception:
            return False
        return shadow_cus(self.cfg, n_seq, self.args.max_completion_length) > 0

    def _step(self, batch, reward_fn, do_optimizer_step=True, last_micro_step=None, return_outputs=False, defer_metrics=False, completions=None):
        """One SC-GRPO micro-step: vision tower -> group rollout -> rewards -> reference / policy passes + backward (-> optimizer).  This is the path
        `SCGRPOTrainer.compute_loss` (the reference's API, REF:586) runs and the one bench.py times.
        reward_fn(completion_ids: np.ndarray [N,C]) -> [N, n_funcs] rewards (decode + plugin functions live with the caller, which owns the tokenizer).
        last_micro_step (default: do_optimizer_step): the data-parallel gradient buckets leave from this call's backward.
        completions: [N, C] ids already rolled out for these prompts (SCGRPOTrainer.training_step rolls all micro-batches of an optimizer step out at once): no
        rollout here, the policy forward runs over the completions.
        return_outputs: the whole loss_and_grads dict (log-probs, advantages, masks, ids, metrics) plus "completion_ids" instead of the metrics alone."""
        import time
        timing = os.environ.get("IADR1_TIMING") == "1"

        def mark():
            if timing:
                torch.cuda.synchronize()
            return time.perf_counter()

        t0 = mark()
        vis = self.vision_policy(batch, save=True)
        t1 = mark()
        # two-phase policy forward: with one whole-batch micro-batch in the shared-prefix layout the rollout's prefill IS the prompt part of
        # the training forward (saved in the training arena), so the policy forward after the rollout only runs the completion rows
        a = self.args
        N = len(batch["input_ids"]) * a.num_generations
        carry = {} if (completions is None and a.share_prefix and a.reuse_prefill and a.num_generations > 1 and a.micro_batch_seqs >= N and N % a.num_generations == 0) else None
        if carry is not None and a.recompute != "off":
            P_ = np.asarray(batch["input_ids"]).shape[1] if a.max_prompt_length is None else min(np.asarray(batch["input_ids"]).shape[1], a.max_prompt_length)
            if self.pol.recompute_wanted(len(batch["input_ids"]) * P_ + N * a.max_completion_length, a.recompute):
                carry = None       # gradient checkpointing: nothing of the rollout is kept, the policy forward runs (checkpointed) before backward
        comp = self.rollout(batch, vis=vis, train_carry=carry, shadow_ref=True) if completions is None else np.asarray(completions)
        ref_lp = self.shadow_logps if completions is None else None
        pol_head = self.shadow_policy_head if completions is None else None
        t2 = mark()
        # rewards are computed on the host inside loss_and_grads, after the first forward passes are in the GPU queue (in the
        # phase-timing mode they are evaluated here so that they get their own column)
        rewards = reward_fn(comp) if timing else (lambda: reward_fn(comp))
        t3 = mark()
        self.last_step_traced = bool(carry and carry.get("traced"))     # the decode steps filled the completion rows of the training arena
        last = do_optimizer_step if last_micro_step is None else last_micro_step
        out = self.loss_and_grads(batch, comp, rewards, backward=True, last_micro_step=last, vis=vis, train_carry=carry, defer_metrics=defer_metrics or do_optimizer_step,
                                  ref_logps=ref_lp, policy_head=pol_head)
        self.last_step_shadowed = ref_lp is not None
        t4 = mark()
        if do_optimizer_step:
            self.optimizer_step()
        if not defer_metrics and "finalize" in out["metrics"]:
            out["metrics"].pop("finalize")()
        t5 = mark()
        if timing:
            print(f"[iadr1 timing] vision {1e3*(t1-t0):.1f} ms | rollout {1e3*(t2-t1):.1f} | rewards {1e3*(t3-t2):.1f} | ref+policy fwd/bwd {1e3*(t4-t3):.1f} | optimizer {1e3*(t5-t4):.1f}", flush=True)
        if return_outputs:
            out["completion_ids"] = comp
            return out
        return out["metrics"]
