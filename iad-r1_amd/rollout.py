"""GRPO group rollout: the in-process replacement for the reference's vLLM engine on a side GPU
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:314-358 construction, :637-683 call).

  * shares the policy's weights (no `_move_model_to_vllm` weight push, REF:569-579);
  * prefill runs once per PROMPT with the training kernels and writes K/V into a paged cache; the G
    completions of a prompt share its full prompt pages through the block table (what vLLM's
    `enable_prefix_caching=True` buys the reference, REF:351);
  * one decode step is a fixed launch sequence over static buffers -- embed, per layer {RMSNorm(+residual),
    skinny qkv GEMM, RoPE, KV append, paged attention, skinny o GEMM, RMSNorm, skinny gate|up, SwiGLU, skinny
    down}, final norm, lm_head, fused top-k/top-p sampler, bookkeeping -- captured ONCE in a hipGraph
    (torch.cuda.CUDAGraph) and replayed max_completion_length-1 times; positions, cache slots, context
    lengths, the RNG step and the EOS state all live on the device.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from .vlm import Engine, TextPlan

BF16, F32 = torch.bfloat16, torch.float32
PAGE = 32
# counters of the events that cost host time on ragged real-world batches (bench.py `real_shapes`): KV-pool (re)builds, hipGraph captures, training-arena
# hand-over re-keys, host reads of the EOS state, decode steps run
STATS = {"pool_builds": 0, "graph_captures": 0, "capture_seconds": 0.0, "trace_rekeys": 0, "eos_polls": 0, "decode_steps": 0, "rollouts": 0}


class Rollout:
    def __init__(self, engine: Engine, max_seqs: int, max_prompt: int, max_new: int, max_prompts: int | None = None, use_graph: bool = True, decode_cus: int = 0,
                 decode_stream=None):
        """decode_cus / decode_stream: the decode replays run on `decode_stream`, a CU-masked stream that owns `decode_cus` CUs (hip.cu_mask_stream), next to the
        shadow pass on the other CUs (overlap.ChunkedRefPass); 0 / None: the current stream, the whole device."""
        self.e = engine
        c = engine.cfg
        dev = engine.dev
        STATS["pool_builds"] += 1
        self.N, self.max_new, self.use_graph, self.max_prompt = max_seqs, max_new, use_graph, max_prompt
        L, H, D, Hq, Hkv, I, V = c.num_hidden_layers, c.hidden_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads, c.intermediate_size, c.vocab_size
        max_prompts = max_prompts or max_seqs
        self.max_pages = max_prompt // PAGE + (PAGE - 1 + max_new + PAGE - 1) // PAGE + 1      # full prompt pages + the private pages of the worst remainder (generate())
        # page pool: shared prompt pages + private pages (tail of the prompt + completion) per sequence
        self.n_pages = max_prompts * ((max_prompt + PAGE - 1) // PAGE) + max_seqs * ((max_new + 2 * PAGE - 1) // PAGE + 1) + 1
        self.kc = torch.zeros(L, self.n_pages, Hkv, PAGE, D, dtype=BF16, device=dev)
        self.vc = torch.zeros(L, self.n_pages, Hkv, D, PAGE, dtype=BF16, device=dev)
        N = max_seqs
        i32, i64 = torch.int32, torch.int64
        self.block_table = torch.zeros(N, self.max_pages, dtype=i32, device=dev)
        self.shared_pages = torch.zeros(max_prompts, dtype=i32, device=dev)     # full prompt pages every sequence of a group shares (attn_decode_group)
        self.group_attn, self.G, self.group_chunks, self.group_ws = False, 1, 1, None
        self.G_seq = 0          # sequences per prompt group of the current rollout (placement hint of attn_decode)
        self.pos = torch.zeros(N, dtype=i32, device=dev)
        self.ctx_len = torch.zeros(N, dtype=i32, device=dev)
        self.slot = torch.zeros(N, dtype=i64, device=dev)
        self.finished = torch.zeros(N, dtype=i32, device=dev)
        self.all_done = torch.zeros(1, dtype=i32, device=dev)                  # written by every decode_advance: 1 when all sequences have finished
        self.done_host = torch.zeros(8, dtype=i32).pin_memory() if dev.type == "cuda" else torch.zeros(8, dtype=i32)   # ring of host copies (generate's EOS poll)
        self.step = torch.zeros(1, dtype=i32, device=dev)
        self.cur_tok = torch.zeros(N, dtype=i64, device=dev)
        self.sampled = torch.zeros(N, dtype=i64, device=dev)
        self.out_tokens = torch.zeros(N, max_new, dtype=i64, device=dev)
        # activations of one decode step
        self.x = torch.empty(N, H, dtype=BF16, device=dev)
        # GEMM inputs live in the decode-packed layout (ops.PackedAct: MFMA B-fragment order), written directly by their
        # producers (RMSNorm, decode attention, the fused SwiGLU epilogue) so that every X fragment load of the skinny GEMMs
        # is one contiguous 1 KiB read instead of 16 half cache lines.  IADR1_DECODE_PACKED=0 keeps row-major (A/B switch).
        self.packed = os.environ.get("IADR1_DECODE_PACKED", "1") != "0" and H % 32 == 0 and (Hq * D) % 32 == 0
        self.fuse_swiglu = I % 64 == 0
        act = (lambda k: ops.PackedAct(N, k, dev)) if self.packed else (lambda k: torch.empty(N, k, dtype=BF16, device=dev))
        self.h = act(H)
        self.qkv = torch.empty(N, c.qkv_width, dtype=BF16, device=dev)
        self.o = act(Hq * D)
        self.br = torch.empty(N, H, dtype=BF16, device=dev)
        # split-K of the two narrow-N projections (fp32 partial slabs summed by the fused residual + RMSNorm).  The o projection runs 16-column blocks of 16 waves,
        # one per CU: K is split until the grid fills the CUs ONCE (3B: 128 tiles x 2; 7B: 224 tiles x 1 -- two slices made 448 blocks = two rounds).  The down
        # projection: 8 slices where the persistent X-resident split kernel takes it (<= 48 k-steps per slice: 3B), otherwise the one-shot 64-column kernel with
        # (H / 64) x ks ~ the CU count (7B: 56 x 4).  Measured on the 7B shapes, decode step in ms: (2, 8) 4.83, (1, 8) 4.80, (2, 4) 4.71, (1, 4) 4.64, (1, 6) 4.96,
        # (1, 2) 5.18 (profiles/r04_decode_ksplit_7b.txt).
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 256
        self.decode_cus, self.decode_stream = int(decode_cus), decode_stream
        if self.decode_cus:
            ncu = self.decode_cus
        if H * Hq * D >= 1 << 20:
            self.ks_o = max(1, min(2, ncu // max(1, H // 16)))
            # (the persistent split kernel takes <= 80 k-steps per slice -- the launcher's own gate, gemm.hip)
            self.ks_down = 8 if (I // 32 + 7) // 8 <= 80 else max(1, min(8, ncu // max(1, H // 64)))
        else:
            self.ks_o, self.ks_down = 1, 1
        if os.environ.get("IADR1_DECODE_KS"):
            self.ks_o, self.ks_down = (int(z) for z in os.environ["IADR1_DECODE_KS"].split(","))
        self.part_o = torch.empty(self.ks_o, N, H, dtype=F32, device=dev)
        self.part_d = torch.empty(self.ks_down, N, H, dtype=F32, device=dev)
        self.gu = torch.empty(N, 2 * I, dtype=BF16, device=dev)
        self.a = act(I) if self.fuse_swiglu else torch.empty(N, I, dtype=BF16, device=dev)
        self.logits = torch.empty(N, V, dtype=F32, device=dev)
        self.cos = torch.empty(N, D // 2, dtype=F32, device=dev)
        self.sin = torch.empty(N, D // 2, dtype=F32, device=dev)
        self.graph = None
        # progress word of the decode step (step * layers + layer, stored by the first kernel of every decoder layer) and the weight prefetcher it paces
        # (wprefetch.WeightPrefetcher, set by the owner; None: no marks in the captured step)
        self.mark = torch.zeros(1, dtype=i32, device=dev)
        self.pf_stop = torch.zeros(1, dtype=i32, device=dev)        # sequence number of the last finished rollout (the prefetcher's stop word)
        self._pf_epoch = 0
        self.wprefetch = None
        self._marks_in_graph = False
        self._marks_clean = True
        self._toks_host, self._toks_event = None, None
        self.trace = None           # training arena filled by the decode steps (generate(train_trace=...)); part of the captured graph
        self.decode_events = None   # bench.py sets a list: (start event, end event, decode steps, sum of prompt lengths over sequences) per call
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=dev)
        self.sampling = dict(temperature=0.9, top_k=50, top_p=0.9, suppress=-1, eos=c.eos_token_id, pad=c.pad_token_id)

    # ---- one decode step (graph body) -------------------------------------------------------------------------
    def _decode_step(self):
        e, c, P = self.e, self.e.cfg, self.e.p
        tr = self.trace      # None, or the training arena the step also fills (Rollout.generate(train_trace=...))

        marks = self.wprefetch is not None
        self._marks_in_graph = marks

        def side(mark_layer=None, **kw):
            """iadr1_side_out_t for one launch of this step: rows base + s * stride + *step of the given arena tensors (None without a trace).
            The structs are host memory read at launch time; they are kept in self._sides so that a re-capture builds them anew.
            mark_layer: this launch opens decoder layer `mark_layer` -- with a weight prefetcher attached it also stores the step's progress mark."""
            mk = dict(mark=self.mark, mark_mul=c.num_hidden_layers, mark_add=mark_layer) if (marks and mark_layer is not None) else {}
            if tr is None:
                if not mk:
                    return None
                so = ops.SideOut.make(self.step, 0, 0, **mk)
            else:
                so = ops.SideOut.make(self.step, tr["base"], tr["stride"], **kw, **mk)
            self._sides.append(so)
            return so

        self._sides = []
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        qw, kw = Hq * D, Hkv * D
        s = self.sampling
        ops.embed_fwd(self.cur_tok, None, P.w("embed"), None, out=self.x)
        # (the rotary table of this step's positions was written by the decode_advance that closed the previous step)
        have_branch = False
        T_ = lambda name, i: None if tr is None else tr[name][i]
        for i in range(c.num_hidden_layers):
            b = f"layers.{i}."
            if not have_branch:      # layer 0 enters with the embedding rows (filled after the rollout)
                ops.rmsnorm_fwd(self.x, P.w(b + "ln1"), c.rms_norm_eps, out=self.h, side=side(mark_layer=i, p1=T_("h1", i), p2=T_("rstd1", i)))
            else:
                ops.rmsnorm_fwd(None, P.w(b + "ln1"), c.rms_norm_eps, res=self.x, res_out=self.x, x32=self.part_d, out=self.h,
                                side=side(mark_layer=i, p0=T_("x_in", i), p1=T_("h1", i), p2=T_("rstd1", i)))
            if P.qkv_rope_packed:   # q|k|v projection + rotary + K/V cache append in one launch
                ops.gemm_qkv_rope_kv(self.h, P.wpk(b + "qkv.w"), P.wpk_bias(b + "qkv.w"), self.qkv, self.cos, self.sin, self.slot, self.kc[i], self.vc[i], Hq, Hkv, D,
                                     side=side(p0=T_("qkv", i)))
            else:
                assert tr is None
                ops.gemm_skinny(self.h, P.wpk(b + "qkv.w"), c.qkv_width, bias=P.w(b + "qkv.b"), out=self.qkv)
                ops.rope_kv_store(self.qkv, self.cos, self.sin, self.slot, self.kc[i], self.vc[i], Hq, Hkv, D)
            self._attention(i, side(p0=T_("o", i), p1=T_("lse", i), ld1=None if tr is None else tr["lse"][i].stride(0)))
            ops.gemm_skinny(self.o, P.wpk(b + "o.w"), c.hidden_size, out=self.part_o, ksplit=self.ks_o)
            ops.rmsnorm_fwd(None, P.w(b + "ln2"), c.rms_norm_eps, res=self.x, res_out=self.x, x32=self.part_o, out=self.h,
                            side=side(p0=T_("x_mid", i), p1=T_("h2", i), p2=T_("rstd2", i)))
            if self.fuse_swiglu:
                ops.gemm_skinny(self.h, P.wpk(b + "gu.w"), 2 * c.intermediate_size, out=self.a, swiglu=True, side=None if (tr is not None and tr["mlp_on_shadow"]) else side(p0=T_("gu", i), p1=T_("a", i)))
            else:
                assert tr is None
                ops.gemm_skinny(self.h, P.wpk(b + "gu.w"), 2 * c.intermediate_size, out=self.gu)
                ops.swiglu_fwd(self.gu, out=self.a)
            ops.gemm_skinny(self.a, P.wpk(b + "down.w"), c.hidden_size, out=self.part_d, ksplit=self.ks_down)
            have_branch = True
        ops.rmsnorm_fwd(None, P.w("norm"), c.rms_norm_eps, res=self.x, res_out=self.x, x32=self.part_d, out=self.h,
                        side=side(p0=None if tr is None else tr["x_last"], p1=None if tr is None else tr["hf"], p2=None if tr is None else tr["rstdf"]))
        ops.gemm_skinny(self.h, P.wpk(P.lm_head_name()), c.vocab_size, out=self.logits)
        self._sample_and_advance()

    def _attention(self, i, side):
        """Decode attention of layer i.  Long shared prompts / many kv heads: one block per (prompt group, kv head) reads the group's full prompt pages once for all
        its sequences (iadr1_attn_decode_group); otherwise one block per (sequence, kv head)."""
        c = self.e.cfg
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        q = self.qkv[:, : Hq * D]
        if self.group_attn:
            ops.attn_decode_group(q, self.kc[i], self.vc[i], self.block_table, self.ctx_len, self.shared_pages[: self.N // self.G], self.G, Hq, Hkv, D, c.attn_scale, out=self.o, side=side,
                                  chunks=self.group_chunks, ws=self.group_ws)
        else:
            ops.attn_decode(q, self.kc[i], self.vc[i], self.block_table, self.ctx_len, Hq, Hkv, D, c.attn_scale, out=self.o, side=side, seqs_per_group=self.G_seq)

    def _sample_and_advance(self):
        s = self.sampling
        ops.sample(self.logits, s["temperature"], s["top_k"], s["top_p"], 0, 0, suppress_token=s["suppress"], step_ptr=self.step, out=self.sampled, seed_ptr=self.seed_dev)
        ops.decode_advance(self.sampled, self.cur_tok, self.out_tokens, self.pos, self.ctx_len, self.slot, self.block_table, self.finished, self.step, s["eos"], s["pad"],
                           all_done=self.all_done, inv_freq=self.e.inv_freq, cos=self.cos, sin=self.sin)

    def _capture(self):
        # warm-up on a side stream (first-call attribute setup must not happen under capture), then capture once
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._decode_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._decode_step()
        self.graph = g
        STATS["graph_captures"] += 1

    # ---- public ---------------------------------------------------------------------------------------------------
    def generate(self, *args, **kw) -> torch.Tensor:
        """`_generate` with this thread's launcher configuration (iadr1_set_decode_cus: the CU count the persistent decode grids are sized for) set for the
        duration of the call only: whatever ends the call, later skinny-GEMM callers of this thread size their grids for the whole device again."""
        try:
            return self._generate(*args, **kw)
        finally:
            if self.e.dev.type == "cuda":
                ops.hip.set_decode_cus(0)

    def _generate(self, plan: TextPlan, img_embeds, G: int, max_new: int, temperature=0.9, top_k=50, top_p=0.9, seed=0, suppress_eos=False,
                  stop_at_eos=True, train_carry=None, train_trace=False, shadow=None) -> torch.Tensor:
        """plan: the Bp left-padded prompts.  Returns completion ids [Bp*G, max_new] (prompt-major order: p0 x G,
        p1 x G, ...), pad after the first EOS.
        shadow: an overlap.ChunkedRefPass -- the frozen reference's teacher-forced pass over the tokens produced so far is enqueued on ITS stream every
        `shadow.steps` decode steps, behind one event per chunk (the caller reads shadow.logp after shadow.join())."""
        e, c = self.e, self.e.cfg
        dev = e.dev
        Bp, S = plan.B, plan.S
        N = Bp * G
        assert N == self.N, f"rollout was built for {self.N} sequences, got {N}"
        assert max_new <= self.max_new
        if dev.type == "cuda":
            ops.hip.set_decode_cus(self.decode_cus)      # this thread's launcher configuration (persistent grids) for the duration of the call: generate() resets it
        self.seed_dev.fill_(int(seed) & 0x7FFFFFFFFFFFFFFF)   # device-resident: a new seed per rollout does not invalidate the captured graph
        sampling = dict(temperature=float(temperature), top_k=int(top_k), top_p=float(top_p),
                        suppress=c.eos_token_id if suppress_eos else -1, eos=c.eos_token_id if stop_at_eos and not suppress_eos else -1, pad=c.pad_token_id)
        if sampling != self.sampling:
            self.sampling = sampling
            self.graph = None  # sampling parameters are kernel arguments frozen in the graph
        want_trace = bool(train_trace and train_carry is not None)
        # overlap mode: the gate|up / SwiGLU rows of the training arena are rebuilt by the shadow pass (overlap.ChunkedRefPass._policy_mlp) instead of stored here
        mlp_on_shadow = bool(want_trace and shadow is not None and shadow.rebuilds_policy_mlp(c, N))
        if not want_trace and self.trace is not None:
            self.trace, self.graph = None, None
        if (self.wprefetch is not None) != self._marks_in_graph:
            self.graph = None      # the progress marks are part of the captured step
        lengths = plan.lengths
        # ---- page tables ------------------------------------------------------------------------------------
        next_page = 1  # page 0 is a scratch page for unused table entries
        bt = np.zeros((N, self.max_pages), dtype=np.int32)
        slot_shared = np.full(Bp * S, -1, dtype=np.int64)
        slot_tail = np.full((G, Bp * S), -1, dtype=np.int64)
        any_tail = False
        for b in range(Bp):
            n = int(lengths[b])
            full = n // PAGE
            shared = list(range(next_page, next_page + full))
            next_page += full
            first = S - n  # left padding
            if full:
                jj = np.arange(full * PAGE)
                slot_shared[b * S + first + jj] = np.asarray(shared, dtype=np.int64)[jj // PAGE] * PAGE + jj % PAGE
            priv = (n - full * PAGE + max_new + PAGE - 1) // PAGE + 1
            for gi in range(G):
                r = b * G + gi
                pages = shared + list(range(next_page, next_page + priv))
                next_page += priv
                assert len(pages) <= self.max_pages and next_page <= self.n_pages, "KV page pool too small"
                bt[r, : len(pages)] = pages
                for j in range(full * PAGE, n):
                    slot_tail[gi, b * S + first + j] = pages[j // PAGE] * PAGE + j % PAGE
                    any_tail = True
        self.block_table.copy_(torch.from_numpy(bt).pin_memory(), non_blocking=True)      # (pinned: a pageable upload synchronises the stream, ops.h2d)
        # group-shared decode attention when it pays.  Measured on MI355X (decode step, 64 sequences, 7B-class shapes, profiles/r02_group_attention.txt; the
        # per-sequence kernel with its blocks placed one prompt group per XCD, the group kernel with one block per (group, kv head)):
        #   LLaVA-1.5 (MHA, 32 kv heads, 832 shared tokens: 2048 per-sequence blocks)      7.69 vs 6.30 ms   -> group kernel
        #   LLaVA-NeXT (8 kv heads, 3168 shared tokens: 512 per-sequence blocks)           6.48 vs 6.53 ms   -> per-sequence kernel
        #   LLaVA-OneVision (4 kv heads, 3936 shared tokens: 256 per-sequence blocks)      5.61 vs 6.98 ms   -> per-sequence kernel
        # Rule: N * Hkv >= 1024 per-sequence blocks (several resident rounds: the copies of a page are then read at different times) and >= 256 shared tokens.
        # IADR1_DECODE_GROUP_ATTN=0|1 forces it.
        if self.G_seq != G:
            self.G_seq, self.graph = G, None
        shared_tok = np.array([int(lengths[b]) // PAGE * PAGE for b in range(Bp)])
        self.shared_pages[:Bp].copy_(torch.from_numpy((shared_tok // PAGE).astype(np.int32)).pin_memory(), non_blocking=True)
        want = os.environ.get("IADR1_DECODE_GROUP_ATTN")
        rows_ok = G * (c.num_attention_heads // c.num_key_value_heads) <= 64
        use = rows_ok and G > 1 and N * c.num_key_value_heads >= 1024 and int(shared_tok.min()) >= 256
        use = rows_ok and (use if want is None else want == "1")
        # few (group, kv head) blocks for a long shared part: split it over chunks of >= 8 pages until ~256 blocks stream (two launches, partial states through a workspace)
        chunks = 1
        if use:
            blocks = Bp * c.num_key_value_heads
            chunks = max(1, min(256 // max(blocks, 1), int(shared_tok.min()) // PAGE // 8, 16))
        if use != self.group_attn or (use and (G != self.G or chunks != self.group_chunks)):
            self.group_attn, self.G, self.group_chunks, self.graph = use, G, chunks, None
            self.group_ws = ops.attn_decode_group_ws(self.N, G, c.num_attention_heads, c.num_key_value_heads, c.head_dim, chunks, dev) if (use and chunks > 1) else None
        slot_shared_d = ops.h2d(slot_shared, dev)
        slot_tail_d = ops.h2d(slot_tail, dev) if any_tail else None
        Hkv, D = c.num_key_value_heads, c.head_dim

        def kv_sink(i, k, v):
            ops.kv_store(k, v, slot_shared_d, self.kc[i], self.vc[i], Hkv, D)
            if slot_tail_d is not None:
                for gi in range(G):
                    ops.kv_store(k, v, slot_tail_d[gi], self.kc[i], self.vc[i], Hkv, D)

        # ---- prefill (once per prompt) ----------------------------------------------------------------------
        if train_carry is not None:
            # the prefill doubles as the prompt part of the policy's training forward (Engine.text_forward two-phase mode): activations
            # saved in the training arena, rows [0, Bp*S) of the shared-prefix batch of T_total = Bp*S + N*max_new rows
            hf, _ = e.text_forward(plan, img_embeds, save=True, kv_sink=kv_sink, rows=(0, Bp * S, Bp * S + N * max_new), carry=train_carry)
            hf = hf[: Bp * S]
            if want_trace:
                # the decode steps also fill the completion rows of the training arena: decode replay k processes completion token k of every
                # sequence = row Bp*S + s*max_new + k; the step counter reads k + 1 during that replay (it counts sampled tokens)
                T_all = Bp * S + N * max_new
                A = e._text_buffers(T_all, True)
                L = c.num_hidden_layers
                tr = {k: [A[k][i, :T_all] for i in range(L)] for k in ("x_in", "h1", "qkv", "o", "x_mid", "h2", "gu", "a")}
                tr.update(rstd1=[A["rstd1"][i, :T_all] for i in range(L)], rstd2=[A["rstd2"][i, :T_all] for i in range(L)],
                          lse=[A["lse"][i].view(-1)[: c.num_attention_heads * T_all].view(c.num_attention_heads, T_all) for i in range(L)],
                          x_last=train_carry["x_last"], hf=train_carry["hf"], rstdf=train_carry["rstdf"], base=Bp * S - 1, stride=max_new)
                key = (A["x_in"].data_ptr(), train_carry["hf"].data_ptr(), T_all, max_new, mlp_on_shadow)
                if self.trace is None or self.trace.get("key") != key:
                    self.graph = None       # the arena pointers are kernel arguments frozen in the graph
                    STATS["trace_rekeys"] += 1
                    # the LAST completion token of a sequence is never a decode input (it is only sampled): its rows are not written by the steps and
                    # nothing in the loss depends on them, but backward reads them -- they must be finite.  Zero the completion block once.
                    for k, v in tr.items():
                        for t_ in (v if isinstance(v, list) else [v]):
                            if isinstance(t_, torch.Tensor):
                                (t_[:, Bp * S:] if (t_.dim() == 2 and t_.shape[0] == c.num_attention_heads and k == "lse") else t_[Bp * S:]).zero_()
                tr["key"], tr["mlp_on_shadow"] = key, mlp_on_shadow
                self.trace = tr
        else:
            hf, _ = e.text_forward(plan, img_embeds, save=False, kv_sink=kv_sink)
        last_rows = torch.arange(Bp, device=dev, dtype=torch.int64) * S + (S - 1)
        lg = e.logits_rows(hf, last_rows)                                    # [Bp, V] fp32
        self.logits.copy_(lg.repeat_interleave(G, 0))
        del hf, lg
        # ---- device state for the first generated token ----------------------------------------------------
        rep = lambda a: np.repeat(a, G)
        first_pos = rep(lengths + plan.rope_deltas)                         # M-RoPE position of the first new token
        self.pos.copy_(torch.from_numpy((first_pos - 1).astype(np.int32)).pin_memory(), non_blocking=True)
        self.ctx_len.copy_(torch.from_numpy(rep(lengths).astype(np.int32)).pin_memory(), non_blocking=True)
        self.finished.zero_()
        self.step.zero_()
        self.out_tokens.fill_(c.pad_token_id)
        self._sample_and_advance()                                           # token 0 from the prefill logits
        # ---- decode ----------------------------------------------------------------------------------------------
        if self.use_graph and self.graph is None:
            state = (self.pos, self.ctx_len, self.slot, self.finished, self.step, self.cur_tok, self.out_tokens, self.cos, self.sin, self.all_done)
            saved = [t.clone() for t in state]
            import time as _time
            torch.cuda.synchronize()
            _t0 = _time.perf_counter()
            self._capture()  # warm-up + capture advance the state twice: restore it (K/V written meanwhile are rewritten by the real steps)
            self._marks_clean = False
            for t, s_ in zip(state, saved):
                t.copy_(s_)
            STATS["capture_seconds"] += _time.perf_counter() - _t0
        bounds = ()
        if shadow is not None:
            shadow.begin(plan, G, max_new, first_pos, self.out_tokens, step_counter=self.step, policy=(e, self.trace) if mlp_on_shadow else None)
            bounds = shadow.boundaries(max_new)
            gate = torch.cuda.Event()
            gate.record()                  # behind the sampling of token 0 (and the prefill): the reference's vision tower, prompt rows and first log-prob start now
            shadow.prompt_phase(gate)
        if self.wprefetch is not None:     # one persistent launch for the whole rollout, on its own CU-masked stream; it polls the progress word the replays store
            Lm = c.num_hidden_layers
            if not self._marks_clean:      # (a capture / warm-up ran decode steps, or an exception cut a rollout short: the word must read 0 when the launch starts)
                self.mark.zero_()
                torch.cuda.current_stream().synchronize()
                self._marks_clean = True
            self._pf_epoch += 1
            self.wprefetch.start(self.mark, 1 * Lm, (max_new - 1) * Lm + Lm - 1, self.pf_stop, self._pf_epoch)       # no stream dependency: see csrc/prefetch.hip
            self._marks_clean = False
        # the CU-masked decode stream (and the host join that goes with it) only when something runs NEXT TO the replays: a rollout without the shadow pass
        # (SCGRPOTrainer.training_step's batched rollouts, step(completions=...), the evaluation harness) has nothing to hide behind the +0.2 ms per step
        ds = self.decode_stream if (shadow is not None or self.wprefetch is not None) else None
        _outer = torch.cuda.current_stream()
        if ds is not None:                 # hand the rest of the rollout over to the CU-masked decode stream (the prefill above ran on the whole device)
            ds.wait_stream(_outer)
            torch.cuda.set_stream(ds)
        try:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if self.decode_events is not None else None
            if ev:
                ev[0].record()
            nsteps = 0
            # EOS live: stop once every sequence has finished, WITHOUT draining the queue.  Every POLL steps the device flag decode_advance maintains is copied
            # to pinned host memory behind the step; the host reads the copy that is LAG polls old (it waits for that copy's event, never for the newest work), so the
            # GPU always has >= (LAG - 1) * POLL queued steps and at most LAG * POLL steps run past the last EOS.  (Round 3 read `finished.all()` every 32 steps: a full
            # drain of the queue each time and 16 wasted steps on average.)
            POLL, LAG = 4, 3
            polls = []
            live = sampling["eos"] >= 0
            for it in range(1, max_new):
                if self.graph is not None:
                    self.graph.replay()
                else:
                    self._decode_step()
                nsteps += 1
                if shadow is not None and it in bounds:      # replay `it` has produced token `it`: rows [.., it) of every sequence and their targets are final
                    shadow.chunk(it, None)                   # (gated on the device step counter: overlap.ChunkedRefPass.chunk)
                if live and it % POLL == 0:
                    k = (it // POLL) % self.done_host.numel()
                    self.done_host[k: k + 1].copy_(self.all_done, non_blocking=True)
                    pe = torch.cuda.Event()
                    pe.record()
                    polls.append((pe, k))
                    if len(polls) >= LAG:
                        ev0, k0 = polls.pop(0)
                        ev0.synchronize()
                        STATS["eos_polls"] += 1
                        if int(self.done_host[k0]):
                            break
            if self.wprefetch is not None:     # behind the last replay (the host joins this stream below): releases the prefetcher, also when EOS ended the rollout early,
                self.pf_stop.fill_(self._pf_epoch)      # and leaves the progress word at 0 for the next rollout's launch
                self.mark.zero_()
                self._marks_clean = ds is not None      # (only the masked decode stream is joined on the host below)
            STATS["decode_steps"] += nsteps
            STATS["rollouts"] += 1
            if ev:
                ev[1].record()
                self.decode_events.append((ev[0], ev[1], nsteps, int(np.sum(lengths)) * G))
        finally:
            # whatever happened in the loop (a failed launch, an exception of the shadow pass): the process's current stream must not stay the CU-masked blocking
            # stream
            if ds is not None:
                torch.cuda.set_stream(_outer)
            if self.wprefetch is not None and not self._marks_clean:
                self.pf_stop.fill_(self._pf_epoch)      # (an exception cut the loop short: release the resident prefetch launch; the next start() zeroes the progress word)
        if ds is not None:
            # Join on the HOST: anything left pending on the outer stream's hardware queue while the replays run -- the dependency packet of `_outer.wait_stream(ds)`
            # (the host is hundreds of replays ahead), or a kernel polling the step counter -- costs every decode launch ~8 us when that queue happens to share a
            # dispatch pipe with the decode queue: 5.0 instead of 3.0 ms per step (tools/decode_mask_probe.py plain 192, profiles/r05_decode_join.txt: event join 1208.8,
            # counter join the same, host join 1202.2 ms per step).  Which queues share a pipe depends on the order in which the process created them, so nothing may
            # be pending: the host waits for the last replay (it needs the tokens next anyway) and enqueues the rest behind it.
            _je = torch.cuda.Event()
            _je.record(ds)
            _je.synchronize()
        toks = self.out_tokens[:, :max_new].clone()
        # the host copy of the tokens leaves BEFORE the shadow pass's tail is enqueued: the caller (rewards, the training batch's plans) waits for the decode replays
        # only, and prepares the next phase while the tail runs (tokens_host)
        if self._toks_host is None or self._toks_host.shape != toks.shape:
            self._toks_host = torch.empty(toks.shape, dtype=toks.dtype).pin_memory()
        self._toks_host.copy_(toks, non_blocking=True)
        self._toks_event = torch.cuda.Event()
        self._toks_event.record()
        if shadow is not None:
            shadow.finish(nsteps + 1, None)        # (the current stream is ordered behind the last replay)
            if shadow.trace is not None:
                print("[iadr1 overlap] side-stream phases (ms after the start of the decode loop):", shadow.report(), flush=True)
        return toks

    def tokens_host(self) -> np.ndarray:
        """The last generate()'s completion ids on the host (waits for the decode replays, not for what was enqueued behind them)."""
        self._toks_event.synchronize()
        return self._toks_host.numpy().copy()
