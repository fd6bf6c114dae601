"""Qwen2.5-VL forward / backward on the HIP kernel library -- the engine behind `model(**inputs).logits`
at /root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:505 and the PA-SFT forward(labels)
(TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:1308-1400).  Explicit (hand-scheduled) backward instead of a
tracing autograd: every saved tensor, every GEMM and its operand layout is chosen here.

Conventions: activations are flat [tokens, width] bf16; a decoder micro-batch of B sequences x S positions
is T = B*S rows (left padding stays in place but belongs to no attention segment); the vision tower runs
ONCE per unique image per step and its output rows are fanned out to every sequence that shows that image
(the reference recomputes it G times on identical pixels -- same values, same summed gradients).
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass, field

import numpy as np
import torch

from . import indexing, llava_ov, ops
from .params import ParamStore, VLMConfig

# IADR1_POISON=1: outputs that skip their zero fill (every row is written by the kernel) start as NaN instead of stale memory, so a row the kernel
# does not write shows up in the tests instead of passing on a lucky allocation
_POISON = bool(os.environ.get("IADR1_POISON"))

BF16, F32 = torch.bfloat16, torch.float32


# ------------------------------------------------------------------------------------------------------------
# plans: host-side integer work + small device tables
# ------------------------------------------------------------------------------------------------------------
@dataclass
class VisionPlan:
    n_patches: int
    win_index: torch.Tensor      # [N/m2] int64 gather (window order <- raster order of merged groups)
    rev_index: torch.Tensor      # inverse permutation
    seg_window: ops.Segments
    seg_full: ops.Segments
    cos: torch.Tensor            # [N, d/2] fp32, already in window order
    sin: torch.Tensor


@dataclass
class SiglipPlan:
    """LLaVA-OneVision vision plan for a list of image sizes (height, width): crops are the attention segments of the SigLIP tower, `pack` is the
    sparse map projector rows -> packed image tokens (iadr1_amd.llava_ov.pack_plan) and `pack_t` its transpose for the backward pass."""
    n_crops: int
    n_patches: int               # n_crops * tokens per crop
    seg: ops.Segments
    pos_ids: torch.Tensor        # [n_patches] int64: position-embedding row of every patch row
    pack: tuple                  # (ptr, idx, w) device tensors; T_img packed tokens
    pack_t: tuple
    n_tokens: int
    lens: list
    crops: list
    # CLIP towers (class token in front of every crop's patch tokens): `assemble` builds the tower's input rows from [patch-embedding rows ; class embedding],
    # `select` drops the class rows again in front of the projector; each with its transpose for the backward pass.  None for SigLIP.
    assemble: tuple = None
    assemble_t: tuple = None
    select: tuple = None
    select_t: tuple = None
    _pos_scatter: tuple = None

    def pos_scatter(self, device, seq):
        """Ordered scatter plan of the learned position rows: tower row r reads position row r % seq (backward: rows_scatter_acc)."""
        if self._pos_scatter is None:
            self._pos_scatter = ops.scatter_plan(np.tile(np.arange(seq, dtype=np.int64), self.n_crops), device)
        return self._pos_scatter


@dataclass
class TextPlan:
    B: int
    S: int
    ids: torch.Tensor            # [T] int64
    img_index: torch.Tensor      # [T] int32, row of the image-embed matrix or -1
    seg: ops.Segments            # one segment per sequence, skipping left padding
    cos: torch.Tensor            # [T, D/2] fp32 (M-RoPE component already selected per frequency)
    sin: torch.Tensor
    rope_deltas: np.ndarray      # [B]
    lengths: np.ndarray          # [B] number of real tokens
    shared: tuple = None         # (ng, P, n, C, G) when built by text_plan_shared
    tail: "TextPlan" = None      # completion rows only (two-phase forward), set by text_plan_shared
    host: tuple = None           # (ids [T], img_index [T]) numpy copies: the backward's ordered scatter plans are built from them (scatter_plans)
    _scatter: tuple = None

    def scatter_plans(self, device):
        """(embedding-row plan, image-row plan) for ops.rows_scatter_acc: which token rows feed which embedding row (by token id) / image-embedding row, as CSR
        lists in ascending order -- the backward of the embedding gather as an ORDERED sum (no atomics).  Built once per plan, on first use, from the host ids."""
        if self._scatter is None:
            ids, img = self.host
            text = np.where(img < 0, ids, -1)
            self._scatter = (ops.scatter_plan(text, device), ops.scatter_plan(img, device))
        return self._scatter


class _Ring:
    """Round-robin static buffers shared between the main stream (writer, reader) and a side stream (reader): `take` hands out the next slot after making
    the current stream wait for the side-stream reader that last used it; `busy(slot, stream)` records that reader."""

    def __init__(self, bufs):
        self.bufs, self.ev, self.k = bufs, [None] * len(bufs), 0

    def take(self, rows):
        k = self.k
        self.k = (k + 1) % len(self.bufs)
        if self.ev[k] is not None:
            torch.cuda.current_stream().wait_event(self.ev[k])
            self.ev[k] = None
        return self.bufs[k][:rows], (self, k)

    def busy(self, k, stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        self.ev[k] = ev


class Engine:
    def __init__(self, params: ParamStore):
        self.p = params
        self.cfg: VLMConfig = params.cfg
        self.dev = params.device
        c = self.cfg
        hd, hr = c.head_dim, c.head_dim_real
        inv = np.zeros(hd // 2, dtype=np.float32)      # padded heads (VLMConfig.head_dim): the rotary pairs past the real width rotate zeros, frequency 0
        inv[: hr // 2] = 1.0 / (c.rope_theta ** (np.arange(0, hr, 2, dtype=np.float32) / hr))
        self.inv_freq = torch.tensor(inv, dtype=F32, device=self.dev)
        self.mrope_comp = torch.tensor(indexing.mrope_component_of_channel(c.mrope_section, hd // 2), dtype=torch.long, device=self.dev)
        vd = c.v_head_dim
        self.v_inv_freq = 1.0 / (10000.0 ** (np.arange(0, vd // 2, 2, dtype=np.float32) / (vd // 2)))
        self.lm_chunk = 4096
        # side stream of the weight-gradient GEMMs (_wgrad).  Same priority as the main stream: measured with the dgrad chain on a high-priority stream
        # (torch priority -1; the range here is (0, -1)): 1294-1306 ms/step against 1272 -- the starved wgrad queue lengthens the join at the end
        self.wgrad_tn_wide = False      # ops.gemm_tn_acc(wide=...): the owner sets it (sft.SFTEngine: True)
        # (which pool stream -- hardware queue -- the side stream is does not matter for two ordinary streams: PA-SFT 3B 356.4 - 358.4 ms per step over six choices)
        self.wgrad_stream = torch.cuda.Stream() if (self.dev.type == "cuda" and os.environ.get("IADR1_WGRAD_STREAM", "1") != "0") else None
        self.keep_logits_bytes = 24 << 30
        # lm_head + log-softmax + gather.  "fused" (default): the linear_logprob kernels -- logits never reach HBM, the backward recomputes them straight into
        # dlogits; "logits": the two-step form (fp32 logit chunks + row kernels; a differentiated pass keeps its logits when they fit `keep_logits_bytes`).
        # Measured on the 3B bench shape: 1261.9 (fused) vs 1263.2 ms per step, and 10 GB of kept logits less.  Policy and frozen reference always run the
        # same form, so a policy that equals the reference has KL == 0 exactly (bit-equal log-probs).
        self.head_mode = os.environ.get("IADR1_HEAD", "fused")
        assert self.head_mode in ("fused", "logits"), self.head_mode
        self._ws = {}

    def _workspace(self, key, shape, dtype):
        """Persistent scratch (lm_head logit chunks are GBs: re-allocating them per call costs ~100 ms of page mapping)."""
        t = self._ws.get(key)
        if t is None or t.shape != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.dev)
            self._ws[key] = t
        return t

    def _workspace_rows(self, key, lead, T, width, dtype):
        """Grow-only scratch [*lead, cap >= T, width] sliced to T rows: token counts vary from micro-batch to micro-batch (image sizes, prompt lengths), and
        `_workspace` would free + re-malloc a multi-GB block on every change of shape (ckpt_x_in is ~4 GB at 7B)."""
        t = self._ws.get(key)
        if t is None or t.shape[:-2] != tuple(lead) or t.shape[-2] < T or t.shape[-1] != width or t.dtype != dtype:
            self._ws[key] = t = None        # release before growing
            t = torch.empty((*lead, T, width), dtype=dtype, device=self.dev)
            self._ws[key] = t
        return t[..., :T, :]

    # ========================================================================================================
    # plans
    # ========================================================================================================
    def vision_plan(self, grids) -> VisionPlan:
        """Host-side index plan of the vision tower for a list of (t, h, w) patch grids.  Plans are immutable and depend on the grids only, so the last few
        are kept: with a pixel cap most batches repeat a handful of grid lists, and building one (window order, rotary table, segment lists, their
        host-to-device copies) costs ~15 ms of host time at the start of a step, when the GPU has nothing else queued."""
        key = tuple(tuple(int(z) for z in g) for g in grids)
        cache = self.__dict__.setdefault("_vision_plans", {})
        plan = cache.get(key)
        if plan is None:
            plan = self._vision_plan_build(list(key))
            if len(cache) >= 16:
                cache.pop(next(iter(cache)))
            cache[key] = plan
        return plan

    def vision_inputs(self, batch):
        """Batch dict -> (grids, vision plan, pixel tensor on the device, first image-embed row of every image).
        Qwen-VL: `image_grid_thw` [(t, h, w)] + `pixel_values` [patches, patch_dim] (fp32 from the processor, cast to bf16 here).
        LLaVA-OneVision: `image_sizes` [(height, width)] + `pixel_values` as the HF processor emits them -- [images, max crops, 3, S, S] padded to the
        largest crop count (each image's first num_crops are real: transformers llava_onevision get_image_features), a list of per-image crop stacks,
        or the crops already concatenated [total crops, 3, S, S]."""
        c = self.cfg
        if c.is_llava:
            if c.llava_family == "llava":     # LLaVA-1.5: one crop per image [images, 3, S, S], no sizes (the processor resized / centre-cropped to S x S)
                n_img = len(batch["pixel_values"]) if isinstance(batch["pixel_values"], (list, tuple)) else int(torch.as_tensor(batch["pixel_values"]).shape[0])
                sizes = [(c.v_image_size, c.v_image_size)] * n_img
            else:
                sizes = [(int(h), int(w)) for h, w in np.asarray(batch["image_sizes"]).reshape(-1, 2)]
            plan_v = self.vision_plan(sizes)
            px = batch["pixel_values"]
            if isinstance(px, (list, tuple)):
                px = torch.cat([torch.as_tensor(z) for z in px], 0)
            px = torch.as_tensor(px)
            if px.dim() == 5:
                px = torch.cat([px[i, :n] for i, n in enumerate(plan_v.crops)], 0)
            assert px.dim() == 4 and px.shape[0] == plan_v.n_crops, (tuple(px.shape), plan_v.n_crops)
            return sizes, plan_v, px.to(self.dev), np.cumsum([0] + list(plan_v.lens))
        grids = [tuple(int(z) for z in g) for g in np.asarray(batch["image_grid_thw"])]
        plan_v = self.vision_plan(grids)
        px = torch.as_tensor(batch["pixel_values"]).to(self.dev)
        px = px if px.dtype == BF16 else ops.cast_f32_to_bf16(px.to(F32).contiguous())
        m2 = c.v_merge**2
        return grids, plan_v, px, np.cumsum([0] + [g[0] * g[1] * g[2] // m2 for g in grids])

    def n_image_tokens(self, g) -> int:
        """Placeholder tokens one image occupies in the prompt: merged patches of a Qwen grid (t, h, w), packed any-resolution features of a
        LLaVA-OneVision image size (height, width)."""
        c = self.cfg
        if c.is_llava:
            cache = self.__dict__.setdefault("_ov_tokens", {})
            key = (int(g[0]), int(g[1]))
            if c.llava_family == "llava":
                return c.v_tokens
            if key not in cache:
                cache[key] = llava_ov.num_image_tokens(key, c.image_grid_pinpoints, c.v_image_size, c.v_side, c.anyres_max)
            return cache[key]
        return int(g[0]) * int(g[1]) * int(g[2]) // c.v_merge**2

    def _siglip_plan_build(self, sizes) -> SiglipPlan:
        c = self.cfg
        dev = self.dev
        up = lambda a: ops.h2d(np.ascontiguousarray(a), dev)
        per, seq = c.v_tokens, c.v_seq
        if c.llava_family == "llava":         # one crop per image, its patch features ARE the image tokens: no packing map
            crops, lens, pack, pack_t, n_tok = [1] * len(sizes), [per] * len(sizes), None, None, len(sizes) * per
        else:
            plan = llava_ov.pack_plan(sizes, c.image_grid_pinpoints, c.v_image_size, c.v_side, c.anyres_max)
            tp = llava_ov.transpose_plan(plan)
            crops, lens, n_tok = plan["crops"], plan["lens"], len(plan["ptr"]) - 1
            pack, pack_t = (up(plan["ptr"]), up(plan["idx"]), up(plan["w"])), (up(tp["ptr"]), up(tp["idx"]), up(tp["w"]))
        nc = sum(crops)
        extra = {}
        if c.v_cls:
            # tower row (crop k, position j): j == 0 -> the class-embedding row (index nc * per of the source), j > 0 -> patch row k * per + j - 1
            j = np.tile(np.arange(seq), nc)
            k = np.repeat(np.arange(nc), seq)
            asm = {"ptr": np.arange(nc * seq + 1, dtype=np.int32), "idx": np.where(j == 0, nc * per, k * per + j - 1).astype(np.int32), "w": np.ones(nc * seq, np.float32), "n_src": nc * per + 1}
            sel = {"ptr": np.arange(nc * per + 1, dtype=np.int32), "idx": (np.repeat(np.arange(nc), per) * seq + 1 + np.tile(np.arange(per), nc)).astype(np.int32),
                   "w": np.ones(nc * per, np.float32), "n_src": nc * seq}
            tri = lambda m: (up(m["ptr"]), up(m["idx"]), up(m["w"]))
            extra = dict(assemble=tri(asm), assemble_t=tri(llava_ov.transpose_plan(asm)), select=tri(sel), select_t=tri(llava_ov.transpose_plan(sel)))
        return SiglipPlan(nc, nc * seq, ops.Segments.from_cu(np.arange(nc + 1) * seq, dev), up(np.tile(np.arange(seq, dtype=np.int64), nc)), pack, pack_t, n_tok, lens, crops, **extra)

    def _vision_plan_build(self, grids) -> VisionPlan:
        c = self.cfg
        if c.is_llava:
            return self._siglip_plan_build([(int(g[0]), int(g[1])) for g in grids])
        grids = [tuple(int(z) for z in g) for g in grids]
        m2 = c.v_merge**2
        if c.v_arch == "qwen2_vl":   # no window reorder, every block attends over the whole image (TF:qwen2_vl ::690-720)
            cu_full = indexing.vision_cu_seqlens(grids)
            pos = indexing.vision_position_ids(grids, c.v_merge).astype(np.float32)
            rot_t = ops.h2d((pos[:, :, None] * self.v_inv_freq[None, None, :]).reshape(pos.shape[0], -1), self.dev)
            seg = ops.Segments.from_cu(cu_full, self.dev)
            return VisionPlan(pos.shape[0], None, None, seg, seg, rot_t.cos().contiguous(), rot_t.sin().contiguous())
        win, cu_win = indexing.vision_window_index(grids, c.v_merge, c.v_window, c.v_patch)
        cu_full = indexing.vision_cu_seqlens(grids)
        pos = indexing.vision_position_ids(grids, c.v_merge).astype(np.float32)  # [N,2]
        n = pos.shape[0]
        rot = (pos[:, :, None] * self.v_inv_freq[None, None, :]).reshape(n, -1)     # [N, d/2]: h freqs then w freqs
        rot = rot.reshape(n // m2, m2, -1)[win].reshape(n, -1)
        rot_t = ops.h2d(rot, self.dev)
        win_t = ops.h2d(win, self.dev)
        rev = ops.h2d(np.argsort(win), self.dev)
        return VisionPlan(n, win_t, rev, ops.Segments.from_cu(cu_win, self.dev), ops.Segments.from_cu(cu_full, self.dev), rot_t.cos().contiguous(), rot_t.sin().contiguous())

    def _positions(self, input_ids, attention_mask, flat_grids):
        """Rotary positions [3, B, S] + per-row delta (position of the next token - number of real tokens).  Qwen-VL: the image-aware M-RoPE index
        (TF:944-1062).  LLaVA-OneVision / Qwen2: ordinary 1-D positions, the same on all three axes; counted over the real tokens, so left padding
        does not shift them (HF numbers the padded columns too -- rotary attention depends on position differences only)."""
        c = self.cfg
        if c.is_llava:
            m = (np.asarray(attention_mask) != 0).astype(np.int64)
            p1 = np.maximum(np.cumsum(m, 1) - 1, 0)
            return np.broadcast_to(p1[None], (3, *p1.shape)).copy(), np.zeros(m.shape[0], dtype=np.int64)
        return indexing.mrope_position_ids(input_ids, attention_mask, flat_grids, c.image_token_id, c.v_merge)

    def text_plan(self, input_ids: np.ndarray, attention_mask: np.ndarray, grids_per_row, img_row_offset) -> TextPlan:
        """input_ids / attention_mask: [B,S] numpy.  grids_per_row[b]: list of (t,h,w) of the images in row b.
        img_row_offset[b]: list, per image of row b, of the first row of that image in the image-embed matrix
        (several sequences may point at the same rows)."""
        c = self.cfg
        B, S = input_ids.shape
        flat_grids = [g for row in grids_per_row for g in row]
        pos, deltas = self._positions(input_ids, attention_mask, flat_grids)
        img_index = np.full((B, S), -1, dtype=np.int32)
        for b in range(B):
            cols = np.flatnonzero((input_ids[b] == c.image_token_id) & (attention_mask[b] != 0))
            k = 0
            for g, off in zip(grids_per_row[b], img_row_offset[b]):
                n = self.n_image_tokens(g)
                img_index[b, cols[k: k + n]] = off + np.arange(n)
                k += n
            if k != len(cols):
                raise ValueError(f"row {b}: {len(cols)} image tokens but grids supply {k}")
        lengths = attention_mask.sum(1).astype(np.int64)
        # A row is  [left padding 0..][real tokens 1..][0.. after the first EOS]  (REF:722-731 builds exactly this):
        # the real tokens form ONE attention segment; slots outside it are never attended to and never read.
        starts, ends = [], []
        for b in range(B):
            nz = np.flatnonzero(attention_mask[b])
            first, last = (int(nz[0]), int(nz[-1]) + 1) if len(nz) else (S, S)
            if last - first != len(nz):
                raise ValueError("attention_mask rows must be one contiguous run of ones (left padding / post-EOS padding only)")
            starts.append(b * S + first)
            ends.append(b * S + last)
        pos_t = ops.h2d(pos.reshape(3, B * S), self.dev)
        sel = pos_t[self.mrope_comp]                                   # [D/2, T]: component per frequency
        ang = sel.t().to(F32) * self.inv_freq[None, :]                  # [T, D/2] fp32, as TF::525-538
        return TextPlan(B, S, ops.h2d(input_ids.reshape(-1).astype(np.int64), self.dev), ops.h2d(img_index.reshape(-1), self.dev),
                        ops.Segments(starts, ends, self.dev), ang.cos().contiguous(), ang.sin().contiguous(), deltas, lengths,
                        host=(input_ids.reshape(-1).astype(np.int64), img_index.reshape(-1).astype(np.int64)))

    def text_plan_shared(self, ids_p: np.ndarray, mask_p: np.ndarray, comp: np.ndarray, cmask: np.ndarray, G: int, grids_per_prompt, img_off_per_prompt) -> TextPlan:
        """Shared-prefix layout of a GRPO micro-batch: the reference feeds [ng*G, P+C] rows whose first P positions are the
        same prompt G times (REF:train/stage_rl/trainer/sc_grpo_trainer.py:626,705-746).  A causal decoder gives those G
        copies identical hidden states, so here every prompt goes through the layers ONCE:
            flat tokens = [ng prompts x P] ++ [ng*G completions x C],
        and completion segments attend to their prompt's keys through `seg_prefix` (include/iadr1_hip.h).  Positions are the
        reference's (M-RoPE over the full [P+C] rows); logit rows for completion token j are `logit_rows()`."""
        c = self.cfg
        ng, P = ids_p.shape
        n, C = comp.shape
        assert n == ng * G
        # M-RoPE positions: the reference's get_rope_index over the full [P+C] rows gives the prompt part its image-aware positions
        # and every later text token `max prompt position + 1 + j` on all three axes (TF:1042-1176) -- so it is evaluated on the ng
        # unique prompts only and the completion part is arithmetic (same values; 0.8 instead of 6 ms of host time per step)
        flat_grids = [g for b in range(ng) for g in grids_per_prompt[b]]
        pos_p, deltas_p = self._positions(ids_p, mask_p, flat_grids)   # [3, ng, P], [ng]
        first = (mask_p.sum(1).astype(np.int64) + np.asarray(deltas_p).reshape(-1).astype(np.int64))              # position of completion token 0
        pos_c = np.repeat(first, G)[:, None] + np.arange(C, dtype=np.int64)[None, :]                               # [n, C]
        pos_flat = np.concatenate([pos_p.reshape(3, ng * P), np.broadcast_to(pos_c.reshape(1, n * C), (3, n * C))], 1)
        deltas = np.repeat(np.asarray(deltas_p).reshape(-1), G)
        mask_full = np.concatenate([np.repeat(mask_p, G, 0), cmask.astype(mask_p.dtype)], 1)
        T = ng * P + n * C
        img_index = np.full(T, -1, dtype=np.int32)
        starts, ends, prefix = [], [], []
        for b in range(ng):
            cols = np.flatnonzero((ids_p[b] == c.image_token_id) & (mask_p[b] != 0))
            k = 0
            for g, off in zip(grids_per_prompt[b], img_off_per_prompt[b]):
                cnt = self.n_image_tokens(g)
                img_index[b * P + cols[k: k + cnt]] = off + np.arange(cnt)
                k += cnt
            if k != len(cols):
                raise ValueError(f"prompt {b}: {len(cols)} image tokens but grids supply {k}")
            nz = np.flatnonzero(mask_p[b])
            if len(nz) == 0 or int(nz[-1]) + 1 - int(nz[0]) != len(nz):
                raise ValueError("prompt attention_mask rows must be one non-empty contiguous run of ones")
            starts.append(b * P + int(nz[0]))
            ends.append(b * P + int(nz[-1]) + 1)
            prefix.append([0, 0, ng + b * G, G])
        base = ng * P
        clen = cmask.astype(np.int64).sum(1)
        if not (np.all(clen >= 1) and np.all(cmask[np.arange(n), clen - 1] != 0) and np.all(cmask[:, 0] != 0)):
            raise ValueError("completion masks must be a non-empty run of ones starting at the first completion token")
        for r in range(n):
            b = r // G
            starts.append(base + r * C)
            ends.append(base + r * C + int(clen[r]))
            prefix.append([starts[b], ends[b] - starts[b], 0, 0])
        ids_flat = np.concatenate([ids_p.reshape(-1), comp.reshape(-1)]).astype(np.int64)
        pos_t = ops.h2d(pos_flat, self.dev)
        ang = pos_t[self.mrope_comp].t().to(F32) * self.inv_freq[None, :]
        plan = TextPlan(n, P + C, ops.h2d(ids_flat, self.dev), ops.h2d(img_index, self.dev),
                        ops.Segments(starts, ends, self.dev, prefix=prefix), ang.cos().contiguous(), ang.sin().contiguous(), deltas, mask_full.sum(1).astype(np.int64))
        plan.shared = (ng, P, n, C, G)
        plan.host = (ids_flat, img_index.astype(np.int64))
        # the completion rows alone (two-phase forward: the prompt rows were already run by the rollout's prefill); segment indices stay absolute
        T0 = ng * P
        plan.tail = TextPlan(n, P + C, plan.ids[T0:], plan.img_index[T0:], ops.Segments(starts[ng:], ends[ng:], self.dev, prefix=prefix[ng:]),
                             plan.cos[T0:], plan.sin[T0:], deltas, plan.lengths)
        return plan

    @staticmethod
    def shared_logit_rows(ng, P, n, C, G):
        """Flat row whose hidden state predicts completion token j of sequence r: j = 0 -> the prompt's last token (one row
        shared by the G sequences of the group), j > 0 -> completion row j-1.  Returns (rows [n*C] int64, first_sel [n] indices
        into rows of the j = 0 entries, first_dest [ng] their flat rows, first_group [n] group of each j = 0 entry)."""
        r = np.arange(n)[:, None]
        j = np.arange(C)[None, :]
        rows = np.where(j == 0, (r // G) * P + (P - 1), ng * P + r * C + j - 1).reshape(-1).astype(np.int64)
        return rows, (np.arange(n) * C).astype(np.int64), (np.arange(ng) * P + (P - 1)).astype(np.int64), (np.arange(n) // G).astype(np.int64)

    # ========================================================================================================
    # vision tower  (TF::408-471)
    # ========================================================================================================
    def vision_forward(self, pixel_values: torch.Tensor, plan: VisionPlan, save: bool):
        """pixel_values: [N, C*T*P*P] fp32 or bf16 on device -> merged image embeds [N/m2, H] bf16 (raster order)."""
        c, P = self.cfg, self.p
        if c.is_llava:
            return self._vision_forward_siglip(pixel_values, plan, save)
        vh, nh, d, m2 = c.v_hidden, c.v_heads, c.v_head_dim, c.v_merge**2
        N = plan.n_patches
        px = pixel_values if pixel_values.dtype == BF16 else ops.cast_f32_to_bf16(pixel_values)
        if c.v_arch == "qwen2_vl":
            return self._vision_forward_q2(px, plan, save)
        x = ops.gemm_nt(px, P.w("visual.patch_embed"))                                   # K1: conv3d(stride==kernel) == GEMM
        x = ops.embed_fwd(plan.win_index, None, x.view(N // m2, m2 * vh), None).view(N, vh)  # K2: window-order gather
        ctx = {"px": px, "layers": []} if save else None
        res, branch, bias = x, None, None
        for i in range(c.v_depth):
            b = f"visual.blocks.{i}."
            seg = plan.seg_full if i in c.v_fullatt else plan.seg_window
            x_in = torch.empty_like(res) if (save and branch is not None) else res
            if branch is None:
                h1, rstd1 = ops.rmsnorm_fwd(res, P.w(b + "norm1"), 1e-6, want_rstd=save)
            else:
                h1, rstd1 = ops.rmsnorm_fwd(branch, P.w(b + "norm1"), 1e-6, res=res, res_out=x_in, want_rstd=save)
            qkv = ops.gemm_nt(h1, P.w(b + "qkv.w"), bias=P.w(b + "qkv.b"))
            ops.rope_(qkv, plan.cos, plan.sin, 2 * nh, d)
            o, lse = ops.attn_fwd(qkv[:, :vh], qkv[:, vh: 2 * vh], qkv[:, 2 * vh:], seg, nh, nh, d, False, d**-0.5, want_lse=save)
            ab = ops.gemm_nt(o, P.w(b + "proj.w"), bias=P.w(b + "proj.b"))
            x_mid = torch.empty_like(x_in) if save else x_in
            h2, rstd2 = ops.rmsnorm_fwd(ab, P.w(b + "norm2"), 1e-6, res=x_in, res_out=x_mid, want_rstd=save)
            gu = ops.gemm_nt(h2, P.w(b + "gu.w"), bias=P.w(b + "gu.b"))
            a = ops.swiglu_fwd(gu)
            branch = ops.gemm_nt(a, P.w(b + "down.w"), bias=P.w(b + "down.b"))
            res = x_mid
            if save:
                ctx["layers"].append((x_in, rstd1, h1, qkv, o, lse, x_mid, rstd2, h2, gu, a, seg))
        x_last = torch.empty_like(res) if save else res
        hq, rstdq = ops.rmsnorm_fwd(branch, P.w("visual.merger.ln_q"), 1e-6, res=res, res_out=x_last, want_rstd=save)
        hq4 = hq.view(N // m2, m2 * vh)
        z = ops.gemm_nt(hq4, P.w("visual.merger.fc1.w"), bias=P.w("visual.merger.fc1.b"))
        ga = ops.gelu_fwd(z)
        mo = ops.gemm_nt(ga, P.w("visual.merger.fc2.w"), bias=P.w("visual.merger.fc2.b"))
        out = ops.embed_fwd(plan.rev_index, None, mo, None)                               # K2^-1: back to raster order
        if save:
            ctx.update(x_last=x_last, rstdq=rstdq, hq4=hq4, z=z, ga=ga, plan=plan)
        return out, ctx

    # ========================================================================================================
    # LLaVA-OneVision vision path: SigLIP tower -> projector -> any-resolution packing
    # (transformers/models/siglip/modeling_siglip.py:116-181,310-356; models/llava_onevision/modeling_llava_onevision.py:131-150,280-348,399-418)
    # ========================================================================================================
    def _siglip_patches(self, crops: torch.Tensor) -> torch.Tensor:
        """Crops [n, 3, S, S] (fp32 or bf16, the processor's layout) -> patch rows [n * tokens, patch_dim_pad] bf16 in the order (c, dy, dx) the
        conv weight is flattened in: the stride == kernel Conv2d of SG:124-130 is then a GEMM.  Pure data movement."""
        c = self.cfg
        n, ch, S, _ = crops.shape
        p, side = c.v_patch, c.v_side
        if S != side * p:      # valid-padding conv: the last S - side * p pixel rows / columns are never read (384 = 27 * 14 + 6 in the SigLIP-so400m tower)
            crops = crops[:, :, :side * p, :side * p]
        rows = crops.reshape(n, ch, side, p, side, p).permute(0, 2, 4, 1, 3, 5).reshape(n * side * side, ch * p * p)
        if rows.dtype == BF16 and c.patch_dim_pad == c.patch_dim:
            return rows.contiguous()
        return ops.cast_f32_to_bf16(rows.to(F32).contiguous(), cpad=c.patch_dim_pad)

    def _vision_forward_siglip(self, crops, plan: SiglipPlan, save: bool):
        """crops: [n_crops, 3, S, S] on device (all images of the batch, base image first per image) -> image tokens [n_tokens, H] bf16 in the order the
        placeholders consume them.  SigLIP tower (LLaVA-OneVision) or CLIP tower (LLaVA-1.5 / NeXT: class token, pre-LayerNorm, QuickGELU, the blocks up to the
        feature layer, class token dropped in front of the projector; transformers models/clip/modeling_clip.py:141-200,  models/llava/modeling_llava.py:144-189)."""
        c, P = self.cfg, self.p
        vh, nh, dp, H = c.v_hidden, c.v_heads, c.v_head_pad, c.hidden_size
        N, seg, eps = plan.n_patches, plan.seg, c.v_ln_eps
        clip = c.v_arch == "clip"
        scale = (vh // nh) ** -0.5          # the TRUE head width (72 / 64), not the padded one
        px = self._siglip_patches(crops)
        n_pe = plan.n_crops * c.v_tokens
        assert px.shape[0] == n_pe, (px.shape, n_pe)
        if clip:
            srcA = torch.empty(n_pe + 1, vh, dtype=BF16, device=self.dev)
            ops.gemm_nt(px, P.w("visual.patch_embed"), out=srcA[:n_pe])
            srcA[n_pe].copy_(P.w("visual.cls"))
            res = ops.rows_gather_sum(srcA, plan.assemble[0], plan.assemble[1], N, weights=plan.assemble[2])       # [class ; patches] per crop
        else:
            res = ops.gemm_nt(px, P.w("visual.patch_embed"), bias=P.w("visual.patch_embed.b"))
        branch = ops.embed_fwd(plan.pos_ids, None, P.w("visual.pos"), None)                    # learned positions, one row per tower row
        ctx = {"px": px, "layers": []} if save else None
        if clip:       # pre_layrnorm on (embeddings + positions): its output is the residual stream the blocks start from
            x_pre = torch.empty_like(res) if save else res
            h0, mu0, rs0 = ops.layernorm_fwd(branch, P.w("visual.pre_ln"), P.w("visual.pre_ln.b"), eps, res=res, res_out=x_pre, want_stats=save)
            if save:
                ctx["pre"] = (x_pre, mu0, rs0)
            res, branch = h0, None
        w3 = nh * dp
        act = ops.quick_gelu_fwd if clip else ops.gelu_tanh_fwd
        for i in range(c.v_run_depth):
            b = f"visual.blocks.{i}."
            if branch is None:
                x_in = res
                h1, mu1, rs1 = ops.layernorm_fwd(res, P.w(b + "norm1"), P.w(b + "norm1.b"), eps, want_stats=save)
            else:
                x_in = torch.empty_like(res) if save else res
                h1, mu1, rs1 = ops.layernorm_fwd(branch, P.w(b + "norm1"), P.w(b + "norm1.b"), eps, res=res, res_out=x_in, want_stats=save)
            qkv = ops.gemm_nt(h1, P.w(b + "qkv.w"), bias=P.w(b + "qkv.b"))
            o, lse = ops.attn_fwd(qkv[:, :w3], qkv[:, w3: 2 * w3], qkv[:, 2 * w3:], seg, nh, nh, dp, False, scale, want_lse=save)
            ab = ops.gemm_nt(o, P.w(b + "proj.w"), bias=P.w(b + "proj.b"))
            x_mid = torch.empty_like(x_in) if save else x_in
            h2, mu2, rs2 = ops.layernorm_fwd(ab, P.w(b + "norm2"), P.w(b + "norm2.b"), eps, res=x_in, res_out=x_mid, want_stats=save)
            z = ops.gemm_nt(h2, P.w(b + "fc1.w"), bias=P.w(b + "fc1.b"))
            a = act(z)
            branch = ops.gemm_nt(a, P.w(b + "fc2.w"), bias=P.w(b + "fc2.b"))
            res = x_mid
            if save:
                ctx["layers"].append((x_in, mu1, rs1, h1, qkv, o, lse, x_mid, mu2, rs2, h2, z, a))
        # hidden state of the feature layer (before any post_layernorm): residual + MLP branch
        feat = torch.empty_like(res)
        ops.hip.call("rmsnorm_fwd", branch, None, 0, None, res, feat, P.w("visual.blocks.0.norm1"), None, None, N, vh, vh, vh, vh, 1e-6, None)
        featp = ops.rows_gather_sum(feat, plan.select[0], plan.select[1], n_pe, weights=plan.select[2]) if clip else feat      # "default" strategy: without the class rows
        z = ops.gemm_nt(featp, P.w("visual.merger.fc1.w"), bias=P.w("visual.merger.fc1.b"))
        ga = ops.gelu_fwd(z)                                                                  # exact GELU (projector_hidden_act "gelu")
        if plan.pack is None:             # LLaVA-1.5: the projector rows are the image tokens
            out = ops.gemm_nt(ga, P.w("visual.merger.fc2.w"), bias=P.w("visual.merger.fc2.b"))
        else:
            src = torch.empty(n_pe + 1, H, dtype=BF16, device=self.dev)
            ops.gemm_nt(ga, P.w("visual.merger.fc2.w"), bias=P.w("visual.merger.fc2.b"), out=src[:n_pe])
            src[n_pe].copy_(P.w("visual.newline"))
            out = ops.rows_gather_sum(src, plan.pack[0], plan.pack[1], plan.n_tokens, weights=plan.pack[2])
        if save:
            ctx.update(feat=featp, z=z, ga=ga, plan=plan)
        return out, ctx

    def _vision_backward_siglip(self, d_out, ctx, only_newline=False):
        c, P = self.cfg, self.p
        vh, nh, dp, H = c.v_hidden, c.v_heads, c.v_head_pad, c.hidden_size
        plan: SiglipPlan = ctx["plan"]
        N, seg = plan.n_patches, plan.seg
        clip = c.v_arch == "clip"
        n_pe = plan.n_crops * c.v_tokens
        scale = (vh // nh) ** -0.5
        w3 = nh * dp
        if plan.pack is None:
            dmo = d_out.contiguous()
        else:
            dsrc = ops.rows_gather_sum(d_out.contiguous(), plan.pack_t[0], plan.pack_t[1], n_pe + 1, weights=plan.pack_t[2])
            ops.colsum_acc(dsrc[n_pe: n_pe + 1], P.g("visual.newline"))
            dmo = dsrc[:n_pe]
        if only_newline:
            return
        ops.colsum_acc(dmo, P.g("visual.merger.fc2.b"))
        dga = ops.gemm_nt(dmo, P.wT("visual.merger.fc2.w"))
        self._wgrad("visual.merger.fc2.w", dmo, ctx["ga"])
        dz = ops.gelu_bwd(dga, ctx["z"])
        ops.colsum_acc(dz, P.g("visual.merger.fc1.b"))
        dres = ops.gemm_nt(dz, P.wT("visual.merger.fc1.w"))        # gradient of the feature layer's hidden state = of the residual stream AND of the last MLP branch
        self._wgrad("visual.merger.fc1.w", dz, ctx["feat"])
        if clip:       # back to the tower's rows: the class rows carry no gradient from the projector
            dres = ops.rows_gather_sum(dres, plan.select_t[0], plan.select_t[1], N, weights=plan.select_t[2])
        act_bwd = ops.quick_gelu_bwd if clip else ops.gelu_tanh_bwd
        for i in reversed(range(c.v_run_depth)):
            b = f"visual.blocks.{i}."
            x_in, mu1, rs1, h1, qkv, o, lse, x_mid, mu2, rs2, h2, z, a = ctx["layers"][i]
            ops.colsum_acc(dres, P.g(b + "fc2.b"))
            da = ops.gemm_nt(dres, P.wT(b + "fc2.w"))
            self._wgrad(b + "fc2.w", dres, a)
            dz = act_bwd(da, z)
            ops.colsum_acc(dz, P.g(b + "fc1.b"))
            dh2 = ops.gemm_nt(dz, P.wT(b + "fc1.w"))
            self._wgrad(b + "fc1.w", dz, h2)
            dx_mid = ops.layernorm_bwd(dh2, x_mid, P.w(b + "norm2"), mu2, rs2, dres=dres, dw=P.g(b + "norm2"), db=P.g(b + "norm2.b"))
            ops.colsum_acc(dx_mid, P.g(b + "proj.b"))
            do = ops.gemm_nt(dx_mid, P.wT(b + "proj.w"))
            self._wgrad(b + "proj.w", dx_mid, o)
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :w3], qkv[:, w3: 2 * w3], qkv[:, 2 * w3:], o, do, lse, seg, nh, nh, dp, False, scale, dqkv[:, :w3], dqkv[:, w3: 2 * w3], dqkv[:, 2 * w3:])
            ops.colsum_acc(dqkv, P.g(b + "qkv.b"))
            dh1 = ops.gemm_nt(dqkv, P.wT(b + "qkv.w"))
            self._wgrad(b + "qkv.w", dqkv, h1)
            dres = ops.layernorm_bwd(dh1, x_in, P.w(b + "norm1"), mu1, rs1, dres=dx_mid, dw=P.g(b + "norm1"), db=P.g(b + "norm1.b"))
        if clip:
            x_pre, mu0, rs0 = ctx["pre"]
            dres = ops.layernorm_bwd(dres, x_pre, P.w("visual.pre_ln"), mu0, rs0, dw=P.g("visual.pre_ln"), db=P.g("visual.pre_ln.b"))
        # dres = gradient of (patch / class embedding + position embedding)
        ops.rows_scatter_acc(dres, plan.pos_scatter(self.dev, c.v_seq), P.g("visual.pos"))
        if clip:
            dsrcA = ops.rows_gather_sum(dres, plan.assemble_t[0], plan.assemble_t[1], n_pe + 1, weights=plan.assemble_t[2])
            ops.colsum_acc(dsrcA[n_pe: n_pe + 1], P.g("visual.cls"))
            self._wgrad("visual.patch_embed", dsrcA[:n_pe], ctx["px"])
        else:
            ops.colsum_acc(dres, P.g("visual.patch_embed.b"))
            self._wgrad("visual.patch_embed", dres, ctx["px"])
        self.join_wgrads()

    def _vision_forward_q2(self, px, plan: VisionPlan, save: bool):
        """Qwen2-VL tower (TF:models/qwen2_vl/modeling_qwen2_vl.py:418-447 block, ::293-301 MLP, ::270-290 merger)."""
        c, P = self.cfg, self.p
        vh, nh, d, m2 = c.v_hidden, c.v_heads, c.v_head_dim, c.v_merge**2
        N, seg = plan.n_patches, plan.seg_full
        res = ops.gemm_nt(px, P.w("visual.patch_embed"))
        ctx = {"px": px, "layers": []} if save else None
        branch = None
        for i in range(c.v_depth):
            b = f"visual.blocks.{i}."
            x_in = torch.empty_like(res) if (save and branch is not None) else res
            h1, mu1, rs1 = ops.layernorm_fwd(res if branch is None else branch, P.w(b + "norm1"), P.w(b + "norm1.b"), 1e-6,
                                             res=None if branch is None else res, res_out=None if branch is None else x_in, want_stats=save)
            qkv = ops.gemm_nt(h1, P.w(b + "qkv.w"), bias=P.w(b + "qkv.b"))
            ops.rope_(qkv, plan.cos, plan.sin, 2 * nh, d)
            o, lse = ops.attn_fwd(qkv[:, :vh], qkv[:, vh: 2 * vh], qkv[:, 2 * vh:], seg, nh, nh, d, False, d**-0.5, want_lse=save)
            ab = ops.gemm_nt(o, P.w(b + "proj.w"), bias=P.w(b + "proj.b"))
            x_mid = torch.empty_like(x_in) if save else x_in
            h2, mu2, rs2 = ops.layernorm_fwd(ab, P.w(b + "norm2"), P.w(b + "norm2.b"), 1e-6, res=x_in, res_out=x_mid, want_stats=save)
            z = ops.gemm_nt(h2, P.w(b + "fc1.w"), bias=P.w(b + "fc1.b"))
            a = ops.quick_gelu_fwd(z)
            branch = ops.gemm_nt(a, P.w(b + "fc2.w"), bias=P.w(b + "fc2.b"))
            res = x_mid
            if save:
                ctx["layers"].append((x_in, mu1, rs1, h1, qkv, o, lse, x_mid, mu2, rs2, h2, z, a))
        x_last = torch.empty_like(res) if save else res
        hq, muq, rsq = ops.layernorm_fwd(branch, P.w("visual.merger.ln_q"), P.w("visual.merger.ln_q.b"), 1e-6, res=res, res_out=x_last, want_stats=save)
        hq4 = hq.view(N // m2, m2 * vh)
        z = ops.gemm_nt(hq4, P.w("visual.merger.fc1.w"), bias=P.w("visual.merger.fc1.b"))
        ga = ops.gelu_fwd(z)
        out = ops.gemm_nt(ga, P.w("visual.merger.fc2.w"), bias=P.w("visual.merger.fc2.b"))
        if save:
            ctx.update(x_last=x_last, muq=muq, rstdq=rsq, hq4=hq4, z=z, ga=ga, plan=plan)
        return out, ctx

    def _vision_backward_q2(self, d_out, ctx):
        c, P = self.cfg, self.p
        vh, nh, d, m2 = c.v_hidden, c.v_heads, c.v_head_dim, c.v_merge**2
        plan: VisionPlan = ctx["plan"]
        N, seg = plan.n_patches, plan.seg_full
        ops.colsum_acc(d_out, P.g("visual.merger.fc2.b"))
        dga = ops.gemm_nt(d_out, P.wT("visual.merger.fc2.w"))
        self._wgrad("visual.merger.fc2.w", d_out, ctx["ga"])
        dz = ops.gelu_bwd(dga, ctx["z"])
        ops.colsum_acc(dz, P.g("visual.merger.fc1.b"))
        dhq4 = ops.gemm_nt(dz, P.wT("visual.merger.fc1.w"))
        self._wgrad("visual.merger.fc1.w", dz, ctx["hq4"])
        dres = ops.layernorm_bwd(dhq4.view(N, vh), ctx["x_last"], P.w("visual.merger.ln_q"), ctx["muq"], ctx["rstdq"],
                                 dw=P.g("visual.merger.ln_q"), db=P.g("visual.merger.ln_q.b"))
        for i in reversed(range(c.v_depth)):
            b = f"visual.blocks.{i}."
            x_in, mu1, rs1, h1, qkv, o, lse, x_mid, mu2, rs2, h2, z, a = ctx["layers"][i]
            ops.colsum_acc(dres, P.g(b + "fc2.b"))
            da = ops.gemm_nt(dres, P.wT(b + "fc2.w"))
            self._wgrad(b + "fc2.w", dres, a)
            dz = ops.quick_gelu_bwd(da, z)
            ops.colsum_acc(dz, P.g(b + "fc1.b"))
            dh2 = ops.gemm_nt(dz, P.wT(b + "fc1.w"))
            self._wgrad(b + "fc1.w", dz, h2)
            dx_mid = ops.layernorm_bwd(dh2, x_mid, P.w(b + "norm2"), mu2, rs2, dres=dres, dw=P.g(b + "norm2"), db=P.g(b + "norm2.b"))
            ops.colsum_acc(dx_mid, P.g(b + "proj.b"))
            do = ops.gemm_nt(dx_mid, P.wT(b + "proj.w"))
            self._wgrad(b + "proj.w", dx_mid, o)
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :vh], qkv[:, vh: 2 * vh], qkv[:, 2 * vh:], o, do, lse, seg, nh, nh, d, False, d**-0.5,
                         dqkv[:, :vh], dqkv[:, vh: 2 * vh], dqkv[:, 2 * vh:])
            ops.rope_(dqkv, plan.cos, plan.sin, 2 * nh, d, backward=True)
            ops.colsum_acc(dqkv, P.g(b + "qkv.b"))
            dh1 = ops.gemm_nt(dqkv, P.wT(b + "qkv.w"))
            self._wgrad(b + "qkv.w", dqkv, h1)
            dres = ops.layernorm_bwd(dh1, x_in, P.w(b + "norm1"), mu1, rs1, dres=dx_mid, dw=P.g(b + "norm1"), db=P.g(b + "norm1.b"))
        self._wgrad("visual.patch_embed", dres, ctx["px"])
        self.join_wgrads()

    def _bw_scratch(self, T):
        """Static scratch of the decoder backward pass: every temporary of a layer lives in buffers allocated once (grow-only in T) -- main-stream-only
        tensors in one buffer each, tensors a weight-gradient GEMM reads on the side stream in small RINGS whose slots are guarded by events.  Before this,
        those tensors came from the caching allocator with `record_stream` marks: a block freed by the main stream could not be reused until the side
        stream had passed it, the host runs a whole backward ahead, so every layer asked for fresh 430 / 860 MiB blocks, and once the card was full the
        allocator answered with hipFree / hipMalloc pairs inside the step (BENCH_r02: 84 device allocations + 153 frees in the timed region)."""
        s = self.__dict__.get("_bw")
        if s is not None and s["cap"] >= T:
            return s
        self.join_wgrads()
        self.__dict__["_bw"] = None
        c = self.cfg
        H, I, qw = c.hidden_size, c.intermediate_size, c.qkv_width
        mk = lambda w: torch.empty(T, w, dtype=BF16, device=self.dev)
        s = {"cap": T, "da": mk(I), "dh": mk(H), "do": mk(c.num_attention_heads * c.head_dim),
             "res": _Ring([mk(H) for _ in range(6)]), "dgu": _Ring([mk(2 * I) for _ in range(3)]), "dqkv": _Ring([mk(qw) for _ in range(3)])}
        self.__dict__["_bw"] = s
        return s

    def _wgrad(self, name, dy, x, slot=None):
        """grad[name] ([N,K] fp32) += dy[T,N]^T . x[T,K]   (ops.gemm_tn_acc: the TN form of the 256 x 256 kernel straight from dy / x; transposed copies + the NT
        kernel only for small / ragged shapes).
        slot: dy lives in a `_Ring` slot (static scratch): the slot is marked busy until the side stream has read it, instead of a record_stream mark.
        Weight gradients are leaves of the backward graph: by default they (transposes + GEMM) are issued on a side stream, so the dgrad chain
        on the main stream never waits for them and the two queues fill each other's tile-quantisation tails and launch bubbles (a 640-tile
        GEMM leaves half the CUs idle in its third round): -35 ms per step.  Measured alternative: only the transposes on the side stream --
        no gain, the win is GEMM/GEMM overlap.  IADR1_WGRAD_STREAM=0 issues everything on one stream (exclusive per-launch timings)."""
        ws = self.wgrad_stream
        if ws is None:
            ops.gemm_tn_acc(dy, x, self.p.g(name), wide=self.wgrad_tn_wide)
            return
        ws.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ws):
            ops.gemm_tn_acc(dy, x, self.p.g(name), wide=self.wgrad_tn_wide)
        if slot is not None:
            slot[0].busy(slot[1], ws)
        else:
            dy.record_stream(ws)     # the caching allocator must not hand these blocks to the main stream before the side stream is done with them
        x.record_stream(ws)

    def join_wgrads(self):
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    def vision_backward(self, d_out: torch.Tensor, ctx, only_newline: bool = False):
        """d_out: [N/m2, H] bf16 gradient of the merged image embeds (raster order).
        only_newline (PA-SFT of the any-resolution LLaVA families with a frozen tower and projector, sft.frozen_parameter_rule): the one trainable tensor of the
        vision side is the row the packing inserts; its gradient needs ctx["plan"] alone."""
        c, P = self.cfg, self.p
        if c.is_llava:
            return self._vision_backward_siglip(d_out, ctx, only_newline)
        assert not only_newline
        vh, nh, d, m2 = c.v_hidden, c.v_heads, c.v_head_dim, c.v_merge**2
        plan: VisionPlan = ctx["plan"]
        N = plan.n_patches
        if c.v_arch == "qwen2_vl":
            return self._vision_backward_q2(d_out, ctx)
        dmo = ops.embed_fwd(plan.win_index, None, d_out, None)                          # inverse of the reverse gather
        ops.colsum_acc(dmo, P.g("visual.merger.fc2.b"))
        dga = ops.gemm_nt(dmo, P.wT("visual.merger.fc2.w"))
        self._wgrad("visual.merger.fc2.w", dmo, ctx["ga"])
        dz = ops.gelu_bwd(dga, ctx["z"])
        ops.colsum_acc(dz, P.g("visual.merger.fc1.b"))
        dhq4 = ops.gemm_nt(dz, P.wT("visual.merger.fc1.w"))
        self._wgrad("visual.merger.fc1.w", dz, ctx["hq4"])
        dres = ops.rmsnorm_bwd(dhq4.view(N, vh), ctx["x_last"], P.w("visual.merger.ln_q"), ctx["rstdq"], dw=P.g("visual.merger.ln_q"))
        for i in reversed(range(c.v_depth)):
            b = f"visual.blocks.{i}."
            x_in, rstd1, h1, qkv, o, lse, x_mid, rstd2, h2, gu, a, seg = ctx["layers"][i]
            ops.colsum_acc(dres, P.g(b + "down.b"))
            da = ops.gemm_nt(dres, P.wT(b + "down.w"))
            self._wgrad(b + "down.w", dres, a)
            dgu = ops.swiglu_bwd(da, gu)
            ops.colsum_acc(dgu, P.g(b + "gu.b"))
            dh2 = ops.gemm_nt(dgu, P.wT(b + "gu.w"))
            self._wgrad(b + "gu.w", dgu, h2)
            dx_mid = ops.rmsnorm_bwd(dh2, x_mid, P.w(b + "norm2"), rstd2, dres=dres, dw=P.g(b + "norm2"))
            ops.colsum_acc(dx_mid, P.g(b + "proj.b"))
            do = ops.gemm_nt(dx_mid, P.wT(b + "proj.w"))
            self._wgrad(b + "proj.w", dx_mid, o)
            dqkv = torch.empty_like(qkv)
            ops.attn_bwd(qkv[:, :vh], qkv[:, vh: 2 * vh], qkv[:, 2 * vh:], o, do, lse, seg, nh, nh, d, False, d**-0.5,
                         dqkv[:, :vh], dqkv[:, vh: 2 * vh], dqkv[:, 2 * vh:])
            ops.rope_(dqkv, plan.cos, plan.sin, 2 * nh, d, backward=True)
            ops.colsum_acc(dqkv, P.g(b + "qkv.b"))
            dh1 = ops.gemm_nt(dqkv, P.wT(b + "qkv.w"))
            self._wgrad(b + "qkv.w", dqkv, h1)
            dres = ops.rmsnorm_bwd(dh1, x_in, P.w(b + "norm1"), rstd1, dres=dx_mid, dw=P.g(b + "norm1"))
        dx = ops.embed_fwd(plan.rev_index, None, dres.view(N // m2, m2 * vh), None).view(N, vh)  # undo the window gather
        self._wgrad("visual.patch_embed", dx, ctx["px"])
        self.join_wgrads()

    # ========================================================================================================
    # text decoder (TF::790-873; layer ::708-757)
    # ========================================================================================================
    def _text_buffers(self, T: int, save: bool):
        """Static activation storage (no allocator traffic in the step; GiB-sized tensors were being hipFree'd and
        re-mapped by the caching allocator every layer).  save=True: one slab per decoder layer, alive until the
        backward of this micro-batch; save=False: one scratch slab reused by every layer."""
        c = self.cfg
        H, I = c.hidden_size, c.intermediate_size
        L = c.num_hidden_layers if save else 1
        key = "act_save" if save else "act_scratch"
        cur = self._ws.get(key)
        if cur is None or cur["T"] < T:
            if cur is not None:
                self._ws[key] = cur = None  # release before growing
            mk = lambda w: torch.empty(L, T, w, dtype=BF16, device=self.dev)
            cur = {"T": T, "x_in": mk(H), "h1": mk(H), "qkv": mk(c.qkv_width), "o": mk(c.num_attention_heads * c.head_dim), "x_mid": mk(H),
                   "h2": mk(H), "gu": mk(2 * I), "a": mk(I), "rstd1": torch.empty(L, T, dtype=F32, device=self.dev),
                   "rstd2": torch.empty(L, T, dtype=F32, device=self.dev), "lse": torch.empty(L, c.num_attention_heads, T, dtype=F32, device=self.dev)}
            self._ws[key] = cur
        return cur

    def saved_activation_bytes(self, T: int) -> int:
        """Bytes of the per-layer activation arena text_forward(save=True) keeps for T token rows (what gradient checkpointing avoids)."""
        c = self.cfg
        per_tok = (5 * c.hidden_size + c.qkv_width + c.num_attention_heads * c.head_dim + 3 * c.intermediate_size) * 2 + (2 + c.num_attention_heads) * 4
        return c.num_hidden_layers * T * per_tok

    def recompute_wanted(self, T: int, mode: str) -> bool:
        """`--gradient_checkpointing` (every reference launch script passes it, REF scripts/train/SC_GRPO/*.sh:56) as a POLICY on a 288 GB part: "off" never
        recomputes; "on" always; "auto" (what the flag selects) recomputes only when the saved activations of the micro-batch would not fit comfortably.
        The decision is STATIC (it used to read the allocator's free memory at call time, so kernels and numerics could differ between ranks, steps
        and runs): the arena of the token-row bucket (T rounded up to 2048 rows) against 60 % of
            total HBM - resident parameter / optimizer bytes of this store - `resident_extra` (the frozen reference, the gradient-exchange staging ring:
            set by the owner of the engine) - a fixed reserve (12 GB: KV pool, workspaces, allocator slack; + 40 GB when a process group exists: RCCL's
            channel buffers and the headroom the first multi-rank run should not have to discover),
        cached per (bucket, mode) and logged once.  3B at 20 480 rows: 71 GB against 0.6 x 186 -> kept, the rollout goes on doubling as the policy forward;
        7B at 20 480 rows: 91 GB against 0.6 x 96 -> one decoder layer is recomputed at a time in backward; 7B under DDP at 10 240 rows: 45 GB against
        0.6 x 56 -> recomputed (<= 235 GB reserved instead of 255).  Every rank computes the same answer from the same sizes."""
        if mode == "on" or os.environ.get("IADR1_RECOMPUTE") == "1":
            return True
        if mode != "auto" or os.environ.get("IADR1_RECOMPUTE") == "0":
            return False
        if self.__dict__.get("_recompute_forced"):      # check_ddp_headroom(): an earlier step of this run peaked too close to the card's capacity
            self._ws.pop("act_save", None)
            return True
        bucket = (int(T) + 2047) // 2048 * 2048
        cache = self.__dict__.setdefault("_recompute_decisions", {})
        if bucket not in cache:
            total = torch.cuda.get_device_properties(self.dev).total_memory
            ddp = False
            try:
                import torch.distributed as dist
                ddp = dist.is_available() and dist.is_initialized()
            except Exception:
                pass
            reserve = (12 << 30) + ((40 << 30) if ddp else 0)
            budget = total - self.p.resident_bytes() - int(self.__dict__.get("resident_extra", 0)) - reserve
            need = self.saved_activation_bytes(bucket)
            cache[bucket] = need > 0.6 * budget
            if os.environ.get("IADR1_QUIET") != "1":
                print(f"[iadr1] gradient checkpointing (auto): {bucket} token rows need {need / 2**30:.1f} GiB of saved activations, budget 0.6 x {budget / 2**30:.1f} GiB "
                      f"-> {'recompute one decoder layer at a time' if cache[bucket] else 'keep'}", file=sys.stderr, flush=True)
        if cache[bucket]:
            # monotone: once one bucket needs recomputation the run keeps recomputing -- micro-batches whose token counts straddle the threshold
            # (ragged any-resolution prompts) would otherwise free and re-allocate the multi-GB arena on alternating steps, with both buffers resident at the peak
            self._recompute_forced = True
            self._ws.pop("act_save", None)          # a stale arena of an earlier (smaller) bucket must not sit next to the checkpoint buffers
        return cache[bucket]

    def check_ddp_headroom(self, mode: str, limit_bytes: int = 235 << 30, group=None):
        """Data-parallel runs only, policy "auto", called once per optimizer step by the trainers: the static budget above cannot see everything a model family
        keeps (LLaVA-OneVision's 40 saved SigLIP crops and its 4 000-token KV pool: 261 GB reserved although the decoder arena alone fits).  If a COMPLETED step
        peaked above `limit_bytes` of reserved memory (235 GB: what VERDICT r3 #5 asks to stay under so that RCCL's channel buffers have room), every later step
        recomputes.  Deterministic for given shapes (the allocator's peak of a whole step, not its free memory at some call), logged once.  Returns True when it
        switched: the caller then releases what still references the arena and calls torch.cuda.empty_cache() (RCCL allocates outside torch's caching allocator).

        The all-reduce below is reached by EVERY rank or by none: only the mode, the environment, the process-group state and the call count -- all
        identical across ranks -- may return before it.  A rank whose own token-row bucket already forced recomputation (recompute_wanted: ragged prompts put ranks in
        different buckets) contributes that as a vote, so after the first check all ranks recompute or none does.  `group`: the engine's process group."""
        if mode != "auto" or os.environ.get("IADR1_RECOMPUTE") in ("0", "1"):
            return False
        n_checked = self.__dict__.get("_headroom_checks", 0)
        if n_checked >= 4:                      # the peak of a step is a property of the shapes: settled after the first few steps, no more synchronising reads
            return False
        try:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                return False
        except Exception:
            return False
        self._headroom_checks = n_checked + 1
        was_forced = bool(self.__dict__.get("_recompute_forced"))
        peak = torch.cuda.max_memory_reserved(self.dev) if self.dev.type == "cuda" else 0
        if self.dev.type == "cuda":
            torch.cuda.reset_peak_memory_stats(self.dev)        #  the peak of THIS step, not of everything since start-up (a checkpoint export, a graph capture)
        # every rank takes the same decision: one rank recomputing next to seven that do not would run different kernels (and a different number of them) per step
        over = torch.tensor([1 if (peak > limit_bytes or was_forced) else 0], dtype=torch.int32, device=self.dev)
        dist.all_reduce(over, op=dist.ReduceOp.MAX, group=group)
        if int(over.item()) and not was_forced:
            self._recompute_forced = True
            self._ws.pop("act_save", None)
            if os.environ.get("IADR1_QUIET") != "1":
                print(f"[iadr1] gradient checkpointing (auto): a step peaked at {peak / 2**30:.1f} GiB reserved on this rank (limit {limit_bytes / 2**30:.0f}; the decision is shared: "
                      "some rank peaked above it or had already switched) under a process group -> decoder activations are recomputed from the next step on", file=sys.stderr, flush=True)
            return True           # the CALLER drops what still points into the arena (rollout trace / captured graph) and then empties the cache
        return False

    def _recompute_layer(self, i, ctx):
        """Gradient checkpointing: the activations of decoder layer i, rebuilt from its checkpointed input rows with the forward's own kernels (bit-identical to
        what save=True would have kept), into one of two single-layer slabs -- two, because the weight-gradient GEMMs of layer i still read this slab on the
        side stream while the main stream rebuilds layer i - 1; the slab is handed out again only after those have finished (event)."""
        c, P = self.cfg, self.p
        H, D, Hq, Hkv = c.hidden_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        qw, kw = Hq * D, Hkv * D
        plan = ctx["plan"]
        x_in = ctx["recompute"]["x_in"][i]
        T = x_in.shape[0]
        rc = self.__dict__.get("_rc")
        if rc is None or rc["T"] < T:
            self.join_wgrads()
            self.__dict__["_rc"] = None
            I = c.intermediate_size
            mk = lambda w: torch.empty(T, w, dtype=BF16, device=self.dev)
            slab = lambda: {"h1": mk(H), "qkv": mk(c.qkv_width), "o": mk(Hq * D), "x_mid": mk(H), "h2": mk(H), "gu": mk(2 * I), "a": mk(I),
                            "rstd1": torch.empty(T, dtype=F32, device=self.dev), "rstd2": torch.empty(T, dtype=F32, device=self.dev),
                            "lse": torch.empty(Hq, T, dtype=F32, device=self.dev)}
            rc = self.__dict__["_rc"] = {"T": T, "ring": _Ring([slab(), slab()])}
        ring = rc["ring"]
        k = ring.k
        ring.k = (k + 1) % 2
        if ring.ev[k] is not None:
            torch.cuda.current_stream().wait_event(ring.ev[k])
            ring.ev[k] = None
        S = ring.bufs[k]
        b = f"layers.{i}."
        eps = float(c.rms_norm_eps)
        h1, rstd1, qkv, o, x_mid, h2, rstd2 = S["h1"][:T], S["rstd1"][:T], S["qkv"][:T], S["o"][:T], S["x_mid"][:T], S["h2"][:T], S["rstd2"][:T]
        lse = S["lse"].view(-1)[: Hq * T].view(Hq, T)
        ops.hip.call("rmsnorm_fwd", x_in, None, 0, None, None, None, P.w(b + "ln1"), h1, rstd1, T, H, H, H, H, eps, None)
        ops.gemm_nt(h1, P.w(b + "qkv.w"), bias=P.w(b + "qkv.b"), out=qkv)
        ops.rope_(qkv, plan.cos, plan.sin, Hq + Hkv, D)
        if not plan.seg.covers(0, T):
            o.zero_()
        ops.hip.call("attn_fwd", qkv[:, :qw], qkv[:, qw: qw + kw], qkv[:, qw + kw:], o, lse, plan.seg.start, plan.seg.end, plan.seg.prefix, plan.seg.n, plan.seg.max_len,
                     plan.seg.n_head, plan.seg.max_tail, T, Hq, Hkv, D, c.qkv_width, c.qkv_width, c.qkv_width, qw, 1, c.attn_scale)
        ab = ops.gemm_nt(o, P.w(b + "o.w"), out=self._workspace_rows("rc_ab", (), T, H, BF16))
        ops.hip.call("rmsnorm_fwd", ab, None, 0, None, x_in, x_mid, P.w(b + "ln2"), h2, rstd2, T, H, H, H, H, eps, None)
        gu, a = ops.gemm_swiglu(h2, P.w(b + "gu.w"), gu_out=S["gu"][:T], a_out=S["a"][:T], keep_gu=True)
        return (x_in, rstd1, h1, qkv, o, lse, x_mid, rstd2, h2, gu, a), (ring, k)

    def text_forward(self, plan: TextPlan, img_embeds, save: bool, kv_sink=None, rows=None, carry=None, recompute: bool = False):
        """Decoder forward over the flat token rows of `plan`.

        recompute (with save=True, single-phase calls): gradient checkpointing -- only the residual-stream rows ENTERING every layer are kept ([L, T, H]);
        text_backward rebuilds one layer's activations at a time (`_recompute_layer`).  The per-layer arena of save=True is not touched.

        rows = (r0, r1, T_total), carry = dict: TWO-PHASE use (save=True only).  This call covers rows [r0, r1) of a T_total-row batch whose
        other rows are produced by another call sharing `carry` and the same activation arena: the rollout's prefill runs the prompt rows
        [0, ng*P) with save=True (they are exactly the prompt region of the shared-prefix training batch, same kernels, same inputs), and
        after the rollout only the completion rows [ng*P, T) go through the layers -- the policy forward no longer recomputes the prompts.
        `plan` then describes the rows of this call (ids / img_index / cos / sin have r1-r0 rows) with ABSOLUTE segment indices; the call
        that completes the batch sets carry["full_plan"] first and gets the ctx for text_backward over all T_total rows."""
        c, P = self.cfg, self.p
        H, D, Hq, Hkv = c.hidden_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        qw, kw = Hq * D, Hkv * D
        Tl = plan.ids.numel()
        r0, r1, T = (0, Tl, Tl) if rows is None else rows
        assert r1 - r0 == Tl and (rows is None or (save and carry is not None))
        part = rows is not None
        recompute = bool(recompute and save and not part)
        keep = save and not recompute          # per-layer arena slabs; otherwise one scratch slab reused by every layer
        CK = self._workspace_rows("ckpt_x_in", (c.num_hidden_layers,), Tl, H, BF16) if recompute else None
        if part and "x0" not in carry:
            # persistent like the activation arena: the decode graph may hold these pointers (rollout side outputs)
            cb = self._ws.get("carry")
            if cb is None or cb["T"] != T:
                cb = self._ws["carry"] = {"T": T, "x0": torch.empty(T, H, dtype=BF16, device=self.dev), "hf": torch.empty(T, H, dtype=BF16, device=self.dev),
                                          "x_last": torch.empty(T, H, dtype=BF16, device=self.dev), "rstdf": torch.empty(T, dtype=F32, device=self.dev)}
            carry.update(x0=cb["x0"], hf=cb["hf"], x_last=cb["x_last"], rstdf=cb["rstdf"])
        x = ops.embed_fwd(plan.ids, plan.img_index if img_embeds is not None else None, P.w("embed"), img_embeds, out=carry["x0"][r0:r1] if part else None)
        B = self._text_buffers(T, keep)
        ctx = {"layers": [], "plan": carry.get("full_plan", plan) if part else plan} if save else None
        if recompute:
            ctx["recompute"] = {"x_in": []}
        res, branch = x, None
        eps = c.rms_norm_eps
        for i in range(c.num_hidden_layers):
            b = f"layers.{i}."
            li = i if keep else 0
            buf = lambda name: B[name][li, r0:r1]
            full = lambda name: B[name][li, :T]
            h1 = buf("h1")
            rstd1 = B["rstd1"][li, r0:r1] if keep else None
            if branch is None:
                x_in = res
                ops.hip.call("rmsnorm_fwd", res, None, 0, None, None, None, P.w(b + "ln1"), h1, rstd1, Tl, H, H, H, H, float(eps), None)
            else:
                x_in = CK[i] if recompute else (buf("x_in") if keep else res)
                ops.hip.call("rmsnorm_fwd", branch, None, 0, None, res, x_in, P.w(b + "ln1"), h1, rstd1, Tl, H, H, H, H, float(eps), None)
            if recompute:
                ctx["recompute"]["x_in"].append(x_in)
            qkv = ops.gemm_nt(h1, P.w(b + "qkv.w"), bias=P.w(b + "qkv.b"), out=buf("qkv"))
            ops.rope_(qkv, plan.cos, plan.sin, Hq + Hkv, D)
            if kv_sink is not None:
                kv_sink(i, qkv[:, qw: qw + kw], qkv[:, qw + kw:])
            o = buf("o")
            if _POISON:
                o.fill_(float("nan"))
            if not plan.seg.covers(r0, r1):
                o.zero_()  # rows outside every segment (left / post-EOS padding) must read as zeros downstream (0 x stale NaN in wgrad otherwise)
            lse = B["lse"][li].view(-1)[: Hq * T].view(Hq, T) if keep else None
            # attention addresses rows absolutely: the keys of a completion segment live in the prompt rows written by the other phase
            qkv_all, o_all = full("qkv"), full("o")
            ops.hip.call("attn_fwd", qkv_all[:, :qw], qkv_all[:, qw: qw + kw], qkv_all[:, qw + kw:], o_all, lse, plan.seg.start, plan.seg.end, plan.seg.prefix, plan.seg.n, plan.seg.max_len,
                         plan.seg.n_head, plan.seg.max_tail, T, Hq, Hkv, D, c.qkv_width, c.qkv_width, c.qkv_width, qw, 1, c.attn_scale)
            ab = ops.gemm_nt(o, P.w(b + "o.w"))
            x_mid = buf("x_mid") if (keep or recompute) else x_in          # (recompute: x_in is a checkpoint, it must not be overwritten)
            h2 = buf("h2")
            rstd2 = B["rstd2"][li, r0:r1] if keep else None
            ops.hip.call("rmsnorm_fwd", ab, None, 0, None, x_in, x_mid, P.w(b + "ln2"), h2, rstd2, Tl, H, H, H, H, float(eps), None)
            # gate|up projection with the activation in its epilogue; the gate|up matrix itself is written only when backward will read it
            gu, a = ops.gemm_swiglu(h2, P.w(b + "gu.w"), gu_out=buf("gu"), a_out=buf("a"), keep_gu=keep)
            branch = ops.gemm_nt(a, P.w(b + "down.w"))
            res = x_mid
            if recompute:
                ctx["layers"].append(None)      # (the scratch slab's x_mid rows are consumed by the next layer's ln1 before its ln2 overwrites them: stream order)
            elif save:
                if not part:
                    ctx["layers"].append((x_in, rstd1, h1, qkv, o, lse, x_mid, rstd2, h2, gu, a))
                else:   # views over ALL rows of the batch (the other phase fills / has filled the rest)
                    ctx["layers"].append((carry["x0"] if i == 0 else full("x_in"), B["rstd1"][li, :T], full("h1"), qkv_all, o_all, lse, full("x_mid"),
                                          B["rstd2"][li, :T], full("h2"), full("gu"), full("a")))
        if part:
            hf, x_last, rstdf = carry["hf"], carry["x_last"], carry["rstdf"]
            ops.hip.call("rmsnorm_fwd", branch, None, 0, None, res, x_last[r0:r1], P.w("norm"), hf[r0:r1], rstdf[r0:r1], Tl, H, H, H, H, float(eps), None)
            ctx.update(x_last=x_last, rstdf=rstdf)
            return hf, ctx
        x_last = torch.empty_like(res) if save else res
        hf, rstdf = ops.rmsnorm_fwd(branch, P.w("norm"), eps, res=res, res_out=x_last, want_rstd=save)
        if save:
            ctx.update(x_last=x_last, rstdf=rstdf)
        return hf, ctx

    def text_context_from_trace(self, plan: TextPlan, rows, carry):
        """The (hf, ctx) pair text_forward(plan, None, save=True, rows=rows, carry=carry) would return for the completion rows [r0, r1), WITHOUT running
        the layers: the rollout's decode steps already wrote those rows of the activation arena (Rollout side outputs, include/iadr1_hip.h
        iadr1_side_out_t).  Only the embedding rows of the completion tokens (layer 0's residual input) are gathered here."""
        c, P = self.cfg, self.p
        Hq, D, Hkv = c.num_attention_heads, c.head_dim, c.num_key_value_heads
        r0, r1, T = rows
        ops.embed_fwd(plan.ids, None, P.w("embed"), None, out=carry["x0"][r0:r1])
        B = self._text_buffers(T, True)
        ctx = {"layers": [], "plan": carry["full_plan"]}
        for i in range(c.num_hidden_layers):
            full = lambda name: B[name][i, :T]
            lse = B["lse"][i].view(-1)[: Hq * T].view(Hq, T)
            ctx["layers"].append((carry["x0"] if i == 0 else full("x_in"), B["rstd1"][i, :T], full("h1"), full("qkv"), full("o"), lse, full("x_mid"),
                                  B["rstd2"][i, :T], full("h2"), full("gu"), full("a")))
        ctx.update(x_last=carry["x_last"], rstdf=carry["rstdf"])
        return carry["hf"], ctx

    def text_backward(self, dhf: torch.Tensor, ctx, dimg32=None, layer_done=None):
        """dhf: gradient of the final-norm output [T,H] bf16.  Accumulates parameter grads; image-embed
        gradients are accumulated (fp32 atomics) into dimg32 [n_img_rows, H]."""
        c, P = self.cfg, self.p
        D, Hq, Hkv = c.head_dim, c.num_attention_heads, c.num_key_value_heads
        qw, kw = Hq * D, Hkv * D
        plan: TextPlan = ctx["plan"]
        T = dhf.shape[0]
        S = self._bw_scratch(T)          # static scratch: no allocator traffic in the loop (see _bw_scratch)
        side = self.wgrad_stream is not None
        take = lambda ring: ring.take(T)
        dres, sl_res = take(S["res"])
        ops.rmsnorm_bwd(dhf, ctx["x_last"], P.w("norm"), ctx["rstdf"], dw=P.g("norm"), out=dres)
        rcomp = ctx.get("recompute")
        for i in reversed(range(c.num_hidden_layers)):
            b = f"layers.{i}."
            rc_slot = None
            if rcomp is not None:
                ctx["layers"][i], rc_slot = self._recompute_layer(i, ctx)
            x_in, rstd1, h1, qkv, o, lse, x_mid, rstd2, h2, gu, a = ctx["layers"][i]
            self._wgrad(b + "down.w", dres, a, slot=sl_res if side else None)          # side stream; the dgrad chain below does not wait for it
            da = ops.gemm_nt(dres, P.wT(b + "down.w"), out=S["da"][:T])
            dgu, sl_gu = take(S["dgu"])
            ops.swiglu_bwd(da, gu, out=dgu)
            self._wgrad(b + "gu.w", dgu, h2, slot=sl_gu if side else None)
            dh2 = ops.gemm_nt(dgu, P.wT(b + "gu.w"), out=S["dh"][:T])
            dx_mid, sl_mid = take(S["res"])
            ops.rmsnorm_bwd(dh2, x_mid, P.w(b + "ln2"), rstd2, dres=dres, dw=P.g(b + "ln2"), out=dx_mid)
            self._wgrad(b + "o.w", dx_mid, o, slot=sl_mid if side else None)
            do = ops.gemm_nt(dx_mid, P.wT(b + "o.w"), out=S["do"][:T])
            dqkv, sl_qkv = take(S["dqkv"])
            # rows of left padding belong to no segment: their gradient is exactly 0 (attn_bwd writes every row that is in a segment)
            if not plan.seg.covers(0, qkv.shape[0]):
                dqkv.zero_()
            elif _POISON:
                dqkv.fill_(float("nan"))
            ops.attn_bwd(qkv[:, :qw], qkv[:, qw: qw + kw], qkv[:, qw + kw:], o, do, lse, plan.seg, Hq, Hkv, D, True, c.attn_scale,
                         dqkv[:, :qw], dqkv[:, qw: qw + kw], dqkv[:, qw + kw:])
            ops.rope_(dqkv, plan.cos, plan.sin, Hq + Hkv, D, backward=True)
            if c.qkv_bias:           # (LLaMA / Mistral: no q/k/v biases -- the fused bias row stays zero and receives no gradient)
                ops.colsum_acc(dqkv, P.g(b + "qkv.b"))
            self._wgrad(b + "qkv.w", dqkv, h1, slot=sl_qkv if side else None)
            if rc_slot is not None and side:
                rc_slot[0].busy(rc_slot[1], self.wgrad_stream)          # the recompute slab is free again once this layer's weight gradients have read it
            dh1 = ops.gemm_nt(dqkv, P.wT(b + "qkv.w"), out=S["dh"][:T])
            dres_new, sl_new = take(S["res"])
            ops.rmsnorm_bwd(dh1, x_in, P.w(b + "ln1"), rstd1, dres=dx_mid, dw=P.g(b + "ln1"), out=dres_new)
            dres, sl_res = dres_new, sl_new
            ctx["layers"][i] = None  # release this layer's activations
            if layer_done is not None:   # this layer's weight gradients are final: the DDP bucket can leave -- ordered after the side stream
                if self.wgrad_stream is not None:
                    with torch.cuda.stream(self.wgrad_stream):
                        layer_done(i)
                else:
                    layer_done(i)
        # backward of the embedding gather: ordered sums per vocabulary row / image-embedding row (host-built CSR, no atomics)
        emb_plan, img_plan = plan.scatter_plans(self.dev)
        ops.rows_scatter_acc(dres, emb_plan, P.g("embed"))
        if dimg32 is not None:
            ops.rows_scatter_acc(dres, img_plan, dimg32)
        self.join_wgrads()

    # ========================================================================================================
    # lm_head + log-softmax + gather (REF sc_grpo_trainer.py:505-513), only on the rows that are consumed
    # ========================================================================================================
    @staticmethod
    def scatter_plan(rows_host: np.ndarray, T: int, device):
        """CSR (ptr [T+1], idx [R]) of `rows`: for each token row the selected rows that read it, in selection order (host integer work)."""
        rows_host = np.asarray(rows_host, dtype=np.int64).reshape(-1)
        order = np.argsort(rows_host, kind="stable").astype(np.int32)
        ptr = np.zeros(T + 1, dtype=np.int32)
        np.cumsum(np.bincount(rows_host, minlength=T), out=ptr[1:])
        return ops.h2d(ptr, device), ops.h2d(order, device)

    def logprobs(self, hf: torch.Tensor, rows: torch.Tensor, targets: torch.Tensor, save: bool, dup=None, rows_host=None, precomputed=None):
        """logp[r] = log_softmax(lm_head(hf[rows[r]]))[targets[r]]  (targets < 0 -> 0).  The [R,V] logits exist
        only as fp32 chunks of `lm_chunk` rows.  `rows` may repeat a hidden row only where `dup` = (sel, dest, group) says so:
        entries rows[sel] all equal dest[group] (shared-prefix layout: the prompt's last token predicts the first token of
        every completion of its group); the backward sums their gradients.
        precomputed: (logp [R], lse [R]) of exactly these rows / targets, already produced by the same fused launch elsewhere (overlap.ChunkedRefPass._policy_head, on
        the side stream under the rollout): nothing is launched here but the gather of the selected rows, which the backward reads."""
        P = self.p
        W = P.w(P.lm_head_name())
        hsel = ops.embed_fwd(rows, None, hf, None)
        R = rows.numel()
        if precomputed is not None:
            logp, lse = precomputed
            assert logp.numel() == R and lse.numel() == R and self.head_mode == "fused"
            ctx = None
            if save:
                ptr, idx = self.scatter_plan(rows_host if rows_host is not None else rows.cpu().numpy(), hf.shape[0], self.dev)
                ctx = {"hsel": hsel, "rows": rows, "targets": targets, "lse": lse, "T": hf.shape[0], "scatter": (ptr, idx), "logits": None}
            return logp, ctx
        logp = torch.empty(R, dtype=F32, device=self.dev)
        lse = torch.empty(R, dtype=F32, device=self.dev)
        V = W.shape[0]
        # with `save`, the fp32 logits of all R rows are kept for the backward (10 GB for 16384 x 151936: nothing on a 288 GB part)
        # instead of being recomputed chunk by chunk; without it only one chunk is ever alive
        fused = self.head_mode == "fused"
        # fused: one linear_logprob launch pair over all R rows (GEMM epilogue -> (max, sum exp) per 64-column slice + target logit, then a per-row merge)
        keep = save and R * V * 4 <= self.keep_logits_bytes and not fused
        lg_all = self._workspace("lm_logits_all", (R, V), F32) if keep else None
        if fused:
            need = ops.linear_logprob_ws_bytes(R, V)
            ws = self._ws.get("lm_logprob_ws")
            if ws is None or ws.numel() < need:                     # grow-only: the row count changes from batch to batch
                ws = self._ws["lm_logprob_ws"] = torch.empty(need + need // 4, dtype=torch.uint8, device=self.dev)
            ops.linear_logprob(hsel, W, targets, logp=logp, lse=lse, ws=ws)
        else:
            lg_buf = None if keep else self._workspace("lm_logits", (min(R, self.lm_chunk), V), F32)
            for r0 in range(0, R, self.lm_chunk):
                r1 = min(R, r0 + self.lm_chunk)
                lg = ops.gemm_nt(hsel[r0:r1], W, out=lg_all[r0:r1] if keep else lg_buf[: r1 - r0])
                lp, ls = ops.logprob_rows(lg, targets[r0:r1])
                logp[r0:r1] = lp
                lse[r0:r1] = ls
        ctx = None
        if save:
            ptr, idx = self.scatter_plan(rows_host if rows_host is not None else rows.cpu().numpy(), hf.shape[0], self.dev)
            ctx = {"hsel": hsel, "rows": rows, "targets": targets, "lse": lse, "T": hf.shape[0], "scatter": (ptr, idx), "logits": lg_all}
        return logp, ctx

    def logprobs_backward(self, g: torch.Tensor, ctx) -> torch.Tensor:
        """g[r] = dLoss/dlogp[r] (fp32).  Returns dLoss/dhf [T,H] bf16; accumulates the lm_head (= embedding when tied) grad."""
        P = self.p
        name = P.lm_head_name()
        W, WT = P.w(name), P.wT(name)
        hsel, rows, targets, lse = ctx["hsel"], ctx["rows"], ctx["targets"], ctx["lse"]
        R, H = hsel.shape
        dhsel = torch.empty(R, H, dtype=BF16, device=self.dev)
        V = W.shape[0]
        nc = min(R, self.lm_chunk)
        kept = ctx.get("logits")
        fused = kept is None and self.head_mode == "fused"
        lg_buf = None if kept is not None or fused else self._workspace("lm_logits", (nc, V), F32)
        dl_buf = self._workspace("lm_dlogits", (nc, V), BF16)
        for r0 in range(0, R, self.lm_chunk):
            r1 = min(R, r0 + self.lm_chunk)
            n = r1 - r0
            if fused:
                dl = ops.linear_logprob_dlogits(hsel[r0:r1], W, targets[r0:r1], lse[r0:r1], g[r0:r1], out=dl_buf[:n])
            else:
                lg = kept[r0:r1] if kept is not None else ops.gemm_nt(hsel[r0:r1], W, out=lg_buf[:n])
                dl = ops.dlogits_rows(lg, targets[r0:r1], lse[r0:r1], g[r0:r1], out=dl_buf[:n])
            ops.gemm_nt(dl, WT, out=dhsel[r0:r1])
            ops.gemm_tn_acc(dl, hsel[r0:r1], self.p.g(name), wide=self.wgrad_tn_wide)          # dW_head += dl^T . h  (no [V, n] transposed copy of the logit gradients)
        # scatter back onto the token rows: dhf[t] = sum of the selected rows that read hidden row t (0 for rows nothing read)
        ptr, idx = ctx["scatter"]
        return ops.rows_gather_sum(dhsel, ptr, idx, ctx["T"])

    def logits_rows(self, hf: torch.Tensor, rows: torch.Tensor) -> torch.Tensor:
        """fp32 logits of a FEW rows (rollout prefill: last prompt position of each prompt)."""
        P = self.p
        hsel = ops.embed_fwd(rows, None, hf, None)
        return ops.gemm_nt(hsel, P.w(P.lm_head_name()), out_dtype=F32)
