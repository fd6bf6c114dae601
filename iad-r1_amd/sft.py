"""PA-SFT step on the HIP engine: model(**batch, labels) -> shifted cross-entropy with ignore_index=-100
(TF:loss/loss_utils.py:32-71 via TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:1389-1393), gradient-accumulation
rescale of /root/reference/train/stage_sft/llamafactory/train/sft/trainer.py:92-107, AdamW with decoupled weight
decay on >=2-D parameters (HF Trainer parameter groups)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch

from . import hip, ops
from .params import ParamStore, VLMConfig
from .sc_grpo import GradReducer
from .vlm import Engine

BF16, F32 = torch.bfloat16, torch.float32


@dataclass
class SFTArgs:
    learning_rate: float = 1e-5
    weight_decay: float = 0.1
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    gradient_accumulation_steps: int = 1
    micro_batch_seqs: int = 16
    frozen: Optional[Callable[[str], bool]] = None      # parameter-store names that do not train (frozen_parameter_rule); None: everything trains
    recompute: str = "off"                              # `--gradient_checkpointing`: "off" | "auto" | "on" (vlm.Engine.recompute_wanted)


# Families the reference's LLaMA-Factory registers as composite models (model/model_utils/visual.py:236-288), by HF `model_type`.  For these -- and only
# these -- full fine-tuning freezes the vision tower and the projector unless told otherwise (hparams/finetuning_args.py:416-423: both default to True;
# no launch script overrides them).  `qwen2_5_vl` and `llava_onevision` are NOT registered there: their whole model trains.
_COMPOSITE = ("qwen2_vl", "llava", "llava_next")


def frozen_parameter_rule(model_type: str, freeze_vision_tower: bool = True, freeze_multi_modal_projector: bool = True) -> Optional[Callable[[str], bool]]:
    """Which tensors of the parameter store PA-SFT leaves untouched: `_setup_full_tuning` (llamafactory/model/adapter.py:39-55) clears requires_grad of every
    parameter whose name contains a key of `get_forbidden_modules` (model_utils/visual.py:153-171) -- qwen2_vl: `visual.patch_embed`, `visual.blocks` (tower),
    `visual.merger` (projector); llava / llava_next: `vision_tower`, `multi_modal_projector` (`image_newline` and the language model train).  Restated on the
    store's names (tower: visual.patch_embed*, visual.blocks.*, visual.pos, visual.cls, visual.pre_ln*; projector: visual.merger.*); pinned to the reference's
    function by tests/golden/sft_freeze.json."""
    if model_type not in _COMPOSITE or not (freeze_vision_tower or freeze_multi_modal_projector):
        return None
    tower = ("visual.patch_embed", "visual.blocks.", "visual.pos", "visual.cls", "visual.pre_ln")

    def frozen(name: str) -> bool:
        if name.startswith("visual.merger."):
            return freeze_multi_modal_projector
        return freeze_vision_tower and name.startswith(tower)
    return frozen


class SFTEngine:
    def __init__(self, cfg: VLMConfig, params: ParamStore, args: SFTArgs, group=None):
        self.cfg, self.args = cfg, args
        self.eng = Engine(params)
        self.eng.wgrad_tn_wide = True       # every weight gradient straight from row-major dY / X (measured on the PA-SFT step: 358.9 -> 351.3 ms, profiles/r05_gemm_tn.txt)
        self.dev = params.device
        self.reducer = GradReducer(params, group)
        self.opt_step = 0
        self.accum = 0
        self.norm2 = torch.zeros(1, dtype=F32, device=self.dev)
        self.norm_scratch = torch.zeros(2048, dtype=F32, device=self.dev)
        # frozen tensors: no optimizer visit, and no backward work that only they would need
        fz = args.frozen
        vis = [n for n in params.slots if n.startswith("visual.")]
        self.segments = params.optimizer_segments(fz)
        self.frozen_ranges = params.frozen_ranges(fz) if fz is not None else []
        live = [n for n in vis if fz is None or not fz(n)]
        # "all": the tower / projector backward runs; "newline": only the row the any-resolution packing inserts trains (LLaVA-NeXT with a frozen tower and
        # projector), its gradient needs the transposed packing map alone; "none": the image embeddings are constants of the step
        self.vision_grads = "all" if any(n != "visual.newline" for n in live) else ("newline" if live else "none")

    def loss_and_grads(self, batch, backward=True, num_items_in_batch=None, last_micro_step=True):
        """batch: input_ids, attention_mask, labels [B,S] (numpy), pixel_values [n,patch_dim], image_grid_thw [n_img,3],
        images_per_row (default 1).  Loss = sum CE over labels != -100 / count (or / num_items_in_batch)."""
        c, e = self.cfg, self.eng
        ids, mask, labels = (np.asarray(batch[k]) for k in ("input_ids", "attention_mask", "labels"))
        B, S = ids.shape
        grids, plan_v, px, rows = e.vision_inputs(batch)
        img, vctx = e.vision_forward(px, plan_v, save=backward and self.vision_grads == "all")
        ipr = batch.get("images_per_row") or [1] * B
        gpr, off, k = [], [], 0
        for n in ipr:
            gpr.append(grids[k: k + n])
            off.append([int(rows[j]) for j in range(k, k + n)])
            k += n
        # shifted targets: position s predicts labels[s+1]; only supervised positions are sent through the lm_head
        tgt = np.full((B, S), -100, dtype=np.int64)
        tgt[:, :-1] = labels[:, 1:]
        n_items = int((tgt != -100).sum()) if num_items_in_batch is None else int(num_items_in_batch)
        dimg32 = torch.zeros(img.shape, dtype=F32, device=self.dev) if (backward and self.vision_grads != "none") else None
        total = torch.zeros((), dtype=F32, device=self.dev)
        mb = max(1, min(self.args.micro_batch_seqs, B))
        starts = list(range(0, B, mb))
        for si, r0 in enumerate(starts):
            r1 = min(B, r0 + mb)
            plan = e.text_plan(ids[r0:r1], mask[r0:r1], gpr[r0:r1], off[r0:r1])
            sel = np.flatnonzero(tgt[r0:r1].reshape(-1) != -100)
            if len(sel) == 0:
                continue
            rows_d = ops.h2d(sel.astype(np.int64), self.dev)
            tgt_d = ops.h2d(tgt[r0:r1].reshape(-1)[sel], self.dev)
            hf, ctx = e.text_forward(plan, img, save=backward, recompute=backward and e.recompute_wanted(plan.ids.numel(), self.args.recompute))
            lp, lctx = e.logprobs(hf, rows_d, tgt_d, save=backward, rows_host=sel)
            total += -lp.sum()
            if backward:
                g = torch.full((len(sel),), -1.0 / n_items, dtype=F32, device=self.dev)
                dhf = e.logprobs_backward(g, lctx)
                hook = self.reducer.layer_ready if (last_micro_step and si == len(starts) - 1) else None
                e.text_backward(dhf, ctx, dimg32, layer_done=hook)
        if backward:
            if self.vision_grads == "all":
                e.vision_backward(ops.f32_bias_to_bf16(dimg32, None), vctx)
            elif self.vision_grads == "newline":
                e.vision_backward(ops.f32_bias_to_bf16(dimg32, None), {"plan": plan_v}, only_newline=True)
            self.accum += 1
            if last_micro_step:
                # a slice without supervised tokens ran no backward and fired no hook: send the layer buckets it skipped now, in the same order,
                # so that every rank issues the same collective sequence whatever its local data was
                self.reducer.all_layers_ready()
        return float(total) / n_items

    def optimizer_step(self):
        a, st = self.args, self.eng.p
        self.reducer.finish()
        self.opt_step += 1
        scale = 1.0 / (self.reducer.world * max(1, self.accum))
        if self.vision_grads == "all":
            for lo, hi in self.frozen_ranges:      # a partly frozen vision side: the backward wrote gradients of frozen tensors too; they count in no norm and no update
                st.grad[lo:hi].zero_()
        hip.call("sumsq", st.grad, st.n_total, self.norm_scratch, self.norm2)      # (frozen ranges hold zeros)
        self.grad_scale = scale          # grad_norm() = sqrt(norm2) * scale: the norm of the averaged gradient, before clipping (HF's `grad_norm` log key)
        for lo, hi, decays in self.segments:
            hip.call("adamw_flat", st.master[lo:hi], st.m[lo:hi], st.v[lo:hi], st.grad[lo:hi], st.flat[lo:hi], hi - lo, a.learning_rate,
                     a.adam_beta1, a.adam_beta2, a.adam_epsilon, a.weight_decay if decays else 0.0, self.opt_step, scale, self.norm2, a.max_grad_norm)
        st.refresh_shadows()
        self.accum = 0
        if self.eng.check_ddp_headroom(a.recompute, group=self.reducer.group):
            torch.cuda.empty_cache()

    def grad_norm(self) -> float:
        """Global L2 norm of the last optimizer step's (averaged, unclipped) gradient; synchronises the device."""
        return float(self.norm2.sqrt().item()) * getattr(self, "grad_scale", 1.0)
