"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the LLaVA-OneVision branch of the reference's model switch
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:124-132 loads `LlavaOnevisionForConditionalGeneration`; its arithmetic lives in
the reference's pinned third-party dependency transformers==4.51.3 (requirements.txt:205), not under /root/reference).  TF: = the installed
transformers 5.15.0, models/llava_onevision/modeling_llava_onevision.py, SG: = models/siglip/modeling_siglip.py.

  vision tower   SG:116-181 (patch conv + learned positions), SG:325-356 (pre-LN block: LayerNorm, biased q/k/v/out attention over the 729
                 tokens of one crop, LayerNorm, fc1 -> GELU(tanh) -> fc2), hidden state of the LAST encoder layer, before post_layernorm
                 (vision_feature_layer = -1, strategy "full": TF:399-408)
  projector      TF:131-150  linear_1 -> exact GELU -> linear_2
  packing        TF:280-348  base features ++ crop grid, unpad (TF:221-262), bilinear shrink above anyres_max_9, image_newline per row
  decoder        Qwen2 (the Qwen2.5-VL decoder of oracle/qwen25vl.py with ordinary 1-D rotary positions = column index, as HF assigns them
                 when no position_ids are passed)

Pinned by tests/golden/llava_ov.npz, captured from a tiny HF LlavaOnevisionForConditionalGeneration by tools/make_golden_llava.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import qwen25vl as oq


def _select_best_resolution(size, pinpoints):
    oh, ow = int(size[0]), int(size[1])
    best, me, mw = None, 0, float("inf")
    for h, w in pinpoints:
        sc = min(w / ow, h / oh)
        dw, dh = int(ow * sc), int(oh * sc)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > me or (eff == me and waste < mw):
            me, mw, best = eff, waste, (h, w)
    return best


def _unpad(t, size):
    """TF:221-262 on a [C, H, W] tensor."""
    oh, ow = int(size[0]), int(size[1])
    ch, cw = t.shape[1:]
    if ow / oh > cw / ch:
        nh = int(round(oh * (cw / ow), 7))
        pad = (ch - nh) // 2
        return t[:, pad: ch - pad, :]
    nw = int(round(ow * (ch / oh), 7))
    pad = (cw - nw) // 2
    return t[:, :, pad: cw - pad]


class LlavaOVOracle(oq.Qwen25VLOracle):
    """cfg: {"text": {...Qwen2...}, "vision": {hidden_size, intermediate_size, depth, num_heads, patch_size, image_size, layer_norm_eps},
    "image_token_id", "image_grid_pinpoints", "anyres_max", ...}; weights keyed by the checkpoint names of the reference's dependency
    (vision_tower.vision_model.*, multi_modal_projector.*, image_newline, language_model.model.*, language_model.lm_head.weight)."""

    def __init__(self, cfg, weights, requires_grad=False, dtype=torch.float32, copy: bool = True):
        w2 = {}
        for k, a in weights.items():
            if k.startswith("language_model.model."):
                w2["model." + k[len("language_model.model."):]] = a
            elif k == "language_model.lm_head.weight":
                w2["lm_head.weight"] = a
            else:
                w2[k] = a
        text_cfg = dict(cfg)
        text_cfg["text"] = dict(cfg["text"], mrope_section=[cfg["text"]["hidden_size"] // cfg["text"]["num_attention_heads"] // 2, 0, 0])
        super().__init__(text_cfg, w2, requires_grad=requires_grad, dtype=dtype, copy=copy)
        self.cfg = text_cfg

    def parameters(self):
        back = lambda k: ("language_model." + k if k.startswith("model.") else ("language_model.lm_head.weight" if k == "lm_head.weight" else k))
        tied = self.cfg.get("tie_word_embeddings", False)
        for k, t in self.w.items():
            if tied and k == "lm_head.weight":
                continue
            yield back(k), t

    # -- SigLIP tower on a stack of crops [n, 3, S, S] -> [n, tokens, vh] -------------------------------------------------------
    def tower(self, crops):
        v, w = self.cfg["vision"], self.w
        p, vh, nh = v["patch_size"], v["hidden_size"], v["num_heads"]
        pre = "vision_tower.vision_model."
        x = F.conv2d(crops.to(w[pre + "embeddings.patch_embedding.weight"].dtype), w[pre + "embeddings.patch_embedding.weight"], w[pre + "embeddings.patch_embedding.bias"], stride=p)
        x = x.flatten(2).transpose(1, 2) + w[pre + "embeddings.position_embedding.weight"][None]      # SG:163-181
        n, L, _ = x.shape
        hd = vh // nh
        for i in range(v["depth"]):
            b = f"{pre}encoder.layers.{i}."
            h = F.layer_norm(x, (vh,), w[b + "layer_norm1.weight"], w[b + "layer_norm1.bias"], v["layer_norm_eps"])
            q, k, vv = (F.linear(h, w[b + f"self_attn.{z}_proj.weight"], w[b + f"self_attn.{z}_proj.bias"]).view(n, L, nh, hd).transpose(1, 2) for z in "qkv")
            pr = torch.softmax((q @ k.transpose(2, 3)) * hd**-0.5, -1, dtype=torch.float32).to(x.dtype)
            a = (pr @ vv).transpose(1, 2).reshape(n, L, vh)
            x = x + F.linear(a, w[b + "self_attn.out_proj.weight"], w[b + "self_attn.out_proj.bias"])
            h = F.layer_norm(x, (vh,), w[b + "layer_norm2.weight"], w[b + "layer_norm2.bias"], v["layer_norm_eps"])
            h = F.gelu(F.linear(h, w[b + "mlp.fc1.weight"], w[b + "mlp.fc1.bias"]), approximate="tanh")
            x = x + F.linear(h, w[b + "mlp.fc2.weight"], w[b + "mlp.fc2.bias"])
        return x

    def project(self, feats):
        w = self.w
        h = F.gelu(F.linear(feats, w["multi_modal_projector.linear_1.weight"], w["multi_modal_projector.linear_1.bias"]))
        return F.linear(h, w["multi_modal_projector.linear_2.weight"], w["multi_modal_projector.linear_2.bias"])

    def visual(self, pixel_values, image_sizes, return_last_hidden=False):
        """pixel_values: [total crops, 3, S, S] (crops of all images in order, base image first); image_sizes: [(h, w)].
        -> packed image tokens [n_tokens, H] in the order the <image> placeholders consume them."""
        v = self.cfg["vision"]
        crop, side = v["image_size"], v["image_size"] // v["patch_size"]
        pins = [tuple(p) for p in self.cfg["image_grid_pinpoints"]]
        feats = self.project(self.tower(pixel_values))                   # [crops, side^2, H]
        out, c0 = [], 0
        newline = self.w["image_newline"]
        for size in image_sizes:
            bh, bw = _select_best_resolution(size, pins)
            gh, gw = bh // crop, bw // crop
            nc = gh * gw + 1
            f = feats[c0: c0 + nc]
            c0 += nc
            base, rest = f[0], f[1:]
            rest = rest.view(gh, gw, side, side, -1).permute(4, 0, 2, 1, 3).contiguous().flatten(1, 2).flatten(2, 3)     # [H, gh*side, gw*side]
            rest = _unpad(rest, size)
            ch, cw = rest.shape[1:]
            ratio = math.sqrt(ch * cw / (self.cfg.get("anyres_max", 9) * side**2))
            if ratio > 1.1:
                rest = F.interpolate(rest[None], [int(ch // ratio), int(cw // ratio)], mode="bilinear")[0]
            rest = torch.cat([rest, newline[:, None, None].expand(*rest.shape[:-1], 1).to(rest.dtype)], -1)
            out.append(torch.cat([base, rest.flatten(1, 2).transpose(0, 1)], 0))
        return torch.cat(out, 0)

    def hidden_states(self, input_ids, attention_mask, pixel_values=None, image_sizes=None, return_hidden=False):
        img = self.visual(pixel_values, image_sizes) if pixel_values is not None else None
        x = self.embed(input_ids, img)
        B, S = input_ids.shape
        pos = torch.arange(S).view(1, 1, S).expand(3, B, S)             # HF: position_ids = cache_position (column index), padding included
        return self.text_model(x, attention_mask, pos, return_hidden)

    def logits(self, input_ids, attention_mask, pixel_values=None, image_sizes=None):
        return self.hidden_states(input_ids, attention_mask, pixel_values, image_sizes) @ self.w["lm_head.weight"].t()

    def per_token_logps(self, input_ids, attention_mask, pixel_values=None, image_sizes=None):
        lg = self.logits(input_ids, attention_mask, pixel_values, image_sizes)[:, :-1]
        return torch.log_softmax(lg.float(), -1).gather(-1, input_ids[:, 1:].unsqueeze(-1)).squeeze(-1)


def ensure_left_padding(input_ids, attention_mask, pad_token_id):
    """REF train/stage_rl/trainer/sc_grpo_trainer.py:516-567 `_ensure_left_padding_data`, applied to every llava batch before the forward of
    `_get_per_token_logps` (REF:502-504): a row whose first pad token starts an all-pad tail (i.e. a row WITHOUT left padding whose completion
    ended early) is rotated -- content to the right end, pads to the left -- while the caller keeps slicing the log-probs at the un-rotated
    columns (REF:735,743) and masking them with the un-rotated completion mask."""
    ids, mask = input_ids.clone(), attention_mask.clone()
    B, S = ids.shape
    for i in range(B):
        pm = input_ids[i] == pad_token_id
        if pm.any():
            first = int(pm.nonzero()[0])
            if bool((input_ids[i, first:] == pad_token_id).all()):
                n = S - first
                ids[i] = torch.cat([torch.full((n,), pad_token_id, dtype=ids.dtype), input_ids[i, :first]])
                mask[i] = torch.cat([torch.zeros(n, dtype=mask.dtype), torch.ones(first, dtype=mask.dtype)])
    return ids, mask


class LlavaOracle(LlavaOVOracle):
    """LLaVA-1.5 (`LlavaForConditionalGeneration`, REF sc_grpo_trainer.py:133-135) and LLaVA-NeXT / 1.6 (`LlavaNextForConditionalGeneration`, REF:130-132):
    cfg["family"] in ("llava", "llava_next").  CL: = transformers models/clip/modeling_clip.py, LV: = models/llava/modeling_llava.py, LN: =
    models/llava_next/modeling_llava_next.py (installed 5.15.0).

      vision tower  CLIP ViT (CL:141-200 embeddings: bias-free patch conv, class token first, learned positions; CL pre_layrnorm; pre-LN blocks with
                    biased q/k/v/out attention over the 1 + side^2 tokens of one crop and fc1 -> QuickGELU -> fc2), hidden state of encoder layer
                    `vision_feature_layer` = -2 (the last block is not run), strategy "default": the class token is dropped (LV:161-172)
      projector     linear_1 -> exact GELU -> linear_2 (LV:96-128)
      llava         one 336-pixel crop per image, its side^2 features replace the image tokens (LV:174-189)
      llava_next    LN:265-330: base features ++ crop grid un-padded to the aspect ratio, `image_newline` after every feature row; no shrink step
      decoder       LLaMA / Mistral = the Qwen2 decoder without q/k/v biases (zeros here, not parameters)

    Pinned by tests/golden/llava15.npz / llava_next.npz (tiny HF models, tools/make_golden_llava.py)."""

    def __init__(self, cfg, weights, requires_grad=False, dtype=torch.float32):
        import numpy as np
        t = cfg["text"]
        hd = t["hidden_size"] // t["num_attention_heads"]
        w = dict(weights)
        self._no_bias = set()
        for i in range(t["num_hidden_layers"]):
            for z, n in (("q", t["num_attention_heads"]), ("k", t["num_key_value_heads"]), ("v", t["num_key_value_heads"])):
                k = f"language_model.model.layers.{i}.self_attn.{z}_proj.bias"
                if k not in w:
                    w[k] = np.zeros(n * hd, dtype=np.float32)
                    self._no_bias.add(k)
        super().__init__(cfg, w, requires_grad=requires_grad, dtype=dtype)
        for k in self._no_bias:                      # constants, not parameters
            kk = "model." + k[len("language_model.model."):]
            self.w[kk] = self.w[kk].detach().requires_grad_(False)

    def parameters(self):
        for k, t_ in super().parameters():
            if k not in self._no_bias:
                yield k, t_

    def tower(self, crops):
        v, w = self.cfg["vision"], self.w
        p, vh, nh = v["patch_size"], v["hidden_size"], v["num_heads"]
        pre = "vision_tower.vision_model."
        x = F.conv2d(crops.to(w[pre + "embeddings.patch_embedding.weight"].dtype), w[pre + "embeddings.patch_embedding.weight"], None, stride=p).flatten(2).transpose(1, 2)
        n = x.shape[0]
        x = torch.cat([w[pre + "embeddings.class_embedding"].expand(n, 1, -1), x], 1) + w[pre + "embeddings.position_embedding.weight"][None]      # CL:188-200
        x = F.layer_norm(x, (vh,), w[pre + "pre_layrnorm.weight"], w[pre + "pre_layrnorm.bias"], v["layer_norm_eps"])
        L, hd = x.shape[1], vh // nh
        run = v["depth"] + 1 + self.cfg.get("vision_feature_layer", -2)          # hidden_states[-2] of depth + 1 entries = the output of block depth - 2
        for i in range(run):
            b = f"{pre}encoder.layers.{i}."
            h = F.layer_norm(x, (vh,), w[b + "layer_norm1.weight"], w[b + "layer_norm1.bias"], v["layer_norm_eps"])
            q, k, vv = (F.linear(h, w[b + f"self_attn.{z}_proj.weight"], w[b + f"self_attn.{z}_proj.bias"]).view(n, L, nh, hd).transpose(1, 2) for z in "qkv")
            pr = torch.softmax((q @ k.transpose(2, 3)) * hd**-0.5, -1, dtype=torch.float32).to(x.dtype)
            x = x + F.linear((pr @ vv).transpose(1, 2).reshape(n, L, vh), w[b + "self_attn.out_proj.weight"], w[b + "self_attn.out_proj.bias"])
            h = F.layer_norm(x, (vh,), w[b + "layer_norm2.weight"], w[b + "layer_norm2.bias"], v["layer_norm_eps"])
            h = F.linear(h, w[b + "mlp.fc1.weight"], w[b + "mlp.fc1.bias"])
            x = x + F.linear(h * torch.sigmoid(1.702 * h), w[b + "mlp.fc2.weight"], w[b + "mlp.fc2.bias"])
        return x[:, 1:]                                                         # "default": without the class token

    def visual(self, pixel_values, image_sizes=None, return_last_hidden=False):
        v = self.cfg["vision"]
        crop, side = v["image_size"], v["image_size"] // v["patch_size"]
        feats = self.project(self.tower(pixel_values))                          # [crops, side^2, H]
        if self.cfg["family"] == "llava":
            return feats.reshape(-1, feats.shape[-1])
        pins = [tuple(p) for p in self.cfg["image_grid_pinpoints"]]
        newline = self.w["image_newline"]
        out, c0 = [], 0
        for size in image_sizes:
            bh, bw = _select_best_resolution(size, pins)
            gh, gw = bh // crop, bw // crop
            f = feats[c0: c0 + gh * gw + 1]
            c0 += gh * gw + 1
            rest = f[1:].view(gh, gw, side, side, -1).permute(4, 0, 2, 1, 3).contiguous().flatten(1, 2).flatten(2, 3)
            rest = _unpad(rest, size)
            rest = torch.cat([rest, newline[:, None, None].expand(*rest.shape[:-1], 1).to(rest.dtype)], -1)
            out.append(torch.cat([f[0], rest.flatten(1, 2).transpose(0, 1)], 0))
        return torch.cat(out, 0)
