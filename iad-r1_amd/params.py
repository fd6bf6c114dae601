"""Model configuration + parameter store for the Qwen2.5-VL engine.

Memory layout (MI355X-first: 288 GB HBM, no sharding): every parameter of the model lives in ONE flat bf16
buffer (GEMM operands), with fused layouts chosen for the kernels -- q|k|v rows in one matrix, gate|up rows
in one matrix (ViT widths zero-padded to a multiple of 64) -- plus, for a trainable copy, flat fp32 master /
Adam m / Adam v / gradient buffers with the same offsets, so the optimizer is one launch per decay group, the
DDP all-reduce walks one buffer in large buckets, and a transposed shadow ([K,N]) of every GEMM weight is kept
for the dgrad GEMMs.  Names on the outside are the checkpoint names the reference loads / saves
(`visual.blocks.N.attn.qkv.weight`, `model.layers.N.self_attn.q_proj.weight`, `lm_head.weight`, ...;
/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:118-119, grpo_ad.py:203)."""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field

import numpy as np
import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


def _rup(x, m):
    return (x + m - 1) // m * m




def _first_set(*values):
    """First value that is not None (a token id of 0 is a value, not 'unset')."""
    return next(v for v in values if v is not None)


@dataclass
class VLMConfig:
    # text decoder
    vocab_size: int
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    rms_norm_eps: float
    rope_theta: float
    mrope_section: tuple
    # vision tower
    v_depth: int
    v_hidden: int
    v_inter: int
    v_heads: int
    v_in_channels: int
    v_patch: int
    v_merge: int
    v_temporal: int
    v_window: int
    v_fullatt: tuple
    # tokens
    image_token_id: int
    vision_start_token_id: int
    vision_end_token_id: int
    eos_token_id: int
    pad_token_id: int
    tie_word_embeddings: bool
    # "qwen2_5_vl": RMSNorm + gated SwiGLU MLP + window attention;  "qwen2_vl": LayerNorm(+bias) + fc1/QuickGELU/fc2, full
    # attention only (TF:models/qwen2_vl/modeling_qwen2_vl.py:418-447) -- the decoder is identical
    v_arch: str = "qwen2_5_vl"
    # "siglip" (LLaVA-OneVision branch of the reference's model switch, sc_grpo_trainer.py:124-132): SigLIP tower -- learned positions, LayerNorm(+bias),
    # biased q/k/v/out attention over the v_tokens of one crop (no rotary), fc1 / GELU(tanh) / fc2 -- Linear-GELU-Linear projector, any-resolution
    # crop grid with `image_newline`, Qwen2 decoder with ordinary 1-D rotary positions
    v_image_size: int = 0            # crop side in pixels (384)
    v_ln_eps: float = 1e-6
    image_grid_pinpoints: tuple = ()
    anyres_max: int | None = 9       # N of vision_aspect_ratio "anyres_max_N" (LLaVA-OneVision); None: no shrink step (LLaVA-NeXT)
    # "clip" (LLaVA-1.5 / LLaVA-NeXT branches, sc_grpo_trainer.py:130-135): CLIP tower -- bias-free patch conv, class token, learned positions, pre-LayerNorm,
    # LayerNorm(+bias) blocks with QuickGELU MLPs, features of encoder layer `v_feature_layer` (-2: the last block is not run) without the class token.
    # llava_family: "onevision" (any-resolution packing with shrink), "next" (without), "llava" (one crop per image, no packing, no image_newline)
    llava_family: str = ""
    v_feature_layer: int = -1
    qkv_bias: bool = True            # q/k/v projections of the decoder carry biases (Qwen2); False: LLaMA / Mistral -- the fused q|k|v bias stays zero and is no parameter

    @property
    def is_llava(self):
        return self.v_arch in ("siglip", "clip")

    @property
    def v_cls(self):
        """Class tokens in front of a crop's patch tokens (CLIP: 1)."""
        return 1 if self.v_arch == "clip" else 0

    @property
    def v_seq(self):
        """Tokens of one crop inside the tower."""
        return self.v_tokens + self.v_cls

    @property
    def v_run_depth(self):
        """Encoder blocks that are run: hidden_states[v_feature_layer] of the depth + 1 recorded states."""
        return self.v_depth + 1 + self.v_feature_layer

    @property
    def v_head_pad(self):
        """Head width the attention kernels run at: SigLIP's 72 is zero-padded to 80 (a built head size) inside the fused q|k|v / out-projection
        weights -- the padded dims are zero in q, k, v and meet zero out-projection columns, so values and gradients are exactly those of width 72."""
        d = self.v_hidden // self.v_heads
        if d in (80, 128):
            return d
        if d in (72, 64):        # SigLIP-so400m: 72, CLIP ViT-L: 64
            return 80
        raise ValueError(f"vision head dim {d}: the attention kernels are built for 128 and 80 (72 and 64 run zero-padded to 80)")

    @property
    def v_side(self):
        return self.v_image_size // self.v_patch

    @property
    def v_tokens(self):
        return self.v_side**2

    @property
    def patch_dim_pad(self):
        return _rup(self.patch_dim, 8)

    @property
    def head_dim_real(self):
        return self.hidden_size // self.num_attention_heads

    @property
    def head_dim(self):
        """Head width the decoder KERNELS run at: 128.  A narrower head (64: the Qwen2-0.5B decoder of LLaVA-OneVision-0.5B) is stored zero-padded to 128 in
        q|k|v rows / o columns with its two rotary halves at [0, r/2) and [64, 64 + r/2) (`head_slots`), so that the kernels' rotary partners (d, d + 64)
        are the model's (d, d + r/2); the zero dims add nothing to q.k, give zero outputs, and keep zero gradients.  `attn_scale` uses the real width."""
        r = self.head_dim_real
        return r if r >= 128 else 128

    @property
    def attn_scale(self):
        return float(self.head_dim_real) ** -0.5

    @property
    def head_slots(self):
        """Index of real head dim d inside the padded head (identity when nothing is padded)."""
        r, D = self.head_dim_real, self.head_dim
        return np.arange(r) if r == D else np.concatenate([np.arange(r // 2), D // 2 + np.arange(r // 2)])

    @property
    def v_head_dim(self):
        return self.v_hidden // self.v_heads

    @property
    def qkv_width(self):
        return (self.num_attention_heads + 2 * self.num_key_value_heads) * self.head_dim

    @property
    def v_inter_pad(self):
        return _rup(self.v_inter, 64)

    @property
    def patch_dim(self):
        return self.v_in_channels * self.v_temporal * self.v_patch * self.v_patch

    @staticmethod
    def from_dict(d: dict) -> "VLMConfig":
        """Accepts the nested {text, vision, ...} form of tests/fixture_util.TINY."""
        t, v = d["text"], d["vision"]
        if v.get("arch") in ("siglip", "clip"):
            clip = v["arch"] == "clip"
            hd = max(128, t["hidden_size"] // t["num_attention_heads"])      # the width the kernels run at (VLMConfig.head_dim): 1-D positions drive every rotary pair
            return VLMConfig(
                vocab_size=t["vocab_size"], hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"], num_hidden_layers=t["num_hidden_layers"],
                num_attention_heads=t["num_attention_heads"], num_key_value_heads=t["num_key_value_heads"], rms_norm_eps=t["rms_norm_eps"], rope_theta=t["rope_theta"],
                mrope_section=(hd // 2, 0, 0), v_depth=v["depth"], v_hidden=v["hidden_size"], v_inter=v["intermediate_size"], v_heads=v["num_heads"],
                v_in_channels=v["in_channels"], v_patch=v["patch_size"], v_merge=1, v_temporal=1, v_window=0, v_fullatt=tuple(range(v["depth"])),
                image_token_id=d["image_token_id"], vision_start_token_id=d.get("vision_start_token_id", -1), vision_end_token_id=d.get("vision_end_token_id", -1),
                eos_token_id=d["eos_token_id"], pad_token_id=d["pad_token_id"], tie_word_embeddings=d.get("tie_word_embeddings", False), v_arch=v["arch"],
                v_image_size=v["image_size"], v_ln_eps=v.get("layer_norm_eps", 1e-5 if clip else 1e-6),
                image_grid_pinpoints=tuple(tuple(p) for p in d.get("image_grid_pinpoints", ())),
                anyres_max=d.get("anyres_max", 9) if not clip else None,
                llava_family={"llava": "llava", "llava_next": "next"}[d["family"]] if clip else "onevision",
                v_feature_layer=d.get("vision_feature_layer", -2 if clip else -1), qkv_bias=bool(t.get("attention_bias", not clip)))
        return VLMConfig(
            vocab_size=t["vocab_size"], hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"],
            num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
            num_key_value_heads=t["num_key_value_heads"], rms_norm_eps=t["rms_norm_eps"], rope_theta=t["rope_theta"],
            mrope_section=tuple(t["mrope_section"]), v_depth=v["depth"], v_hidden=v["hidden_size"], v_inter=v["intermediate_size"],
            v_heads=v["num_heads"], v_in_channels=v["in_channels"], v_patch=v["patch_size"], v_merge=v["spatial_merge_size"],
            v_temporal=v["temporal_patch_size"], v_window=v["window_size"], v_fullatt=tuple(v["fullatt_block_indexes"]),
            image_token_id=d["image_token_id"], vision_start_token_id=d["vision_start_token_id"],
            vision_end_token_id=d["vision_end_token_id"], eos_token_id=d["eos_token_id"], pad_token_id=d["pad_token_id"],
            tie_word_embeddings=d.get("tie_word_embeddings", False), v_arch=v.get("arch", "qwen2_5_vl"),
        )

    @staticmethod
    def from_hf_config(c: dict) -> "VLMConfig":
        """config.json of a Qwen2.5-VL / Qwen2-VL checkpoint (flat 4.51-style or nested text_config 5.x-style) or of a LLaVA-OneVision one."""
        t = c.get("text_config", c)
        v = c["vision_config"]
        if c.get("model_type") in ("llava", "llava_next"):
            # LLaVA-1.5 / LLaVA-NeXT: the published config.json files list only what differs from the defaults of the text model's config class
            dflt = {"llama": dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32, rms_norm_eps=1e-6, rope_theta=10000.0),
                    "mistral": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8, rms_norm_eps=1e-6, rope_theta=10000.0)}
            mt = t.get("model_type", "llama")
            if mt not in dflt:
                raise ValueError(f"{c['model_type']}: text model type {mt!r} not built (llama and mistral are)")
            tt = {**dflt[mt], **{k: v_ for k, v_ in t.items() if v_ is not None}}
            if tt.get("attention_bias") or tt.get("mlp_bias"):
                raise ValueError("llava: decoders with attention / MLP biases are not the published LLaVA-1.5 / NeXT configurations")
            if c.get("vision_feature_select_strategy", "default") != "default" or not isinstance(c.get("vision_feature_layer", -2), int) or c.get("projector_hidden_act", "gelu") != "gelu":
                raise ValueError("llava: built for vision_feature_select_strategy 'default', a single vision_feature_layer and the 'gelu' projector")
            if v.get("hidden_act", "quick_gelu") != "quick_gelu":
                raise ValueError("llava: the CLIP tower's MLP activation built here is quick_gelu")
            eos = tt.get("eos_token_id", 2)
            d = {"family": c["model_type"],
                 "text": {"vocab_size": tt.get("vocab_size", c.get("vocab_size", 32064)), "hidden_size": tt["hidden_size"], "intermediate_size": tt["intermediate_size"],
                          "num_hidden_layers": tt["num_hidden_layers"], "num_attention_heads": tt["num_attention_heads"],
                          "num_key_value_heads": tt.get("num_key_value_heads", tt["num_attention_heads"]), "rms_norm_eps": tt["rms_norm_eps"],
                          "rope_theta": float((tt.get("rope_parameters") or {}).get("rope_theta", tt["rope_theta"])), "attention_bias": False},
                 "vision": {"arch": "clip", "depth": v.get("num_hidden_layers", 12), "hidden_size": v.get("hidden_size", 768), "intermediate_size": v.get("intermediate_size", 3072),
                            "num_heads": v.get("num_attention_heads", 12), "in_channels": v.get("num_channels", 3), "patch_size": v.get("patch_size", 32),
                            "image_size": v.get("image_size", 224), "layer_norm_eps": v.get("layer_norm_eps", 1e-5)},
                 "vision_feature_layer": c.get("vision_feature_layer", -2), "image_grid_pinpoints": c.get("image_grid_pinpoints", ()) if c["model_type"] == "llava_next" else (),
                 "image_token_id": c.get("image_token_index", c.get("image_token_id", 32000)), "eos_token_id": eos[0] if isinstance(eos, (list, tuple)) else eos,
                 "pad_token_id": _first_set(c.get("pad_token_id"), tt.get("pad_token_id"), 0), "tie_word_embeddings": False}
            return VLMConfig.from_dict(d)
        if c.get("model_type") == "llava_onevision" or "image_grid_pinpoints" in c:
            asp = str(c.get("vision_aspect_ratio", "anyres_max_9"))
            eos = t.get("eos_token_id", c.get("eos_token_id", 151645))
            d = {"text": {"vocab_size": t.get("vocab_size", 152000), "hidden_size": t["hidden_size"], "intermediate_size": t["intermediate_size"], "num_hidden_layers": t["num_hidden_layers"],
                          "num_attention_heads": t["num_attention_heads"], "num_key_value_heads": t.get("num_key_value_heads", t["num_attention_heads"]),
                          "rms_norm_eps": t.get("rms_norm_eps", 1e-6), "rope_theta": float((t.get("rope_parameters") or {}).get("rope_theta", t.get("rope_theta", 1e6)))},
                 "vision": {"arch": "siglip", "depth": v.get("num_hidden_layers", 26), "hidden_size": v.get("hidden_size", 1152), "intermediate_size": v.get("intermediate_size", 4304),
                            "num_heads": v.get("num_attention_heads", 16), "in_channels": v.get("num_channels", 3), "patch_size": v.get("patch_size", 14),
                            "image_size": v.get("image_size", 384), "layer_norm_eps": v.get("layer_norm_eps", 1e-6)},
                 "image_grid_pinpoints": c["image_grid_pinpoints"], "anyres_max": int(asp.rsplit("_", 1)[-1]) if asp.startswith("anyres_max_") else 9,
                 "image_token_id": c.get("image_token_index", c.get("image_token_id", 151646)), "eos_token_id": eos[0] if isinstance(eos, (list, tuple)) else eos,
                 "pad_token_id": _first_set(t.get("pad_token_id"), c.get("pad_token_id"), 151643), "tie_word_embeddings": bool(c.get("tie_word_embeddings", t.get("tie_word_embeddings", False)))}
            if c.get("vision_feature_layer", -1) != -1 or c.get("vision_feature_select_strategy", "full") != "full":
                raise ValueError("llava_onevision: vision_feature_layer = -1 with the 'full' select strategy is the configuration built here")
            return VLMConfig.from_dict(d)
        rope = t.get("rope_parameters") or t.get("rope_scaling") or c.get("rope_scaling") or {}
        eos = t.get("eos_token_id", c.get("eos_token_id", 151645))
        if "embed_dim" in v:  # Qwen2-VL: ViT width is `embed_dim`, vision `hidden_size` is the merger output
            v = dict(v, hidden_size=v["embed_dim"], intermediate_size=int(v["embed_dim"] * v.get("mlp_ratio", 4)), window_size=0,
                     fullatt_block_indexes=list(range(v["depth"])), arch="qwen2_vl")
        return VLMConfig(
            vocab_size=t["vocab_size"], hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"],
            num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
            num_key_value_heads=t.get("num_key_value_heads", t["num_attention_heads"]), rms_norm_eps=t.get("rms_norm_eps", 1e-6),
            rope_theta=float(rope.get("rope_theta", t.get("rope_theta", 1e6))), mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))),
            v_depth=v["depth"], v_hidden=v["hidden_size"], v_inter=v["intermediate_size"], v_heads=v["num_heads"],
            v_in_channels=v.get("in_channels", v.get("in_chans", 3)), v_patch=v["patch_size"], v_merge=v["spatial_merge_size"],
            v_temporal=v["temporal_patch_size"], v_window=v["window_size"], v_fullatt=tuple(v["fullatt_block_indexes"]),
            image_token_id=c.get("image_token_id", 151655), vision_start_token_id=c.get("vision_start_token_id", 151652),
            vision_end_token_id=c.get("vision_end_token_id", 151653), eos_token_id=eos[0] if isinstance(eos, (list, tuple)) else eos,
            pad_token_id=_first_set(t.get("pad_token_id"), c.get("pad_token_id"), 151643), tie_word_embeddings=c.get("tie_word_embeddings", False),
            v_arch=v.get("arch", "qwen2_5_vl"),
        )

    @staticmethod
    def qwen25vl_3b() -> "VLMConfig":
        """Qwen2.5-VL-3B-Instruct shapes (public config.json)."""
        return VLMConfig(
            vocab_size=151936, hidden_size=2048, intermediate_size=11008, num_hidden_layers=36, num_attention_heads=16,
            num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=1e6, mrope_section=(16, 24, 24), v_depth=32, v_hidden=1280,
            v_inter=3420, v_heads=16, v_in_channels=3, v_patch=14, v_merge=2, v_temporal=2, v_window=112, v_fullatt=(7, 15, 23, 31),
            image_token_id=151655, vision_start_token_id=151652, vision_end_token_id=151653, eos_token_id=151645, pad_token_id=151643,
            tie_word_embeddings=True,
        )

    @staticmethod
    def qwen2vl_2b() -> "VLMConfig":
        """Qwen2-VL-2B-Instruct shapes (public config.json; BASELINE.json config 1)."""
        return VLMConfig(
            vocab_size=151936, hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12,
            num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=1e6, mrope_section=(16, 24, 24), v_depth=32, v_hidden=1280,
            v_inter=5120, v_heads=16, v_in_channels=3, v_patch=14, v_merge=2, v_temporal=2, v_window=0, v_fullatt=tuple(range(32)),
            image_token_id=151655, vision_start_token_id=151652, vision_end_token_id=151653, eos_token_id=151645, pad_token_id=151643,
            tie_word_embeddings=True, v_arch="qwen2_vl",
        )

    @staticmethod
    def qwen25vl_7b() -> "VLMConfig":
        c = VLMConfig.qwen25vl_3b()
        c.vocab_size, c.hidden_size, c.intermediate_size, c.num_hidden_layers = 152064, 3584, 18944, 28
        c.num_attention_heads, c.num_key_value_heads, c.tie_word_embeddings = 28, 4, False
        return c


def _llava_ov_7b() -> VLMConfig:
    """LLaVA-OneVision-Qwen2-7B-SI shapes (public config.json; BASELINE.json config 5): SigLIP-so400m/14-384 tower as LLaVA-OV ships it (26 layers,
    width 1152, 16 heads of 72, MLP 4304), Qwen2-7B decoder (width 3584, 28 layers, 28/4 heads, MLP 18944, untied head), any-resolution pinpoints
    384 x {1..6} by 384 x {1..6}, anyres_max_9."""
    pins = tuple((384 * i, 384 * j) for i in range(1, 7) for j in range(1, 7))
    return VLMConfig.from_dict({
        "text": {"vocab_size": 152064, "hidden_size": 3584, "intermediate_size": 18944, "num_hidden_layers": 28, "num_attention_heads": 28, "num_key_value_heads": 4,
                 "rms_norm_eps": 1e-6, "rope_theta": 1e6},
        "vision": {"arch": "siglip", "depth": 26, "hidden_size": 1152, "intermediate_size": 4304, "num_heads": 16, "in_channels": 3, "patch_size": 14, "image_size": 384,
                   "layer_norm_eps": 1e-6},
        "image_grid_pinpoints": pins, "anyres_max": 9, "image_token_id": 151646, "eos_token_id": 151645, "pad_token_id": 151643, "tie_word_embeddings": False})


VLMConfig.llava_ov_7b = staticmethod(_llava_ov_7b)


def _llava15_7b() -> VLMConfig:
    """llava-1.5-7b-hf shapes (public config.json): CLIP ViT-L/14-336 tower, vicuna-7b (LLaMA: 32 layers, width 4096, 32 heads, MLP 11008, vocabulary 32064)."""
    return VLMConfig.from_hf_config({
        "model_type": "llava", "image_token_index": 32000, "pad_token_id": 32001, "projector_hidden_act": "gelu", "vision_feature_layer": -2, "vision_feature_select_strategy": "default",
        "text_config": {"model_type": "llama", "max_position_embeddings": 4096, "rms_norm_eps": 1e-05, "vocab_size": 32064},
        "vision_config": {"model_type": "clip_vision_model", "hidden_size": 1024, "image_size": 336, "intermediate_size": 4096, "num_attention_heads": 16, "num_hidden_layers": 24, "patch_size": 14}})


def _llava_next_7b() -> VLMConfig:
    """llava-v1.6-mistral-7b-hf shapes (public config.json): the same CLIP tower, Mistral-7B (GQA 32 / 8, MLP 14336), five any-resolution pinpoints."""
    return VLMConfig.from_hf_config({
        "model_type": "llava_next", "image_token_index": 32000, "projector_hidden_act": "gelu", "vision_feature_layer": -2, "vision_feature_select_strategy": "default",
        "image_grid_pinpoints": [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]],
        "text_config": {"model_type": "mistral", "intermediate_size": 14336, "max_position_embeddings": 32768, "num_key_value_heads": 8, "rms_norm_eps": 1e-05, "rope_theta": 1000000.0,
                        "sliding_window": None, "vocab_size": 32064},
        "vision_config": {"model_type": "clip_vision_model", "hidden_size": 1024, "image_size": 336, "intermediate_size": 4096, "num_attention_heads": 16, "num_hidden_layers": 24, "patch_size": 14}})


VLMConfig.llava15_7b = staticmethod(_llava15_7b)
VLMConfig.llava_next_7b = staticmethod(_llava_next_7b)


@dataclass
class _Slot:
    name: str
    shape: tuple
    offset: int
    decay: bool
    gemm: bool  # has a transposed shadow
    t_offset: int = -1


class ParamStore:
    """Flat parameter buffers + named views in the engine's fused layout."""

    def __init__(self, cfg: VLMConfig, device, trainable: bool, with_transposes: bool | None = None, with_decode_pack: bool | None = None, decode_weights: str | None = None):
        self.cfg, self.device, self.trainable = cfg, torch.device(device), trainable
        self.with_transposes = trainable if with_transposes is None else with_transposes
        self.with_decode_pack = trainable if with_decode_pack is None else with_decode_pack
        c = cfg
        H, I, D = c.hidden_size, c.intermediate_size, c.head_dim
        vh, vip = c.v_hidden, c.v_inter_pad
        specs = []  # (name, shape, decay, gemm)

        def add(name, shape, decay, gemm):
            specs.append((name, tuple(shape), decay, gemm))

        q2 = c.v_arch == "qwen2_vl"
        assert c.v_arch in ("qwen2_5_vl", "qwen2_vl", "siglip", "clip"), c.v_arch
        self.extra = {}      # checkpoint tensors this engine does not use (post_layernorm, SigLIP pooling head, CLIP blocks past the feature layer): kept as loaded, written back on save
        if c.is_llava:
            dp, nh = c.v_head_pad, c.v_heads
            add("visual.patch_embed", (vh, c.patch_dim_pad), True, True)
            if c.v_arch == "clip":          # bias-free patch conv, class token, pre-LayerNorm (transformers models/clip/modeling_clip.py:141-200)
                add("visual.cls", (vh,), True, False)
                add("visual.pre_ln", (vh,), False, False)
                add("visual.pre_ln.b", (vh,), False, False)
            else:
                add("visual.patch_embed.b", (vh,), False, False)
            add("visual.pos", (c.v_seq, vh), True, False)
            for i in range(c.v_run_depth):
                b = f"visual.blocks.{i}."
                add(b + "norm1", (vh,), False, False)
                add(b + "norm1.b", (vh,), False, False)
                add(b + "qkv.w", (3 * nh * dp, vh), True, True)
                add(b + "qkv.b", (3 * nh * dp,), False, False)
                add(b + "proj.w", (vh, nh * dp), True, True)
                add(b + "proj.b", (vh,), False, False)
                add(b + "norm2", (vh,), False, False)
                add(b + "norm2.b", (vh,), False, False)
                add(b + "fc1.w", (c.v_inter, vh), True, True)
                add(b + "fc1.b", (c.v_inter,), False, False)
                add(b + "fc2.w", (vh, c.v_inter), True, True)
                add(b + "fc2.b", (vh,), False, False)
            add("visual.merger.fc1.w", (H, vh), True, True)
            add("visual.merger.fc1.b", (H,), False, False)
            add("visual.merger.fc2.w", (H, H), True, True)
            add("visual.merger.fc2.b", (H,), False, False)
            if c.llava_family != "llava":
                add("visual.newline", (H,), True, False)
        else:
            add("visual.patch_embed", (vh, c.patch_dim), True, True)
        # Weight-decay group = transformers.Trainer.get_decay_parameter_names (both trainers of the reference create their optimizer through it): parameters of
        # nn.LayerNorm modules and names matching bias / layernorm / rmsnorm / .norm. / _norm. are exempt, everything else decays -- including the RMSNorm gains of
        # the Qwen2.5-VL vision tower (`norm1`, `norm2`, `merger.ln_q`: a Qwen2RMSNorm under names the patterns miss), `image_newline` and CLIP's
        # `class_embedding`.  Pinned per family by tests/golden/sft_freeze.json (decay_parameters).
        rms_gain_decays = not q2
        for i in range(0 if c.is_llava else c.v_depth):
            b = f"visual.blocks.{i}."
            add(b + "norm1", (vh,), rms_gain_decays, False)
            add(b + "qkv.w", (3 * vh, vh), True, True)
            add(b + "qkv.b", (3 * vh,), False, False)
            add(b + "proj.w", (vh, vh), True, True)
            add(b + "proj.b", (vh,), False, False)
            add(b + "norm2", (vh,), rms_gain_decays, False)
            if q2:
                add(b + "norm1.b", (vh,), False, False)
                add(b + "norm2.b", (vh,), False, False)
                add(b + "fc1.w", (c.v_inter, vh), True, True)
                add(b + "fc1.b", (c.v_inter,), False, False)
                add(b + "fc2.w", (vh, c.v_inter), True, True)
                add(b + "fc2.b", (vh,), False, False)
                continue
            add(b + "gu.w", (2 * vip, vh), True, True)
            add(b + "gu.b", (2 * vip,), False, False)
            add(b + "down.w", (vh, vip), True, True)
            add(b + "down.b", (vh,), False, False)
        mu = c.v_merge**2
        if not c.is_llava:
            add("visual.merger.ln_q", (vh,), rms_gain_decays, False)
            if q2:
                add("visual.merger.ln_q.b", (vh,), False, False)
            add("visual.merger.fc1.w", (vh * mu, vh * mu), True, True)
            add("visual.merger.fc1.b", (vh * mu,), False, False)
            add("visual.merger.fc2.w", (H, vh * mu), True, True)
            add("visual.merger.fc2.b", (H,), False, False)
        add("embed", (c.vocab_size, H), True, True)
        for i in range(c.num_hidden_layers):
            b = f"layers.{i}."
            add(b + "ln1", (H,), False, False)
            add(b + "qkv.w", (c.qkv_width, H), True, True)
            add(b + "qkv.b", (c.qkv_width,), False, False)
            add(b + "o.w", (H, c.num_attention_heads * D), True, True)
            add(b + "ln2", (H,), False, False)
            add(b + "gu.w", (2 * I, H), True, True)
            add(b + "down.w", (H, I), True, True)
        add("norm", (H,), False, False)
        if not c.tie_word_embeddings:
            add("lm_head", (c.vocab_size, H), True, True)

        # decayed tensors first, then the no-decay group (two AdamW launches)
        order = [s for s in specs if s[2]] + [s for s in specs if not s[2]]
        self.slots: dict[str, _Slot] = {}
        off = toff = 0
        for name, shape, decay, gemm in order:
            n = int(np.prod(shape))
            s = _Slot(name, shape, off, decay, gemm)
            off += _rup(n, 64)
            if gemm and self.with_transposes:
                s.t_offset = toff
                toff += _rup(n, 64)
            self.slots[name] = s
        self.n_total = off
        self.n_decay = sum(_rup(int(np.prod(s[1])), 64) for s in order if s[2])
        self.flat = torch.zeros(self.n_total, dtype=BF16, device=self.device)
        self.flat_t = torch.zeros(toff, dtype=BF16, device=self.device) if self.with_transposes else None
        if trainable:
            self.master = torch.zeros(self.n_total, dtype=F32, device=self.device)
            self.m = torch.zeros(self.n_total, dtype=F32, device=self.device)
            self.v = torch.zeros(self.n_total, dtype=F32, device=self.device)
            self.grad = torch.zeros(self.n_total, dtype=F32, device=self.device)
        self._views, self._tviews, self._gviews = {}, {}, {}
        # decode-packed (MFMA fragment order) shadows of the weights the rollout streams every step
        self._pk = {}
        # opt-in FP8 weight stream of the rollout (BASELINE config 5; IADR1_DECODE_WEIGHTS=fp8 or ParamStore(..., decode_weights="fp8")): the three big
        # streams of a decode step -- gate|up, down, lm_head -- are kept as OCP e4m3 + one fp32 scale per output row instead of bf16 packs; q|k|v and o
        # (latency-bound, 10 MB) stay bf16.  Training passes never read these copies.
        self.decode_fp8 = (decode_weights or os.environ.get("IADR1_DECODE_WEIGHTS", "bf16")) == "fp8"
        assert not self.decode_fp8 or (c.hidden_size % 64 == 0 and c.intermediate_size % 64 == 0 and c.vocab_size % 64 == 0), "fp8 decode weights need widths that are multiples of 64"
        self._pk8 = {}
        if self.with_decode_pack:
            names = [f"layers.{i}.{k}" for i in range(c.num_hidden_layers) for k in ("qkv.w", "o.w", "gu.w", "down.w")] + [self.lm_head_name()]
            if self.decode_fp8:
                names8 = [n for n in names if n.endswith(("gu.w", "down.w")) or n == self.lm_head_name()]
                names = [n for n in names if n not in names8]
                tot8 = sum(_rup(int(np.prod(self.slots[n].shape)), 64) for n in names8)
                self.flat_pk8 = torch.zeros(tot8, dtype=torch.uint8, device=self.device)
                o8 = 0
                for n in names8:
                    k = int(np.prod(self.slots[n].shape))
                    self._pk8[n] = (self.flat_pk8[o8: o8 + k], torch.zeros(self.slots[n].shape[0], dtype=F32, device=self.device))
                    o8 += _rup(k, 64)
            tot = sum(_rup(int(np.prod(self.slots[n].shape)), 64) for n in names)
            self.flat_pk = torch.zeros(tot, dtype=BF16, device=self.device)
            o = 0
            for n in names:
                k = int(np.prod(self.slots[n].shape))
                self._pk[n] = self.flat_pk[o: o + k]
                o += _rup(k, 64)
            self._pkb = {f"layers.{i}.qkv.w": torch.zeros(c.qkv_width, dtype=BF16, device=self.device) for i in range(c.num_hidden_layers)}

    def resident_bytes(self) -> int:
        """Bytes this store keeps in HBM for its whole life: bf16 parameters, transposed and decode-packed shadows, fp32 master / Adam moments / gradient."""
        n = 0
        for name in ("flat", "flat_t", "master", "m", "v", "grad", "flat_pk", "flat_pk8"):
            t = getattr(self, name, None)
            if t is not None:
                n += t.numel() * t.element_size()
        return n

    # ---- views -------------------------------------------------------------------------------------------
    def w(self, name: str) -> torch.Tensor:
        v = self._views.get(name)
        if v is None:
            s = self.slots[name]
            v = self.flat[s.offset: s.offset + int(np.prod(s.shape))].view(*s.shape)
            self._views[name] = v
        return v

    def wT(self, name: str) -> torch.Tensor:
        """Transposed shadow [K, N] of GEMM weight `name` ([N, K])."""
        v = self._tviews.get(name)
        if v is None:
            s = self.slots[name]
            assert s.t_offset >= 0, f"{name}: no transposed shadow"
            v = self.flat_t[s.t_offset: s.t_offset + int(np.prod(s.shape))].view(s.shape[1], s.shape[0])
            self._tviews[name] = v
        return v

    def g(self, name: str) -> torch.Tensor:
        v = self._gviews.get(name)
        if v is None:
            s = self.slots[name]
            v = self.grad[s.offset: s.offset + int(np.prod(s.shape))].view(*s.shape)
            self._gviews[name] = v
        return v

    def lm_head_name(self):
        return "embed" if self.cfg.tie_word_embeddings else "lm_head"

    def wpk(self, name: str) -> torch.Tensor:
        """Decode-packed shadow (flat) of GEMM weight `name`; an (FP8 pack, row scales) pair for the streams kept in FP8 (ops.gemm_skinny takes either)."""
        return self._pk8[name] if name in self._pk8 else self._pk[name]

    def wpk_bias(self, name: str) -> torch.Tensor:
        """Bias permuted like the rope-ordered decode pack of q|k|v weight `name`."""
        return self._pkb[name]

    @property
    def qkv_rope_packed(self) -> bool:
        return self.cfg.head_dim == 128

    def refresh_decode_pack(self):
        fuse = self.cfg.intermediate_size % 64 == 0
        c = self.cfg
        for name, (dst8, sc8) in self._pk8.items():
            ops.pack_weight_fp8(self.w(name), out=dst8, out_scale=sc8, gateup=name.endswith(".gu.w") and self.cfg.intermediate_size % 64 == 0)
        for name, dst in self._pk.items():
            if self.qkv_rope_packed and name.endswith(".qkv.w"):
                # rotary partners share a tile: the decode q|k|v GEMM applies rope and appends K/V in its epilogue
                ops.pack_qkv_rope(self.w(name), self.w(name[:-1] + "b"), c.num_attention_heads, c.num_key_value_heads, c.head_dim, out=dst, out_bias=self._pkb[name])
            elif fuse and name.endswith(".gu.w"):
                # gate/up tiles interleaved: the decode GEMM applies SwiGLU in its epilogue
                ops.pack_gateup(self.w(name), out=dst)
            else:
                ops.pack_weight(self.w(name), out=dst)

    def refresh_transposes(self):
        if not self.with_transposes:
            return
        for name, s in self.slots.items():
            if s.t_offset >= 0:
                ops.transpose(self.w(name), out=self.wT(name))

    def sync_master_from_bf16(self):
        from . import hip
        hip.call("cast_bf16_to_f32", self.flat, self.master, self.n_total)

    # ---- checkpoint-name <-> fused layout -----------------------------------------------------------------------------
    def _assign(self, name, tensor):
        dst = self.w(name)
        src = torch.as_tensor(tensor)
        assert tuple(src.shape) == tuple(dst.shape), (name, tuple(src.shape), tuple(dst.shape))
        dst.copy_(src.to(BF16))

    # names of the LLaVA-OneVision checkpoints as transformers 4.51.3 (the reference's pin) writes them; 5.x prefixes are accepted on load
    @staticmethod
    def _llava_canonical(name: str) -> str:
        if name.startswith("model.vision_tower."):
            return "vision_tower.vision_model." + name[len("model.vision_tower."):]
        if name.startswith("model.multi_modal_projector."):
            return name[len("model."):]
        if name == "model.image_newline":
            return "image_newline"
        if name.startswith("model.language_model."):
            return "language_model.model." + name[len("model.language_model."):]
        if name == "lm_head.weight":
            return "language_model.lm_head.weight"
        return name

    def _load_named_llava(self, sd: dict):
        c = self.cfg
        sd = {self._llava_canonical(k): v for k, v in sd.items()}
        t = lambda k: torch.as_tensor(sd.pop(k)).float()
        vh, nh, d, dp = c.v_hidden, c.v_heads, c.v_hidden // c.v_heads, c.v_head_pad
        pre = "vision_tower.vision_model."
        pe = torch.zeros(vh, c.patch_dim_pad)
        pe[:, : c.patch_dim] = t(pre + "embeddings.patch_embedding.weight").reshape(vh, -1)
        self._assign("visual.patch_embed", pe)
        if c.v_arch == "clip":
            self._assign("visual.cls", t(pre + "embeddings.class_embedding"))
            self._assign("visual.pre_ln", t(pre + "pre_layrnorm.weight"))
            self._assign("visual.pre_ln.b", t(pre + "pre_layrnorm.bias"))
        else:
            self._assign("visual.patch_embed.b", t(pre + "embeddings.patch_embedding.bias"))
        self._assign("visual.pos", t(pre + "embeddings.position_embedding.weight"))

        def pad_rows(w):          # [nh*d, K] -> [nh*dp, K], zero rows for the padded dims of every head
            out = torch.zeros(nh, dp, *w.shape[1:])
            out[:, :d] = w.reshape(nh, d, *w.shape[1:])
            return out.reshape(nh * dp, *w.shape[1:])

        for i in range(c.v_run_depth):      # (the blocks past the feature layer stay in `extra`)
            s_, b = f"{pre}encoder.layers.{i}.", f"visual.blocks.{i}."
            self._assign(b + "norm1", t(s_ + "layer_norm1.weight"))
            self._assign(b + "norm1.b", t(s_ + "layer_norm1.bias"))
            self._assign(b + "norm2", t(s_ + "layer_norm2.weight"))
            self._assign(b + "norm2.b", t(s_ + "layer_norm2.bias"))
            self._assign(b + "qkv.w", torch.cat([pad_rows(t(s_ + f"self_attn.{z}_proj.weight")) for z in "qkv"], 0))
            self._assign(b + "qkv.b", torch.cat([pad_rows(t(s_ + f"self_attn.{z}_proj.bias")) for z in "qkv"], 0))
            self._assign(b + "proj.w", pad_rows(t(s_ + "self_attn.out_proj.weight").t().contiguous()).t().contiguous())
            self._assign(b + "proj.b", t(s_ + "self_attn.out_proj.bias"))
            self._assign(b + "fc1.w", t(s_ + "mlp.fc1.weight"))
            self._assign(b + "fc1.b", t(s_ + "mlp.fc1.bias"))
            self._assign(b + "fc2.w", t(s_ + "mlp.fc2.weight"))
            self._assign(b + "fc2.b", t(s_ + "mlp.fc2.bias"))
        self._assign("visual.merger.fc1.w", t("multi_modal_projector.linear_1.weight"))
        self._assign("visual.merger.fc1.b", t("multi_modal_projector.linear_1.bias"))
        self._assign("visual.merger.fc2.w", t("multi_modal_projector.linear_2.weight"))
        self._assign("visual.merger.fc2.b", t("multi_modal_projector.linear_2.bias"))
        if c.llava_family != "llava":
            self._assign("visual.newline", t("image_newline"))
        lm = "language_model.model."
        self._assign("embed", t(lm + "embed_tokens.weight"))
        slots_t, Dr, Dk = torch.from_numpy(c.head_slots), c.head_dim_real, c.head_dim

        def pad_heads(w):         # [heads * Dr, ...] -> [heads * Dk, ...]: every head's dims scattered to their slots of the padded head (VLMConfig.head_dim)
            if Dr == Dk:
                return w
            nh_ = w.shape[0] // Dr
            out = torch.zeros(nh_, Dk, *w.shape[1:])
            out[:, slots_t] = w.reshape(nh_, Dr, *w.shape[1:])
            return out.reshape(nh_ * Dk, *w.shape[1:])

        for i in range(c.num_hidden_layers):
            s_, b = f"{lm}layers.{i}.", f"layers.{i}."
            self._assign(b + "ln1", t(s_ + "input_layernorm.weight"))
            self._assign(b + "ln2", t(s_ + "post_attention_layernorm.weight"))
            self._assign(b + "qkv.w", torch.cat([pad_heads(t(s_ + f"self_attn.{z}_proj.weight")) for z in "qkv"], 0))
            if c.qkv_bias:
                self._assign(b + "qkv.b", torch.cat([pad_heads(t(s_ + f"self_attn.{z}_proj.bias")) for z in "qkv"], 0))
            else:
                self.w(b + "qkv.b").zero_()
            self._assign(b + "o.w", pad_heads(t(s_ + "self_attn.o_proj.weight").t().contiguous()).t().contiguous())
            self._assign(b + "gu.w", torch.cat([t(s_ + "mlp.gate_proj.weight"), t(s_ + "mlp.up_proj.weight")], 0))
            self._assign(b + "down.w", t(s_ + "mlp.down_proj.weight"))
        self._assign("norm", t(lm + "norm.weight"))
        if not c.tie_word_embeddings:
            self._assign("lm_head", t("language_model.lm_head.weight"))
        else:
            sd.pop("language_model.lm_head.weight", None)
        self.extra = {k: torch.as_tensor(v).clone() for k, v in sd.items()}      # post_layernorm, pooling head, ...: not on this path
        self.finalize()

    def _export_named_llava(self, source: str = "param") -> dict:
        c = self.cfg
        get = (lambda n: self.w(n).float().cpu()) if source == "param" else (lambda n: self.g(n).float().cpu())
        vh, nh, d, dp = c.v_hidden, c.v_heads, c.v_hidden // c.v_heads, c.v_head_pad
        unpad = lambda w: w.reshape(nh, dp, *w.shape[1:])[:, :d].reshape(nh * d, *w.shape[1:]).clone()
        pre, out = "vision_tower.vision_model.", {}
        out[pre + "embeddings.patch_embedding.weight"] = get("visual.patch_embed")[:, : c.patch_dim].reshape(vh, c.v_in_channels, c.v_patch, c.v_patch).clone()
        if c.v_arch == "clip":
            out[pre + "embeddings.class_embedding"] = get("visual.cls")
            out[pre + "pre_layrnorm.weight"], out[pre + "pre_layrnorm.bias"] = get("visual.pre_ln"), get("visual.pre_ln.b")
        else:
            out[pre + "embeddings.patch_embedding.bias"] = get("visual.patch_embed.b")
        out[pre + "embeddings.position_embedding.weight"] = get("visual.pos")
        for i in range(c.v_run_depth):
            s_, b = f"{pre}encoder.layers.{i}.", f"visual.blocks.{i}."
            out[s_ + "layer_norm1.weight"], out[s_ + "layer_norm1.bias"] = get(b + "norm1"), get(b + "norm1.b")
            out[s_ + "layer_norm2.weight"], out[s_ + "layer_norm2.bias"] = get(b + "norm2"), get(b + "norm2.b")
            qw, qb = get(b + "qkv.w"), get(b + "qkv.b")
            for j, z in enumerate("qkv"):
                out[s_ + f"self_attn.{z}_proj.weight"] = unpad(qw[j * nh * dp: (j + 1) * nh * dp])
                out[s_ + f"self_attn.{z}_proj.bias"] = unpad(qb[j * nh * dp: (j + 1) * nh * dp])
            out[s_ + "self_attn.out_proj.weight"] = unpad(get(b + "proj.w").t().contiguous()).t().contiguous()
            out[s_ + "self_attn.out_proj.bias"] = get(b + "proj.b")
            out[s_ + "mlp.fc1.weight"], out[s_ + "mlp.fc1.bias"] = get(b + "fc1.w"), get(b + "fc1.b")
            out[s_ + "mlp.fc2.weight"], out[s_ + "mlp.fc2.bias"] = get(b + "fc2.w"), get(b + "fc2.b")
        out["multi_modal_projector.linear_1.weight"], out["multi_modal_projector.linear_1.bias"] = get("visual.merger.fc1.w"), get("visual.merger.fc1.b")
        out["multi_modal_projector.linear_2.weight"], out["multi_modal_projector.linear_2.bias"] = get("visual.merger.fc2.w"), get("visual.merger.fc2.b")
        if c.llava_family != "llava":
            out["image_newline"] = get("visual.newline")
        lm = "language_model.model."
        out[lm + "embed_tokens.weight"] = get("embed")
        hq, hk = c.num_attention_heads * c.head_dim, c.num_key_value_heads * c.head_dim
        slots_t, Dr, Dk = torch.from_numpy(c.head_slots), c.head_dim_real, c.head_dim
        unpad_heads = lambda w: w.clone() if Dr == Dk else w.reshape(w.shape[0] // Dk, Dk, *w.shape[1:])[:, slots_t].reshape(w.shape[0] // Dk * Dr, *w.shape[1:]).clone()
        for i in range(c.num_hidden_layers):
            s_, b = f"{lm}layers.{i}.", f"layers.{i}."
            out[s_ + "input_layernorm.weight"], out[s_ + "post_attention_layernorm.weight"] = get(b + "ln1"), get(b + "ln2")
            qw, qb = get(b + "qkv.w"), get(b + "qkv.b")
            out[s_ + "self_attn.q_proj.weight"], out[s_ + "self_attn.k_proj.weight"], out[s_ + "self_attn.v_proj.weight"] = unpad_heads(qw[:hq]), unpad_heads(qw[hq: hq + hk]), unpad_heads(qw[hq + hk:])
            if c.qkv_bias:
                out[s_ + "self_attn.q_proj.bias"], out[s_ + "self_attn.k_proj.bias"], out[s_ + "self_attn.v_proj.bias"] = unpad_heads(qb[:hq]), unpad_heads(qb[hq: hq + hk]), unpad_heads(qb[hq + hk:])
            out[s_ + "self_attn.o_proj.weight"] = unpad_heads(get(b + "o.w").t().contiguous()).t().contiguous()
            gu = get(b + "gu.w")
            out[s_ + "mlp.gate_proj.weight"], out[s_ + "mlp.up_proj.weight"] = gu[: c.intermediate_size].clone(), gu[c.intermediate_size:].clone()
            out[s_ + "mlp.down_proj.weight"] = get(b + "down.w")
        out[lm + "norm.weight"] = get("norm")
        out["language_model.lm_head.weight"] = get(self.lm_head_name())
        if source == "param":
            out.update({k: v.float().cpu() for k, v in self.extra.items()})
        return out

    def load_named(self, sd: dict):
        """`sd`: checkpoint-name -> array/tensor (HF Qwen2.5-VL names)."""
        c = self.cfg
        if c.is_llava:
            return self._load_named_llava(sd)
        assert c.head_dim == c.head_dim_real, "Qwen-VL checkpoints have 128-wide decoder heads"
        t = lambda k: torch.as_tensor(sd[k]).float()
        vi, vip = c.v_inter, c.v_inter_pad
        self._assign("visual.patch_embed", t("visual.patch_embed.proj.weight").reshape(c.v_hidden, -1))
        for i in range(c.v_depth):
            s, b = f"visual.blocks.{i}.", f"visual.blocks.{i}."
            self._assign(b + "norm1", t(s + "norm1.weight"))
            self._assign(b + "norm2", t(s + "norm2.weight"))
            self._assign(b + "qkv.w", t(s + "attn.qkv.weight"))
            self._assign(b + "qkv.b", t(s + "attn.qkv.bias"))
            self._assign(b + "proj.w", t(s + "attn.proj.weight"))
            self._assign(b + "proj.b", t(s + "attn.proj.bias"))
            if c.v_arch == "qwen2_vl":
                self._assign(b + "norm1.b", t(s + "norm1.bias"))
                self._assign(b + "norm2.b", t(s + "norm2.bias"))
                self._assign(b + "fc1.w", t(s + "mlp.fc1.weight"))
                self._assign(b + "fc1.b", t(s + "mlp.fc1.bias"))
                self._assign(b + "fc2.w", t(s + "mlp.fc2.weight"))
                self._assign(b + "fc2.b", t(s + "mlp.fc2.bias"))
                continue
            gu = torch.zeros(2 * vip, c.v_hidden)
            gu[:vi] = t(s + "mlp.gate_proj.weight")
            gu[vip: vip + vi] = t(s + "mlp.up_proj.weight")
            gb = torch.zeros(2 * vip)
            gb[:vi] = t(s + "mlp.gate_proj.bias")
            gb[vip: vip + vi] = t(s + "mlp.up_proj.bias")
            dn = torch.zeros(c.v_hidden, vip)
            dn[:, :vi] = t(s + "mlp.down_proj.weight")
            self._assign(b + "gu.w", gu)
            self._assign(b + "gu.b", gb)
            self._assign(b + "down.w", dn)
            self._assign(b + "down.b", t(s + "mlp.down_proj.bias"))
        self._assign("visual.merger.ln_q", t("visual.merger.ln_q.weight"))
        if c.v_arch == "qwen2_vl":
            self._assign("visual.merger.ln_q.b", t("visual.merger.ln_q.bias"))
        self._assign("visual.merger.fc1.w", t("visual.merger.mlp.0.weight"))
        self._assign("visual.merger.fc1.b", t("visual.merger.mlp.0.bias"))
        self._assign("visual.merger.fc2.w", t("visual.merger.mlp.2.weight"))
        self._assign("visual.merger.fc2.b", t("visual.merger.mlp.2.bias"))
        self._assign("embed", t("model.embed_tokens.weight"))
        for i in range(c.num_hidden_layers):
            s, b = f"model.layers.{i}.", f"layers.{i}."
            self._assign(b + "ln1", t(s + "input_layernorm.weight"))
            self._assign(b + "ln2", t(s + "post_attention_layernorm.weight"))
            self._assign(b + "qkv.w", torch.cat([t(s + "self_attn.q_proj.weight"), t(s + "self_attn.k_proj.weight"), t(s + "self_attn.v_proj.weight")], 0))
            self._assign(b + "qkv.b", torch.cat([t(s + "self_attn.q_proj.bias"), t(s + "self_attn.k_proj.bias"), t(s + "self_attn.v_proj.bias")], 0))
            self._assign(b + "o.w", t(s + "self_attn.o_proj.weight"))
            self._assign(b + "gu.w", torch.cat([t(s + "mlp.gate_proj.weight"), t(s + "mlp.up_proj.weight")], 0))
            self._assign(b + "down.w", t(s + "mlp.down_proj.weight"))
        self._assign("norm", t("model.norm.weight"))
        if not c.tie_word_embeddings:
            self._assign("lm_head", t("lm_head.weight"))
        self.finalize()

    def finalize(self):
        if self.trainable:
            self.sync_master_from_bf16()
        self.refresh_shadows()

    def optimizer_segments(self, frozen=None):
        """[(lo, hi, decays)]: the contiguous ranges of the flat parameter buffer an optimizer step visits -- everything (the decayed group, then the rest) when
        nothing is frozen; with `frozen(name) -> bool` the runs of adjacent trainable tensors inside each group (frozen tensors take part in neither the update
        nor the weight decay, like parameters without requires_grad under an HF optimizer)."""
        if frozen is None:
            return [(lo, hi, d) for lo, hi, d in ((0, self.n_decay, True), (self.n_decay, self.n_total, False)) if hi > lo]
        segs = []
        for s in sorted(self.slots.values(), key=lambda s: s.offset):
            if frozen(s.name):
                continue
            lo, hi = s.offset, s.offset + _rup(int(np.prod(s.shape)), 64)
            if segs and segs[-1][1] == lo and segs[-1][2] == s.decay:
                segs[-1] = (segs[-1][0], hi, s.decay)
            else:
                segs.append((lo, hi, s.decay))
        return segs

    def frozen_ranges(self, frozen):
        """[(lo, hi)] of the flat buffer held by frozen tensors (merged)."""
        out = []
        for s in sorted(self.slots.values(), key=lambda s: s.offset):
            if frozen(s.name):
                lo, hi = s.offset, s.offset + _rup(int(np.prod(s.shape)), 64)
                if out and out[-1][1] == lo:
                    out[-1] = (out[-1][0], hi)
                else:
                    out.append((lo, hi))
        return out

    def refresh_shadows(self):
        """After any change of the bf16 parameters (load / optimizer step): transposed + decode-packed copies.
        For a trainable store on the GPU the ~450 small launches (one per tensor) are captured into a hipGraph at the first call and replayed: every
        pointer is fixed for the life of the store, and the host time of launching them one by one (about 20 ms for the 3B model) is what kept the
        optimizer tail from overlapping with the next step (sc_grpo.optimizer_step(overlap=True))."""
        if not (self.trainable and self.device.type == "cuda"):
            self.refresh_transposes()
            self.refresh_decode_pack()
            return
        if getattr(self, "_shadow_graph", None) is None:
            self.refresh_transposes()          # eager once: first-use initialisation (function attributes, lazily built buffers) must not happen under capture
            self.refresh_decode_pack()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.refresh_transposes()
                self.refresh_decode_pack()
            self._shadow_graph = g
        self._shadow_graph.replay()

    def export_named(self, source: str = "param") -> dict:
        """Inverse of load_named (bf16 params, or fp32 `grad` views for tests): checkpoint-name -> CPU tensor."""
        c = self.cfg
        if c.is_llava:
            return self._export_named_llava(source)
        get = (lambda n: self.w(n).float().cpu()) if source == "param" else (lambda n: self.g(n).float().cpu())
        vi, vip = c.v_inter, c.v_inter_pad
        out = {}
        out["visual.patch_embed.proj.weight"] = get("visual.patch_embed").view(c.v_hidden, c.v_in_channels, c.v_temporal, c.v_patch, c.v_patch)
        for i in range(c.v_depth):
            s, b = f"visual.blocks.{i}.", f"visual.blocks.{i}."
            out[s + "norm1.weight"], out[s + "norm2.weight"] = get(b + "norm1"), get(b + "norm2")
            out[s + "attn.qkv.weight"], out[s + "attn.qkv.bias"] = get(b + "qkv.w"), get(b + "qkv.b")
            out[s + "attn.proj.weight"], out[s + "attn.proj.bias"] = get(b + "proj.w"), get(b + "proj.b")
            if c.v_arch == "qwen2_vl":
                out[s + "norm1.bias"], out[s + "norm2.bias"] = get(b + "norm1.b"), get(b + "norm2.b")
                out[s + "mlp.fc1.weight"], out[s + "mlp.fc1.bias"] = get(b + "fc1.w"), get(b + "fc1.b")
                out[s + "mlp.fc2.weight"], out[s + "mlp.fc2.bias"] = get(b + "fc2.w"), get(b + "fc2.b")
                continue
            gu, gb, dn = get(b + "gu.w"), get(b + "gu.b"), get(b + "down.w")
            out[s + "mlp.gate_proj.weight"], out[s + "mlp.up_proj.weight"] = gu[:vi].clone(), gu[vip: vip + vi].clone()
            out[s + "mlp.gate_proj.bias"], out[s + "mlp.up_proj.bias"] = gb[:vi].clone(), gb[vip: vip + vi].clone()
            out[s + "mlp.down_proj.weight"], out[s + "mlp.down_proj.bias"] = dn[:, :vi].clone(), get(b + "down.b")
        out["visual.merger.ln_q.weight"] = get("visual.merger.ln_q")
        if c.v_arch == "qwen2_vl":
            out["visual.merger.ln_q.bias"] = get("visual.merger.ln_q.b")
        out["visual.merger.mlp.0.weight"], out["visual.merger.mlp.0.bias"] = get("visual.merger.fc1.w"), get("visual.merger.fc1.b")
        out["visual.merger.mlp.2.weight"], out["visual.merger.mlp.2.bias"] = get("visual.merger.fc2.w"), get("visual.merger.fc2.b")
        out["model.embed_tokens.weight"] = get("embed")
        hq, hk = c.num_attention_heads * c.head_dim, c.num_key_value_heads * c.head_dim
        for i in range(c.num_hidden_layers):
            s, b = f"model.layers.{i}.", f"layers.{i}."
            out[s + "input_layernorm.weight"], out[s + "post_attention_layernorm.weight"] = get(b + "ln1"), get(b + "ln2")
            qw, qb = get(b + "qkv.w"), get(b + "qkv.b")
            out[s + "self_attn.q_proj.weight"], out[s + "self_attn.k_proj.weight"], out[s + "self_attn.v_proj.weight"] = qw[:hq].clone(), qw[hq: hq + hk].clone(), qw[hq + hk:].clone()
            out[s + "self_attn.q_proj.bias"], out[s + "self_attn.k_proj.bias"], out[s + "self_attn.v_proj.bias"] = qb[:hq].clone(), qb[hq: hq + hk].clone(), qb[hq + hk:].clone()
            out[s + "self_attn.o_proj.weight"] = get(b + "o.w")
            gu = get(b + "gu.w")
            out[s + "mlp.gate_proj.weight"], out[s + "mlp.up_proj.weight"] = gu[: c.intermediate_size].clone(), gu[c.intermediate_size:].clone()
            out[s + "mlp.down_proj.weight"] = get(b + "down.w")
        out["model.norm.weight"] = get("norm")
        if not c.tie_word_embeddings:
            out["lm_head.weight"] = get("lm_head")
        return out

    def init_random(self, seed: int = 0, std: float = 0.02):
        """Random-init weights of the architecture (benchmarks; no checkpoints exist offline): N(0, std) matrices,
        unit norm gains, zero biases; padded regions stay zero."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        c = self.cfg
        for name, s in self.slots.items():
            v = self.w(name)
            if name.endswith(("norm1", "norm2", "ln1", "ln2", "ln_q")) or name == "norm":
                v.fill_(1.0)
            elif name.endswith(".b"):
                v.zero_()
            else:
                v.copy_((torch.randn(v.shape, generator=g, device=self.device, dtype=F32) * std).to(BF16))
        vi, vip = c.v_inter, c.v_inter_pad
        if vip != vi and c.v_arch == "qwen2_5_vl":
            for i in range(c.v_depth):
                b = f"visual.blocks.{i}."
                self.w(b + "gu.w")[vi:vip].zero_()
                self.w(b + "gu.w")[vip + vi:].zero_()
                self.w(b + "down.w")[:, vi:].zero_()
        if c.is_llava:       # zero padding: the 592 - 588 extra patch columns and the 80 - 72 padded dims of every vision head
            d, dp, nh = c.v_hidden // c.v_heads, c.v_head_pad, c.v_heads
            self.w("visual.patch_embed")[:, c.patch_dim:].zero_()
            for i in range(c.v_run_depth):
                b = f"visual.blocks.{i}."
                self.w(b + "qkv.w").view(3 * nh, dp, -1)[:, d:].zero_()
                self.w(b + "qkv.b").view(3 * nh, dp)[:, d:].zero_()
                self.w(b + "proj.w").view(-1, nh, dp)[:, :, d:].zero_()
        if c.head_dim != c.head_dim_real:       # padded decoder heads (VLMConfig.head_dim): everything outside the real dims' slots is zero
            keep = torch.zeros(c.head_dim, dtype=torch.bool, device=self.device)
            keep[torch.from_numpy(c.head_slots).to(self.device)] = True
            nh_all = c.num_attention_heads + 2 * c.num_key_value_heads
            for i in range(c.num_hidden_layers):
                b = f"layers.{i}."
                self.w(b + "qkv.w").view(nh_all, c.head_dim, -1)[:, ~keep].zero_()
                self.w(b + "qkv.b").view(nh_all, c.head_dim)[:, ~keep].zero_()
                self.w(b + "o.w").view(-1, c.num_attention_heads, c.head_dim)[:, :, ~keep].zero_()
        self.finalize()

    def copy_from(self, other: "ParamStore"):
        self.flat.copy_(other.flat)
        self.finalize()
