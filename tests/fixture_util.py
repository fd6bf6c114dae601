"""Shared, reference-free builders for the tiny parity workloads.

Everything here is deterministic numpy (legacy ``RandomState``), so that the golden
generator (``tools/make_golden.py``, which imports the reference in the build container), the
oracle tests (CPU) and the HIP parity tests (GPU box, no reference present) all see the *same*
weights and batches without shipping weight blobs.

Weight names follow the on-disk Qwen2.5-VL checkpoint layout the reference loads with
``from_pretrained`` (``visual.blocks.N.attn.qkv.weight``, ``model.layers.N.self_attn.q_proj.weight``,
``lm_head.weight`` ...; /root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:118-119).
"""
from __future__ import annotations

import zlib

import numpy as np

# A Qwen2.5-VL-shaped model small enough for the CPU oracle but using the *real* head sizes the
# HIP attention kernels are built for (text head_dim 128, vision head_dim 80), a ragged vision MLP
# width (324 ~ the real 3420, not a multiple of 8) and both windowed and full ViT blocks.
TINY = {
    "text": {
        "vocab_size": 640,
        "hidden_size": 256,
        "intermediate_size": 512,
        "num_hidden_layers": 2,
        "num_attention_heads": 2,
        "num_key_value_heads": 1,
        "rms_norm_eps": 1e-6,
        "rope_theta": 1000000.0,
        "mrope_section": [16, 24, 24],
    },
    "vision": {
        "depth": 4,
        "hidden_size": 160,
        "intermediate_size": 324,
        "num_heads": 2,
        "in_channels": 3,
        "patch_size": 14,
        "spatial_merge_size": 2,
        "temporal_patch_size": 2,
        "window_size": 112,
        "out_hidden_size": 256,
        "fullatt_block_indexes": [1, 3],
    },
    "image_token_id": 630,
    "video_token_id": 631,
    "vision_start_token_id": 628,
    "vision_end_token_id": 629,
    "eos_token_id": 1,
    "pad_token_id": 2,
    "tie_word_embeddings": True,
}


# A second tiny config with the structural differences of Qwen2.5-VL-7B (BASELINE.json config 4): UNTIED lm_head and a
# GQA group of 7 query heads per kv head (28/4 in the real model), so hidden = 7*128.
TINY7 = {
    "text": dict(TINY["text"], hidden_size=896, intermediate_size=1152, num_attention_heads=7, num_key_value_heads=1, vocab_size=768),
    "vision": dict(TINY["vision"], out_hidden_size=896, depth=2, fullatt_block_indexes=[1]),
    "image_token_id": 760, "video_token_id": 761, "vision_start_token_id": 758, "vision_end_token_id": 759,
    "eos_token_id": 1, "pad_token_id": 2, "tie_word_embeddings": False,
}


# Qwen2-VL structure (BASELINE.json config 1, Qwen2-VL-2B PA-SFT): the vision tower uses LayerNorm (with bias), a
# fc1 -> QuickGELU -> fc2 MLP (mlp_ratio 4), no window attention, and a LayerNorm merger
# (TF:models/qwen2_vl/modeling_qwen2_vl.py:270-300, 418-447); the decoder is the same as Qwen2.5-VL's.
# "hidden_size" is the ViT width (HF `embed_dim`), "out_hidden_size" the merger output (HF vision `hidden_size`).
TINY_Q2 = {
    "text": dict(TINY["text"]),
    "vision": {
        "arch": "qwen2_vl", "depth": 3, "hidden_size": 160, "intermediate_size": 640, "num_heads": 2, "in_channels": 3,
        "patch_size": 14, "spatial_merge_size": 2, "temporal_patch_size": 2, "window_size": 0, "out_hidden_size": 256,
        "fullatt_block_indexes": [0, 1, 2],
    },
    "image_token_id": 630, "video_token_id": 631, "vision_start_token_id": 628, "vision_end_token_id": 629,
    "eos_token_id": 1, "pad_token_id": 2, "tie_word_embeddings": True,
}


def param_shapes(cfg: dict) -> dict[str, tuple[int, ...]]:
    """Checkpoint-name -> shape for a Qwen2.5-VL config dict like ``TINY``."""
    t, v = cfg["text"], cfg["vision"]
    h, inter = t["hidden_size"], t["intermediate_size"]
    hd = h // t["num_attention_heads"]
    kvd = hd * t["num_key_value_heads"]
    vh, vi = v["hidden_size"], v["intermediate_size"]
    p = v["patch_size"]
    mu = v["spatial_merge_size"] ** 2
    s: dict[str, tuple[int, ...]] = {}
    s["visual.patch_embed.proj.weight"] = (vh, v["in_channels"], v["temporal_patch_size"], p, p)
    for i in range(v["depth"]):
        b = f"visual.blocks.{i}."
        s[b + "norm1.weight"] = (vh,)
        s[b + "norm2.weight"] = (vh,)
        s[b + "attn.qkv.weight"] = (3 * vh, vh)
        s[b + "attn.qkv.bias"] = (3 * vh,)
        s[b + "attn.proj.weight"] = (vh, vh)
        s[b + "attn.proj.bias"] = (vh,)
        if v.get("arch") == "qwen2_vl":
            s[b + "norm1.bias"] = (vh,)
            s[b + "norm2.bias"] = (vh,)
            s[b + "mlp.fc1.weight"] = (vi, vh)
            s[b + "mlp.fc1.bias"] = (vi,)
            s[b + "mlp.fc2.weight"] = (vh, vi)
            s[b + "mlp.fc2.bias"] = (vh,)
            continue
        s[b + "mlp.gate_proj.weight"] = (vi, vh)
        s[b + "mlp.gate_proj.bias"] = (vi,)
        s[b + "mlp.up_proj.weight"] = (vi, vh)
        s[b + "mlp.up_proj.bias"] = (vi,)
        s[b + "mlp.down_proj.weight"] = (vh, vi)
        s[b + "mlp.down_proj.bias"] = (vh,)
    s["visual.merger.ln_q.weight"] = (vh,)
    if v.get("arch") == "qwen2_vl":
        s["visual.merger.ln_q.bias"] = (vh,)
    s["visual.merger.mlp.0.weight"] = (vh * mu, vh * mu)
    s["visual.merger.mlp.0.bias"] = (vh * mu,)
    s["visual.merger.mlp.2.weight"] = (v["out_hidden_size"], vh * mu)
    s["visual.merger.mlp.2.bias"] = (v["out_hidden_size"],)
    s["model.embed_tokens.weight"] = (t["vocab_size"], h)
    for i in range(t["num_hidden_layers"]):
        b = f"model.layers.{i}."
        s[b + "input_layernorm.weight"] = (h,)
        s[b + "self_attn.q_proj.weight"] = (h, h)
        s[b + "self_attn.q_proj.bias"] = (h,)
        s[b + "self_attn.k_proj.weight"] = (kvd, h)
        s[b + "self_attn.k_proj.bias"] = (kvd,)
        s[b + "self_attn.v_proj.weight"] = (kvd, h)
        s[b + "self_attn.v_proj.bias"] = (kvd,)
        s[b + "self_attn.o_proj.weight"] = (h, h)
        s[b + "post_attention_layernorm.weight"] = (h,)
        s[b + "mlp.gate_proj.weight"] = (inter, h)
        s[b + "mlp.up_proj.weight"] = (inter, h)
        s[b + "mlp.down_proj.weight"] = (h, inter)
    s["model.norm.weight"] = (h,)
    if not cfg.get("tie_word_embeddings", False):
        s["lm_head.weight"] = (t["vocab_size"], h)
    return s


def _bf16_round(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (so bf16 kernels and the fp32 oracle share weights)."""
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def make_weights(cfg: dict, seed: int = 0, std: float = 0.05, bf16_exact: bool = True) -> dict[str, np.ndarray]:
    """Seeded weights keyed by checkpoint name.  Norm gains ~ 1 + 0.1 N(0,1), biases ~ 0.02 N(0,1),
    matrices ~ std N(0,1).  Values are bf16-representable when ``bf16_exact``."""
    out = {}
    for name, shape in param_shapes(cfg).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        x = rs.standard_normal(shape).astype(np.float32)
        if name.endswith("norm.weight") or "norm1.weight" in name or "norm2.weight" in name or name.endswith("ln_q.weight") or "layernorm.weight" in name:
            x = 1.0 + 0.1 * x
        elif name.endswith(".bias"):
            x = 0.02 * x
        else:
            x = std * x
        out[name] = _bf16_round(x) if bf16_exact else x
    return out


def perturb_weights(w: dict[str, np.ndarray], seed: int, scale: float = 0.02) -> dict[str, np.ndarray]:
    """A second, nearby parameter set (the 'policy' next to the frozen 'ref') so that KL != 0."""
    out = {}
    for name, a in w.items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 104729 * seed) & 0x7FFFFFFF)
        out[name] = _bf16_round(a + scale * np.abs(a).mean() * rs.standard_normal(a.shape).astype(np.float32))
    return out


def synth_pixel_values(grid_thw: list[tuple[int, int, int]], cfg: dict, seed: int = 1234) -> np.ndarray:
    """Processor-shaped patches ``[sum(t*h*w), C*T*P*P]`` fp32, bf16-representable, ~N(0,1)
    (what Qwen2VLImageProcessor emits after normalisation)."""
    v = cfg["vision"]
    k = v["in_channels"] * v["temporal_patch_size"] * v["patch_size"] ** 2
    n = sum(t * h * w for t, h, w in grid_thw)
    rs = np.random.RandomState(seed)
    return _bf16_round(rs.standard_normal((n, k)).astype(np.float32))


def n_image_tokens(grid: tuple[int, int, int], cfg: dict) -> int:
    t, h, w = grid
    return t * h * w // cfg["vision"]["spatial_merge_size"] ** 2


def synth_prompt(grid: tuple[int, int, int], n_text: int, cfg: dict, seed: int) -> list[int]:
    """`<prefix> <|vision_start|> <|image_pad|>*N <|vision_end|> <text>*n_text` token ids."""
    rs = np.random.RandomState(seed)
    lo, hi = 3, cfg["vision_start_token_id"]  # avoid pad/eos/special ids
    prefix = rs.randint(lo, hi, size=3).tolist()
    body = rs.randint(lo, hi, size=n_text).tolist()
    return (
        prefix
        + [cfg["vision_start_token_id"]]
        + [cfg["image_token_id"]] * n_image_tokens(grid, cfg)
        + [cfg["vision_end_token_id"]]
        + body
    )


def left_pad(rows: list[list[int]], pad_id: int) -> tuple[np.ndarray, np.ndarray]:
    m = max(len(r) for r in rows)
    ids = np.full((len(rows), m), pad_id, dtype=np.int64)
    mask = np.zeros((len(rows), m), dtype=np.int64)
    for i, r in enumerate(rows):
        ids[i, m - len(r):] = r
        mask[i, m - len(r):] = 1
    return ids, mask


def synth_completions(n: int, max_len: int, cfg: dict, seed: int, eos_rows: dict[int, int] | None = None) -> list[list[int]]:
    """``n`` ragged completions; rows listed in ``eos_rows`` end with EOS at that index."""
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        ln = max_len if (eos_rows is None or i not in eos_rows) else eos_rows[i] + 1
        ids = rs.randint(3, cfg["vision_start_token_id"], size=ln).tolist()
        if eos_rows is not None and i in eos_rows:
            ids[-1] = cfg["eos_token_id"]
        out.append(ids)
    return out


# ------------------------------------------------------------------------------------------------------------------------------------
# An offline Qwen2-VL processor (no hub access, no tokenizer files): a character-level tokenizer with the Qwen2-VL special tokens, the
# HF PIL image processor of the family, and the Qwen2-VL chat template (ChatML turns, `<|vision_start|><|image_pad|><|vision_end|>` per
# image, default system turn).  Used by the golden generator (through the reference's own `maybe_apply_chat_template`) and by the host-side
# test of `iadr1_amd.trainer.prepare_batch`, so that chat rendering, tokenisation with left padding and image patching really execute.
# ------------------------------------------------------------------------------------------------------------------------------------
QWEN2VL_CHAT_TEMPLATE = (
    "{% set image_count = namespace(value=0) %}{% for message in messages %}{% if loop.first and message['role'] != 'system' %}<|im_start|>system\n"
    "You are a helpful assistant.<|im_end|>\n{% endif %}<|im_start|>{{ message['role'] }}\n{% if message['content'] is string %}{{ message['content'] }}<|im_end|>\n"
    "{% else %}{% for content in message['content'] %}{% if content['type'] == 'image' or 'image' in content or 'image_url' in content %}"
    "{% set image_count.value = image_count.value + 1 %}{% if add_vision_id %}Picture {{ image_count.value }}: {% endif %}<|vision_start|><|image_pad|><|vision_end|>"
    "{% elif 'text' in content %}{{ content['text'] }}{% endif %}{% endfor %}<|im_end|>\n{% endif %}{% endfor %}{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}"
)
QWEN2VL_SPECIAL = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|vision_start|>", "<|vision_end|>", "<|image_pad|>", "<|video_pad|>"]


def local_qwen2vl_processor(max_pixels=None, min_pixels=None):
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    from transformers.models.qwen2_vl.processing_qwen2_vl import Qwen2VLProcessor
    import transformers
    BaseVideoProcessor = transformers.BaseVideoProcessor     # offline (no torchvision) this is the placeholder class the processor's type check looks up

    words = QWEN2VL_SPECIAL + [chr(c) for c in range(32, 127)] + ["\n"]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<|endoftext|>"))
    tok.pre_tokenizer = pre_tokenizers.Split("", "isolated")
    t = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="<|endoftext|>", additional_special_tokens=QWEN2VL_SPECIAL)

    class _NoVideo(BaseVideoProcessor):      # the family's video processor needs torchvision (absent offline); images never touch it
        _auto_class = None                   # (save_pretrained of the processor skips it)

    proc = Qwen2VLProcessor(image_processor=Qwen2VLImageProcessorPil(), tokenizer=t, video_processor=_NoVideo.__new__(_NoVideo), chat_template=QWEN2VL_CHAT_TEMPLATE)
    if max_pixels is not None:               # REF train/stage_rl/trainer/sc_grpo_trainer.py:192-193
        proc.image_processor.max_pixels = max_pixels
        proc.image_processor.min_pixels = min_pixels
    return proc


def synth_pil_image(width, height, seed):
    from PIL import Image
    return Image.fromarray(np.random.RandomState(seed).randint(0, 256, (height, width, 3)).astype(np.uint8))


def prepare_examples():
    """Dataset rows as train/stage_rl/grpo_ad.py::make_conversation emits them (conversational prompt + image list), plus a plain-string prompt."""
    img = lambda: {"type": "image"}
    return [
        [{"prompt": [{"role": "user", "content": [img(), {"type": "text", "text": "Is there any defect in the object?"}]}], "image": [synth_pil_image(448, 448, 1)], "solution": "s0"},
         {"prompt": [{"role": "system", "content": "You are an inspector."}, {"role": "user", "content": [img(), img(), {"type": "text", "text": "Compare the two images."}]}],
          "image": [synth_pil_image(300, 200, 2), synth_pil_image(448, 336, 3)], "solution": "s1"}],
        [{"prompt": "<|im_start|>user\n<|vision_start|><|image_pad|><|vision_end|>plain string prompt<|im_end|>\n<|im_start|>assistant\n", "image": [synth_pil_image(224, 224, 4)], "solution": "s2"}],
    ]


# ------------------------------------------------------------------------------------------------------------------------------------
# LLaVA-OneVision structure (BASELINE.json config 5: LLaVA-OV-SI-7B; the reference's model switch sc_grpo_trainer.py:124-132): SigLIP tower with
# the real head size 72 (2 heads -> width 144), LayerNorm + GELU(tanh) MLP, learned positions, 4 x 4 tokens per 56-pixel crop; Linear-GELU-Linear
# projector; any-resolution crop grid; Qwen2 decoder (untied head, 1-D rotary).  Checkpoint names as transformers 4.51.3 (the reference's pin) writes them.
# ------------------------------------------------------------------------------------------------------------------------------------
TINY_OV = {
    "text": dict(TINY["text"]),
    "vision": {"arch": "siglip", "depth": 2, "hidden_size": 144, "intermediate_size": 256, "num_heads": 2, "in_channels": 3, "patch_size": 14, "image_size": 56,
               "layer_norm_eps": 1e-6},
    "image_grid_pinpoints": [[56, 56], [56, 112], [112, 56], [112, 112], [168, 112], [112, 168], [168, 168], [224, 224], [280, 280]],
    "anyres_max": 9,
    "image_token_id": 630, "video_token_id": 631, "vision_start_token_id": 628, "vision_end_token_id": 629,
    "eos_token_id": 1, "pad_token_id": 2, "tie_word_embeddings": False,
}


# the same structure with 64-wide decoder heads (the Qwen2-0.5B decoder of LLaVA-OneVision-0.5B: 14 heads of 64): 4 query heads, 2 kv heads on width 256
TINY_OV64 = dict(TINY_OV, text=dict(TINY_OV["text"], num_attention_heads=4, num_key_value_heads=2))


# ... and with the decoder geometry of LLaVA-OneVision-7B (Qwen2-7B: 28 query heads on 4 kv heads = a GQA group of 7; BASELINE config 5): 7 query heads, 1 kv head, width 896
TINY_OV7 = dict(TINY_OV, text=dict(TINY_OV["text"], hidden_size=896, intermediate_size=1152, num_attention_heads=7, num_key_value_heads=1))


def param_shapes_ov(cfg: dict) -> dict[str, tuple[int, ...]]:
    t, v = cfg["text"], cfg["vision"]
    h, inter = t["hidden_size"], t["intermediate_size"]
    hd = h // t["num_attention_heads"]
    kvd = hd * t["num_key_value_heads"]
    vh, vi, p = v["hidden_size"], v["intermediate_size"], v["patch_size"]
    npos = (v["image_size"] // p) ** 2
    s: dict[str, tuple[int, ...]] = {}
    pre = "vision_tower.vision_model."
    s[pre + "embeddings.patch_embedding.weight"] = (vh, v["in_channels"], p, p)
    s[pre + "embeddings.patch_embedding.bias"] = (vh,)
    s[pre + "embeddings.position_embedding.weight"] = (npos, vh)
    for i in range(v["depth"]):
        b = f"{pre}encoder.layers.{i}."
        for ln in ("layer_norm1", "layer_norm2"):
            s[b + ln + ".weight"] = (vh,)
            s[b + ln + ".bias"] = (vh,)
        for z in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[b + f"self_attn.{z}.weight"] = (vh, vh)
            s[b + f"self_attn.{z}.bias"] = (vh,)
        s[b + "mlp.fc1.weight"], s[b + "mlp.fc1.bias"] = (vi, vh), (vi,)
        s[b + "mlp.fc2.weight"], s[b + "mlp.fc2.bias"] = (vh, vi), (vh,)
    s["multi_modal_projector.linear_1.weight"], s["multi_modal_projector.linear_1.bias"] = (h, vh), (h,)
    s["multi_modal_projector.linear_2.weight"], s["multi_modal_projector.linear_2.bias"] = (h, h), (h,)
    s["image_newline"] = (h,)
    s["language_model.model.embed_tokens.weight"] = (t["vocab_size"], h)
    for i in range(t["num_hidden_layers"]):
        b = f"language_model.model.layers.{i}."
        s[b + "input_layernorm.weight"] = (h,)
        s[b + "self_attn.q_proj.weight"], s[b + "self_attn.q_proj.bias"] = (h, h), (h,)
        s[b + "self_attn.k_proj.weight"], s[b + "self_attn.k_proj.bias"] = (kvd, h), (kvd,)
        s[b + "self_attn.v_proj.weight"], s[b + "self_attn.v_proj.bias"] = (kvd, h), (kvd,)
        s[b + "self_attn.o_proj.weight"] = (h, h)
        s[b + "post_attention_layernorm.weight"] = (h,)
        s[b + "mlp.gate_proj.weight"], s[b + "mlp.up_proj.weight"], s[b + "mlp.down_proj.weight"] = (inter, h), (inter, h), (h, inter)
    s["language_model.model.norm.weight"] = (h,)
    if not cfg.get("tie_word_embeddings", False):
        s["language_model.lm_head.weight"] = (t["vocab_size"], h)
    return s


def make_weights_ov(cfg: dict, seed: int = 0, std: float = 0.05) -> dict[str, np.ndarray]:
    out = {}
    for name, shape in param_shapes_ov(cfg).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        x = rs.standard_normal(shape).astype(np.float32)
        if "layer_norm" in name and name.endswith(".weight") or name.endswith("layernorm.weight") or name.endswith("norm.weight"):
            x = 1.0 + 0.1 * x
        elif name.endswith(".bias"):
            x = 0.02 * x
        elif name == "image_newline":
            x = 0.06 * x
        else:
            x = std * x
        out[name] = _bf16_round(x)
    return out


def synth_crops(n_crops: int, cfg: dict, seed: int) -> np.ndarray:
    """Processor-shaped crops [n, 3, S, S] fp32, bf16-representable, ~N(0,1) (normalised pixels)."""
    s = cfg["vision"]["image_size"]
    return _bf16_round(np.random.RandomState(seed).standard_normal((n_crops, cfg["vision"]["in_channels"], s, s)).astype(np.float32))


LLAVA_OV_SPECIAL = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<image>", "<video>"]
LLAVA_OV_CHAT_TEMPLATE = (     # the llava-onevision-qwen2 chat template's structure: ChatML turns, every image placeholder of a turn before its text
    "{% for message in messages %}<|im_start|>{{ message['role'] }}\n{% if message['content'] is string %}{{ message['content'] }}{% else %}"
    "{% for content in message['content'] %}{% if content['type'] == 'image' %}<image>{% endif %}{% endfor %}"
    "{% for content in message['content'] %}{% if content['type'] == 'text' %}\n{{ content['text'] }}{% endif %}{% endfor %}{% endif %}<|im_end|>\n{% endfor %}"
    "{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}"
)


def local_llava_ov_processor(cfg: dict = None):
    """Offline transformers LlavaOnevisionProcessor (character-level tokenizer, the family's PIL image processor with the tiny structure's crop size and
    pinpoints): any-resolution cropping and the expansion of every `<image>` to the packed feature count really execute."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    import transformers
    from transformers import LlavaOnevisionImageProcessorPil, PreTrainedTokenizerFast
    from transformers.models.llava_onevision.processing_llava_onevision import LlavaOnevisionProcessor

    cfg = cfg or TINY_OV
    v = cfg["vision"]
    words = LLAVA_OV_SPECIAL + [chr(c) for c in range(32, 127)] + ["\n"]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<|endoftext|>"))
    tok.pre_tokenizer = pre_tokenizers.Split("", "isolated")
    t = PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="<|im_end|>", pad_token="<|endoftext|>", additional_special_tokens=LLAVA_OV_SPECIAL)

    class _NoVideo(transformers.BaseVideoProcessor):
        _auto_class = None

    ip = LlavaOnevisionImageProcessorPil(size={"height": v["image_size"], "width": v["image_size"]}, image_grid_pinpoints=[list(p) for p in cfg["image_grid_pinpoints"]])
    return LlavaOnevisionProcessor(image_processor=ip, tokenizer=t, video_processor=_NoVideo.__new__(_NoVideo), num_image_tokens=(v["image_size"] // v["patch_size"]) ** 2,
                                   vision_feature_select_strategy="full", chat_template=LLAVA_OV_CHAT_TEMPLATE, vision_aspect_ratio=f"anyres_max_{cfg.get('anyres_max', 9)}")


# ------------------------------------------------------------------------------------------------------------------------------------
# LLaVA-1.5 / LLaVA-NeXT structures (the remaining branches of the reference's model switch, sc_grpo_trainer.py:130-135): CLIP tower (class token, pre-LayerNorm,
# QuickGELU MLP, 64-wide heads, bias-free patch conv, feature layer -2 without the class token), Linear-GELU-Linear projector, LLaMA (MHA) / Mistral (GQA)
# decoder without q/k/v biases; NeXT adds the any-resolution crop grid + image_newline.  Checkpoint names as transformers 4.51.3 writes them.
# ------------------------------------------------------------------------------------------------------------------------------------
TINY_LLAVA15 = {
    "family": "llava",
    "text": {"vocab_size": 640, "hidden_size": 256, "intermediate_size": 512, "num_hidden_layers": 2, "num_attention_heads": 2, "num_key_value_heads": 2,
             "rms_norm_eps": 1e-5, "rope_theta": 10000.0, "attention_bias": False},
    "vision": {"arch": "clip", "depth": 3, "hidden_size": 128, "intermediate_size": 256, "num_heads": 2, "in_channels": 3, "patch_size": 14, "image_size": 56,
               "layer_norm_eps": 1e-5},
    "vision_feature_layer": -2, "vision_feature_select_strategy": "default",
    "image_token_id": 630, "video_token_id": 631, "vision_start_token_id": 628, "vision_end_token_id": 629, "eos_token_id": 1, "pad_token_id": 2, "tie_word_embeddings": False,
}
TINY_LLAVA_NEXT = dict(TINY_LLAVA15, family="llava_next", text=dict(TINY_LLAVA15["text"], num_key_value_heads=1, rope_theta=1000000.0),
                       image_grid_pinpoints=[[56, 112], [112, 56], [112, 112], [168, 56], [56, 168]])


def param_shapes_llava(cfg: dict) -> dict[str, tuple[int, ...]]:
    t, v = cfg["text"], cfg["vision"]
    h, inter = t["hidden_size"], t["intermediate_size"]
    hd = h // t["num_attention_heads"]
    kvd = hd * t["num_key_value_heads"]
    vh, vi, p = v["hidden_size"], v["intermediate_size"], v["patch_size"]
    npos = (v["image_size"] // p) ** 2 + 1
    s: dict[str, tuple[int, ...]] = {}
    pre = "vision_tower.vision_model."
    s[pre + "embeddings.class_embedding"] = (vh,)
    s[pre + "embeddings.patch_embedding.weight"] = (vh, v["in_channels"], p, p)
    s[pre + "embeddings.position_embedding.weight"] = (npos, vh)
    s[pre + "pre_layrnorm.weight"], s[pre + "pre_layrnorm.bias"] = (vh,), (vh,)
    for i in range(v["depth"]):
        b = f"{pre}encoder.layers.{i}."
        for ln in ("layer_norm1", "layer_norm2"):
            s[b + ln + ".weight"], s[b + ln + ".bias"] = (vh,), (vh,)
        for z in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[b + f"self_attn.{z}.weight"], s[b + f"self_attn.{z}.bias"] = (vh, vh), (vh,)
        s[b + "mlp.fc1.weight"], s[b + "mlp.fc1.bias"] = (vi, vh), (vi,)
        s[b + "mlp.fc2.weight"], s[b + "mlp.fc2.bias"] = (vh, vi), (vh,)
    s[pre + "post_layernorm.weight"], s[pre + "post_layernorm.bias"] = (vh,), (vh,)
    s["multi_modal_projector.linear_1.weight"], s["multi_modal_projector.linear_1.bias"] = (h, vh), (h,)
    s["multi_modal_projector.linear_2.weight"], s["multi_modal_projector.linear_2.bias"] = (h, h), (h,)
    if cfg["family"] == "llava_next":
        s["image_newline"] = (h,)
    s["language_model.model.embed_tokens.weight"] = (t["vocab_size"], h)
    for i in range(t["num_hidden_layers"]):
        b = f"language_model.model.layers.{i}."
        s[b + "input_layernorm.weight"] = (h,)
        s[b + "self_attn.q_proj.weight"], s[b + "self_attn.k_proj.weight"], s[b + "self_attn.v_proj.weight"], s[b + "self_attn.o_proj.weight"] = (h, h), (kvd, h), (kvd, h), (h, h)
        s[b + "post_attention_layernorm.weight"] = (h,)
        s[b + "mlp.gate_proj.weight"], s[b + "mlp.up_proj.weight"], s[b + "mlp.down_proj.weight"] = (inter, h), (inter, h), (h, inter)
    s["language_model.model.norm.weight"] = (h,)
    s["language_model.lm_head.weight"] = (t["vocab_size"], h)
    return s


def make_weights_llava(cfg: dict, seed: int = 0, std: float = 0.05) -> dict[str, np.ndarray]:
    out = {}
    for name, shape in param_shapes_llava(cfg).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        x = rs.standard_normal(shape).astype(np.float32)
        if ("layer_norm" in name or "layrnorm" in name or "layernorm" in name or name.endswith("norm.weight")) and name.endswith(".weight"):
            x = 1.0 + 0.1 * x
        elif name.endswith(".bias"):
            x = 0.02 * x
        elif name in ("image_newline",) or name.endswith("class_embedding"):
            x = 0.06 * x
        else:
            x = std * x
        out[name] = _bf16_round(x)
    return out


LLAVA_SPECIAL = ["<unk>", "<s>", "</s>", "<image>", "<pad>"]
LLAVA15_CHAT_TEMPLATE = (      # llava-1.5-hf chat template's structure: "USER: <image>\n<text> ASSISTANT:"
    "{% for message in messages %}{% if message['role'] != 'system' %}{{ message['role'].upper() + ': '}}{% endif %}"
    "{% for content in message['content'] | selectattr('type', 'equalto', 'image') %}{{ '<image>\n' }}{% endfor %}"
    "{% if message['role'] == 'system' %}{{ message['content'][0]['text'] if message['content'] is not string else message['content'] }}{{ ' ' }}{% else %}"
    "{% for content in message['content'] | selectattr('type', 'equalto', 'text') %}{{ content['text'] + ' '}}{% endfor %}{% endif %}{% endfor %}"
    "{% if add_generation_prompt %}{{ 'ASSISTANT:' }}{% endif %}"
)
LLAVA_NEXT_CHAT_TEMPLATE = (   # llava-v1.6-mistral-7b-hf: "[INST] <image>\n<text> [/INST]"
    "{% for message in messages %}{% if message['role'] == 'user' %}{{ '[INST] ' }}{% for content in message['content'] | selectattr('type', 'equalto', 'image') %}{{ '<image>\n' }}{% endfor %}"
    "{% for content in message['content'] | selectattr('type', 'equalto', 'text') %}{{ content['text'] }}{% endfor %}{{ ' [/INST]' }}{% endif %}{% endfor %}"
)


def local_llava_processor(cfg: dict):
    """Offline transformers LlavaProcessor (cfg family "llava": CLIP image processor, one resized / centre-cropped crop per image) or LlavaNextProcessor ("llava_next":
    any-resolution crops) with a character-level tokenizer (bos <s>, eos </s>); every `<image>` is expanded by the processor to the image token count of the family."""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import CLIPImageProcessorPil, LlavaNextImageProcessorPil, PreTrainedTokenizerFast
    from transformers.models.llava.processing_llava import LlavaProcessor
    from transformers.models.llava_next.processing_llava_next import LlavaNextProcessor

    v = cfg["vision"]
    words = LLAVA_SPECIAL + [chr(c) for c in range(32, 127)] + ["\n"]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split("", "isolated")
    t = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", pad_token="<pad>", unk_token="<unk>", additional_special_tokens=["<image>"])
    size, crop = {"shortest_edge": v["image_size"]}, {"height": v["image_size"], "width": v["image_size"]}
    if cfg["family"] == "llava":
        return LlavaProcessor(image_processor=CLIPImageProcessorPil(size=size, crop_size=crop), tokenizer=t, patch_size=v["patch_size"], vision_feature_select_strategy="default",
                              chat_template=LLAVA15_CHAT_TEMPLATE, num_additional_image_tokens=1)
    ip = LlavaNextImageProcessorPil(size=size, crop_size=crop, image_grid_pinpoints=[list(p) for p in cfg["image_grid_pinpoints"]])
    return LlavaNextProcessor(image_processor=ip, tokenizer=t, patch_size=v["patch_size"], vision_feature_select_strategy="default", chat_template=LLAVA_NEXT_CHAT_TEMPLATE,
                              num_additional_image_tokens=1)
