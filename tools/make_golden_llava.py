#!/usr/bin/env python3
"""Golden vectors of the LLaVA-OneVision branch (BASELINE.json config 5), captured in the build container from
  * a tiny random-init transformers `LlavaOnevisionForConditionalGeneration` (the arithmetic of the reference's pinned dependency), and
  * the reference's own `SCGRPOTrainer.compute_loss` driven with that model under a model id containing "llava_ov", so that the llava-only
    code path `_ensure_left_padding_data` (REF train/stage_rl/trainer/sc_grpo_trainer.py:502-504,516-567) is part of what is pinned.
Writes tests/golden/llava_ov.npz and tests/golden/sc_grpo_llava_ov.npz.  Run here only (imports /root/reference)."""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402  (reference import helpers, mock trainer)

fx = mg.fx
OUT = mg.OUT


def hf_name(k):
    if k.startswith("vision_tower.vision_model."):
        return "model.vision_tower." + k[len("vision_tower.vision_model."):]
    if k.startswith("multi_modal_projector."):
        return "model." + k
    if k == "image_newline":
        return "model.image_newline"
    if k.startswith("language_model.model."):
        return "model.language_model." + k[len("language_model.model."):]
    if k == "language_model.lm_head.weight":
        return "lm_head.weight"
    raise KeyError(k)


def build_hf(cfg, weights):
    from transformers import LlavaOnevisionConfig, LlavaOnevisionForConditionalGeneration
    from transformers.models.qwen2.configuration_qwen2 import Qwen2Config
    from transformers.models.siglip.configuration_siglip import SiglipVisionConfig
    t, v = cfg["text"], cfg["vision"]
    vc = SiglipVisionConfig(hidden_size=v["hidden_size"], intermediate_size=v["intermediate_size"], num_hidden_layers=v["depth"], num_attention_heads=v["num_heads"],
                            image_size=v["image_size"], patch_size=v["patch_size"], layer_norm_eps=v["layer_norm_eps"])
    tc = Qwen2Config(vocab_size=t["vocab_size"], hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"], num_hidden_layers=t["num_hidden_layers"],
                     num_attention_heads=t["num_attention_heads"], num_key_value_heads=t["num_key_value_heads"], rms_norm_eps=t["rms_norm_eps"], rope_theta=t["rope_theta"],
                     tie_word_embeddings=cfg["tie_word_embeddings"], max_position_embeddings=4096)
    c = LlavaOnevisionConfig(vision_config=vc, text_config=tc, image_token_index=cfg["image_token_id"], image_grid_pinpoints=cfg["image_grid_pinpoints"],
                             vision_feature_layer=-1, vision_feature_select_strategy="full", vision_aspect_ratio=f"anyres_max_{cfg['anyres_max']}", tie_word_embeddings=cfg["tie_word_embeddings"])
    m = LlavaOnevisionForConditionalGeneration(c)
    m.config._attn_implementation = "eager"
    sd = m.state_dict()
    new = {hf_name(k): torch.from_numpy(np.asarray(a)).float() for k, a in weights.items()}
    missing = [k for k in sd if k not in new and "post_layernorm" not in k and ".head." not in k]
    assert not missing, missing[:5]
    m.load_state_dict(new, strict=False)
    return m.float()


def tiny_ov_batch(cfg, sizes, n_texts, seed):
    import iadr1_amd  # noqa: F401  (the product's host plan gives the token counts the processor would reserve)
    from iadr1_amd import llava_ov as lo
    v = cfg["vision"]
    side = v["image_size"] // v["patch_size"]
    rs = np.random.RandomState(seed)
    rows, crops = [], 0
    for sz, nt in zip(sizes, n_texts):
        n_img = lo.num_image_tokens(sz, cfg["image_grid_pinpoints"], v["image_size"], side, cfg["anyres_max"])
        rows.append(rs.randint(3, 600, 3).tolist() + [cfg["image_token_id"]] * n_img + rs.randint(3, 600, nt).tolist())
        crops += lo.num_crops(sz, cfg["image_grid_pinpoints"], v["image_size"])
    ids, mask = fx.left_pad(rows, cfg["pad_token_id"])
    return ids, mask, fx.synth_crops(crops, cfg, seed), crops


def gen_forward(cfg=None, name="llava_ov.npz"):
    cfg = cfg or fx.TINY_OV
    w = fx.make_weights_ov(cfg, 0)
    m = build_hf(cfg, w).eval()
    sizes = [(80, 100), (400, 380)]            # 2 x 2 crop grid; 5 x 5 grid (26 crops: shrunk by bilinear interpolation above anyres_max_9)
    ids, mask, pv, ncrops = tiny_ov_batch(cfg, sizes, [6, 11], seed=31)
    with torch.no_grad():
        feats = m.model.get_image_features(torch.from_numpy(pv), torch.tensor(sizes), vision_feature_layer=-1, vision_feature_select_strategy="full").pooler_output
        out = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), pixel_values=torch.from_numpy(pv), image_sizes=torch.tensor(sizes))
        lp = torch.log_softmax(out.logits[:, :-1].float(), -1).gather(-1, torch.from_numpy(ids)[:, 1:].unsqueeze(-1)).squeeze(-1)
    np.savez_compressed(os.path.join(OUT, name), meta=json.dumps({**mg.meta(), "sizes": sizes, "n_text": [6, 11], "seed": 31, "crops": ncrops}),
                        input_ids=ids, attention_mask=mask, image_features=torch.cat(list(feats), 0).numpy(), feature_lens=np.array([f.shape[0] for f in feats]),
                        logits_last=out.logits[:, -1].numpy(), per_token_logps=lp.numpy())
    print(name + ": feature lens", [f.shape[0] for f in feats], "ids", ids.shape)


hf_name_llava = hf_name      # (the installed transformers 5.x names the CLIP tower of Llava / LlavaNext like the SigLIP tower of LlavaOnevision)


def build_hf_llava(cfg, weights):
    """Tiny `LlavaForConditionalGeneration` (family "llava") / `LlavaNextForConditionalGeneration` ("llava_next") with the fixture's weights."""
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration, LlavaNextConfig, LlavaNextForConditionalGeneration, MistralConfig
    t, v = cfg["text"], cfg["vision"]
    vc = CLIPVisionConfig(hidden_size=v["hidden_size"], intermediate_size=v["intermediate_size"], num_hidden_layers=v["depth"], num_attention_heads=v["num_heads"],
                          image_size=v["image_size"], patch_size=v["patch_size"], layer_norm_eps=v["layer_norm_eps"], hidden_act="quick_gelu", projection_dim=64)
    kw = dict(vocab_size=t["vocab_size"], hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"], num_hidden_layers=t["num_hidden_layers"],
              num_attention_heads=t["num_attention_heads"], num_key_value_heads=t["num_key_value_heads"], rms_norm_eps=t["rms_norm_eps"], rope_theta=t["rope_theta"],
              tie_word_embeddings=False, max_position_embeddings=4096)
    if cfg["family"] == "llava":
        c = LlavaConfig(vision_config=vc, text_config=LlamaConfig(attention_bias=False, **kw), image_token_index=cfg["image_token_id"], vision_feature_layer=-2,
                        vision_feature_select_strategy="default", projector_hidden_act="gelu", tie_word_embeddings=False)
        m = LlavaForConditionalGeneration(c)
    else:
        c = LlavaNextConfig(vision_config=vc, text_config=MistralConfig(sliding_window=None, **kw), image_token_index=cfg["image_token_id"], vision_feature_layer=-2,
                            vision_feature_select_strategy="default", projector_hidden_act="gelu", image_grid_pinpoints=cfg["image_grid_pinpoints"], tie_word_embeddings=False)
        m = LlavaNextForConditionalGeneration(c)
    m.config._attn_implementation = "eager"
    sd = m.state_dict()
    new = {hf_name_llava(k): torch.from_numpy(np.asarray(a)).float() for k, a in weights.items()}
    unknown = [k for k in new if k not in sd]
    missing = [k for k in sd if k not in new and "position_ids" not in k]
    assert not unknown and not missing, (unknown[:5], missing[:5])
    m.load_state_dict(new, strict=False)
    return m.float()


def tiny_llava_batch(cfg, sizes, n_texts, seed):
    """Left-padded prompts with the image tokens the family's processor reserves: side^2 for LLaVA-1.5 (one crop), the packed any-resolution count for NeXT."""
    import iadr1_amd  # noqa: F401
    from iadr1_amd import llava_ov as lo
    v = cfg["vision"]
    side = v["image_size"] // v["patch_size"]
    rs = np.random.RandomState(seed)
    rows, crops = [], 0
    for sz, nt in zip(sizes, n_texts):
        if cfg["family"] == "llava":
            n_img, nc = side * side, 1
        else:
            n_img, nc = lo.num_image_tokens(sz, cfg["image_grid_pinpoints"], v["image_size"], side, None), lo.num_crops(sz, cfg["image_grid_pinpoints"], v["image_size"])
        rows.append(rs.randint(3, 600, 3).tolist() + [cfg["image_token_id"]] * n_img + rs.randint(3, 600, nt).tolist())
        crops += nc
    ids, mask = fx.left_pad(rows, cfg["pad_token_id"])
    return ids, mask, fx.synth_crops(crops, cfg, seed), crops


def gen_forward_llava(cfg, name, sizes):
    w = fx.make_weights_llava(cfg, 0)
    m = build_hf_llava(cfg, w).eval()
    ids, mask, pv, ncrops = tiny_llava_batch(cfg, sizes, [6, 11], seed=33)
    kw = {}
    px = torch.from_numpy(pv)
    if cfg["family"] == "llava_next":          # processor layout: [images, max crops, 3, S, S], zero-padded; + the original sizes
        import iadr1_amd  # noqa: F401
        from iadr1_amd import llava_ov as lo
        ncs = [lo.num_crops(s_, cfg["image_grid_pinpoints"], cfg["vision"]["image_size"]) for s_ in sizes]
        px5 = torch.zeros(len(sizes), max(ncs), *px.shape[1:])
        o = 0
        for i, n in enumerate(ncs):
            px5[i, :n] = px[o: o + n]
            o += n
        px, kw = px5, {"image_sizes": torch.tensor(sizes)}
    with torch.no_grad():
        out = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), pixel_values=px, **kw)
        feats = m.model.get_image_features(px, vision_feature_layer=-2, vision_feature_select_strategy="default", **kw).pooler_output
        lp = torch.log_softmax(out.logits[:, :-1].float(), -1).gather(-1, torch.from_numpy(ids)[:, 1:].unsqueeze(-1)).squeeze(-1)
    feats = list(feats)
    np.savez_compressed(os.path.join(OUT, name), meta=json.dumps({**mg.meta(), "family": cfg["family"], "sizes": sizes, "n_text": [6, 11], "seed": 33, "crops": ncrops}),
                        input_ids=ids, attention_mask=mask, image_features=torch.cat([f.reshape(-1, f.shape[-1]) for f in feats], 0).numpy(),
                        feature_lens=np.array([f.reshape(-1, f.shape[-1]).shape[0] for f in feats]), logits_last=out.logits[:, -1].numpy(), per_token_logps=lp.numpy())
    print(name + ": feature lens", [f.reshape(-1, f.shape[-1]).shape[0] for f in feats], "ids", ids.shape)


class OVProcessor(mg.FakeProcessor):
    """Mock processor for the llava branch: hands over pixel_values / image_sizes; `tokenizer.pad_token_id` is what _ensure_left_padding_data reads."""

    def __init__(self, cfg, batch, texts):
        self.cfg, self.batch = cfg, dict(batch)
        self.pad_token_id, self.eos_token_id = cfg["pad_token_id"], cfg["eos_token_id"]
        self.tokenizer = self
        self._texts = texts
        self.chat_template = "x"


def gen_sc_grpo(SCGRPOTrainer, reward, family="llava_ov", cfg_name="TINY_OV", fname=None, perturb=0.25):
    """family "llava_ov": LLaVA-OneVision; "llava" / "llava_next": LLaVA-1.5 / NeXT under the model ids the reference's switch routes to them.
    cfg_name: the llava_ov fixture (TINY_OV; TINY_OV7 = the 7:1 head geometry of LLaVA-OneVision-7B's decoder, BASELINE config 5)."""
    if family == "llava_ov":
        cfg = getattr(fx, cfg_name)
        w_ref = fx.make_weights_ov(cfg, 0)
    else:
        cfg = fx.TINY_LLAVA15 if family == "llava" else fx.TINY_LLAVA_NEXT
        w_ref = fx.make_weights_llava(cfg, 0)
    w_pol = fx.perturb_weights(w_ref, seed=1, scale=perturb)
    bh = build_hf if family == "llava_ov" else build_hf_llava
    ref, pol = bh(cfg, w_ref).eval(), bh(cfg, w_pol).train()
    for p in ref.parameters():
        p.requires_grad_(False)
    G, C, seed = 4, 10, 41
    sizes = [(120, 100)]
    if family == "llava_ov":
        ids, mask, pv, ncrops = tiny_ov_batch(cfg, sizes, [9], seed)
    else:
        ids, mask, pv, ncrops = tiny_llava_batch(cfg, sizes, [9], seed)
    batch = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "pixel_values": torch.from_numpy(pv)[None], "image_sizes": torch.tensor(sizes)}
    if family == "llava":
        batch = {"input_ids": batch["input_ids"], "attention_mask": batch["attention_mask"], "pixel_values": torch.from_numpy(pv)}
    eos_rows = {1: 6, 3: 2}                    # rows 1 and 3 end early -> they are the rows the reference rotates
    comps = fx.synth_completions(G, C, cfg, seed + 100, eos_rows)
    texts = [mg.CANNED[i % len(mg.CANNED)] for i in range(G)]
    t = mg.make_trainer(SCGRPOTrainer, reward, cfg, ref, {**batch, "mm_token_type_ids": torch.zeros(1, ids.shape[1], dtype=torch.int32)}, comps, texts, G, C)
    t.processing_class = OVProcessor(cfg, batch, texts)
    t.model_id = {"llava_ov": "tiny-llava_ov-si", "llava": "tiny-llava_1_5-7b", "llava_next": "tiny-llava_next-mistral"}[family]
    inputs = [{"prompt": [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "q"}]}], "image": [object()], "solution": mg.SOLUTION}]
    with contextlib.redirect_stdout(io.StringIO()):
        loss, loc = mg.capture_locals(lambda: t.compute_loss(pol, inputs), "compute_loss")
    loss.backward()
    if family == "llava_ov":
        inv = {hf_name(k): k for k in fx.param_shapes_ov(cfg)}
        keep = ["language_model.model.norm.weight", "language_model.model.layers.1.self_attn.k_proj.bias", "multi_modal_projector.linear_2.bias", "image_newline",
                "vision_tower.vision_model.encoder.layers.1.layer_norm2.weight", "vision_tower.vision_model.embeddings.patch_embedding.bias"]
    else:
        inv = {hf_name_llava(k): k for k in fx.param_shapes_llava(cfg)}
        keep = ["language_model.model.norm.weight", "language_model.model.layers.1.self_attn.k_proj.weight", "multi_modal_projector.linear_2.bias",
                "vision_tower.vision_model.encoder.layers.1.layer_norm2.weight", "vision_tower.vision_model.embeddings.class_embedding",
                "vision_tower.vision_model.pre_layrnorm.bias"] + (["image_newline"] if family == "llava_next" else [])
    grads = {inv[k]: p.grad for k, p in pol.named_parameters() if p.grad is not None and k in inv}
    out = {"meta": json.dumps({**mg.meta(), "G": G, "C": C, "sizes": sizes, "n_text": 9, "seed": seed, "beta": 0.04, "eos_rows": eos_rows, "perturb_scale": perturb, "crops": ncrops,
                               "model_id": t.model_id, "config": "fixture_util." + (cfg_name if family == "llava_ov" else "")}),
           "prompt_completion_ids": loc["prompt_completion_ids"].numpy(), "attention_mask": loc["attention_mask"].numpy(), "completion_mask": loc["completion_mask"].numpy(),
           "per_token_logps": loc["per_token_logps"].detach().numpy(), "ref_per_token_logps": loc["ref_per_token_logps"].numpy(), "rewards_per_func": loc["rewards_per_func"].numpy(),
           "advantages": loc["advantages"].numpy(), "loss": np.float64(loss.item()), "metric_kl": np.float64(t._metrics["kl"][0]),
           "metric_completion_length": np.float64(t._metrics["completion_length"][0]), "metric_reward": np.float64(t._metrics["reward"][0]),
           "grad_norm_names": np.array(sorted(grads)), "grad_norms": np.array([float(grads[k].norm()) for k in sorted(grads)], dtype=np.float64)}
    for k in keep:
        out["grad::" + k] = grads[k].numpy()
    fname = fname or {"llava_ov": "sc_grpo_llava_ov.npz", "llava": "sc_grpo_llava15.npz", "llava_next": "sc_grpo_llava_next.npz"}[family]
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(f"{fname}: loss={loss.item():.8f} kl={t._metrics['kl'][0]:.6f} len={t._metrics['completion_length'][0]}")


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if sys.argv[1:] == ["ov7"]:       # only the config-5 head-geometry golden
        reward, _, _, SCGRPOTrainer, _ = mg.import_reference()
        gen_sc_grpo(SCGRPOTrainer, reward, "llava_ov", cfg_name="TINY_OV7", fname="sc_grpo_llava_ov_g7.npz", perturb=0.08)
        sys.exit(0)
    gen_forward()
    gen_forward(fx.TINY_OV64, "llava_ov_hd64.npz")        # 64-wide decoder heads (LLaVA-OneVision-0.5B's Qwen2-0.5B structure)
    gen_forward_llava(fx.TINY_LLAVA15, "llava15.npz", [(56, 56), (56, 56)])
    gen_forward_llava(fx.TINY_LLAVA_NEXT, "llava_next.npz", [(80, 100), (150, 60)])
    reward, _, _, SCGRPOTrainer, _ = mg.import_reference()
    gen_sc_grpo(SCGRPOTrainer, reward)
    gen_sc_grpo(SCGRPOTrainer, reward, "llava")
    gen_sc_grpo(SCGRPOTrainer, reward, "llava_next")
    gen_sc_grpo(SCGRPOTrainer, reward, "llava_ov", cfg_name="TINY_OV7", fname="sc_grpo_llava_ov_g7.npz", perturb=0.08)
