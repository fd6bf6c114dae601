#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference and the installed transformers); nothing
here travels to the GPU box.  The fixtures it writes are *data*: inputs + outputs of the
reference's own functions:

  rewards.json        train/stage_rl/reward.py:13-101  accuracy_reward / consistency_reward,
                      reward_process/type_reward.py:155-232 compute_reward,
                      reward_process/location_reward.py:1-49 map_location_to_region
  pad.json            trl/trl/trainer/utils.py:418-478 pad
  sc_grpo_g4.npz,     train/stage_rl/trainer/sc_grpo_trainer.py:586-819 SCGRPOTrainer.compute_loss on a
  sc_grpo_g8.npz      tiny random-init Qwen2_5_VLForConditionalGeneration (mocked processor / vLLM)
  logps_padded.npz    sc_grpo_trainer.py:384-514 _get_per_token_logps on a left-padded 2-prompt batch
  vision_index.json   transformers.vision_utils get_vision_window_index / get_vision_position_ids,
                      Qwen2_5_VLModel.get_rope_index
  greedy.npz          HF generate(do_sample=False) token ids (the rollout's bit-exact target)
  sft.npz             HF forward(labels=...) loss (transformers loss_utils.ForCausalLMLoss) and a
                      3-step torch.optim.AdamW loss curve (llamafactory sft trainer semantics)

Recipe for importing SCGRPOTrainer without vllm / sentence_transformers: SURVEY.md section 8(c).
Goldens are captured against transformers == the version recorded in each fixture's `meta`.
"""
from __future__ import annotations

import contextlib
import importlib.machinery
import io
import json
import os
import sys
import types
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(REF, "trl"))
sys.path.insert(0, os.path.join(REF, "train", "stage_rl"))

import fixture_util as fx  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    st_models = _stub("sentence_transformers.models")
    _stub("sentence_transformers", SentenceTransformer=object, models=st_models)
    import trl  # noqa: F401  (vendored trl from the reference tree)

    _stub("vllm", LLM=object, SamplingParams=object)
    import reward  # reference train/stage_rl/reward.py
    from reward_process import location_reward, type_reward
    from trainer import SCGRPOTrainer
    from trl.trainer.utils import pad

    return reward, type_reward, location_reward, SCGRPOTrainer, pad


def meta():
    import transformers

    return {"transformers": transformers.__version__, "torch": torch.__version__, "reference": "Yanhui-Lee/IAD-R1 @ 2025-12-26"}


# ----------------------------------------------------------------------------------------------
# rewards
# ----------------------------------------------------------------------------------------------
TYPE_STRINGS = [
    "scratch", "Scratch", "surface scratch", "scratch mark", "linear scratch", "a long scratch mark on the surface",
    "scrach", "scratchh", "contamination", "Contamination", "stain", "dirt", "color anomaly", "colour anomaly",
    "surface contamination", "impurity", "foreign object", "presence of foreign objects", "foreign body", "debris",
    "unwanted object", "missing parts", "missing part", "notch", "gap", "chip", "surface notch", "deformation",
    "shape distortion", "warping", "bending", "bent component", "bent", "hole", "opening", "puncture", "cavity",
    "void", "through-hole", "through hole", "damage", "structural damage", "breakage", "fracture", "broken", "rupture",
    "surface damage", "abrasion", "wear", "surface wear", "wear mark", "grinding damage", "surface anomalies",
    "surface anomaly", "structural anomalies", "Structural Anomaly", "structural anomaly.", "crack", "misalignment",
    "discoloration", "", "   ", "!!!", "hole, damage", "scratch and stain", "perforation!", "twisting ", "irregularity",
    "geometric  distortion", "material damage", "deterioration", "aperture", "penetration defect", "score mark",
    "linear anomaly", "extraneous material", "foreign element", "foreign matter", "contaminant object",
    "surface discontinuity", "surface erosion", "shape deviation",
]
LOCATION_STRINGS = [
    "top left", "top-left corner", "upper left", "left", "center", "centre", "middle", "right", "upper right",
    "top right", "bottom left", "lower left", "bottom", "lower right", "bottom right corner", "top", "upper",
    "lower", "the left side near the top", "right edge", "", "everywhere", "topleft", "Bottom-Right", "TOP",
    "left and right", "upper bottom",
]


def build_reward_cases():
    rs = np.random.RandomState(20250929)
    comps, sols = [], []

    def yes_comp(t, loc, ans="yes", think="looks odd"):
        return f"<think>{think}</think><location>{loc}</location><type>{t}</type><answer>{ans}</answer>"

    def no_comp(ans="no", think="looks fine"):
        return f"<think>{think}</think><answer>{ans}</answer>"

    def yes_sol(t, loc):
        return f"<think>gt</think><location>{loc}</location><type>{t}</type><answer>yes</answer>"

    # systematic: every type string against a handful of gt types
    gts = ["scratch", "Contamination", "hole", "structural damage", "surface anomalies", "Structural Anomalies", "bent component", "foreign object", "wear"]
    for t in TYPE_STRINGS:
        for g in gts:
            comps.append(yes_comp(t, "top left"))
            sols.append(yes_sol(g, "upper left"))
    # locations
    for a in LOCATION_STRINGS:
        for b in LOCATION_STRINGS[::3]:
            comps.append(yes_comp("scratch", a))
            sols.append(yes_sol("scratch", b))
    # answer variants / malformed
    odd = [
        no_comp(), no_comp("No"), no_comp(" no "), no_comp("yes"), no_comp("maybe"), "no", "<answer>no</answer>",
        "<think>x</think>\n<answer>no</answer>", "<think>a</think><answer>no</answer> trailing", "prefix <think>a</think><answer>no</answer>",
        "<think>a</think><location>l</location><answer>no</answer>", "<think>a</think><type>t</type><answer>no</answer>",
        yes_comp("scratch", "left", "yes"), yes_comp("scratch", "left", "Yes"), yes_comp("scratch", "left", "no"),
        yes_comp("scratch", "left", ""), "<think>a</think><type>scratch</type><location>left</location><answer>yes</answer>",
        "<think>a</think><location>left</location><type>scratch</type>", "<location>left</location><type>scratch</type><answer>yes</answer>",
        "<think>line1\nline2</think><location>top\nleft</location><type>scratch</type><answer>yes</answer>",
        "<think>a</think> <location>left</location><type>scratch</type><answer>yes</answer>",
        "<think></think><location></location><type></type><answer></answer>", "", "<answer>yes</answer><answer>no</answer>",
        "<think>a</think><answer>no</answer><answer>yes</answer>", "<THINK>a</THINK><ANSWER>no</ANSWER>",
        "<think>a</think><location>left</location><type>scratch</type><answer>yes</answer><type>hole</type>",
    ]
    sol_variants = [
        "<think>g</think><answer>no</answer>", "<answer>no</answer>", "no", "No", " NO ", yes_sol("scratch", "left"),
        "<answer>yes</answer>", "yes", "<think>g</think><location>left</location><answer>yes</answer>",
        "<think>g</think><type>scratch</type><answer>yes</answer>", "<answer>Yes</answer>",
    ]
    for c in odd:
        for s in sol_variants:
            comps.append(c)
            sols.append(s)
    # random mixes
    for _ in range(150):
        t = TYPE_STRINGS[rs.randint(len(TYPE_STRINGS))]
        g = TYPE_STRINGS[rs.randint(len(TYPE_STRINGS))]
        a = LOCATION_STRINGS[rs.randint(len(LOCATION_STRINGS))]
        b = LOCATION_STRINGS[rs.randint(len(LOCATION_STRINGS))]
        ans = ["yes", "no", "Yes", "unsure"][rs.randint(4)]
        comps.append(yes_comp(t, a, ans))
        sols.append(yes_sol(g, b))
    return comps, sols


def gen_rewards(reward, type_reward, location_reward):
    comps, sols = build_reward_cases()
    wrapped = [[{"role": "assistant", "content": c}] for c in comps]
    with contextlib.redirect_stdout(io.StringIO()):
        acc = reward.accuracy_reward(wrapped, sols)
        fmt = reward.consistency_reward(wrapped, sols)
    assert len(acc) == len(comps) and len(fmt) == len(comps)
    calc = type_reward.AnomalyRewardCalculator()
    type_pairs = [(a, b) for a in TYPE_STRINGS for b in TYPE_STRINGS[::2]]
    type_scores = [calc.compute_reward(a, b) for a, b in type_pairs]
    loc_pairs = [(a, b) for a in LOCATION_STRINGS for b in LOCATION_STRINGS]
    loc_scores = [location_reward.map_location_to_region(a, b) for a, b in loc_pairs]
    # the reference quirk of SURVEY Appendix B.8: gt neither yes/no -> consistency_reward emits nothing
    with contextlib.redirect_stdout(io.StringIO()):
        short = reward.consistency_reward([[{"role": "assistant", "content": "x"}]] * 2, ["maybe", "<answer>no</answer>"])
    obj = {
        "meta": meta(),
        "completions": comps, "solutions": sols, "accuracy": acc, "format": fmt,
        "type_pairs": type_pairs, "type_scores": type_scores,
        "location_pairs": loc_pairs, "location_scores": loc_scores,
        "format_len_quirk": len(short),
    }
    with open(os.path.join(OUT, "rewards.json"), "w") as f:
        json.dump(obj, f, indent=0)
    print(f"rewards.json: {len(comps)} reward cases, {len(type_pairs)} type pairs, {len(loc_pairs)} location pairs")


def gen_pad(pad):
    cases = []
    rs = np.random.RandomState(7)
    specs = [
        ([[1, 2, 3], [4, 5]], 0, "right", None), ([[1, 2, 3], [4, 5]], 0, "left", None),
        ([[1], [2, 3, 4, 5], [6, 7]], 9, "right", None), ([[1], [2, 3, 4, 5], [6, 7]], 9, "left", 4),
        ([[1, 2, 3, 4, 5]], -1, "right", 4), ([[[1, 2], [3, 4]], [[5, 6]]], 0, "right", None),
        ([[[1, 2], [3, 4]], [[5, 6]]], 0, "left", None),
    ]
    for _ in range(5):
        n = rs.randint(1, 6)
        specs.append(([rs.randint(0, 99, size=rs.randint(1, 9)).tolist() for _ in range(n)], int(rs.randint(0, 5)), ["left", "right"][rs.randint(2)], [None, 8][rs.randint(2)]))
    for rows, val, side, mult in specs:
        out = pad([torch.tensor(r) for r in rows], padding_value=val, padding_side=side, pad_to_multiple_of=mult)
        cases.append({"rows": rows, "padding_value": val, "padding_side": side, "pad_to_multiple_of": mult, "out": out.tolist()})
    with open(os.path.join(OUT, "pad.json"), "w") as f:
        json.dump({"meta": meta(), "cases": cases}, f)
    print(f"pad.json: {len(cases)} cases")


# ----------------------------------------------------------------------------------------------
# tiny HF model
# ----------------------------------------------------------------------------------------------
def hf_name(name: str) -> str:
    if name.startswith("visual."):
        return "model." + name
    if name.startswith("model."):
        return "model.language_model." + name[len("model."):]
    return name


def build_hf_model_qwen2vl(cfg: dict, weights: dict[str, np.ndarray]):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration

    t, v = cfg["text"], cfg["vision"]
    hf_cfg = Qwen2VLConfig(
        text_config=dict(
            vocab_size=t["vocab_size"], hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"],
            num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
            num_key_value_heads=t["num_key_value_heads"], rms_norm_eps=t["rms_norm_eps"],
            rope_parameters={"rope_type": "default", "rope_theta": t["rope_theta"], "mrope_section": t["mrope_section"]},
            pad_token_id=cfg["pad_token_id"], eos_token_id=cfg["eos_token_id"], bos_token_id=None,
        ),
        vision_config=dict(
            depth=v["depth"], embed_dim=v["hidden_size"], hidden_size=v["out_hidden_size"], mlp_ratio=v["intermediate_size"] // v["hidden_size"],
            num_heads=v["num_heads"], in_channels=v["in_channels"], patch_size=v["patch_size"], spatial_merge_size=v["spatial_merge_size"],
            temporal_patch_size=v["temporal_patch_size"],
        ),
        image_token_id=cfg["image_token_id"], video_token_id=cfg["video_token_id"],
        vision_start_token_id=cfg["vision_start_token_id"], vision_end_token_id=cfg["vision_end_token_id"],
        tie_word_embeddings=cfg["tie_word_embeddings"],
    )
    hf_cfg._attn_implementation = "eager"
    model = Qwen2VLForConditionalGeneration(hf_cfg)
    sd = {hf_name(k): torch.from_numpy(a.copy()) for k, a in weights.items()}
    if cfg["tie_word_embeddings"]:
        sd["lm_head.weight"] = sd["model.language_model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in m for m in missing), missing
    model.config._attn_implementation = "eager"
    model.float()
    return model


def build_hf_model(cfg: dict, weights: dict[str, np.ndarray]):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration

    if cfg["vision"].get("arch") == "qwen2_vl":
        return build_hf_model_qwen2vl(cfg, weights)
    t, v = cfg["text"], cfg["vision"]
    hf_cfg = Qwen2_5_VLConfig(
        text_config=dict(
            vocab_size=t["vocab_size"], hidden_size=t["hidden_size"], intermediate_size=t["intermediate_size"],
            num_hidden_layers=t["num_hidden_layers"], num_attention_heads=t["num_attention_heads"],
            num_key_value_heads=t["num_key_value_heads"], rms_norm_eps=t["rms_norm_eps"],
            rope_parameters={"rope_type": "default", "rope_theta": t["rope_theta"], "mrope_section": t["mrope_section"]},
            pad_token_id=cfg["pad_token_id"], eos_token_id=cfg["eos_token_id"], bos_token_id=None,
        ),
        vision_config=dict(
            depth=v["depth"], hidden_size=v["hidden_size"], intermediate_size=v["intermediate_size"], num_heads=v["num_heads"],
            in_channels=v["in_channels"], patch_size=v["patch_size"], spatial_merge_size=v["spatial_merge_size"],
            temporal_patch_size=v["temporal_patch_size"], window_size=v["window_size"], out_hidden_size=v["out_hidden_size"],
            fullatt_block_indexes=list(v["fullatt_block_indexes"]),
        ),
        image_token_id=cfg["image_token_id"], video_token_id=cfg["video_token_id"],
        vision_start_token_id=cfg["vision_start_token_id"], vision_end_token_id=cfg["vision_end_token_id"],
        tie_word_embeddings=cfg["tie_word_embeddings"],
    )
    hf_cfg._attn_implementation = "eager"
    model = Qwen2_5_VLForConditionalGeneration(hf_cfg)
    sd = {hf_name(k): torch.from_numpy(a.copy()) for k, a in weights.items()}
    if cfg["tie_word_embeddings"]:
        sd["lm_head.weight"] = sd["model.language_model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in m for m in missing), missing
    model.config._attn_implementation = "eager"
    model.float()
    return model


CANNED = [
    "<think>surface looks uniform</think><answer>no</answer>",
    "<think>a line on the left</think><location>upper left</location><type>scratch</type><answer>yes</answer>",
    "<think>dark blob</think><location>center</location><type>stain</type><answer>yes</answer>",
    "<think>hmm</think><location>bottom right</location><type>hole</type><answer>no</answer>",
    "no tags at all",
    "<think>x</think><location>top left corner</location><type>surface scratch</type><answer>yes</answer>",
    "<think>y</think><location>left</location><type>structural anomaly</type><answer>yes</answer>",
    "<think>z</think><location>top</location><type>scrach</type><answer>yes</answer> extra",
]
SOLUTION = "<think>gt</think><location>top left</location><type>scratch</type><answer>yes</answer>"


class FakeProcessor:
    """Stands in for AutoProcessor: returns preset tensors (the reference only forwards them)."""

    def __init__(self, cfg, batch, completions_text, n_completion):
        self.cfg = cfg
        # transformers>=5 derives M-RoPE from `mm_token_type_ids` (the pinned
        # 4.51.3 derives it from input_ids).  The trainer swaps in prompt+completion input_ids but
        # forwards the processor's other tensors untouched, so the mock emits the type ids already at
        # prompt+completion length (completion tokens are text = 0).
        batch = dict(batch)
        tt = batch["mm_token_type_ids"]
        batch["mm_token_type_ids"] = torch.cat([tt, torch.zeros(tt.shape[0], n_completion, dtype=tt.dtype)], 1)
        self.batch = batch
        self.pad_token_id = cfg["pad_token_id"]
        self.eos_token_id = cfg["eos_token_id"]
        self.tokenizer = self
        self._texts = completions_text
        self.chat_template = "x"

    def apply_chat_template(self, conversation, **kw):
        return "PROMPT"

    def __call__(self, text=None, images=None, **kw):
        from transformers import BatchFeature

        return BatchFeature({k: v.clone() for k, v in self.batch.items()})

    def batch_decode(self, ids, skip_special_tokens=True):
        return list(self._texts[: len(ids)])


class FakeLLM:
    def __init__(self, completions):
        self.completions = completions

    def generate(self, prompts, sampling_params=None, use_tqdm=False):
        assert len(prompts) == len(self.completions), (len(prompts), len(self.completions))
        return [types.SimpleNamespace(outputs=[types.SimpleNamespace(token_ids=list(c))]) for c in self.completions]


def make_trainer(SCGRPOTrainer, reward, cfg, model_ref, batch, completions, texts, G, max_completion_length):
    t = SCGRPOTrainer.__new__(SCGRPOTrainer)
    dev = torch.device("cpu")
    t.accelerator = types.SimpleNamespace(device=dev, process_index=0, is_main_process=True, gather_for_metrics=lambda x: x, unwrap_model=lambda m: m)
    t.processing_class = FakeProcessor(cfg, batch, texts, max(len(c) for c in completions))
    t.use_vllm = True
    t.llm = FakeLLM(completions)
    t.sampling_params = None
    t.num_generations = G
    t.max_prompt_length = 4096
    t.max_completion_length = max_completion_length
    t.beta = 0.04
    t.ref_model = model_ref
    t.model_id = "tiny-qwen2.5-vl"
    t.reward_funcs = [reward.accuracy_reward, reward.consistency_reward]
    t.reward_processing_classes = [None, None]
    t._metrics = defaultdict(list)
    t._last_loaded_step = 0
    t.state = types.SimpleNamespace(global_step=0)
    t.args = types.SimpleNamespace(device=dev, past_index=-1, ds3_gather_for_generation=True)
    t.is_deepspeed_enabled = False
    t._past = None
    return t


def capture_locals(fn, code_name):
    """Run fn(); return (result, f_locals of the frame named code_name at its return)."""
    grabbed = {}

    def prof(frame, event, arg):
        if event == "return" and frame.f_code.co_name == code_name:
            grabbed.update(frame.f_locals)

    sys.setprofile(prof)
    try:
        res = fn()
    finally:
        sys.setprofile(None)
    return res, grabbed


def tiny_batch(cfg, grids, n_texts, seed):
    rows = [fx.synth_prompt(g, n, cfg, seed + i) for i, (g, n) in enumerate(zip(grids, n_texts))]
    ids, mask = fx.left_pad(rows, cfg["pad_token_id"])
    pv = fx.synth_pixel_values(grids, cfg, seed=seed)
    return {
        "input_ids": torch.from_numpy(ids),
        "attention_mask": torch.from_numpy(mask),
        "pixel_values": torch.from_numpy(pv),
        "image_grid_thw": torch.tensor(grids, dtype=torch.long),
        "mm_token_type_ids": torch.from_numpy((ids == cfg["image_token_id"]).astype(np.int32)),
    }


GRAD_FULL = [
    "model.norm.weight", "model.layers.1.self_attn.k_proj.bias", "model.layers.0.input_layernorm.weight",
    "visual.merger.ln_q.weight", "visual.blocks.0.attn.qkv.bias", "visual.blocks.3.norm2.weight", "visual.merger.mlp.2.bias",
]


def gen_sc_grpo(SCGRPOTrainer, reward, G, C, eos_rows, name, seed, perturb_scale=0.02, truncate=0, cfg=None, cfg_name="fixture_util.TINY", grad_full=None):
    """perturb_scale: distance policy <-> frozen reference (0.02: KL ~ 3e-3; 0.25: KL ~ 0.1, where a relative tolerance on KL and loss is a real
    check).  truncate > 0: max_prompt_length = P - truncate, i.e. the reference's left truncation (sc_grpo_trainer.py:630-634) cuts that many
    leading text tokens of the prompt (pixel tensors untouched, M-RoPE positions recomputed on the truncated ids).
    cfg: the fixture configuration (default TINY; TINY7 = the 7B structure: untied lm_head, GQA group 7 -- BASELINE config 4)."""
    cfg = cfg or fx.TINY
    w_ref = fx.make_weights(cfg, seed=0)
    w_pol = fx.perturb_weights(w_ref, seed=1, scale=perturb_scale)
    ref = build_hf_model(cfg, w_ref).eval()
    pol = build_hf_model(cfg, w_pol).train()
    for p in ref.parameters():
        p.requires_grad_(False)
    grid = (1, 16, 12)
    batch = tiny_batch(cfg, [grid], [9], seed)
    comps = fx.synth_completions(G, C, cfg, seed + 100, eos_rows)
    texts = [CANNED[i % len(CANNED)] for i in range(G)]
    t = make_trainer(SCGRPOTrainer, reward, cfg, ref, batch, comps, texts, G, C)
    P_full = batch["input_ids"].shape[1]
    if truncate:
        t.max_prompt_length = P_full - truncate
        # the mock processor hands `mm_token_type_ids` over at prompt+completion length; the trainer truncates ids / mask only, so cut it here the same way
        tt = t.processing_class.batch["mm_token_type_ids"]
        t.processing_class.batch["mm_token_type_ids"] = tt[:, truncate:]
    inputs = [{"prompt": [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "q"}]}], "image": [object()], "solution": SOLUTION}]
    # the trainer opens str images with PIL; pass non-str objects straight through (sc_grpo_trainer.py:610)
    with contextlib.redirect_stdout(io.StringIO()):
        loss, loc = capture_locals(lambda: t.compute_loss(pol, inputs), "compute_loss")
    loss.backward()
    grads = {k: p.grad for k, p in pol.named_parameters() if p.grad is not None}
    inv = {hf_name(k): k for k in fx.param_shapes(cfg)}
    gnorm = {inv[k]: float(g.norm()) for k, g in grads.items() if k in inv}
    out = {
        "meta": json.dumps({**meta(), "G": G, "C": C, "grid": grid, "n_text": 9, "seed": seed, "beta": 0.04, "eos_rows": eos_rows or {}, "perturb_scale": perturb_scale, "truncate": truncate,
                            "max_prompt_length": int(t.max_prompt_length), "config": cfg_name, "weights": f"fixture_util.make_weights({cfg_name.split('.')[-1]},0) / perturb_weights(.,1,scale={perturb_scale})"}),
        "completion_ids": loc["completion_ids"].numpy(),
        "prompt_completion_ids": loc["prompt_completion_ids"].numpy(),
        "attention_mask": loc["attention_mask"].numpy(),
        "completion_mask": loc["completion_mask"].numpy(),
        "per_token_logps": loc["per_token_logps"].detach().numpy(),
        "ref_per_token_logps": loc["ref_per_token_logps"].numpy(),
        "per_token_kl": loc["per_token_kl"].detach().numpy(),
        "rewards_per_func": loc["rewards_per_func"].numpy(),
        "advantages": loc["advantages"].numpy(),
        "loss": np.float64(loss.item()),
        "metric_completion_length": np.float64(t._metrics["completion_length"][0]),
        "metric_reward": np.float64(t._metrics["reward"][0]),
        "metric_reward_std": np.float64(t._metrics["reward_std"][0]),
        "metric_kl": np.float64(t._metrics["kl"][0]),
        "metric_rewards_accuracy": np.float64(t._metrics["rewards/accuracy_reward"][0]),
        "metric_rewards_format": np.float64(t._metrics["rewards/consistency_reward"][0]),
        "grad_norm_names": np.array(sorted(gnorm)),
        "grad_norms": np.array([gnorm[k] for k in sorted(gnorm)], dtype=np.float64),
        "completions_text": np.array(texts),
        "solution": np.array(SOLUTION),
    }
    for k in (grad_full or GRAD_FULL):
        out["grad::" + k] = grads[hf_name(k)].numpy()
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(f"{name}: loss={loss.item():.8f} reward={t._metrics['reward'][0]:.4f} kl={t._metrics['kl'][0]:.6f} len={t._metrics['completion_length'][0]}")


def tiny_reward_model(tokenizer, seed=5):
    """A 2-layer Qwen2 sequence classifier (num_labels=1) on the character-level test tokenizer: the `PreTrainedModel` kind of reward function."""
    from transformers import Qwen2Config, Qwen2ForSequenceClassification
    torch.manual_seed(seed)
    cfg = Qwen2Config(vocab_size=len(tokenizer), hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      max_position_embeddings=512, num_labels=1, pad_token_id=None, tie_word_embeddings=False)
    m = Qwen2ForSequenceClassification(cfg)
    m.config._name_or_path = "local/tiny-reward-model"
    return m.float().eval()


def gen_reward_model(SCGRPOTrainer, reward):
    """The reference's compute_loss with a reward MODEL among the reward functions (sc_grpo_trainer.py:760-772): records what the model was fed (texts through
    the reward tokenizer's chat template, right padding) and the rewards it returned, plus the model's weights so the test can rebuild it."""
    G, C = 4, 10
    cfg = fx.TINY
    w_ref = fx.make_weights(cfg, seed=0)
    ref = build_hf_model(cfg, w_ref).eval()
    pol = build_hf_model(cfg, fx.perturb_weights(w_ref, seed=1, scale=0.25)).train()
    for p in ref.parameters():
        p.requires_grad_(False)
    grid = (1, 16, 12)
    batch = tiny_batch(cfg, [grid], [9], 31)
    comps = fx.synth_completions(G, C, cfg, 131, {1: 6})
    texts = [CANNED[i % len(CANNED)] for i in range(G)]
    t = make_trainer(SCGRPOTrainer, reward, cfg, ref, batch, comps, texts, G, C)
    tok = fx.local_qwen2vl_processor().tokenizer
    tok.chat_template = fx.QWEN2VL_CHAT_TEMPLATE           # (the processor holds the template; a reward tokenizer carries its own)
    rm = tiny_reward_model(tok)
    # what the reference's constructor does to a reward model / its tokenizer (sc_grpo_trainer.py:250-261)
    if tok.pad_token_id is None:
        tok.pad_token = tok.eos_token
    rm.config.pad_token_id = tok.pad_token_id
    t.reward_funcs = [reward.accuracy_reward, rm]
    t.reward_processing_classes = [None, tok]
    seen = {}
    orig = tok.__class__.__call__

    def spy(self, text=None, *a, **kw):
        out = orig(self, text, *a, **kw)
        seen["texts"], seen["ids"], seen["mask"] = list(text), out["input_ids"].numpy().copy(), out["attention_mask"].numpy().copy()
        return out
    tok.__class__.__call__ = spy
    inputs = [{"prompt": [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "Is there a defect?"}]}], "image": [object()], "solution": SOLUTION}]
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            loss, loc = capture_locals(lambda: t.compute_loss(pol, inputs), "compute_loss")
    finally:
        tok.__class__.__call__ = orig
    out = {
        "meta": json.dumps({**meta(), "G": G, "C": C, "reward_model": "tools/make_golden.py tiny_reward_model (Qwen2ForSequenceClassification, weights stored here)",
                            "tokenizer": "fixture_util.local_qwen2vl_processor().tokenizer with chat_template = QWEN2VL_CHAT_TEMPLATE"}),
        "rewards_per_func": loc["rewards_per_func"].numpy(),
        "advantages": loc["advantages"].numpy(),
        "texts": np.array(seen["texts"]), "input_ids": seen["ids"], "attention_mask": seen["mask"],
        "completions_text": np.array(texts), "solution": np.array(SOLUTION),
        "metric_names": np.array(sorted(k for k in t._metrics if k.startswith("rewards/"))),
        "loss": np.float64(loss.item()),
    }
    for k, v in rm.state_dict().items():
        out["rm::" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "reward_model.npz"), **out)
    print("reward_model:", loc["rewards_per_func"].numpy().tolist(), out["metric_names"].tolist())


def gen_logps_padded(SCGRPOTrainer):
    """_get_per_token_logps on two different left-padded prompts with two different grids + the
    intermediate activations used to pin the oracle layer by layer."""
    cfg = fx.TINY
    w = fx.make_weights(cfg, seed=0)
    model = build_hf_model(cfg, w).eval()
    grids = [(1, 16, 12), (1, 8, 8)]
    batch = tiny_batch(cfg, grids, [5, 17], seed=77)
    comp = np.array(fx.synth_completions(2, 6, cfg, 5)).astype(np.int64)
    ids = torch.cat([batch["input_ids"], torch.from_numpy(comp)], 1)
    mask = torch.cat([batch["attention_mask"], torch.ones(2, 6, dtype=torch.long)], 1)
    inputs = dict(batch, input_ids=ids, attention_mask=mask, mm_token_type_ids=(ids == cfg["image_token_id"]).int())
    holder = types.SimpleNamespace(model_id="tiny-qwen2.5-vl")
    with torch.no_grad():
        logps = SCGRPOTrainer._get_per_token_logps(holder, model, **inputs)
        vis = model.model.visual(inputs["pixel_values"], grid_thw=inputs["image_grid_thw"])
        pos, deltas = model.model.get_rope_index(ids, mm_token_type_ids=inputs["mm_token_type_ids"], image_grid_thw=inputs["image_grid_thw"], attention_mask=mask)
        out = model(**inputs, output_hidden_states=True)
    np.savez_compressed(
        os.path.join(OUT, "logps_padded.npz"),
        meta=json.dumps({**meta(), "grids": grids, "n_text": [5, 17], "seed": 77}),
        input_ids=ids.numpy(), attention_mask=mask.numpy(), pixel_values=inputs["pixel_values"].numpy(),
        image_grid_thw=inputs["image_grid_thw"].numpy(), per_token_logps=logps.numpy(),
        image_embeds=vis.pooler_output.numpy(), vit_last_hidden=vis.last_hidden_state.numpy(),
        position_ids=pos.numpy(), rope_deltas=deltas.numpy(),
        logits=out.logits.numpy().astype(np.float32),
        hidden_0=out.hidden_states[0].numpy(), hidden_1=out.hidden_states[1].numpy(), hidden_last=out.hidden_states[-1].numpy(),
    )
    print("logps_padded.npz:", tuple(logps.shape))


def gen_logps_7b_like(SCGRPOTrainer):
    """Same as logps_padded but on TINY7 (untied lm_head, GQA group 7: the structural deltas of Qwen2.5-VL-7B)."""
    cfg = fx.TINY7
    model = build_hf_model(cfg, fx.make_weights(cfg, seed=0)).eval()
    grids = [(1, 8, 12), (1, 16, 8)]
    batch = tiny_batch(cfg, grids, [7, 3], seed=91)
    comp = np.array(fx.synth_completions(2, 5, cfg, 6)).astype(np.int64)
    ids = torch.cat([batch["input_ids"], torch.from_numpy(comp)], 1)
    mask = torch.cat([batch["attention_mask"], torch.ones(2, 5, dtype=torch.long)], 1)
    inputs = dict(batch, input_ids=ids, attention_mask=mask, mm_token_type_ids=(ids == cfg["image_token_id"]).int())
    holder = types.SimpleNamespace(model_id="tiny-qwen2.5-vl")
    with torch.no_grad():
        logps = SCGRPOTrainer._get_per_token_logps(holder, model, **inputs)
    np.savez_compressed(os.path.join(OUT, "logps_7b_like.npz"), meta=json.dumps({**meta(), "grids": grids, "n_text": [7, 3], "seed": 91, "config": "fixture_util.TINY7"}),
                        input_ids=ids.numpy(), attention_mask=mask.numpy(), image_grid_thw=inputs["image_grid_thw"].numpy(), per_token_logps=logps.numpy())
    print("logps_7b_like.npz:", tuple(logps.shape))


def gen_vision_index():
    from transformers import vision_utils as vu

    cfg = fx.TINY
    model = build_hf_model(cfg, fx.make_weights(cfg, seed=0))
    cases = []
    for grids in ([(1, 32, 32)], [(1, 16, 12)], [(1, 8, 8), (1, 20, 14)], [(1, 6, 34), (1, 18, 18), (1, 4, 4)]):
        g = torch.tensor(grids)
        wi, cu = vu.get_vision_window_index(g, spatial_merge_size=2, window_size=112, patch_size=14)
        pid = vu.get_vision_position_ids(g, 2)
        cs = vu.get_vision_cu_seqlens(g)
        cases.append({"grid_thw": grids, "window_index": wi.tolist(), "cu_window_seqlens": cu.tolist(), "position_ids": pid.tolist(), "cu_seqlens": cs.tolist()})
    rope = []
    for grids, n_text in (([(1, 32, 32)], [251]), ([(1, 16, 12), (1, 8, 8)], [4, 30])):
        b = tiny_batch(cfg, grids, n_text, seed=3)
        pos, d = model.model.get_rope_index(b["input_ids"], mm_token_type_ids=b["mm_token_type_ids"], image_grid_thw=b["image_grid_thw"], attention_mask=b["attention_mask"])
        rope.append({"grid_thw": grids, "input_ids": b["input_ids"].tolist(), "attention_mask": b["attention_mask"].tolist(), "position_ids": pos.tolist(), "rope_deltas": d.tolist()})
    with open(os.path.join(OUT, "vision_index.json"), "w") as f:
        json.dump({"meta": meta(), "window": cases, "rope_index": rope}, f)
    print("vision_index.json:", len(cases), "window cases,", len(rope), "rope cases")


def gen_greedy():
    cfg = fx.TINY
    model = build_hf_model(cfg, fx.make_weights(cfg, seed=0)).eval()
    grids = [(1, 16, 12), (1, 8, 8)]
    b = tiny_batch(cfg, grids, [5, 17], seed=77)
    with torch.no_grad():
        out = model.generate(**b, do_sample=False, max_new_tokens=12, min_new_tokens=12, use_cache=True, pad_token_id=cfg["pad_token_id"], eos_token_id=None)
        # also the per-step top-2 logit margin (fp32), so a bf16 device path can tell a real
        # mismatch from a near-tie
        ids = out
        mask = torch.cat([b["attention_mask"], torch.ones(2, 12, dtype=torch.long)], 1)
        logits = model(input_ids=ids, attention_mask=mask, pixel_values=b["pixel_values"], image_grid_thw=b["image_grid_thw"], mm_token_type_ids=(ids == cfg["image_token_id"]).int()).logits
        P = b["input_ids"].shape[1]
        step_logits = logits[:, P - 1 : P - 1 + 12].float()
        top2 = step_logits.topk(2, -1).values
    np.savez_compressed(
        os.path.join(OUT, "greedy.npz"), meta=json.dumps({**meta(), "grids": grids, "n_text": [5, 17], "seed": 77, "new_tokens": 12}),
        prompt_ids=b["input_ids"].numpy(), prompt_mask=b["attention_mask"].numpy(), sequences=out.numpy(),
        margin=(top2[..., 0] - top2[..., 1]).numpy(),
    )
    print("greedy.npz:", out[:, P:].tolist(), "min margin", float((top2[..., 0] - top2[..., 1]).min()))


def hf_param_groups(model):
    """(decay, no_decay) parameter lists exactly as transformers.Trainer.create_optimizer forms them (get_decay_parameter_names of the installed version)."""
    from transformers import Trainer
    names = set(Trainer.get_decay_parameter_names(Trainer.__new__(Trainer), model))
    decay = [p for n, p in model.named_parameters() if n in names and p.requires_grad]
    no_decay = [p for n, p in model.named_parameters() if n not in names and p.requires_grad]
    return decay, no_decay


LR20 = 5e-5


def curve_bf16_weights(model, inputs, groups_fn, steps=20, lr=LR20, trainable=None):
    """The reference's mixed precision (`--bf16` + DeepSpeed: bf16 parameters in the forward / backward, fp32 master copy under AdamW) on the fp32 HF model: before
    every step the model's parameters are set to bf16(master); the gradients go to the master copy.  At this learning rate an Adam step (5e-5) is smaller than
    half a bf16 spacing of a typical weight (|w| ~ 0.05: 1.2e-4), so the bf16 copy moves in stair steps -- the curve differs from the pure-fp32 one by far more than any
    kernel error, and it is THIS curve a bf16 run of the reference follows.  Activations stay fp32 here (their bf16 rounding is the small part: a few 1e-3 of the loss)."""
    params = [(n, p) for n, p in model.named_parameters() if (trainable is None or trainable(n))]
    master = [p.detach().clone().requires_grad_(True) for _, p in params]
    dec, nod = groups_fn(model)
    ids_dec = {id(p) for p in dec}
    opt = torch.optim.AdamW([{"params": [m for m, (_, p) in zip(master, params) if id(p) in ids_dec], "weight_decay": 0.1},
                             {"params": [m for m, (_, p) in zip(master, params) if id(p) not in ids_dec], "weight_decay": 0.0}], lr=lr, betas=(0.9, 0.999), eps=1e-8)
    losses = []
    for _ in range(steps):
        with torch.no_grad():
            for m, (_, p) in zip(master, params):
                p.copy_(m.to(torch.bfloat16).float())
        model.zero_grad()
        loss = model(**inputs).loss
        loss.backward()
        for m, (_, p) in zip(master, params):
            m.grad = p.grad.detach().clone()
        opt.step()
        losses.append(loss.item())
    return losses


def gen_sft():
    """PA-SFT numeric oracle: HF forward(labels) loss + 3 AdamW steps (lr 1e-3 for visible motion,
    wd 0.1 as PA_SFT_*.sh:38-44, betas/eps = torch defaults = HF Trainer defaults)."""
    cfg = fx.TINY
    model = build_hf_model(cfg, fx.make_weights(cfg, seed=0)).train()
    grids = [(1, 16, 12), (1, 8, 8)]
    b = tiny_batch(cfg, grids, [5, 17], seed=11)
    resp = np.array(fx.synth_completions(2, 8, cfg, 9)).astype(np.int64)
    ids = torch.cat([b["input_ids"], torch.from_numpy(resp)], 1)
    mask = torch.cat([b["attention_mask"], torch.ones(2, 8, dtype=torch.long)], 1)
    labels = ids.clone()
    labels[:, : b["input_ids"].shape[1]] = -100  # prompt tokens masked (llamafactory supervised.py:34-87)
    inputs = dict(input_ids=ids, attention_mask=mask, pixel_values=b["pixel_values"], image_grid_thw=b["image_grid_thw"], mm_token_type_ids=(ids == cfg["image_token_id"]).int(), labels=labels)
    decay, no_decay = hf_param_groups(model)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.1}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = model(**inputs).loss
        loss.backward()
        opt.step()
        losses.append(loss.item())
    # the north star's "loss curve": 20 AdamW steps from the same start at lr 5e-5 (each step moves a weight by <= 5e-5 against |w| ~ 0.05; the reference script uses 1e-5 / 2e-5: the loss falls
    # visibly without the trajectory turning chaotic), same groups / weight decay
    model = build_hf_model(cfg, fx.make_weights(cfg, seed=0)).train()
    decay, no_decay = hf_param_groups(model)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.1}, {"params": no_decay, "weight_decay": 0.0}], lr=LR20, betas=(0.9, 0.999), eps=1e-8)
    losses20 = []
    for _ in range(20):
        opt.zero_grad()
        loss = model(**inputs).loss
        loss.backward()
        opt.step()
        losses20.append(loss.item())
    losses20_bf16w = curve_bf16_weights(build_hf_model(cfg, fx.make_weights(cfg, seed=0)).train(), inputs, hf_param_groups)
    np.savez_compressed(
        os.path.join(OUT, "sft.npz"), meta=json.dumps({**meta(), "grids": grids, "n_text": [5, 17], "seed": 11, "lr": 1e-3, "wd": 0.1, "lr20": LR20, "no_decay": "transformers.Trainer.get_decay_parameter_names (tests/golden/sft_freeze.json: decay_parameters)"}),
        input_ids=ids.numpy(), attention_mask=mask.numpy(), labels=labels.numpy(), pixel_values=b["pixel_values"].numpy(),
        image_grid_thw=b["image_grid_thw"].numpy(), losses=np.array(losses, dtype=np.float64), losses20=np.array(losses20, dtype=np.float64),
        losses20_bf16w=np.array(losses20_bf16w, dtype=np.float64),
    )
    print("sft.npz: losses", losses, "\n  20 steps at lr", LR20, [round(x, 4) for x in losses20], "\n  bf16 weights + fp32 master", [round(x, 4) for x in losses20_bf16w])


def gen_qwen2vl(SCGRPOTrainer):
    """BASELINE.json config 1 (Qwen2-VL PA-SFT, 4 samples): tiny Qwen2VLForConditionalGeneration -- image embeds,
    per-token logps through the reference's `_get_per_token_logps`, and a 3-step AdamW loss curve on 4 samples."""
    cfg = fx.TINY_Q2
    model = build_hf_model(cfg, fx.make_weights(cfg, seed=0)).eval()
    grids = [(1, 16, 12), (1, 8, 8), (1, 4, 6), (1, 10, 10)]
    b = tiny_batch(cfg, grids, [5, 17, 9, 2], seed=13)
    resp = np.array(fx.synth_completions(4, 8, cfg, 10)).astype(np.int64)
    ids = torch.cat([b["input_ids"], torch.from_numpy(resp)], 1)
    mask = torch.cat([b["attention_mask"], torch.ones(4, 8, dtype=torch.long)], 1)
    mm = (ids == cfg["image_token_id"]).int()
    holder = types.SimpleNamespace(model_id="tiny-qwen2-vl")
    with torch.no_grad():
        logps = SCGRPOTrainer._get_per_token_logps(holder, model, input_ids=ids, attention_mask=mask, pixel_values=b["pixel_values"],
                                                   image_grid_thw=b["image_grid_thw"], mm_token_type_ids=mm)
        vis = model.model.visual(b["pixel_values"], grid_thw=b["image_grid_thw"])
        embeds = vis.pooler_output if hasattr(vis, "pooler_output") else vis
    labels = ids.clone()
    labels[:, : b["input_ids"].shape[1]] = -100
    inputs = dict(input_ids=ids, attention_mask=mask, pixel_values=b["pixel_values"], image_grid_thw=b["image_grid_thw"], mm_token_type_ids=mm, labels=labels)
    model.train()
    decay, no_decay = hf_param_groups(model)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.1}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = model(**inputs).loss
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.savez_compressed(
        os.path.join(OUT, "qwen2vl_sft.npz"), meta=json.dumps({**meta(), "grids": grids, "n_text": [5, 17, 9, 2], "seed": 13, "lr": 1e-3, "wd": 0.1, "config": "fixture_util.TINY_Q2"}),
        input_ids=ids.numpy(), attention_mask=mask.numpy(), labels=labels.numpy(), pixel_values=b["pixel_values"].numpy(),
        image_grid_thw=b["image_grid_thw"].numpy(), losses=np.array(losses, dtype=np.float64), per_token_logps=logps.numpy(),
        image_embeds=torch.as_tensor(embeds).numpy(),
    )
    print("qwen2vl_sft.npz: losses", losses, "logps", tuple(logps.shape))


def gen_prepare():
    """Host-side batch construction of compute_loss, REF train/stage_rl/trainer/sc_grpo_trainer.py:600-622, run with the reference's own
    `maybe_apply_chat_template` (trl/trl/data_utils.py:172-227) and an offline Qwen2-VL processor (tests/fixture_util.local_qwen2vl_processor):
    rendered prompt text, left-padded token ids / mask, patch grids and pixel statistics per micro-batch -> tests/golden/prepare.json."""
    from trl.data_utils import maybe_apply_chat_template
    proc = fx.local_qwen2vl_processor(max_pixels=480000, min_pixels=3136)
    cases = []
    for inputs in fx.prepare_examples():
        prompts_text = [maybe_apply_chat_template(example, proc)["prompt"] for example in inputs]
        images = []
        for x in inputs:
            imgs = x["image"]
            images.extend(imgs)           # PIL objects pass straight through the reference's loader (REF:606-612 opens str paths only)
        enc = proc(text=prompts_text, images=images, return_tensors="pt", padding=True, padding_side="left", add_special_tokens=False)
        pv = enc["pixel_values"].double()
        cases.append({"prompts_text": prompts_text, "input_ids": enc["input_ids"].tolist(), "attention_mask": enc["attention_mask"].tolist(),
                      "image_grid_thw": enc["image_grid_thw"].tolist(), "pixel_shape": list(enc["pixel_values"].shape),
                      "pixel_sum": float(pv.sum()), "pixel_abs_sum": float(pv.abs().sum()), "pixel_head": enc["pixel_values"][0, :8].tolist(),
                      "pixel_tail": enc["pixel_values"][-1, -8:].tolist()})
    with open(os.path.join(OUT, "prepare.json"), "w") as f:
        json.dump({"meta": {**meta(), "max_pixels": 480000, "min_pixels": 3136}, "cases": cases}, f)
    print("prepare.json:", [(len(c["input_ids"]), len(c["input_ids"][0]), c["image_grid_thw"]) for c in cases])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    reward, type_reward, location_reward, SCGRPOTrainer, pad = import_reference()
    only = set(sys.argv[1:])
    if not only or "rewards" in only:
        gen_rewards(reward, type_reward, location_reward)
    if not only or "pad" in only:
        gen_pad(pad)
    if not only or "prepare" in only:
        gen_prepare()
    if not only or "index" in only:
        gen_vision_index()
    if not only or "grpo" in only:
        gen_sc_grpo(SCGRPOTrainer, reward, G=4, C=10, eos_rows={1: 6, 3: 0}, name="sc_grpo_g4.npz", seed=21)
        gen_sc_grpo(SCGRPOTrainer, reward, G=8, C=12, eos_rows={0: 11, 2: 3, 5: 7}, name="sc_grpo_g8.npz", seed=22)
    if not only or "grpo_far" in only:
        gen_sc_grpo(SCGRPOTrainer, reward, G=8, C=12, eos_rows={1: 9, 4: 2, 6: 5}, name="sc_grpo_g8_far.npz", seed=23, perturb_scale=0.25)
        gen_sc_grpo(SCGRPOTrainer, reward, G=4, C=10, eos_rows={0: 4, 2: 8}, name="sc_grpo_trunc.npz", seed=24, perturb_scale=0.25, truncate=2)
    if not only or "grpo7" in only:
        # BASELINE config 4's structure (untied head, 7 query heads per kv head) through the reference's compute_loss: policy far from the reference (KL ~ 0.2)
        gen_sc_grpo(SCGRPOTrainer, reward, G=8, C=12, eos_rows={1: 9, 4: 2, 6: 5}, name="sc_grpo_7b_like.npz", seed=25, perturb_scale=0.08, cfg=fx.TINY7, cfg_name="fixture_util.TINY7",
                    grad_full=["model.norm.weight", "model.layers.1.self_attn.k_proj.bias", "model.layers.1.self_attn.q_proj.bias", "model.layers.0.input_layernorm.weight",
                               "model.layers.0.self_attn.v_proj.weight", "visual.merger.ln_q.weight", "visual.blocks.1.norm2.weight", "visual.merger.mlp.2.bias"])
    if not only or "grpo_q2" in only:
        # the Qwen2-VL structure (LayerNorm / QuickGELU ViT without windows; the reference's SC_GRPO_Qwen_Instruct_2_VL.sh) through the reference's compute_loss
        gen_sc_grpo(SCGRPOTrainer, reward, G=4, C=10, eos_rows={0: 7, 2: 3}, name="sc_grpo_qwen2vl.npz", seed=26, perturb_scale=0.25, cfg=fx.TINY_Q2, cfg_name="fixture_util.TINY_Q2",
                    grad_full=["model.norm.weight", "model.layers.1.self_attn.k_proj.bias", "model.layers.0.input_layernorm.weight", "visual.merger.ln_q.weight",
                               "visual.blocks.0.attn.qkv.bias", "visual.blocks.2.norm2.weight", "visual.blocks.1.norm1.bias", "visual.merger.mlp.2.bias"])
    if not only or "reward_model" in only:
        gen_reward_model(SCGRPOTrainer, reward)
    if not only or "logps" in only:
        gen_logps_padded(SCGRPOTrainer)
    if not only or "logps7" in only:
        gen_logps_7b_like(SCGRPOTrainer)
    if not only or "greedy" in only:
        gen_greedy()
    if not only or "sft" in only:
        gen_sft()
    if not only or "qwen2vl" in only:
        gen_qwen2vl(SCGRPOTrainer)


if __name__ == "__main__":
    main()
