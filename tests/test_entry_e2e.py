"""The two reference entry points run END TO END on the GPU from what a user has on disk: an HF checkpoint directory (config.json in the layout the
published checkpoints use + safetensors in the published names), a dataset manifest, image files.  Only the processor is a stand-in (the published
tokenizer / video processor files cannot be fetched offline): `AutoProcessor.from_pretrained` is pointed at the offline Qwen2-VL processor of
tests/fixture_util.py, which really renders the chat template, tokenises and patches the images."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fixture_util as fx  # noqa: E402
import iadr1_amd  # noqa: E402,F401
from iadr1_amd.params import ParamStore, VLMConfig  # noqa: E402
from iadr1_amd.trainer import load_checkpoint, save_checkpoint  # noqa: E402

DEV = torch.device("cuda", 0)


def _load(rel):
    spec = importlib.util.spec_from_file_location(os.path.basename(rel)[:-3] + "_e2e", os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _tiny_with_processor_ids(proc):
    tid = proc.tokenizer.convert_tokens_to_ids
    return dict(fx.TINY, image_token_id=tid("<|image_pad|>"), video_token_id=tid("<|video_pad|>"), vision_start_token_id=tid("<|vision_start|>"),
                vision_end_token_id=tid("<|vision_end|>"), eos_token_id=tid("<|im_end|>"), pad_token_id=tid("<|endoftext|>"))


def _qwen25vl_config_json(d, nested):
    """config.json as Qwen2.5-VL checkpoints carry it: flat (transformers 4.51.3, the reference's pin) or with `text_config` (5.x)."""
    t, v = d["text"], d["vision"]
    text = {"vocab_size": t["vocab_size"], "hidden_size": t["hidden_size"], "intermediate_size": t["intermediate_size"], "num_hidden_layers": t["num_hidden_layers"],
            "num_attention_heads": t["num_attention_heads"], "num_key_value_heads": t["num_key_value_heads"], "rms_norm_eps": t["rms_norm_eps"], "rope_theta": t["rope_theta"],
            "rope_scaling": {"type": "mrope", "mrope_section": t["mrope_section"]}, "hidden_act": "silu", "max_position_embeddings": 128000}
    vis = {"depth": v["depth"], "hidden_size": v["hidden_size"], "intermediate_size": v["intermediate_size"], "num_heads": v["num_heads"], "in_chans": v["in_channels"],
           "out_hidden_size": v["out_hidden_size"], "patch_size": v["patch_size"], "spatial_merge_size": v["spatial_merge_size"], "temporal_patch_size": v["temporal_patch_size"],
           "window_size": v["window_size"], "fullatt_block_indexes": v["fullatt_block_indexes"], "hidden_act": "silu", "tokens_per_second": 2}
    top = {"architectures": ["Qwen2_5_VLForConditionalGeneration"], "model_type": "qwen2_5_vl", "image_token_id": d["image_token_id"], "video_token_id": d["video_token_id"],
           "vision_start_token_id": d["vision_start_token_id"], "vision_end_token_id": d["vision_end_token_id"], "eos_token_id": d["eos_token_id"],
           "pad_token_id": d["pad_token_id"], "tie_word_embeddings": d["tie_word_embeddings"], "vision_config": vis, "torch_dtype": "bfloat16"}
    if nested:
        top["text_config"] = dict(text, model_type="qwen2_5_vl_text", eos_token_id=d["eos_token_id"], pad_token_id=d["pad_token_id"])
    else:
        top.update(text)
    return top


def _write_checkpoint(path, d, nested=False, seed=0):
    cfg = VLMConfig.from_dict(d)
    s = ParamStore(cfg, DEV, trainable=False)
    s.load_named(fx.make_weights(d, seed))
    save_checkpoint(s, path, _qwen25vl_config_json(d, nested))
    return cfg, s


@pytest.fixture
def offline_processor(monkeypatch):
    proc = fx.local_qwen2vl_processor()
    proc.save_pretrained = lambda *a, **k: None
    import transformers
    monkeypatch.setattr(transformers.AutoProcessor, "from_pretrained", classmethod(lambda cls, *a, **k: proc))
    return proc


@pytest.mark.parametrize("nested", [False, True])
def test_checkpoint_directory_of_the_published_layout_loads(tmp_path, offline_processor, nested):
    """config.json in both published layouts -> the same VLMConfig; every tensor of the safetensors file comes back bit-identical."""
    d = _tiny_with_processor_ids(offline_processor)
    cfg, s = _write_checkpoint(str(tmp_path / "Qwen2.5-VL-tiny"), d, nested)
    cfg2, s2 = load_checkpoint(str(tmp_path / "Qwen2.5-VL-tiny"), DEV, trainable=False)
    assert cfg2 == cfg
    a, b = s.export_named(), s2.export_named()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)


def test_sc_grpo_entry_point_end_to_end(tmp_path, offline_processor, monkeypatch, capsys):
    """`train/stage_rl/grpo_ad.py` with the flags of scripts/train/SC_GRPO/SC_GRPO_Qwen_Instruct_2_5_VL_3B.sh (paths swapped): manifest rows -> 1-image
    prompts (REF grpo_ad.py:135-181) -> chat template + PIL images + processor -> group rollout, reward plugins on the decoded text, SC-GRPO steps, HF-layout
    save.  Two optimizer steps over a 4-row manifest with gradient accumulation 2; the saved directory loads back and differs from the start."""
    d = _tiny_with_processor_ids(offline_processor)
    src = str(tmp_path / "Qwen2.5-VL-tiny-Instruct")
    _, s0 = _write_checkpoint(src, d)
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    rows = []
    for i in range(4):
        fx.synth_pil_image(112 + 28 * (i % 2), 84, 40 + i).save(str(img_dir / f"part_{i}.png"))
        rows.append({"image": f"part_{i}.png", "problem": "Are there any defects in the query image?",
                     "solution": "<think>x</think><location>top left</location><type>scratch</type><answer>Yes</answer>" if i % 2 else "<answer>No</answer>"})
    manifest = tmp_path / "expert_ad_tiny.json"
    manifest.write_text(json.dumps(rows))
    out = str(tmp_path / "out")
    m = _load("train/stage_rl/grpo_ad.py")

    def text_checksum_reward(prompts, completions, **kw):      # the shipped rewards score 0 on the noise a random-weight model writes: no advantage, no update
        assert set(kw) >= {"solution", "image", "problem"} and len(prompts) == len(completions) == 4
        return [float(sum(map(ord, c[0]["content"])) % 7) for c in completions]

    from iadr1_amd import rewards
    monkeypatch.setitem(rewards.REWARD_FUNCS, "format", text_checksum_reward)
    m.main(["--model_name_or_path", src, "--dataset_name", str(manifest), "--image_path", str(img_dir), "--output_dir", out, "--max_prompt_length", "1024",
            "--max_completion_length", "8", "--num_generations", "4", "--per_device_train_batch_size", "1", "--gradient_accumulation_steps", "2", "--learning_rate", "1e-3",
            "--num_train_epochs", "1", "--logging_steps", "1", "--save_steps", "100", "--bf16", "--attn_implementation", "flash_attention_2", "--max_pixels", "401408",
            "--reward_funcs", "accuracy", "format", "--use_vllm_for_gen", "true", "--use_system_prompt", "false", "--single_img", "1", "--report_to", "none",
            "--deepspeed", "local_scripts/zero3.json", "--run_name", "e2e"])
    logs = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert [r["step"] for r in logs] == [1, 2] and all(np.isfinite(r["loss"]) and r["grad_norm"] > 0 and r["completion_length"] == 8.0 for r in logs)
    assert {"rewards/accuracy_reward", "rewards/text_checksum_reward", "reward", "reward_std", "kl", "learning_rate"} <= set(logs[0])
    assert os.path.exists(os.path.join(out, "model.safetensors")) and os.path.exists(os.path.join(out, "config.json"))
    cfg2, s2 = load_checkpoint(out, DEV, trainable=False)
    assert cfg2 == VLMConfig.from_dict(d)
    assert not torch.equal(s2.flat, s0.flat) and bool(torch.isfinite(s2.flat.float()).all())


def test_pa_sft_entry_point_end_to_end(tmp_path, offline_processor):
    """`train/stage_sft/train.py` with the flags of scripts/train/PA_SFT/PA_SFT_Qwen_Instruct_2_5_VL_3B.sh: dataset_info.json + sharegpt manifest + --image_dir,
    qwen2_vl template, 2 optimizer steps; the loss log is written and finite, the saved checkpoint loads back and differs from the start."""
    d = _tiny_with_processor_ids(offline_processor)
    src = str(tmp_path / "Qwen2.5-VL-tiny-Instruct")
    _, s0 = _write_checkpoint(src, d)
    data_dir, img_dir = tmp_path / "data", tmp_path / "imgs"
    data_dir.mkdir()
    img_dir.mkdir()
    rows = []
    for i in range(4):
        fx.synth_pil_image(112, 84 + 28 * (i % 2), 60 + i).save(str(img_dir / f"s_{i}.png"))
        rows.append({"messages": [{"role": "user", "content": "<image>Are there any defects in the query image?"},
                                  {"role": "assistant", "content": "<think>the surface is clean</think><answer>No</answer>" if i % 2 else "<think>a scratch</think><answer>Yes</answer>"}],
                     "images": [f"s_{i}.png"]})
    (data_dir / "expert_ad.json").write_text(json.dumps(rows))
    (data_dir / "dataset_info.json").write_text(json.dumps({"expert_ad": {"file_name": "expert_ad.json", "formatting": "sharegpt", "columns": {"messages": "messages", "images": "images"},
                                                                          "tags": {"role_tag": "role", "content_tag": "content", "user_tag": "user", "assistant_tag": "assistant"}}}))
    out = str(tmp_path / "sft_out")
    m = _load("train/stage_sft/train.py")
    m.main(["--stage", "sft", "--do_train", "--model_name_or_path", src, "--dataset", "expert_ad", "--dataset_dir", str(data_dir), "--image_dir", str(img_dir), "--template", "qwen2_vl",
            "--finetuning_type", "full", "--output_dir", out, "--overwrite_output_dir", "--per_device_train_batch_size", "1", "--gradient_accumulation_steps", "2",
            "--learning_rate", "1e-3", "--lr_scheduler_type", "cosine", "--warmup_steps", "1", "--num_train_epochs", "1", "--cutoff_len", "2048", "--logging_steps", "1",
            "--save_steps", "500", "--bf16", "--deepspeed", "ds_z3.json", "--plot_loss", "--ddp_timeout", "9000"])
    log = [json.loads(l) for l in open(os.path.join(out, "trainer_log.jsonl"))]
    assert [r["current_steps"] for r in log] == [1, 2] and all(np.isfinite(r["loss"]) and r["loss"] > 0 for r in log)
    cfg2, s2 = load_checkpoint(out, DEV, trainable=False)
    assert cfg2 == VLMConfig.from_dict(d) and not torch.equal(s2.flat, s0.flat) and bool(torch.isfinite(s2.flat.float()).all())


def test_pa_sft_entry_point_end_to_end_llava_onevision(tmp_path, monkeypatch):
    """`train/stage_sft/train.py` with the flags of scripts/train/PA_SFT/PA_SFT_LLaVA_OneVision_SI_7B.sh (--template llava_next_qwen) on a LLaVA-OneVision checkpoint
    directory: any-resolution crops from the transformers LlavaOnevisionProcessor, `<image>` expanded to the packed feature count, SigLIP tower + projector + packing
    + Qwen2 decoder trained for 2 optimizer steps; a mismatching template is refused."""
    proc = fx.local_llava_ov_processor()
    proc.save_pretrained = lambda *a, **k: None
    import transformers
    monkeypatch.setattr(transformers.AutoProcessor, "from_pretrained", classmethod(lambda cls, *a, **k: proc))
    d = dict(fx.TINY_OV, image_token_id=proc.tokenizer.convert_tokens_to_ids("<image>"), eos_token_id=proc.tokenizer.eos_token_id, pad_token_id=proc.tokenizer.pad_token_id)
    t, v = d["text"], d["vision"]
    hf = {"model_type": "llava_onevision", "architectures": ["LlavaOnevisionForConditionalGeneration"], "image_token_index": d["image_token_id"],
          "image_grid_pinpoints": [list(p) for p in d["image_grid_pinpoints"]], "vision_aspect_ratio": "anyres_max_9", "vision_feature_layer": -1,
          "vision_feature_select_strategy": "full", "tie_word_embeddings": False,
          "text_config": {"model_type": "qwen2", "vocab_size": t["vocab_size"], "hidden_size": t["hidden_size"], "intermediate_size": t["intermediate_size"],
                          "num_hidden_layers": t["num_hidden_layers"], "num_attention_heads": t["num_attention_heads"], "num_key_value_heads": t["num_key_value_heads"],
                          "rms_norm_eps": t["rms_norm_eps"], "rope_theta": t["rope_theta"], "eos_token_id": d["eos_token_id"], "pad_token_id": d["pad_token_id"]},
          "vision_config": {"model_type": "siglip_vision_model", "num_hidden_layers": v["depth"], "hidden_size": v["hidden_size"], "intermediate_size": v["intermediate_size"],
                            "num_attention_heads": v["num_heads"], "num_channels": v["in_channels"], "patch_size": v["patch_size"], "image_size": v["image_size"], "layer_norm_eps": 1e-6}}
    src = str(tmp_path / "llava-onevision-tiny-si")
    s0 = ParamStore(VLMConfig.from_dict(d), DEV, trainable=False)
    s0.load_named(fx.make_weights_ov(d, 0))
    save_checkpoint(s0, src, hf)
    data_dir, img_dir = tmp_path / "data", tmp_path / "imgs"
    data_dir.mkdir()
    img_dir.mkdir()
    rows = []
    for i, (w, h) in enumerate(((100, 80), (100, 120), (400, 150), (90, 90))):
        fx.synth_pil_image(w, h, 70 + i).save(str(img_dir / f"s_{i}.png"))
        rows.append({"messages": [{"role": "user", "content": "<image>Are there any defects in the query image?"},
                                  {"role": "assistant", "content": "<think>a scratch</think><answer>Yes</answer>" if i % 2 else "<answer>No</answer>"}], "images": [f"s_{i}.png"]})
    (data_dir / "expert_ad.json").write_text(json.dumps(rows))
    (data_dir / "dataset_info.json").write_text(json.dumps({"Expert_AD_Stage_1": {"file_name": "expert_ad.json", "formatting": "sharegpt", "columns": {"messages": "messages", "images": "images"},
                                                                                  "tags": {"role_tag": "role", "content_tag": "content", "user_tag": "user", "assistant_tag": "assistant"}}}))
    out = str(tmp_path / "sft_out")
    m = _load("train/stage_sft/train.py")
    flags = ["--deepspeed", "scripts/train/zero3.json", "--stage", "sft", "--do_train", "--model_name_or_path", src, "--dataset", "Expert_AD_Stage_1", "--dataset_dir", str(data_dir),
             "--image_dir", str(img_dir), "--finetuning_type", "full", "--output_dir", out, "--overwrite_cache", "--overwrite_output_dir", "--warmup_steps", "1",
             "--weight_decay", "0.1", "--per_device_train_batch_size", "1", "--gradient_accumulation_steps", "2", "--ddp_timeout", "90000", "--learning_rate", "1e-3",
             "--lr_scheduler_type", "cosine", "--logging_steps", "1", "--cutoff_len", "8192", "--save_steps", "500", "--plot_loss", "--num_train_epochs", "1", "--bf16"]
    with pytest.raises(ValueError, match="does not belong to the model family"):
        m.main(flags + ["--template", "qwen2_vl"])
    m.main(flags + ["--template", "llava_next_qwen"])
    log = [json.loads(l) for l in open(os.path.join(out, "trainer_log.jsonl"))]
    assert [r["current_steps"] for r in log] == [1, 2] and all(np.isfinite(r["loss"]) and r["loss"] > 0 for r in log)
    cfg2, s2 = load_checkpoint(out, DEV, trainable=False)
    assert cfg2.is_llava and not torch.equal(s2.flat, s0.flat) and bool(torch.isfinite(s2.flat.float()).all())


def _llava_config_json(d):
    """config.json of a LLaVA-1.5 (`llava`) / LLaVA-NeXT (`llava_next`) checkpoint in the published layout: a text_config that names its model type and lists
    only what differs from that class's defaults would do; the tiny structures list everything."""
    t, v = d["text"], d["vision"]
    c = {"model_type": d["family"], "architectures": ["LlavaForConditionalGeneration" if d["family"] == "llava" else "LlavaNextForConditionalGeneration"],
         "image_token_index": d["image_token_id"], "pad_token_id": d["pad_token_id"], "projector_hidden_act": "gelu", "vision_feature_layer": -2, "vision_feature_select_strategy": "default",
         "tie_word_embeddings": False,
         "text_config": {"model_type": "llama" if d["family"] == "llava" else "mistral", "vocab_size": t["vocab_size"], "hidden_size": t["hidden_size"],
                         "intermediate_size": t["intermediate_size"], "num_hidden_layers": t["num_hidden_layers"], "num_attention_heads": t["num_attention_heads"],
                         "num_key_value_heads": t["num_key_value_heads"], "rms_norm_eps": t["rms_norm_eps"], "rope_theta": t["rope_theta"], "eos_token_id": d["eos_token_id"], "sliding_window": None},
         "vision_config": {"model_type": "clip_vision_model", "hidden_size": v["hidden_size"], "image_size": v["image_size"], "intermediate_size": v["intermediate_size"],
                           "num_attention_heads": v["num_heads"], "num_hidden_layers": v["depth"], "patch_size": v["patch_size"], "projection_dim": 64}}
    if d["family"] == "llava_next":
        c["image_grid_pinpoints"] = [list(p) for p in d["image_grid_pinpoints"]]
    return c


@pytest.mark.parametrize("family,model_dir,template", [("llava", "llava_1_5-tiny-hf", "llava"), ("llava_next", "llava_next-mistral-tiny-hf", "llava_next_mistral")])
def test_llava15_and_next_entry_points_end_to_end(tmp_path, monkeypatch, capsys, family, model_dir, template):
    """scripts/train/SC_GRPO/SC_GRPO_LLaVA_1_5.sh / _1_6.sh and scripts/train/PA_SFT/PA_SFT_LLaVA_1_5.sh / _1_6.sh, paths swapped: checkpoint directory in the published
    config layout (model id containing the substring the reference's switch looks for), offline transformers Llava / LlavaNext processor, manifest + image files;
    two optimizer steps each; the saved checkpoints load back (incl. the CLIP block and post_layernorm this path never touches) and differ from the start."""
    cfg0 = fx.TINY_LLAVA15 if family == "llava" else fx.TINY_LLAVA_NEXT
    proc = fx.local_llava_processor(cfg0)
    proc.save_pretrained = lambda *a, **k: None
    import transformers
    monkeypatch.setattr(transformers.AutoProcessor, "from_pretrained", classmethod(lambda cls, *a, **k: proc))
    tok = proc.tokenizer
    d = dict(cfg0, image_token_id=tok.convert_tokens_to_ids("<image>"), eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id, vision_start_token_id=-1, vision_end_token_id=-1)
    src = str(tmp_path / model_dir)
    cfg = VLMConfig.from_dict(d)
    s0 = ParamStore(cfg, DEV, trainable=False)
    w0 = fx.make_weights_llava(d, 0)
    s0.load_named(w0)
    save_checkpoint(s0, src, _llava_config_json(d))
    cfg_l, s1 = load_checkpoint(src, DEV, trainable=False)
    assert cfg_l == cfg
    back = s1.export_named()
    assert set(back) == set(w0) and all(np.array_equal(back[k].float().numpy().reshape(-1), v_.reshape(-1)) for k, v_ in w0.items())
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    rl_rows, sft_rows = [], []
    for i, (w_, h_) in enumerate(((100, 80), (100, 120), (150, 60), (90, 90))):
        fx.synth_pil_image(w_, h_, 80 + i).save(str(img_dir / f"p_{i}.png"))
        rl_rows.append({"image": f"p_{i}.png", "problem": "Are there any defects in the query image?", "solution": "<answer>Yes</answer>" if i % 2 else "<answer>No</answer>"})
        sft_rows.append({"messages": [{"role": "user", "content": "<image>Are there any defects in the query image?"},
                                      {"role": "assistant", "content": "<think>a scratch</think><answer>Yes</answer>" if i % 2 else "<answer>No</answer>"}], "images": [f"p_{i}.png"]})
    # ---- SC-GRPO
    manifest = tmp_path / "rl.json"
    manifest.write_text(json.dumps(rl_rows))
    from iadr1_amd import rewards
    monkeypatch.setitem(rewards.REWARD_FUNCS, "format", lambda prompts, completions, **kw: [float(sum(map(ord, c[0]["content"])) % 7) for c in completions])
    out_rl = str(tmp_path / "out_rl")
    _load("train/stage_rl/grpo_ad.py").main(["--model_name_or_path", src, "--dataset_name", str(manifest), "--image_path", str(img_dir), "--output_dir", out_rl, "--max_prompt_length", "1024",
                                             "--max_completion_length", "8", "--num_generations", "4", "--per_device_train_batch_size", "1", "--gradient_accumulation_steps", "2",
                                             "--learning_rate", "1e-3", "--num_train_epochs", "1", "--logging_steps", "1", "--save_steps", "100", "--bf16", "--reward_funcs", "accuracy", "format",
                                             "--use_vllm_for_gen", "true", "--use_system_prompt", "false", "--single_img", "1", "--deepspeed", "zero3.json"])
    logs = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert [r["step"] for r in logs] == [1, 2] and all(np.isfinite(r["loss"]) and r["grad_norm"] > 0 for r in logs)
    cfg2, s2 = load_checkpoint(out_rl, DEV, trainable=False)
    assert cfg2 == cfg and not torch.equal(s2.flat, s0.flat) and bool(torch.isfinite(s2.flat.float()).all())
    kept = "vision_tower.vision_model.post_layernorm.weight"
    assert np.array_equal(s2.export_named()[kept].float().numpy(), w0[kept])          # untouched tensors travel through load -> train -> save
    # ---- PA-SFT
    data_dir = tmp_path / "data"
    data_dir.mkdir()
    (data_dir / "expert_ad.json").write_text(json.dumps(sft_rows))
    (data_dir / "dataset_info.json").write_text(json.dumps({"Expert_AD_Stage_1": {"file_name": "expert_ad.json", "formatting": "sharegpt", "columns": {"messages": "messages", "images": "images"},
                                                                                  "tags": {"role_tag": "role", "content_tag": "content", "user_tag": "user", "assistant_tag": "assistant"}}}))
    out_sft = str(tmp_path / "out_sft")
    _load("train/stage_sft/train.py").main(["--stage", "sft", "--do_train", "--model_name_or_path", src, "--dataset", "Expert_AD_Stage_1", "--dataset_dir", str(data_dir), "--image_dir", str(img_dir),
                                            "--template", template, "--finetuning_type", "full", "--output_dir", out_sft, "--overwrite_output_dir", "--warmup_steps", "1",
                                            "--per_device_train_batch_size", "1", "--gradient_accumulation_steps", "2", "--learning_rate", "1e-3", "--lr_scheduler_type", "cosine",
                                            "--logging_steps", "1", "--cutoff_len", "4096", "--save_steps", "500", "--num_train_epochs", "1", "--bf16", "--deepspeed", "zero3.json"])
    log = [json.loads(l) for l in open(os.path.join(out_sft, "trainer_log.jsonl"))]
    assert [r["current_steps"] for r in log] == [1, 2] and all(np.isfinite(r["loss"]) and r["loss"] > 0 for r in log)
    assert {"lr", "epoch", "percentage", "total_steps", "elapsed_time", "remaining_time"} <= set(log[-1]) and log[-1]["percentage"] == 100.0 and log[-1]["lr"] == 0.0
    res, state = json.load(open(os.path.join(out_sft, "train_results.json"))), json.load(open(os.path.join(out_sft, "trainer_state.json")))
    assert res == json.load(open(os.path.join(out_sft, "all_results.json"))) and abs(res["train_loss"] - np.mean([r["loss"] for r in log])) < 1e-3
    assert state["global_step"] == 2 and [h["step"] for h in state["log_history"]] == [1, 2, 2]
    cfg3, s3 = load_checkpoint(out_sft, DEV, trainable=False)
    assert cfg3 == cfg and not torch.equal(s3.flat, s0.flat) and bool(torch.isfinite(s3.flat.float()).all())
    # the reference's PA-SFT of these two registered families trains the language model (and `image_newline`) with the CLIP tower and the projector frozen
    # (LLaMA-Factory defaults, tests/golden/sft_freeze.json)
    for n in s3.slots:
        same = torch.equal(s3.w(n), s0.w(n))
        if n.startswith("visual.") and n != "visual.newline":
            assert same, n
        elif len(s3.slots[n].shape) >= 2:
            assert not same, n


def test_evaluation_script_end_to_end(tmp_path, offline_processor, monkeypatch):
    """scripts/Inference/IAD-R1-Inference/hip_qwen_detect_format.py (the reference's vLLM_Qwen_detect_format.py with the rollout engine as the generator): checkpoint
    directory + MMAD-format question file + image files -> greedy answers -> answers json (one entry per image, letters from `get_ans`) + the accuracy csv, in the
    reference's `result/<name>/<dataset>/` layout; a second run resumes from the answers file (nothing left to do) unless --reproduce."""
    d = _tiny_with_processor_ids(offline_processor)
    src = str(tmp_path / "Qwen2.5-VL-tiny-Instruct")
    _write_checkpoint(src, d)
    data = tmp_path / "Industrial_test"
    (data / "MVTec" / "bottle").mkdir(parents=True)
    chat = {}
    for i in range(3):
        rel = f"MVTec/bottle/{'good' if i == 0 else 'broken'}_{i}.png"
        fx.synth_pil_image(112, 84 + 28 * (i % 2), 90 + i).save(str(data / rel))
        chat[rel] = {"conversation": [{"Question": "Is there any defect in the object?", "Answer": "B" if i == 0 else "A", "Options": {"A": "Yes.", "B": "No."}, "type": "Anomaly Detection"}],
                     "similar_templates": [], "random_templates": []}
    qfile = tmp_path / "test_data_format.json"
    qfile.write_text(json.dumps(chat))
    monkeypatch.chdir(tmp_path)
    m = _load("scripts/Inference/IAD-R1-Inference/hip_qwen_detect_format.py")
    argv = ["x", "--model-path", src, "--test_dataset", "test_data", "--json_path", str(qfile), "--data_path", str(data), "--batch_size", "2", "--name", "QwenTiny"]
    monkeypatch.setattr(sys, "argv", argv)
    monkeypatch.setattr("iadr1_amd.evaluate.GreedyGenerator.__init__", _short_generator_init(), raising=True)
    m.main()
    out_dir = tmp_path / "result" / "QwenTiny" / "test_data"
    ans_path = out_dir / "answers_0_shot_Qwen2.5-VL-tiny-Instruct_vllm.json"
    answers = json.load(open(ans_path))
    assert [a["image"] for a in answers] == list(chat) and all(a["gpt_answer"] for a in answers) and all(a["question_type"] == "Anomaly Detection" for a in answers)
    assert os.path.exists(str(ans_path).replace(".json", "_accuracy.csv"))
    m.main()                                                   # resume: every image already answered
    assert json.load(open(ans_path)) == answers


@pytest.mark.parametrize("script,family", [("hip_llava_detect_format.py", "llava_ov"), ("hip_llava_detect_format.py", "llava_next"), ("hip_llava_1_5_detect_format.py", "llava")])
def test_llava_evaluation_scripts_end_to_end(tmp_path, monkeypatch, script, family):
    """scripts/Inference/IAD-R1-Inference/hip_llava_detect_format.py (= the reference's vLLM_LLaVA_detect_format.py: LLaVA-OneVision and LLaVA-1.6) and
    hip_llava_1_5_detect_format.py (= vLLM_LLaVA_1_5_detect_format.py) on the rollout engine: checkpoint directory + MMAD-format question file + image files
    (one of them greyscale: the LLaVA scripts convert to RGB) -> 1-shot prompts (two images) through the offline transformers processor of the family ->
    greedy answers -> answers json + accuracy csv in the reference's layout; resume; the wrong script for a checkpoint is refused."""
    import transformers
    from PIL import Image
    if family == "llava_ov":
        proc = fx.local_llava_ov_processor()
        tok = proc.tokenizer
        d = dict(fx.TINY_OV, image_token_id=tok.convert_tokens_to_ids("<image>"), eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id)
        t, v = d["text"], d["vision"]
        hf = {"model_type": "llava_onevision", "architectures": ["LlavaOnevisionForConditionalGeneration"], "image_token_index": d["image_token_id"],
              "image_grid_pinpoints": [list(p) for p in d["image_grid_pinpoints"]], "vision_aspect_ratio": "anyres_max_9", "vision_feature_layer": -1,
              "vision_feature_select_strategy": "full", "tie_word_embeddings": False,
              "text_config": {"model_type": "qwen2", "vocab_size": t["vocab_size"], "hidden_size": t["hidden_size"], "intermediate_size": t["intermediate_size"],
                              "num_hidden_layers": t["num_hidden_layers"], "num_attention_heads": t["num_attention_heads"], "num_key_value_heads": t["num_key_value_heads"],
                              "rms_norm_eps": t["rms_norm_eps"], "rope_theta": t["rope_theta"], "eos_token_id": d["eos_token_id"], "pad_token_id": d["pad_token_id"]},
              "vision_config": {"model_type": "siglip_vision_model", "num_hidden_layers": v["depth"], "hidden_size": v["hidden_size"], "intermediate_size": v["intermediate_size"],
                                "num_attention_heads": v["num_heads"], "num_channels": v["in_channels"], "patch_size": v["patch_size"], "image_size": v["image_size"], "layer_norm_eps": 1e-6}}
        weights, name = fx.make_weights_ov(d, 0), "llava-onevision-tiny-si"
    else:
        cfg0 = fx.TINY_LLAVA15 if family == "llava" else fx.TINY_LLAVA_NEXT
        proc = fx.local_llava_processor(cfg0)
        tok = proc.tokenizer
        d = dict(cfg0, image_token_id=tok.convert_tokens_to_ids("<image>"), eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id, vision_start_token_id=-1, vision_end_token_id=-1)
        hf, weights, name = _llava_config_json(d), fx.make_weights_llava(d, 0), ("llava_1_5-tiny-hf" if family == "llava" else "llava_next-mistral-tiny-hf")
    proc.save_pretrained = lambda *a, **k: None
    monkeypatch.setattr(transformers.AutoProcessor, "from_pretrained", classmethod(lambda cls, *a, **k: proc))
    src = str(tmp_path / name)
    s0 = ParamStore(VLMConfig.from_dict(d), DEV, trainable=False)
    s0.load_named(weights)
    save_checkpoint(s0, src, hf)
    data = tmp_path / "Industrial_test"
    (data / "MVTec" / "bottle").mkdir(parents=True)
    fx.synth_pil_image(100, 80, 5).save(str(data / "MVTec" / "bottle" / "template.png"))
    chat = {}
    for i in range(3):
        rel = f"MVTec/bottle/{'good' if i == 0 else 'broken'}_{i}.png"
        im = fx.synth_pil_image(100 + 20 * i, 80, 90 + i)
        (im.convert("L") if i == 1 else im).save(str(data / rel))          # a greyscale file: Image.open(...).convert("RGB") of the LLaVA scripts
        chat[rel] = {"conversation": [{"Question": "Is there any defect in the object?", "Answer": "B" if i == 0 else "A", "Options": {"A": "Yes.", "B": "No."}, "type": "Anomaly Detection"}],
                     "similar_templates": ["MVTec/bottle/template.png"], "random_templates": ["MVTec/bottle/template.png"]}
    qfile = tmp_path / "test_data_format.json"
    qfile.write_text(json.dumps(chat))
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr("iadr1_amd.evaluate.GreedyGenerator.__init__", _short_generator_init(), raising=True)
    seen = []
    from iadr1_amd import evaluate
    orig_gen = evaluate.GreedyGenerator.generate
    monkeypatch.setattr(evaluate.GreedyGenerator, "generate", lambda self, batch: (seen.append((batch["images_per_prompt"], np.asarray(batch["input_ids"]).shape)), orig_gen(self, batch))[1])
    m = _load("scripts/Inference/IAD-R1-Inference/" + script)
    argv = ["x", "--model-path", src, "--test_dataset", "test_data", "--json_path", str(qfile), "--data_path", str(data), "--batch_size", "2"]      # defaults: 1-shot, name "LlaVA"
    monkeypatch.setattr(sys, "argv", argv)
    m.main()
    ans_path = tmp_path / "result" / "LlaVA" / "test_data" / f"answers_1_shot_{name}_vllm.json"
    answers = json.load(open(ans_path))
    assert [a["image"] for a in answers] == list(chat) and all(a["gpt_answer"] for a in answers) and all(a["question_type"] == "Anomaly Detection" for a in answers)
    assert [ipp for ipp, _ in seen] == [[2, 2], [2]]                                    # template + query image per prompt, batches of 2 and 1
    assert os.path.exists(str(ans_path).replace(".json", "_accuracy.csv"))
    m.main()                                                                             # resume: nothing left
    assert json.load(open(ans_path)) == answers and len(seen) == 2
    other = "hip_llava_1_5_detect_format.py" if script == "hip_llava_detect_format.py" else "hip_llava_detect_format.py"
    with pytest.raises(SystemExit, match="config.json says otherwise"):
        _load("scripts/Inference/IAD-R1-Inference/" + other).main()
    with pytest.raises(SystemExit, match="config.json says otherwise"):
        _load("scripts/Inference/IAD-R1-Inference/hip_qwen_detect_format.py").main()


def _short_generator_init():
    """The script decodes up to 512 tokens per answer; a random-weight model never stops early, so the test caps the length (same code path)."""
    from iadr1_amd import evaluate
    orig = evaluate.GreedyGenerator.__init__

    def init(self, cfg, store, max_new_tokens=512):
        orig(self, cfg, store, max_new_tokens=12)
    return init
