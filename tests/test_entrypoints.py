"""CPU checks of the drop-in API surface: the two entry points accept the flag sets of the reference's launch scripts,
dataset rows are turned into the same conversation structure, and the trainer reproduces the reference's
constructor error behaviour (/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:108-111,141-145,587-588)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(rel):
    spec = importlib.util.spec_from_file_location(rel.replace("/", "_"), os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _launch_scripts():
    """tests/golden/launch_flags.json: the argv every one of the reference's 14 launch scripts passes to its entry point (extracted from
    scripts/train/{PA_SFT,SC_GRPO}/*.sh by tools/make_golden_launch_flags.py; $VARIABLES replaced by placeholders)."""
    return json.load(open(os.path.join(ROOT, "tests", "golden", "launch_flags.json")))["scripts"]


@pytest.mark.parametrize("entry", ["train/stage_rl/grpo_ad.py", "train/stage_sft/train.py"])
def test_every_reference_launch_script_parses(entry):
    m = _load(entry)
    scripts = [s for s in _launch_scripts() if s["entry"] == entry]
    assert len(scripts) == 7
    for s in scripts:
        a = m.build_parser().parse_args(s["argv"])       # argparse exits (SystemExit) on an unknown flag
        if entry.endswith("grpo_ad.py"):
            assert a.num_generations == 4 and a.max_completion_length == 512 and a.max_pixels == 480000 and a.beta == 0.04 and a.learning_rate == 1e-6, s["script"]
            assert a.gradient_accumulation_steps == 2 and a.per_device_train_batch_size == 1 and a.save_steps == 100
        else:
            assert a.image_dir == "/data/Expert-AD" and a.weight_decay == 0.1 and a.warmup_steps == 100 and a.lr_scheduler_type == "cosine", s["script"]
            assert a.cutoff_len in (4096, 8192) and a.template in ("qwen2_vl", "llava", "llava_next_mistral", "llava_next_qwen")


def test_image_dir_is_where_relative_image_paths_resolve(tmp_path):
    """llamafactory hparams/data_args.py:44,136-137 (defaults to dataset_dir) and data/aligner.py:52-53 (joined only when that file exists)."""
    m = _load("train/stage_sft/train.py")
    (tmp_path / "data").mkdir()
    (tmp_path / "imgs").mkdir()
    (tmp_path / "imgs" / "a.png").write_bytes(b"x")
    rows = [{"messages": [{"role": "user", "content": "<image>q"}, {"role": "assistant", "content": "a"}], "images": ["a.png", "missing.png"]}]
    (tmp_path / "data" / "d.json").write_text(json.dumps(rows))
    got = m.load_sharegpt(str(tmp_path / "data" / "d.json"), str(tmp_path / "data"), str(tmp_path / "imgs"))
    assert got[0]["images"] == [str(tmp_path / "imgs" / "a.png"), "missing.png"]
    got = m.load_sharegpt(str(tmp_path / "data" / "d.json"), str(tmp_path / "data"))                 # default: dataset_dir, where neither file exists
    assert got[0]["images"] == ["a.png", "missing.png"]


def test_config_yaml_sets_defaults_cli_overrides_env_is_exported(tmp_path, monkeypatch):
    """TrlParser.parse_args_and_config, REF trl/trl/scripts/utils.py:165-223."""
    m = _load("train/stage_rl/grpo_ad.py")
    y = tmp_path / "c.yaml"
    y.write_text("env:\n  IADR1_TEST_ENV_FROM_YAML: 17\nmodel_name_or_path: /m/q\ndataset_name: d.json\noutput_dir: out\nnum_generations: 6\nbeta: 0.1\nreward_funcs: [accuracy]\n")
    monkeypatch.delenv("IADR1_TEST_ENV_FROM_YAML", raising=False)
    a = m.parse_args_and_config(m.build_parser(), ["--config", str(y), "--beta", "0.02"])
    assert a.model_name_or_path == "/m/q" and a.num_generations == 6 and a.beta == 0.02 and a.reward_funcs == ["accuracy"]
    assert os.environ["IADR1_TEST_ENV_FROM_YAML"] == "17"
    monkeypatch.delenv("IADR1_TEST_ENV_FROM_YAML", raising=False)
    y.write_text("model_name_or_path: /m/q\ndataset_name: d.json\noutput_dir: out\nno_such_flag: 3\n")
    with pytest.raises(ValueError, match="not used by the parser"):
        m.parse_args_and_config(m.build_parser(), ["--config", str(y)])
    with pytest.raises(SystemExit):      # required flags stay required without a config
        m.parse_args_and_config(m.build_parser(), ["--beta", "0.1"])


class _CharProcessor:
    """One token per character, no images: enough to run encode_example without a tokenizer download."""

    class tokenizer:
        @staticmethod
        def encode(text, add_special_tokens=False):
            return [ord(c) for c in text]

    image_processor = None


def test_pa_sft_encode_example_masks_and_truncates_per_turn(tmp_path):
    m = _load("train/stage_sft/train.py")
    data = [{"messages": [{"role": "user", "content": "what is this"}, {"role": "assistant", "content": "a nut"}, {"role": "user", "content": "broken?"},
                          {"role": "assistant", "content": "yes, scratched"}], "images": []},
            {"messages": [{"role": "user", "content": "dangling"}], "images": []}]
    (tmp_path / "d.json").write_text(json.dumps(data))
    rows = m.load_sharegpt(str(tmp_path / "d.json"), str(tmp_path))
    assert len(rows) == 1 and rows[0]["images"] is None          # the odd-length row is dropped, as the reference's aligner + filter do
    (tmp_path / "dataset_info.json").write_text(json.dumps({"ead": {"file_name": "d.json", "formatting": "sharegpt", "columns": {"messages": "messages", "images": "images"},
                                                                    "tags": {"role_tag": "role", "content_tag": "content", "user_tag": "user", "assistant_tag": "assistant"}}}))
    assert m.load_sharegpt("ead", str(tmp_path)) == rows
    with pytest.raises(ValueError, match="Undefined dataset"):
        m.load_sharegpt("nope", str(tmp_path))
    sys_block = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"
    ids, labels, pix, grids = m.encode_example(_CharProcessor(), rows[0], 4096)
    assert "".join(map(chr, ids)) == (sys_block + "<|im_start|>user\nwhat is this<|im_end|>\n<|im_start|>assistant\na nut<|im_end|>\n"
                                      "<|im_start|>user\nbroken?<|im_end|>\n<|im_start|>assistant\nyes, scratched<|im_end|>\n") and pix is None and grids == []
    assert "".join(chr(l) for l in labels if l != -100) == "a nut<|im_end|>\nyes, scratched<|im_end|>\n"
    assert all(l == -100 or l == i for i, l in zip(ids, labels))
    ids2, labels2, _, _ = m.encode_example(_CharProcessor(), rows[0], 4096, mask_history=True)
    assert ids2 == ids and "".join(chr(l) for l in labels2 if l != -100) == "yes, scratched<|im_end|>\n"
    # the budget is spent turn by turn (reference infer_seqlen), not by chopping the tail of the concatenation: a short answer survives whole and
    # its prompt is trimmed from the right
    ids3, labels3, _, _ = m.encode_example(_CharProcessor(), rows[0], 60)
    assert "".join(map(chr, ids3)) == (sys_block + "<|im_start|>user\nwhat is this<|im_end|>\n<|im_start|>assistant\n")[:44] + "a nut<|im_end|>\n"
    assert "".join(chr(l) for l in labels3 if l != -100) == "a nut<|im_end|>\n"


def test_make_conversation_structure():
    m = _load("train/stage_rl/grpo_ad.py")
    row = {"problem": "Is there any defect?", "image": "a/b.png", "solution": "<answer>no</answer>"}
    out = m.make_conversation(row, "/data", False, 1)
    assert out["image"] == ["/data/a/b.png"] and out["solution"] == row["solution"]
    assert out["prompt"][0]["role"] == "user" and out["prompt"][0]["content"][0] == {"type": "image"}
    assert out["prompt"][0]["content"][1]["text"].endswith("Is there any defect?") and "expert in detecting defects" in out["prompt"][0]["content"][1]["text"]
    out2 = m.make_conversation({"problem": "q", "image": ["r.png", {"path": "t.png"}]}, "/d", True, 0)
    assert [c["role"] for c in out2["prompt"]] == ["system", "user"] and len(out2["image"]) == 2 and "<think>" in out2["prompt"][0]["content"]
    with pytest.raises(TypeError):
        m.make_conversation({"problem": "q", "image": [3]}, "/d", False, 1)


def test_prompt_templates_and_make_conversation_match_the_reference_golden(golden_dir):
    """SURVEY section 8 row a20, pinned: the four prompt templates (single_img 1 / 0 x system / question) are byte-equal to the constants inside the reference's
    main() (REF train/stage_rl/grpo_ad.py:72-118, read from its AST), and make_conversation returns what the reference's nested function returns for every row
    form it accepts (REF:135-181, compiled from the AST and executed by tools/make_golden_prompts.py).  Stated divergences, both stricter / fixed here: a row
    without an image makes the reference return None (datasets.map then keeps the row without a prompt and the trainer fails on it later) -- here ValueError at once;
    `image` given as ONE dict hits an unbound local in the reference (:151) -- here it is accepted like a one-element list."""
    import copy
    import json
    m = _load("train/stage_rl/grpo_ad.py")
    g = json.load(open(os.path.join(golden_dir, "prompts.json")))
    for k in ("1", "0"):
        assert m.PROMPTS[int(k)]["system"] == g["templates"][k]["system"] and m.PROMPTS[int(k)]["question"] == g["templates"][k]["question"], k
    seen = {"returns": 0, "TypeError": 0, "None": 0, "UnboundLocalError": 0}
    for c in g["cases"]:
        row = copy.deepcopy(g["rows"][c["row"]])
        call = lambda: m.make_conversation(copy.deepcopy(row), "/data/Expert-AD", c["use_system_prompt"], c["single_img"])
        if c.get("raises") == "TypeError":
            with pytest.raises(TypeError):
                call()
            seen["TypeError"] += 1
        elif c.get("raises") == "UnboundLocalError":          # the reference's bug for a single-dict image; fixed here
            out = call()
            assert out["image"] == [os.path.join("/data/Expert-AD", row["image"]["path"])] and len(out["prompt"][-1]["content"]) == 2
            seen["UnboundLocalError"] += 1
        elif c["returns"] is None:
            with pytest.raises(ValueError, match="without an image"):
                call()
            seen["None"] += 1
        else:
            out = call()
            assert out["prompt"] == c["returns"]["prompt"] and out["image"] == c["returns"]["image"], c
            # datasets.map merges the returned columns into the row; `messages` is dropped afterwards (REF:183-185)
            assert {k: v for k, v in out.items() if k not in ("prompt", "image")} == {k: v for k, v in row.items() if k not in ("messages", "image")}
            seen["returns"] += 1
    assert seen["returns"] >= 16 and min(seen.values()) >= 4, seen


def test_trainer_constructor_errors_match_reference():
    import iadr1_amd  # noqa: F401
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer
    with pytest.raises(ValueError, match="Invalid `torch_dtype`"):
        SCGRPOTrainer("/x/Qwen2.5-VL-3B", [], args=GRPOConfig(model_init_kwargs={"torch_dtype": 7}))
    with pytest.raises(ValueError, match="already instantiated"):
        SCGRPOTrainer((None, {}), [], args=GRPOConfig(model_init_kwargs={"a": 1}))
    with pytest.raises(ValueError, match="Qwen2.5-VL"):
        SCGRPOTrainer("/x/llava-1.5-7b", [], args=GRPOConfig())
    with pytest.raises(ValueError, match="LoRA is not part of this path"):
        SCGRPOTrainer((None, {}), [], args=GRPOConfig(), peft_config=object())
    with pytest.raises(ValueError, match="LoRA is not part of this path"):
        _load("train/stage_rl/grpo_ad.py").main(["--model_name_or_path", "/x", "--output_dir", "o", "--dataset_name", "d.json", "--use_peft"])


def test_reward_model_as_reward_function_matches_the_reference():
    """"a PreTrainedModel reward is also accepted" (sc_grpo_trainer.py:228-262 set-up, :760-772 scoring, :804-809 metric key).
    tests/golden/reward_model.npz holds what the reference's compute_loss fed a tiny sequence classifier (texts rendered by the reward tokenizer's chat
    template over prompt + completion, right padding, no added special tokens) and the rewards it got back; the trainer's host path must reproduce both."""
    import types
    import numpy as np
    import torch
    from transformers import Qwen2Config, Qwen2ForSequenceClassification
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd import rewards as R
    from iadr1_amd.trainer import SCGRPOTrainer, resolve_reward_funcs, reward_func_name
    g = np.load(os.path.join(ROOT, "tests", "golden", "reward_model.npz"))
    tok = fx.local_qwen2vl_processor().tokenizer
    tok.chat_template = fx.QWEN2VL_CHAT_TEMPLATE
    cfg = Qwen2Config(vocab_size=len(tok), hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      max_position_embeddings=512, num_labels=1, pad_token_id=None, tie_word_embeddings=False)
    rm = Qwen2ForSequenceClassification(cfg).float()
    rm.load_state_dict({k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("rm::")})
    rm.config._name_or_path = "local/tiny-reward-model"
    funcs, classes = resolve_reward_funcs([R.accuracy_reward, rm], [None, tok])
    assert rm.config.pad_token_id == tok.pad_token_id and not rm.training and classes[1] is tok
    assert [reward_func_name(f) for f in funcs] == ["accuracy_reward", "tiny-reward-model"]
    assert sorted("rewards/" + reward_func_name(f) for f in funcs) == g["metric_names"].tolist()
    with pytest.raises(ValueError, match="must match the number of reward functions"):
        resolve_reward_funcs([R.accuracy_reward, rm], [tok])
    with pytest.raises(ValueError, match="expected a callable"):
        resolve_reward_funcs([3], None)

    texts = g["completions_text"].tolist()
    t = SCGRPOTrainer.__new__(SCGRPOTrainer)
    t.args = types.SimpleNamespace(num_generations=len(texts))
    t.state = types.SimpleNamespace(global_step=0)
    t.processing_class = types.SimpleNamespace(batch_decode=lambda ids, skip_special_tokens=True: list(texts))
    t.reward_funcs, t.reward_processing_classes = funcs, classes
    seen = {}
    orig = tok.__class__.__call__

    def spy(self, text=None, *a, **kw):
        out = orig(self, text, *a, **kw)
        seen["texts"], seen["ids"], seen["mask"] = list(text), out["input_ids"].numpy().copy(), out["attention_mask"].numpy().copy()
        return out
    tok.__class__.__call__ = spy
    try:
        inputs = [{"prompt": [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "Is there a defect?"}]}], "image": [object()], "solution": str(g["solution"])}]
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            got = t._rewards(inputs, np.zeros((len(texts), 3), np.int64))
    finally:
        tok.__class__.__call__ = orig
    assert seen["texts"] == g["texts"].tolist()
    assert np.array_equal(seen["ids"], g["input_ids"]) and np.array_equal(seen["mask"], g["attention_mask"])
    assert got.dtype == np.float32 and got.shape == g["rewards_per_func"].shape
    assert np.array_equal(got[:, 0], g["rewards_per_func"][:, 0])
    np.testing.assert_allclose(got[:, 1], g["rewards_per_func"][:, 1], rtol=1e-5, atol=1e-6)


def test_pa_sft_frozen_parameter_rule_matches_the_reference():
    """iadr1_amd.sft.frozen_parameter_rule vs tests/golden/sft_freeze.json -- the output of the reference's own get_forbidden_modules + the name test of
    _setup_full_tuning (llamafactory/model/model_utils/visual.py:153-171, model/adapter.py:39-55) on the HF parameter names of the tiny fixture models, for the
    default flags (what every launch script runs) and the three other vision-tower / projector combinations.  A store tensor is frozen iff every HF parameter
    it is built from is frozen (fused q|k|v, gate|up: all members fall on the same side)."""
    import torch
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd.params import ParamStore, VLMConfig
    from iadr1_amd.sft import frozen_parameter_rule
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "sft_freeze.json")))["cases"]
    fixtures = {"qwen2_vl": (fx.TINY_Q2, fx.make_weights), "qwen2_5_vl": (fx.TINY, fx.make_weights), "llava_onevision": (fx.TINY_OV, fx.make_weights_ov),
                "llava": (fx.TINY_LLAVA15, fx.make_weights_llava), "llava_next": (fx.TINY_LLAVA_NEXT, fx.make_weights_llava)}
    seen = set()
    for c in cases:
        fl = c["flags"]
        if fl["train_mm_proj_only"]:
            continue            # (refused by the entry point: what it freezes follows the transformers version's parameter naming)
        mt = c["model_type"]
        seen.add(mt)
        rule = frozen_parameter_rule(mt, fl["freeze_vision_tower"], fl["freeze_multi_modal_projector"])
        cfgd, mk = fixtures[mt]
        cfg = VLMConfig.from_dict(cfgd)
        store = ParamStore(cfg, torch.device("cpu"), trainable=False, with_decode_pack=False)
        # frozen HF names of the golden, normalised to the classic checkpoint naming the fixtures use
        def norm(n):
            n = n[len("model."):] if n.startswith(("model.visual.", "model.vision_tower.", "model.multi_modal_projector.", "model.image_newline")) else n
            return "model." + n[len("model.language_model."):] if n.startswith("model.language_model.") else n
        frozen_hf = {norm(n) for n in c["frozen"]}
        if rule is None:
            assert not frozen_hf, (mt, fl)
            continue
        fz_store = {n for n in store.slots if rule(n)}
        vis_store = {n for n in store.slots if n.startswith("visual.")}
        assert fz_store <= vis_store
        tower = {n for n in vis_store if not n.startswith("visual.merger.") and n != "visual.newline"}
        proj = {n for n in vis_store if n.startswith("visual.merger.")}
        assert fz_store == (tower if fl["freeze_vision_tower"] else set()) | (proj if fl["freeze_multi_modal_projector"] else set())
        # the same partition on the reference's side: every frozen HF name belongs to the tower / projector, the language model and image_newline never do
        keys = {"qwen2_vl": (("visual.patch_embed", "visual.blocks"), ("visual.merger",))}.get(mt, (("vision_tower",), ("multi_modal_projector",)))
        for n in frozen_hf:
            assert any(k in n for k in keys[0] + keys[1]), n
        assert bool(fz_store & tower) == any(any(k in n for k in keys[0]) for n in frozen_hf)
        assert bool(fz_store & proj) == any(any(k in n for k in keys[1]) for n in frozen_hf)
        assert not any("image_newline" in n or "lm_head" in n or "embed_tokens" in n for n in frozen_hf)
    assert seen == set(fixtures)
    # the entry point: defaults as LLaMA-Factory's, --train_mm_proj_only refused
    m = _load("train/stage_sft/train.py")
    a = m.build_parser().parse_args(["--model_name_or_path", "/m", "--dataset", "d", "--output_dir", "o"])
    assert a.freeze_vision_tower is True and a.freeze_multi_modal_projector is True and a.train_mm_proj_only is False
    a = m.build_parser().parse_args(["--model_name_or_path", "/m", "--dataset", "d", "--output_dir", "o", "--freeze_vision_tower", "false"])
    assert a.freeze_vision_tower is False


def test_weight_decay_group_is_the_hf_trainers():
    """Both trainers of the reference build their optimizer in transformers.Trainer.create_optimizer: the weight-decay group is get_decay_parameter_names(model)
    (tests/golden/sft_freeze.json: decay_parameters, from the installed transformers on the tiny HF models of all five families).  Every tensor of the parameter
    store must sit on the same side as the HF parameters it is built from -- e.g. the RMSNorm gains of the Qwen2.5-VL vision tower DO decay (their names escape
    HF's norm patterns), image_newline and CLIP's class embedding do, LayerNorm gains and every bias do not."""
    import numpy as np
    import torch
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd.params import ParamStore, VLMConfig
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "sft_freeze.json")))
    fixtures = {"qwen2_vl": fx.TINY_Q2, "qwen2_5_vl": fx.TINY, "llava_onevision": fx.TINY_OV, "llava": fx.TINY_LLAVA15, "llava_next": fx.TINY_LLAVA_NEXT}

    def norm(n):            # checkpoint naming differs between transformers generations (model.language_model.layers / language_model.model.layers / model.layers ...)
        while n.startswith(("model.", "language_model.")):
            n = n.split(".", 1)[1]
        return n.replace("vision_tower.vision_model.", "vision_tower.")
    for mt, cfgd in fixtures.items():
        cfg = VLMConfig.from_dict(cfgd)
        store = ParamStore(cfg, torch.device("cpu"), trainable=False, with_decode_pack=False, with_transposes=False)
        slots = sorted(store.slots.values(), key=lambda s: s.offset)
        assert len(slots) < 250
        store.flat.zero_()
        for i, sl in enumerate(slots):          # every element of slot i holds i + 1 (exact in bf16 below 256)
            store.w(sl.name).fill_(float(i + 1))
        exported = store.export_named()
        canon = {k: v for k, v in exported.items()}
        decay_hf = {norm(n) for n in d["decay_parameters"][mt]}
        all_hf = {norm(n) for n in d["parameters"][mt]}
        seen = set()
        for name, t in canon.items():
            key = norm(name)
            ids = {int(v) for v in np.unique(t.float().numpy()) if v != 0}       # (zero: padding rows / lanes of the fused layouts, tensors the store does not hold)
            if key not in all_hf or not ids:
                continue
            assert len(ids) == 1, (mt, name, ids)
            sl = slots[ids.pop() - 1]
            assert sl.decay == (key in decay_hf), (mt, name, sl.name, sl.decay)
            seen.add(sl.name)
        missing = {sl.name for sl in slots} - seen      # every tensor of the store was placed (bias-free decoders keep an all-zero fused bias row no checkpoint holds)
        assert all(n.endswith(".qkv.b") and n.startswith("layers.") and not cfg.qkv_bias for n in missing), (mt, sorted(missing))


def test_entry_point_defaults_are_the_references():
    """Flags a launch script does not pass take the reference's defaults: transformers TrainingArguments (SC-GRPO inherits them through trl's GRPOConfig) and
    LLaMA-Factory's DataArguments / ModelArguments (hparams/data_args.py:41-57, model_args.py:62).  The SC-GRPO scripts themselves override four of them --
    checked below against the argv extracted from the reference's scripts (tests/golden/launch_flags.json)."""
    import iadr1_amd  # noqa: F401
    from iadr1_amd.trainer import GRPOConfig
    rl = _load("train/stage_rl/grpo_ad.py").build_parser().parse_args(["--model_name_or_path", "/m", "--output_dir", "o", "--dataset_name", "d.json"])
    want_rl = dict(num_train_epochs=3.0, per_device_train_batch_size=8, gradient_accumulation_steps=1, learning_rate=1e-6, weight_decay=0.0, max_grad_norm=1.0,
                   lr_scheduler_type="linear", warmup_steps=0, logging_steps=500, save_steps=500, seed=42, max_steps=-1, num_generations=8, max_prompt_length=512,
                   max_completion_length=256, beta=0.04, temperature=0.9)
    for k, v in want_rl.items():
        assert getattr(rl, k) == v, (k, getattr(rl, k))
        assert getattr(GRPOConfig(), k) == v, (k, getattr(GRPOConfig(), k))
    flags = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "launch_flags.json")))["scripts"]
    grpo_scripts = [f for f in flags if "SC_GRPO" in f["script"]]
    assert len(grpo_scripts) == 7
    for f in grpo_scripts:      # what every scripts/train/SC_GRPO/*.sh passes (the comments in trainer.py / grpo_ad.py say so)
        got = {f["argv"][i]: f["argv"][i + 1] for i in range(len(f["argv"]) - 1) if f["argv"][i].startswith("--")}
        assert (got["--num_train_epochs"], got["--logging_steps"], got["--save_steps"], got["--per_device_train_batch_size"]) == ("1", "1", "100", "1"), f["script"]
    sft = _load("train/stage_sft/train.py").build_parser().parse_args(["--model_name_or_path", "/m", "--dataset", "d", "--output_dir", "o"])
    want_sft = dict(num_train_epochs=3.0, per_device_train_batch_size=8, gradient_accumulation_steps=1, learning_rate=5e-5, weight_decay=0.0, max_grad_norm=1.0,
                    lr_scheduler_type="linear", warmup_steps=0, logging_steps=500, save_steps=500, seed=42, max_steps=-1, cutoff_len=2048, dataset_dir="data",
                    image_resolution=512 * 512, train_on_prompt=False, mask_history=False, freeze_vision_tower=True, freeze_multi_modal_projector=True)
    for k, v in want_sft.items():
        assert getattr(sft, k) == v, (k, getattr(sft, k))


def test_llava_rotation_trigger_follows_the_reference_on_the_ids():
    """the llava left-padding fix-up (REF:516-567) decides from the IDS -- first pad token with an all-pad suffix -- not from the EOS mask.
    `right_padding_shift` against the oracle's restatement of the reference function on: early EOS + pads, full-length rows, left-padded rows, a pad inside
    the text, a completion shorter than C WITHOUT an EOS (rotated by the reference), and pad == EOS (the EOS itself moves into the padding)."""
    import numpy as np
    import torch
    import iadr1_amd  # noqa: F401
    from iadr1_amd.sc_grpo import right_padding_shift
    from oracle.llava_ov import ensure_left_padding
    pad, eos = 2, 1
    rows = {
        "early_eos": [5, 6, 7, 8, 9, eos, pad, pad], "full": [5, 6, 7, 8, 9, 10, 11, 12], "left_padded": [pad, pad, 5, 6, 7, eos, pad, pad],
        "pad_inside": [5, pad, 7, 8, 9, eos, pad, pad], "short_no_eos": [5, 6, 7, 8, 9, 10, pad, pad], "all_but_one": [5, pad, pad, pad, pad, pad, pad, pad],
    }
    for name, r in rows.items():
        ids = torch.tensor([r])
        rot, _ = ensure_left_padding(ids, torch.ones_like(ids), pad)
        pl = right_padding_shift(np.asarray(r), pad)
        want = [pad] * pl + r[: len(r) - pl]
        assert rot[0].tolist() == want, (name, pl, rot[0].tolist())
    assert right_padding_shift(np.asarray(rows["early_eos"]), pad) == 2 and right_padding_shift(np.asarray(rows["short_no_eos"]), pad) == 2
    assert right_padding_shift(np.asarray(rows["left_padded"]), pad) == 0 and right_padding_shift(np.asarray(rows["pad_inside"]), pad) == 0
    # pad == EOS: the first EOS starts the all-pad suffix
    r = [5, 6, 7, 1, 1, 1]
    assert right_padding_shift(np.asarray(r), 1) == 3 and ensure_left_padding(torch.tensor([r]), torch.ones(1, 6, dtype=torch.long), 1)[0][0].tolist() == [1, 1, 1, 5, 6, 7]


def test_batch_prefetcher_keeps_order_and_propagates_errors():
    """iadr1_amd.prefetch.BatchPrefetcher on the host alone (device None: no upload): results come back per submission, in order; an exception of the
    preparation surfaces at `.result()` (the step that would have consumed the batch), not silently on the worker thread."""
    import threading
    import iadr1_amd  # noqa: F401
    from iadr1_amd.prefetch import BatchPrefetcher
    seen = []

    def prepare(inputs):
        seen.append((threading.current_thread().name, inputs[0]))
        if inputs[0] == "bad":
            raise ValueError("processor failed")
        return {"input_ids": [len(x) for x in inputs], "tag": inputs[0]}
    pf = BatchPrefetcher(prepare, None)
    futs = [pf.submit([f"row{i}", "x" * i]) for i in range(5)] + [pf.submit(["bad"])]
    got = [BatchPrefetcher.ready(f.result()) for f in futs[:5]]
    assert [g["tag"] for g in got] == [f"row{i}" for i in range(5)] and got[3]["input_ids"] == [4, 3]
    assert all(name.startswith("iadr1-prefetch") for name, _ in seen) and [t for _, t in seen] == [f"row{i}" for i in range(5)] + ["bad"]
    with pytest.raises(ValueError, match="processor failed"):
        futs[5].result()
    pf.shutdown()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch it."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for sub in ("iad-r1_amd", "train", "scripts"):
        for dp, _, files in os.walk(os.path.join(root, sub)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    txt = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M):
                        offenders.append(os.path.join(dp, f))
    assert not offenders, offenders


def test_pad_matches_the_reference_known_answers():
    """§8 a18: trl `pad` (trl/trl/trainer/utils.py:418-478) restated in iadr1_amd.sc_grpo.pad, against tests/golden/pad.json -- the cases of the
    reference's own TestPad (trl/tests/test_utils.py:47-130) run through the reference's function by tools/make_golden.py."""
    import numpy as np
    import iadr1_amd  # noqa: F401
    from iadr1_amd.sc_grpo import pad, right_pad
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "pad.json")))["cases"]
    assert len(cases) >= 8
    for c in cases:
        got = pad([np.asarray(r) for r in c["rows"]], c["padding_value"], c["padding_side"], c["pad_to_multiple_of"])
        assert got.tolist() == c["out"], c
    assert right_pad([[5, 6, 7], [8]], 2).tolist() == [[5, 6, 7], [8, 2, 2]] and right_pad([[1], []], 0).tolist() == [[1], [0]]


def test_prepare_batch_matches_the_reference_host_path():
    """§8 a19 + a23 executed for real: iadr1_amd.trainer.prepare_batch (chat template per example, PIL images, ONE processor call with left padding,
    REF sc_grpo_trainer.py:600-622) on an offline Qwen2-VL processor, against tests/golden/prepare.json = the same rows through the reference's
    `maybe_apply_chat_template` and the same processor call (tools/make_golden.py::gen_prepare).  Covers a conversational prompt with the default
    system turn, one with an explicit system turn and TWO images (1-shot), a plain-string prompt (passes through untouched), a resized image
    (300x200 -> multiples of 28) and the left padding of the shorter row."""
    import numpy as np
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd.trainer import maybe_apply_chat_template, prepare_batch
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "prepare.json")))
    proc = fx.local_qwen2vl_processor(max_pixels=g["meta"]["max_pixels"], min_pixels=g["meta"]["min_pixels"])
    for inputs, want in zip(fx.prepare_examples(), g["cases"]):
        assert [maybe_apply_chat_template(ex, proc) for ex in inputs] == want["prompts_text"]
        b = prepare_batch(proc, inputs)
        assert b["input_ids"].tolist() == want["input_ids"] and b["attention_mask"].tolist() == want["attention_mask"]          # integer work: bit-exact
        assert [list(x) for x in b["image_grid_thw"]] == want["image_grid_thw"] and b["images_per_prompt"] == [len(ex["image"]) for ex in inputs]
        pv = np.asarray(b["pixel_values"], dtype=np.float64)
        assert list(pv.shape) == want["pixel_shape"]
        assert abs(pv.sum() - want["pixel_sum"]) <= 1e-6 * want["pixel_abs_sum"] and abs(np.abs(pv).sum() - want["pixel_abs_sum"]) <= 1e-6 * want["pixel_abs_sum"]
        np.testing.assert_allclose(pv[0, :8], want["pixel_head"], rtol=1e-6)
        np.testing.assert_allclose(pv[-1, -8:], want["pixel_tail"], rtol=1e-6)
    first = g["cases"][0]
    assert sorted(row[0] for row in first["attention_mask"]) == [0, 1] and all(row[-1] == 1 for row in first["attention_mask"])   # the shorter row is padded on the LEFT
    assert first["prompts_text"][0].startswith("<|im_start|>system\nYou are a helpful assistant.") and first["prompts_text"][1].startswith("<|im_start|>system\nYou are an inspector.")


def test_prepare_batch_on_a_llava_onevision_processor():
    """The LLaVA-OneVision side of prepare_batch (BASELINE config 5): the transformers LlavaOnevisionProcessor (offline: character tokenizer + the
    family's PIL image processor) crops by the any-resolution rule and expands every `<image>` to the packed feature count ITSELF; the host-side
    restatement of pack_image_features (iadr1_amd.llava_ov) must reserve exactly the same number of placeholders and crops for every image --
    a no-shrink case, an un-padded case and one shrunk by the bilinear interpolation (26 crops > anyres_max_9)."""
    import numpy as np
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd import llava_ov as lo
    from iadr1_amd.trainer import prepare_batch
    proc = fx.local_llava_ov_processor()
    v, pins = fx.TINY_OV["vision"], fx.TINY_OV["image_grid_pinpoints"]
    side = v["image_size"] // v["patch_size"]
    sizes = [(80, 100), (120, 100), (150, 400)]                                   # (height, width)
    q = lambda n: [{"type": "image"}, {"type": "text", "text": "Is there any defect?" + " more" * n}]
    rows = [{"prompt": [{"role": "user", "content": q(i)}], "image": [fx.synth_pil_image(w, h, 10 + i)], "solution": "s"} for i, (h, w) in enumerate(sizes)]
    b = prepare_batch(proc, rows)
    assert b["image_sizes"] == [list(s) for s in sizes] and b["images_per_prompt"] == [1, 1, 1] and "image_grid_thw" not in b
    image_id = proc.tokenizer.convert_tokens_to_ids("<image>")
    want_tokens = [lo.num_image_tokens(s, pins, v["image_size"], side, 9) for s in sizes]
    want_crops = [lo.num_crops(s, pins, v["image_size"]) for s in sizes]
    assert (b["input_ids"] == image_id).sum(1).tolist() == want_tokens == [70, 106, 184]
    px = np.asarray(b["pixel_values"])
    assert px.shape == (3, max(want_crops), 3, v["image_size"], v["image_size"]) and want_crops == [5, 7, 26]
    for i, n in enumerate(want_crops):                                           # crops past an image's own count are the processor's zero padding
        assert np.abs(px[i, n:]).max(initial=0.0) == 0.0 and np.abs(px[i, n - 1]).max() > 0.0
    m = b["attention_mask"]
    assert m[:, -1].all() and (m[:, 0] == 0).sum() == 2 and (np.diff(m, axis=1) >= 0).all()                # left padding
    plan = lo.pack_plan(sizes, pins, v["image_size"], side, 9)
    assert plan["lens"] == want_tokens and plan["crops"] == want_crops and plan["n_src"] == sum(want_crops) * side * side + 1


def test_from_hf_config_parses_the_published_config_layouts():
    """`VLMConfig.from_hf_config` on config.json files in the layouts the published checkpoints use (field names and nesting as on the model cards:
    Qwen2.5-VL flat 4.x layout with `rope_scaling.mrope_section` and `in_chans`; Qwen2-VL with `embed_dim` / `mlp_ratio`; LLaVA-OneVision with
    `text_config` / `vision_config` / `image_grid_pinpoints` / `vision_aspect_ratio`) gives the shapes the static constructors (bench.py) use."""
    import iadr1_amd  # noqa: F401
    from iadr1_amd.params import VLMConfig
    q25_3b = {"architectures": ["Qwen2_5_VLForConditionalGeneration"], "model_type": "qwen2_5_vl", "bos_token_id": 151643, "eos_token_id": 151645, "vision_start_token_id": 151652,
              "vision_end_token_id": 151653, "vision_token_id": 151654, "image_token_id": 151655, "video_token_id": 151656, "hidden_act": "silu", "hidden_size": 2048,
              "intermediate_size": 11008, "max_position_embeddings": 128000, "num_attention_heads": 16, "num_hidden_layers": 36, "num_key_value_heads": 2, "rms_norm_eps": 1e-06,
              "rope_theta": 1000000.0, "tie_word_embeddings": True, "torch_dtype": "bfloat16", "vocab_size": 151936, "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]},
              "vision_config": {"depth": 32, "hidden_act": "silu", "hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16, "in_chans": 3, "out_hidden_size": 2048,
                                "patch_size": 14, "spatial_merge_size": 2, "spatial_patch_size": 14, "window_size": 112, "fullatt_block_indexes": [7, 15, 23, 31],
                                "tokens_per_second": 2, "temporal_patch_size": 2}}
    assert VLMConfig.from_hf_config(q25_3b) == VLMConfig.qwen25vl_3b()
    q25_7b = dict(q25_3b, hidden_size=3584, intermediate_size=18944, num_attention_heads=28, num_hidden_layers=28, num_key_value_heads=4, tie_word_embeddings=False, vocab_size=152064,
                  vision_config=dict(q25_3b["vision_config"], out_hidden_size=3584))
    assert VLMConfig.from_hf_config(q25_7b) == VLMConfig.qwen25vl_7b()
    nested = {k: v for k, v in q25_3b.items() if k not in ("hidden_size", "intermediate_size", "num_attention_heads", "num_hidden_layers", "num_key_value_heads", "rms_norm_eps",
                                                             "rope_theta", "vocab_size", "rope_scaling")}
    nested["text_config"] = {k: q25_3b[k] for k in ("hidden_size", "intermediate_size", "num_attention_heads", "num_hidden_layers", "num_key_value_heads", "rms_norm_eps", "vocab_size")}
    nested["text_config"]["rope_parameters"] = {"rope_type": "default", "mrope_section": [16, 24, 24], "rope_theta": 1000000.0}
    assert VLMConfig.from_hf_config(nested) == VLMConfig.qwen25vl_3b()                      # transformers 5.x layout of the same checkpoint
    q2_2b = {"architectures": ["Qwen2VLForConditionalGeneration"], "model_type": "qwen2_vl", "eos_token_id": 151645, "vision_start_token_id": 151652, "vision_end_token_id": 151653,
             "image_token_id": 151655, "video_token_id": 151656, "hidden_size": 1536, "intermediate_size": 8960, "num_attention_heads": 12, "num_hidden_layers": 28,
             "num_key_value_heads": 2, "rms_norm_eps": 1e-06, "rope_theta": 1000000.0, "tie_word_embeddings": True, "vocab_size": 151936,
             "rope_scaling": {"type": "mrope", "mrope_section": [16, 24, 24]},
             "vision_config": {"depth": 32, "embed_dim": 1280, "mlp_ratio": 4, "num_heads": 16, "in_chans": 3, "hidden_size": 1536, "patch_size": 14, "spatial_merge_size": 2,
                               "spatial_patch_size": 14, "temporal_patch_size": 2}}
    assert VLMConfig.from_hf_config(q2_2b) == VLMConfig.qwen2vl_2b()
    pins = [[384 * i, 384 * j] for i in range(1, 7) for j in range(1, 7)]
    ov_7b = {"architectures": ["LlavaOnevisionForConditionalGeneration"], "model_type": "llava_onevision", "image_token_index": 151646, "video_token_index": 151647,
             "image_grid_pinpoints": pins, "vision_aspect_ratio": "anyres_max_9", "vision_feature_layer": -1, "vision_feature_select_strategy": "full", "tie_word_embeddings": False,
             "projector_hidden_act": "gelu",
             "text_config": {"model_type": "qwen2", "vocab_size": 152064, "hidden_size": 3584, "intermediate_size": 18944, "num_hidden_layers": 28, "num_attention_heads": 28,
                             "num_key_value_heads": 4, "rms_norm_eps": 1e-06, "rope_theta": 1000000.0, "eos_token_id": 151645},
             "vision_config": {"model_type": "siglip_vision_model", "hidden_size": 1152, "image_size": 384, "intermediate_size": 4304, "num_attention_heads": 16, "num_hidden_layers": 26,
                               "patch_size": 14, "vision_use_head": False}}
    import dataclasses
    want = dataclasses.replace(VLMConfig.llava_ov_7b(), vision_start_token_id=-1, vision_end_token_id=-1)
    assert VLMConfig.from_hf_config(ov_7b) == want
    with pytest.raises(ValueError, match="vision_feature_layer"):
        VLMConfig.from_hf_config(dict(ov_7b, vision_feature_layer=-2))


def test_pa_sft_llava_next_qwen_template_matches_the_reference():
    """The text path of scripts/train/PA_SFT/PA_SFT_LLaVA_OneVision_SI_*.sh (--template llava_next_qwen) against tests/golden/sft_llava.json = the reference's own
    template + LlavaNextPlugin on the same offline LlavaOnevisionProcessor (tools/make_golden_sft_llava.py): area-cap regularisation, the processor's crops and
    image sizes, `<image>` expansion to the packed feature count, ChatML turn pairs with the default / an explicit system prompt, a text-only row; and through
    train.py::encode_example: ids / labels with the image tokens intact and the per-image crop stacks the engine consumes."""
    import re
    import numpy as np
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd import llava_ov as lo
    from iadr1_amd import sft_data as sd
    from iadr1_amd.params import VLMConfig
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "sft_llava.json")))
    assert g["meta"]["default_system"] == sd.QWEN2_VL_DEFAULT_SYSTEM and g["meta"]["image_token"] == "<image>"
    proc = fx.local_llava_ov_processor()
    cfg = VLMConfig.from_dict(fx.TINY_OV)
    tokens_of = lambda size: lo.num_image_tokens(size, cfg.image_grid_pinpoints, cfg.v_image_size, cfg.v_side, cfg.anyres_max)
    collapse = lambda s: re.sub(r"(?:<image>)+", lambda m: "<image*%d>" % (len(m.group(0)) // 7), s)
    char_tok = type("T", (), {"encode": staticmethod(lambda text, add_special_tokens=False: [ord(c) for c in text])})
    for c in g["cases"]:
        pil = [sd.regularize_image_base(fx.synth_pil_image(w, h, seed), c["image_resolution"]) for w, h, seed in c["images"]]
        assert [[im.width, im.height] for im in pil] == c["regularized_sizes"]
        sizes = []
        if pil:
            feats = proc.image_processor(images=pil, return_tensors="pt")
            sizes = [list(map(int, s)) for s in feats["image_sizes"].tolist()]
            assert sizes == c["image_sizes"] and list(feats["pixel_values"].shape) == c["pixel_shape"]
            assert abs(float(feats["pixel_values"].double().abs().sum()) - c["pixel_abs_sum"]) <= 1e-6 * c["pixel_abs_sum"]
        msgs = sd.expand_image_placeholders_llava(c["messages"], sizes, tokens_of)
        assert [{**m, "content": collapse(m["content"])} for m in msgs] == c["expanded"]
        pairs = sd.encode_turns(char_tok, sd.llava_next_qwen_turn_texts([{**m, "content": collapse(m["content"])} for m in msgs], c["system"]))
        assert [[list(s), list(t)] for s, t in pairs] == c["pairs_collapsed"]
    with pytest.raises(ValueError, match="does not match"):
        sd.expand_image_placeholders_llava([{"role": "user", "content": "no placeholder"}], [(80, 100)], tokens_of)
    # the entry point's encoder on the real tokenizer of the processor
    m = _load("train/stage_sft/train.py")
    image_id = proc.tokenizer.convert_tokens_to_ids("<image>")
    row = {"prompt": [{"role": "user", "content": "<image>Any defect?"}], "response": [{"role": "assistant", "content": "<answer>No</answer>"}], "system": "",
           "images": [fx.synth_pil_image(100, 80, 1)]}
    ids, labels, pixels, grids = m.encode_example(proc, row, 4096, image_token_id=image_id, template="llava_next_qwen", cfg=cfg)
    assert grids == [(80, 100)] and ids.count(image_id) == 70 and len(pixels) == 1 and tuple(pixels[0].shape) == (5, 3, 56, 56)
    sup = [t for t in labels if t != -100]
    assert sup == proc.tokenizer.encode("<answer>No</answer><|im_end|>\n", add_special_tokens=False) and len(ids) == len(labels)
    with pytest.raises(ValueError, match="truncates image placeholder"):
        m.encode_example(proc, row, 60, image_token_id=image_id, template="llava_next_qwen", cfg=cfg)


@pytest.mark.parametrize("template,cfg_name", [("llava", "TINY_LLAVA15"), ("llava_next_mistral", "TINY_LLAVA_NEXT")])
def test_pa_sft_llava_and_llava_next_mistral_templates_match_the_reference(template, cfg_name):
    """The text path of scripts/train/PA_SFT/PA_SFT_LLaVA_1_5.sh (--template llava: vicuna turns, fixed image_seqlen per image) and PA_SFT_LLaVA_1_6.sh
    (--template llava_next_mistral: BOS + [INST] turns with the system prompt folded into the first one, any-resolution feature counts) against the reference's own
    template / plugin code on offline transformers Llava / LlavaNext processors (tests/golden/sft_llava.json "other_templates")."""
    import re
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd import llava_ov as lo
    from iadr1_amd import sft_data as sd
    from iadr1_amd.params import VLMConfig
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "sft_llava.json")))["other_templates"][template]
    cfg_d = getattr(fx, cfg_name)
    cfg = VLMConfig.from_dict(cfg_d)
    proc = fx.local_llava_processor(cfg_d)
    if template == "llava":
        assert g["default_system"] == sd.LLAVA_DEFAULT_SYSTEM
        tokens_of = lambda size: cfg.v_tokens
    else:
        assert g["default_system"] == ""
        tokens_of = lambda size: lo.num_image_tokens(size, cfg.image_grid_pinpoints, cfg.v_image_size, cfg.v_side, cfg.anyres_max)
    collapse = lambda s: re.sub(r"(?:<image>)+", lambda m: "<image*%d>" % (len(m.group(0)) // 7), s)
    char_tok = type("T", (), {"encode": staticmethod(lambda text, add_special_tokens=False: [ord(c) for c in text]), "bos_token_id": 1, "eos_token_id": 2})
    for c in g["cases"]:
        pil = [sd.regularize_image_base(fx.synth_pil_image(w, h, seed), c["image_resolution"]) for w, h, seed in c["images"]]
        sizes = [(cfg.v_image_size, cfg.v_image_size)] * len(pil)
        if pil:
            feats = proc.image_processor(images=pil, return_tensors="pt")
            assert list(feats["pixel_values"].shape) == c["pixel_shape"] and abs(float(feats["pixel_values"].double().abs().sum()) - c["pixel_abs_sum"]) <= 1e-6 * c["pixel_abs_sum"]
            if template != "llava":
                sizes = [list(map(int, s)) for s in feats["image_sizes"].tolist()]
                assert sizes == c["image_sizes"]
        msgs = sd.expand_image_placeholders_llava(c["messages"], sizes, tokens_of)
        assert [{**m, "content": collapse(m["content"])} for m in msgs] == c["expanded"]
        pairs = sd.encode_turns(char_tok, sd.TURN_TEXTS[template]([{**m, "content": collapse(m["content"])} for m in msgs], c["system"]))
        assert [[list(s), list(t)] for s, t in pairs] == c["pairs_collapsed"]
    m = _load("train/stage_sft/train.py")
    image_id = proc.tokenizer.convert_tokens_to_ids("<image>")
    row = {"prompt": [{"role": "user", "content": "<image>Any defect?"}], "response": [{"role": "assistant", "content": "No"}], "system": "", "images": [fx.synth_pil_image(100, 80, 1)]}
    ids, labels, pixels, grids = m.encode_example(proc, row, 4096, image_token_id=image_id, template=template, cfg=cfg)
    assert ids.count(image_id) == (16 if template == "llava" else 70) and len(pixels) == 1 and tuple(pixels[0].shape) == ((1, 3, 56, 56) if template == "llava" else (5, 3, 56, 56))
    sup = [t for t in labels if t != -100]
    assert sup[-1] == proc.tokenizer.eos_token_id and len(ids) == len(labels) and (template == "llava" or ids[0] == proc.tokenizer.bos_token_id)


def test_combine_batches_and_trim_completions():
    """Host half of GRPOConfig.batch_rollouts (one group rollout for all micro-batches of an optimizer step): prompts of different lengths are LEFT-padded into one
    batch with their image tensors / grids concatenated in prompt order; completions are cut to the longest one of a micro-batch, as the reference's right padding
    of the vLLM outputs does (REF:680-683).  The any-resolution processors pad the crop dimension per call: such batches are not combined (None)."""
    import numpy as np
    import torch
    from iadr1_amd.trainer import combine_batches, trim_completions
    a = {"input_ids": np.array([[7, 8, 9]]), "attention_mask": np.ones((1, 3), dtype=np.int64), "pixel_values": torch.ones(4, 6), "image_grid_thw": [(1, 2, 2)]}
    b = {"input_ids": torch.tensor([[0, 5, 6, 7, 8], [1, 2, 3, 4, 5]]), "attention_mask": torch.tensor([[0, 1, 1, 1, 1], [1, 1, 1, 1, 1]]), "pixel_values": torch.zeros(8, 6),
         "image_grid_thw": torch.tensor([[1, 2, 2], [1, 2, 2]]), "images_per_prompt": [1, 1]}
    c = combine_batches([a, b], pad_token_id=99)
    assert c["input_ids"].tolist() == [[99, 99, 7, 8, 9], [0, 5, 6, 7, 8], [1, 2, 3, 4, 5]] and c["attention_mask"].tolist() == [[0, 0, 1, 1, 1], [0, 1, 1, 1, 1], [1, 1, 1, 1, 1]]
    assert c["pixel_values"].shape == (12, 6) and float(c["pixel_values"][:4].sum()) == 24.0 and c["image_grid_thw"] == [(1, 2, 2)] * 3 and c["images_per_prompt"] == [1, 1, 1]
    ov1 = {"input_ids": np.array([[1, 2]]), "attention_mask": np.ones((1, 2), dtype=np.int64), "pixel_values": torch.zeros(1, 5, 3, 4, 4), "image_sizes": [(8, 8)]}
    ov2 = dict(ov1, pixel_values=torch.zeros(1, 3, 3, 4, 4))
    assert combine_batches([ov1, ov2], 0) is None and combine_batches([ov1, ov1], 0)["image_sizes"] == [(8, 8), (8, 8)]
    t = np.array([[5, 6, 2, 0, 0, 0], [5, 2, 0, 0, 0, 0]])
    assert trim_completions(t, eos_token_id=2).tolist() == [[5, 6, 2], [5, 2, 0]]
    assert trim_completions(np.full((2, 4), 9), eos_token_id=2).shape == (2, 4) and trim_completions(np.array([[2, 0, 0]]), 2).shape == (1, 1)
