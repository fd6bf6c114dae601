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


# the flags of scripts/train/SC_GRPO/SC_GRPO_Qwen_Instruct_2_5_VL_3B.sh:40-63 (values are placeholders)
SC_GRPO_FLAGS = ("--deepspeed zero3.json --output_dir out --model_name_or_path /m/Qwen2.5-VL-3B --dataset_name d.json --image_path /data "
                 "--use_vllm_for_gen true --use_system_prompt false --max_prompt_length 4096 --max_completion_length 512 --num_generations 4 "
                 "--per_device_train_batch_size 1 --gradient_accumulation_steps 2 --logging_steps 1 --bf16 --report_to wandb "
                 "--gradient_checkpointing true --attn_implementation flash_attention_2 --max_pixels 480000 --save_steps 100 "
                 "--num_train_epochs 1 --run_name x --single_img 1").split()
# scripts/train/PA_SFT/PA_SFT_Qwen_Instruct_2_5_VL_3B.sh:25-50
PA_SFT_FLAGS = ("--deepspeed zero3.json --stage sft --do_train --model_name_or_path /m/q --dataset expert_ad --template qwen2_vl --finetuning_type full "
                "--output_dir out --overwrite_cache --overwrite_output_dir --warmup_steps 100 --weight_decay 0.1 --per_device_train_batch_size 1 "
                "--gradient_accumulation_steps 2 --ddp_timeout 90000 --learning_rate 1e-5 --lr_scheduler_type cosine --logging_steps 1 --cutoff_len 4096 "
                "--save_steps 500 --plot_loss --num_train_epochs 1 --bf16").split()


def test_sc_grpo_cli_accepts_reference_flags():
    m = _load("train/stage_rl/grpo_ad.py")
    a = m.build_parser().parse_args(SC_GRPO_FLAGS)
    assert a.num_generations == 4 and a.max_completion_length == 512 and a.max_pixels == 480000 and a.beta == 0.04 and a.learning_rate == 1e-6


def test_pa_sft_cli_accepts_reference_flags():
    m = _load("train/stage_sft/train.py")
    a = m.build_parser().parse_args(PA_SFT_FLAGS)
    assert a.learning_rate == 1e-5 and a.weight_decay == 0.1 and a.warmup_steps == 100 and a.cutoff_len == 4096 and a.lr_scheduler_type == "cosine"


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


def test_trainer_constructor_errors_match_reference():
    import iadr1_amd  # noqa: F401
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer
    with pytest.raises(ValueError, match="Invalid `torch_dtype`"):
        SCGRPOTrainer("/x/Qwen2.5-VL-3B", [], args=GRPOConfig(model_init_kwargs={"torch_dtype": 7}))
    with pytest.raises(ValueError, match="already instantiated"):
        SCGRPOTrainer((None, {}), [], args=GRPOConfig(model_init_kwargs={"a": 1}))
    with pytest.raises(ValueError, match="Qwen2.5-VL"):
        SCGRPOTrainer("/x/llava-1.5-7b", [], args=GRPOConfig())


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
