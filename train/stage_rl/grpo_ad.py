#!/usr/bin/env python3
"""SC-GRPO entry point, CLI-compatible with the reference's `train/stage_rl/grpo_ad.py`
(/root/reference/train/stage_rl/grpo_ad.py:31-65 script arguments, :72-118 prompt templates, :126-129 reward
registry, :135-181 dataset row -> conversation, :188-207 trainer construction / train / save), driving the
MI355X engine.  Launch exactly like the reference scripts do, one process per GPU:

    torchrun --nproc_per_node=8 --master-addr 127.0.0.1 train/stage_rl/grpo_ad.py \
        --model_name_or_path <Qwen2.5-VL checkpoint dir> --dataset_name data.json --image_path /data \
        --num_generations 8 --max_prompt_length 4096 --max_completion_length 512 --output_dir out ...

Flags of the reference that configure machinery this engine does not have (DeepSpeed, vLLM placement, wandb,
gradient checkpointing, attention implementation) are accepted and ignored.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

FORMAT_RULES = (
    "If you find anomalies in the test image, structure your response with the following format:"
    "<think>[Your process of observation and reasoning is here]</think>"
    "<location>[The location of the anomaly in the image]</location>"
    "<type>[The type of anomaly in the image]</type><answer>[Your final answer is here(yes or no)]</answer>"
    "If no anomalies are detected in the test image, structure your response with the following format:"
    "<think>[Your process of observation and reasoning is here]</think>"
    "<answer>[Your final answer is here(yes or no)]</answer>"
)
# 0-shot (single image) and 1-shot (reference + test image) wording of the reference, REF grpo_ad.py:72-118
PROMPTS = {
    1: {
        "system": "You are an expert in detecting anomalies in image. Your task is to detect if there are any anomalies in the test image."
                  + FORMAT_RULES + "{Question}",
        "question": "You are an expert in detecting defects in image. Your task is to detect if there are any defects in the test image.{Question}",
    },
    0: {
        "system": "You are an expert in detecting anomalies in images. I will provide you with two images: a reference image (first) showing a normal object without defects, and a test image (second) that needs inspection."
                  "Your task is to compare these images and determine if there are any anomalies in the test image. Use the reference image as a baseline for what is considered normal."
                  + FORMAT_RULES +
                  "Remember that the first image is always the reference (normal) image, and the second image is the test image that needs inspection.{Question}",
        "question": "You are an expert in detecting defects in image. I will provide you with two images: a reference image (first) showing a normal object without defects, and a test image (second) that needs inspection."
                    "Your task is to compare these images and determine if there are any anomalies in the test image. Use the reference image as a baseline for what is considered normal.{Question}",
    },
}


def str2bool(v):
    return str(v).lower() not in ("false", "0", "no")


def build_parser():
    p = argparse.ArgumentParser(allow_abbrev=False)
    # GRPOScriptArguments
    p.add_argument("--dataset_name", required=True)
    p.add_argument("--dataset_train_split", default="train")
    p.add_argument("--reward_funcs", nargs="+", default=["accuracy", "format"])
    p.add_argument("--use_vllm_for_gen", default="true")
    p.add_argument("--use_system_prompt", default="false")
    p.add_argument("--image_path", default="/data")
    p.add_argument("--max_pixels", type=int, default=12845056)
    p.add_argument("--min_pixels", type=int, default=3136)
    p.add_argument("--single_img", type=int, default=1)
    # ModelConfig
    p.add_argument("--model_name_or_path", required=True)
    p.add_argument("--attn_implementation", default="flash_attention_2")
    p.add_argument("--torch_dtype", default=None)
    # GRPOConfig / TrainingArguments subset; defaults = theirs (trl/trl/trainer/grpo_config.py, transformers TrainingArguments: 3 epochs, batch 8, logging / saving every 500
    # steps, linear schedule without warm-up).  These are only the fall-backs: all seven scripts/train/SC_GRPO/*.sh pass --num_train_epochs 1
    # --logging_steps 1 --save_steps 100 --per_device_train_batch_size 1 (checked against tests/golden/launch_flags.json in tests/test_entrypoints.py)
    p.add_argument("--output_dir", required=True)
    p.add_argument("--per_device_train_batch_size", type=int, default=8)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--num_generations", type=int, default=8)
    p.add_argument("--max_prompt_length", type=int, default=512)
    p.add_argument("--max_completion_length", type=int, default=256)
    p.add_argument("--beta", type=float, default=0.04)
    p.add_argument("--temperature", type=float, default=0.9)
    p.add_argument("--learning_rate", type=float, default=1e-6)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--lr_scheduler_type", default="linear")
    p.add_argument("--warmup_steps", type=int, default=0)
    p.add_argument("--num_train_epochs", type=float, default=3.0)
    p.add_argument("--max_steps", type=int, default=-1)
    p.add_argument("--logging_steps", type=int, default=500)
    p.add_argument("--save_steps", type=int, default=500)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--micro_batch_seqs", type=int, default=64)
    p.add_argument("--decode_weights", default="bf16", choices=["bf16", "fp8"],
                   help="fp8: the rollout streams the gate|up / down / lm_head weights as e4m3 + per-row scales (BASELINE config 5); the loss and its gradients stay bf16")
    p.add_argument("--run_name", default=None)
    # ModelConfig's LoRA switch: refused with the reason (the trainer's peft_config error), not silently ignored
    p.add_argument("--use_peft", nargs="?", default=False, const=True)
    # accepted for script compatibility, no effect here
    # decoder activations recomputed in backward when they would not fit comfortably in HBM (iadr1_amd.vlm.Engine.recompute_wanted); every reference script passes true
    p.add_argument("--gradient_checkpointing", nargs="?", default=False, const=True, type=lambda v: str(v).lower() in ("1", "true", "yes"))
    for flag in ("--deepspeed", "--report_to", "--bf16", "--ddp_timeout", "--push_to_hub"):
        p.add_argument(flag, nargs="?", default=None, const=True)
    return p


def parse_args_and_config(parser: argparse.ArgumentParser, argv=None):
    """`TrlParser.parse_args_and_config` (REF trl/trl/scripts/utils.py:165-223, used at grpo_ad.py:210-213): `--config file.yaml` supplies values that
    REPLACE the parser defaults (and make required flags optional), command-line flags override them, an `env:` block is exported to os.environ, and
    a YAML key no flag knows is an error (HfArgumentParser raises on unconsumed strings unless asked to return them)."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if "--config" in argv:
        i = argv.index("--config")
        argv.pop(i)
        if i >= len(argv):
            raise ValueError("--config needs a YAML file path")
        import yaml
        with open(argv.pop(i)) as f:
            cfg = yaml.safe_load(f) or {}
        if "env" in cfg:
            env = cfg.pop("env") or {}
            if not isinstance(env, dict):
                raise ValueError("`env` field should be a dict in the YAML file.")
            for k, v in env.items():
                os.environ[k] = str(v)
        for action in parser._actions:
            if action.dest in cfg:
                v = cfg.pop(action.dest)
                action.default = action.type(v) if (action.type is not None and v is not None and not isinstance(v, (list, bool))) else v
                action.required = False
        if cfg:
            raise ValueError(f"Some keys of the --config file are not used by the parser: {sorted(cfg)}")
    return parser.parse_args(argv)


def make_conversation(example: dict, image_path: str, use_system_prompt: bool, single_img: int) -> dict:
    """Dataset row {problem, image, solution, ...} -> {"prompt": chat, "image": [abs paths]} (REF grpo_ad.py:135-181)."""
    img = example.get("image")
    if not img:
        raise ValueError("row without an image")
    items = img if isinstance(img, list) else [img]
    paths = []
    for it in items:
        if isinstance(it, str):
            paths.append(os.path.join(image_path, it))
        elif isinstance(it, dict):
            paths.append(os.path.join(image_path, it["path"]))
        else:
            raise TypeError("Unsupported Format.")
    tmpl = PROMPTS[single_img]
    pics = [{"type": "image"} for _ in paths]
    if use_system_prompt:
        prompt = [{"role": "system", "content": tmpl["system"]}, {"role": "user", "content": pics + [{"type": "text", "text": example["problem"]}]}]
    else:
        prompt = [{"role": "user", "content": pics + [{"type": "text", "text": tmpl["question"].format(Question=example["problem"])}]}]
    out = {k: v for k, v in example.items() if k not in ("messages",)}
    out.update(prompt=prompt, image=paths)
    return out


def load_rows(path: str):
    with open(path) as f:
        txt = f.read().strip()
    return json.loads(txt) if txt.startswith("[") else [json.loads(l) for l in txt.splitlines() if l.strip()]


PEFT_UNSUPPORTED = ("--use_peft / peft_config: LoRA is not part of this path.  In the reference it cannot run either: SCGRPOTrainer wraps the policy with get_peft_model "
                    "(sc_grpo_trainer.py:149-150) but _move_model_to_vllm (:569-579) hands the wrapped state_dict (base_model.model.* / lora_A / lora_B names, adapters not "
                    "merged) to vLLM's load_weights, and generation without vLLM raises (:720); none of its launch scripts passes --use_peft")


def main(argv=None):
    a = parse_args_and_config(build_parser(), argv)
    if a.single_img not in (0, 1):
        raise ValueError("The single_img parameter can only be 0 or 1")
    if str2bool(a.use_peft):
        raise ValueError(PEFT_UNSUPPORTED)
    if not a.dataset_name.endswith((".json", ".jsonl")):
        raise ValueError("dataset_name must be a .json/.jsonl manifest (the reference loads it with load_dataset('json'))")
    import torch
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    import iadr1_amd  # noqa: F401
    from iadr1_amd.rewards import REWARD_FUNCS
    if a.decode_weights == "fp8":
        os.environ["IADR1_DECODE_WEIGHTS"] = "fp8"       # read by the parameter stores the trainer builds
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer

    rows = [make_conversation(r, a.image_path, str2bool(a.use_system_prompt), a.single_img) for r in load_rows(a.dataset_name)]
    cfg = GRPOConfig(**{k: getattr(a, k) for k in GRPOConfig.__dataclass_fields__ if hasattr(a, k) and getattr(a, k) is not None and k not in ("report_to", "bf16", "push_to_hub")})
    trainer = SCGRPOTrainer(model=a.model_name_or_path, reward_funcs=[REWARD_FUNCS[n] for n in a.reward_funcs], args=cfg, train_dataset=rows,
                            attn_implementation=a.attn_implementation, max_pixels=a.max_pixels, min_pixels=a.min_pixels, use_vllm_for_gen=str2bool(a.use_vllm_for_gen))
    trainer.train()
    if int(os.environ.get("RANK", "0")) == 0:
        trainer.save_model(a.output_dir)


if __name__ == "__main__":
    main()
