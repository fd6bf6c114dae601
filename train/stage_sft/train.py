#!/usr/bin/env python3
"""PA-SFT entry point (reference: /root/reference/train/stage_sft/train.py:18-28 -> llamafactory run_exp ->
run_sft, train/sft/workflow.py:40-132), accepting the LLaMA-Factory flags the reference's PA_SFT_*.sh scripts pass
(scripts/train/PA_SFT/*.sh:25-50) and driving the MI355X SFT engine (iadr1_amd.sft).

Data: the `sharegpt` manifests registered in data/dataset_info.json (columns `messages` / `images`); rendering goes
through the checkpoint's own chat template (the `qwen2_vl` template of the reference expands `<image>` to
<|vision_start|><|image_pad|>xN<|vision_end|>, llamafactory/data/mm_plugin.py:850-896 -- the HF processor does the
same expansion), prompt turns are masked with -100 (processors/supervised.py:34-87), cutoff_len truncation, loss
curve written to <output_dir>/trainer_log.jsonl (train/callbacks.py:279-318)."""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build_parser():
    p = argparse.ArgumentParser(allow_abbrev=False)
    p.add_argument("--stage", default="sft")
    p.add_argument("--do_train", nargs="?", const=True, default=True)
    p.add_argument("--model_name_or_path", required=True)
    p.add_argument("--dataset", required=True, help="name in dataset_info.json, or a path to a sharegpt json")
    p.add_argument("--dataset_dir", default="data")
    p.add_argument("--template", default="qwen2_vl")
    p.add_argument("--finetuning_type", default="full")
    p.add_argument("--output_dir", required=True)
    p.add_argument("--per_device_train_batch_size", type=int, default=1)
    p.add_argument("--gradient_accumulation_steps", type=int, default=2)
    p.add_argument("--learning_rate", type=float, default=1e-5)
    p.add_argument("--weight_decay", type=float, default=0.1)
    p.add_argument("--lr_scheduler_type", default="cosine")
    p.add_argument("--warmup_steps", type=int, default=100)
    p.add_argument("--num_train_epochs", type=float, default=1.0)
    p.add_argument("--max_steps", type=int, default=-1)
    p.add_argument("--cutoff_len", type=int, default=4096)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--logging_steps", type=int, default=1)
    p.add_argument("--save_steps", type=int, default=500)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--micro_batch_seqs", type=int, default=16)
    p.add_argument("--train_on_prompt", nargs="?", const=True, default=False, type=lambda v: str(v).lower() in ("1", "true", "yes"))
    p.add_argument("--mask_history", nargs="?", const=True, default=False, type=lambda v: str(v).lower() in ("1", "true", "yes"))
    for flag in ("--deepspeed", "--bf16", "--plot_loss", "--overwrite_cache", "--overwrite_output_dir", "--ddp_timeout", "--preprocessing_num_workers",
                 "--report_to", "--gradient_checkpointing", "--flash_attn", "--image_max_pixels", "--image_min_pixels"):
        p.add_argument(flag, nargs="?", default=None, const=True)
    return p


def load_sharegpt(name: str, dataset_dir: str):
    path = name
    cols = {"messages": "messages", "images": "images"}
    info_p = os.path.join(dataset_dir, "dataset_info.json")
    if not os.path.exists(path) and os.path.exists(info_p):
        info = json.load(open(info_p))[name]
        path = info["file_name"] if os.path.isabs(info["file_name"]) else os.path.join(dataset_dir, info["file_name"])
        cols.update(info.get("columns", {}))
    rows = json.load(open(path))
    return [{"messages": r[cols["messages"]], "images": r.get(cols["images"], [])} for r in rows]


def encode_example(proc, row, cutoff_len, train_on_prompt=False, mask_history=False, image_token_id=151655):
    """-> (input_ids, labels, pixel_values, grids).  Per-turn truncation and masking are iadr1_amd.sft_data.supervised_labels (the reference's
    _encode_supervised_example, llamafactory/data/processors/supervised.py:33-87): answers supervised, prompts -100, turns trimmed against cutoff_len."""
    from iadr1_amd.sft_data import supervised_labels
    from PIL import Image
    msgs = []
    for m in row["messages"]:
        role = {"human": "user", "gpt": "assistant"}.get(m.get("from", m.get("role")), m.get("from", m.get("role")))
        text = m.get("value", m.get("content"))
        parts, segs = [], text.split("<image>")
        for i, s in enumerate(segs):
            if i:
                parts.append({"type": "image"})
            if s:
                parts.append({"type": "text", "text": s})
        msgs.append({"role": role, "content": parts})
    images = [Image.open(p) if isinstance(p, str) else p for p in row["images"]]
    turns, prev, full = [], 0, None      # (prompt_ids, answer_ids) per turn, the unit the reference truncates and masks by
    for t in range(len(msgs)):
        if msgs[t]["role"] != "assistant":
            continue
        upto = proc.apply_chat_template(msgs[: t + 1], tokenize=False)
        before = proc.apply_chat_template(msgs[:t], tokenize=False, add_generation_prompt=True)
        full = proc(text=[upto], images=images or None, return_tensors="pt", add_special_tokens=False)
        n_before = proc(text=[before], images=images or None, return_tensors="pt", add_special_tokens=False)["input_ids"].shape[1]
        cur = full["input_ids"][0].tolist()
        turns.append((cur[prev:n_before], cur[n_before:]))
        prev = len(cur)
    ids, labels = supervised_labels(turns, cutoff_len, train_on_prompt=train_on_prompt, mask_history=mask_history)
    if images and ids.count(image_token_id) != sum(1 for tn in turns for part in tn for x in part if x == image_token_id):
        raise ValueError("cutoff_len=%d truncates image placeholder tokens; raise --cutoff_len" % cutoff_len)
    return ids, labels, (full["pixel_values"] if images else None), (full["image_grid_thw"].tolist() if images else [])


def main(argv=None):
    a = build_parser().parse_args(argv)
    if a.stage != "sft" or a.finetuning_type != "full":
        raise ValueError("only --stage sft --finetuning_type full is part of the IAD-R1 PA-SFT path")
    import numpy as np
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    group = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
        group = dist.group.WORLD
    import iadr1_amd  # noqa: F401
    from transformers import AutoProcessor

    from iadr1_amd.sft import SFTArgs, SFTEngine
    from iadr1_amd.trainer import load_checkpoint, save_checkpoint

    cfg, store = load_checkpoint(a.model_name_or_path, dev, trainable=True)
    proc = AutoProcessor.from_pretrained(a.model_name_or_path)
    eng = SFTEngine(cfg, store, SFTArgs(learning_rate=a.learning_rate, weight_decay=a.weight_decay, max_grad_norm=a.max_grad_norm,
                                        gradient_accumulation_steps=a.gradient_accumulation_steps, micro_batch_seqs=a.micro_batch_seqs), group=group)
    rows = load_sharegpt(a.dataset, a.dataset_dir)[rank::world]
    bs, ga = a.per_device_train_batch_size, a.gradient_accumulation_steps
    total = a.max_steps if a.max_steps > 0 else int(math.ceil(max(1, len(rows) // (bs * ga)) * a.num_train_epochs))
    os.makedirs(a.output_dir, exist_ok=True)
    log = open(os.path.join(a.output_dir, "trainer_log.jsonl"), "a") if rank == 0 else None
    pad = cfg.pad_token_id
    i, t0 = 0, time.time()
    for step in range(total):
        lr = a.learning_rate * (step + 1) / a.warmup_steps if step < a.warmup_steps else (
            a.learning_rate * 0.5 * (1 + math.cos(math.pi * (step - a.warmup_steps) / max(1, total - a.warmup_steps))) if a.lr_scheduler_type == "cosine" else a.learning_rate)
        eng.args.learning_rate = lr
        losses = []
        for k in range(ga):
            enc = [encode_example(proc, rows[(i + j) % len(rows)], a.cutoff_len, a.train_on_prompt, a.mask_history, cfg.image_token_id) for j in range(bs)]
            i += bs
            S = (max(len(e[0]) for e in enc) + 7) // 8 * 8  # pad_to_multiple_of=8 (sft/workflow.py:60), right padding
            ids = np.full((bs, S), pad, dtype=np.int64)
            mask = np.zeros((bs, S), dtype=np.int64)
            labels = np.full((bs, S), -100, dtype=np.int64)
            for r, (x, y, _, _) in enumerate(enc):
                ids[r, : len(x)], mask[r, : len(x)], labels[r, : len(y)] = x, 1, y
            pv = torch.cat([e[2] for e in enc if e[2] is not None], 0)
            grids = [tuple(g) for e in enc for g in e[3]]
            batch = {"input_ids": ids, "attention_mask": mask, "labels": labels, "pixel_values": pv, "image_grid_thw": grids,
                     "images_per_row": [len(e[3]) for e in enc]}
            losses.append(eng.loss_and_grads(batch, last_micro_step=(k == ga - 1)))
        eng.optimizer_step()
        if log and (step + 1) % a.logging_steps == 0:
            log.write(json.dumps({"current_steps": step + 1, "total_steps": total, "loss": float(np.mean(losses)), "lr": lr, "elapsed_time": round(time.time() - t0, 1)}) + "\n")
            log.flush()
        if rank == 0 and a.save_steps and (step + 1) % a.save_steps == 0:
            save_checkpoint(store, os.path.join(a.output_dir, f"checkpoint-{step + 1}"), json.load(open(os.path.join(a.model_name_or_path, "config.json"))))
    if rank == 0:
        save_checkpoint(store, a.output_dir, json.load(open(os.path.join(a.model_name_or_path, "config.json"))))
        proc.save_pretrained(a.output_dir)


if __name__ == "__main__":
    main()
