#!/usr/bin/env python3
"""PA-SFT entry point (reference: /root/reference/train/stage_sft/train.py:18-28 -> llamafactory run_exp ->
run_sft, train/sft/workflow.py:40-132), accepting the LLaMA-Factory flags the reference's PA_SFT_*.sh scripts pass
(scripts/train/PA_SFT/*.sh:25-50) and driving the MI355X SFT engine (iadr1_amd.sft).

Data: the `sharegpt` manifests registered in data/dataset_info.json; alignment, image regularisation, `<image>` expansion, the `qwen2_vl`
template, per-turn truncation and label masking are the reference's, restated in iadr1_amd.sft_data and pinned by goldens produced by the
reference's own functions (tests/golden/sft_data.json, sft_text.json); the HF processor supplies the tokenizer and the image processor.  Loss
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
    # defaults = the reference's (transformers TrainingArguments / LLaMA-Factory DataArguments: data_args.py:41-57, model_args.py:62); the launch scripts override most of them
    p.add_argument("--stage", default="sft")
    p.add_argument("--do_train", nargs="?", const=True, default=True)
    p.add_argument("--model_name_or_path", required=True)
    p.add_argument("--dataset", required=True, help="name in dataset_info.json, or a path to a sharegpt json")
    p.add_argument("--dataset_dir", default="data")
    p.add_argument("--image_dir", default=None, help="folder the manifests' relative image paths live under; defaults to --dataset_dir (llamafactory hparams/data_args.py:44,136-137)")
    p.add_argument("--template", default="qwen2_vl")
    p.add_argument("--finetuning_type", default="full")
    p.add_argument("--output_dir", required=True)
    p.add_argument("--per_device_train_batch_size", type=int, default=8)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--learning_rate", type=float, default=5e-5)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--lr_scheduler_type", default="linear")
    p.add_argument("--warmup_steps", type=int, default=0)
    p.add_argument("--num_train_epochs", type=float, default=3.0)
    p.add_argument("--max_steps", type=int, default=-1)
    p.add_argument("--cutoff_len", type=int, default=2048)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--logging_steps", type=int, default=500)
    p.add_argument("--save_steps", type=int, default=500)
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--disable_shuffling", nargs="?", const=True, default=False, type=lambda v: str(v).lower() in ("1", "true", "yes"))
    p.add_argument("--micro_batch_seqs", type=int, default=16)
    # llamafactory hparams/finetuning_args.py:416-427 (defaults kept: no launch script passes them, so the registered families train with a frozen tower + projector)
    tf = lambda v: str(v).lower() in ("1", "true", "yes")
    p.add_argument("--freeze_vision_tower", nargs="?", const=True, default=True, type=tf)
    p.add_argument("--freeze_multi_modal_projector", nargs="?", const=True, default=True, type=tf)
    p.add_argument("--train_mm_proj_only", nargs="?", const=True, default=False, type=tf)
    p.add_argument("--image_resolution", type=int, default=512 * 512)
    p.add_argument("--resume_from_checkpoint", default=None, help="checkpoint-N directory with a training state; default: the last one under output_dir unless --overwrite_output_dir")
    p.add_argument("--train_on_prompt", nargs="?", const=True, default=False, type=lambda v: str(v).lower() in ("1", "true", "yes"))
    p.add_argument("--mask_history", nargs="?", const=True, default=False, type=lambda v: str(v).lower() in ("1", "true", "yes"))
    for flag in ("--deepspeed", "--bf16", "--plot_loss", "--overwrite_cache", "--overwrite_output_dir", "--ddp_timeout", "--preprocessing_num_workers",
                 "--report_to", "--flash_attn", "--image_max_pixels", "--image_min_pixels"):
        p.add_argument(flag, nargs="?", default=None, const=True)
    # decoder activations recomputed in backward when they would not fit comfortably in HBM (iadr1_amd.vlm.Engine.recompute_wanted); the PA_SFT scripts do not pass it
    p.add_argument("--gradient_checkpointing", nargs="?", default=False, const=True, type=tf)
    return p


def load_sharegpt(name: str, dataset_dir: str, image_dir: str | None = None):
    """Rows of a sharegpt-format dataset, aligned the way the reference aligns them (iadr1_amd.sft_data.align_sharegpt).  `name` is an entry of
    <dataset_dir>/dataset_info.json (file_name / formatting / columns / tags, LLaMA-Factory schema) or, failing that, a path to a json file in the
    README's Expert-AD layout (README.md:71-99: "messages" with role / content, "images")."""
    from iadr1_amd.sft_data import ShareGPTSchema, align_sharegpt
    info_p = os.path.join(dataset_dir, "dataset_info.json")
    if not os.path.exists(name) and os.path.exists(info_p):
        info = json.load(open(info_p))
        if name not in info:
            raise ValueError("Undefined dataset %s in dataset_info.json." % name)
        entry = info[name]
        path = entry["file_name"] if os.path.isabs(entry["file_name"]) else os.path.join(dataset_dir, entry["file_name"])
        schema = ShareGPTSchema.from_dataset_info(entry)
    else:
        path = name
        schema = ShareGPTSchema(messages="messages", images="images", role_tag="role", content_tag="content", user_tag="user", assistant_tag="assistant")
    # a relative image path is joined with image_dir when that file exists, otherwise left as it is (llamafactory data/aligner.py:52-53)
    rows = [align_sharegpt(r, schema, image_dir=image_dir or dataset_dir) for r in json.load(open(path))]
    return [r for r in rows if r["prompt"]]      # the reference filters out the rows its aligner emptied (odd turn counts, roles out of order)


TEMPLATES = ("qwen2_vl", "llava_next_qwen", "llava", "llava_next_mistral")
# template -> model family it belongs to (VLMConfig.llava_family; "" = Qwen2-VL / Qwen2.5-VL)
TEMPLATE_FAMILY = {"qwen2_vl": "", "llava_next_qwen": "onevision", "llava": "llava", "llava_next_mistral": "next"}


def encode_example(proc, row, cutoff_len, train_on_prompt=False, mask_history=False, image_token_id=151655, image_resolution=512 * 512, template="qwen2_vl", cfg=None):
    """Aligned row -> (input_ids, labels, pixel_values, grids), the reference's supervised preprocessing end to end (iadr1_amd.sft_data).
    "qwen2_vl": images regularised (mm_plugin.py:108-123,810-824), "<image>" expanded to the vision tokens of its patch grid (mm_plugin.py:850-896), ChatML
    turns tokenised piecewise (template.py:85-160,1120-1133), per-turn budget and label mask (processors/supervised.py:33-87); grids = [t, h, w] per image.
    The llava* templates (the LLaVA scripts): the base plugin's area cap only; "llava_next_qwen" / "llava_next_mistral": the any-resolution image processor's crops,
    "<image>" -> packed feature count copies of the image token (LlavaNextPlugin, mm_plugin.py:327-366); "llava": one crop per image, "<image>" -> image_seqlen =
    (image_size / patch_size)^2 copies (LlavaPlugin, mm_plugin.py:287-311).  pixel_values = list of per-image crop stacks, grids = (height, width) per image."""
    from iadr1_amd.sft_data import TURN_TEXTS, encode_turns, expand_image_placeholders, expand_image_placeholders_llava, regularize_image, regularize_image_base, supervised_labels
    family = TEMPLATE_FAMILY[template]
    images = []
    for im in row["images"] or []:
        if isinstance(im, str):
            from PIL import Image
            im = Image.open(im)
        images.append(regularize_image_base(im, image_resolution) if family else regularize_image(im, image_resolution))
    feats = proc.image_processor(images=images, return_tensors="pt") if images else None
    if family:
        from iadr1_amd import llava_ov
        if family == "llava":
            grids = [(cfg.v_image_size, cfg.v_image_size)] * len(images)
            tokens_of = lambda size: cfg.v_tokens
            pixels = [feats["pixel_values"][i: i + 1] for i in range(len(images))] if images else None
        else:
            grids = [tuple(int(v) for v in s) for s in feats["image_sizes"].tolist()] if images else []
            tokens_of = lambda size: llava_ov.num_image_tokens(size, cfg.image_grid_pinpoints, cfg.v_image_size, cfg.v_side, cfg.anyres_max)
            pixels = [feats["pixel_values"][i, : llava_ov.num_crops(g, cfg.image_grid_pinpoints, cfg.v_image_size)] for i, g in enumerate(grids)] if images else None
        msgs = expand_image_placeholders_llava(row["prompt"] + row["response"], grids, tokens_of)
    else:
        grids = feats["image_grid_thw"].tolist() if images else []
        msgs = expand_image_placeholders(row["prompt"] + row["response"], grids, merge_size=getattr(proc.image_processor, "merge_size", 2))
        pixels = feats["pixel_values"] if images else None
    turns = encode_turns(proc.tokenizer, TURN_TEXTS[template](msgs, row["system"]))
    ids, labels = supervised_labels(turns, cutoff_len, train_on_prompt=train_on_prompt, mask_history=mask_history)
    if images and ids.count(image_token_id) != sum(part.count(image_token_id) for tn in turns for part in tn):
        raise ValueError("cutoff_len=%d truncates image placeholder tokens; raise --cutoff_len" % cutoff_len)
    return ids, labels, pixels, grids


def main(argv=None):
    a = build_parser().parse_args(argv)
    if a.stage != "sft" or a.finetuning_type != "full":
        raise ValueError("only --stage sft --finetuning_type full is part of the IAD-R1 PA-SFT path")
    if a.template not in TEMPLATES:
        raise ValueError(f"--template {a.template}: the IAD-R1 PA-SFT scripts use {TEMPLATES}; other llamafactory templates are not part of this path")
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

    from iadr1_amd.sft import SFTArgs, SFTEngine, frozen_parameter_rule
    from iadr1_amd.trainer import last_checkpoint, load_checkpoint, load_training_state, save_checkpoint, save_training_state

    cfg, store = load_checkpoint(a.model_name_or_path, dev, trainable=True)
    if cfg.llava_family != TEMPLATE_FAMILY[a.template]:
        raise ValueError(f"--template {a.template} does not belong to the model family of {a.model_name_or_path}")
    proc = AutoProcessor.from_pretrained(a.model_name_or_path)
    if a.train_mm_proj_only:
        raise ValueError("--train_mm_proj_only: not part of this path (no launch script uses it; the reference's rule matches parameter NAMES against 'model' / "
                         "'language_model', so what it freezes depends on the transformers version's naming)")
    with open(os.path.join(a.model_name_or_path, "config.json")) as f:
        model_type = json.load(f).get("model_type")
    frozen = frozen_parameter_rule(model_type, a.freeze_vision_tower, a.freeze_multi_modal_projector)
    if frozen is not None and rank == 0:
        n_fz = sum(1 for n in store.slots if frozen(n))
        print(f"[pa-sft] model_type {model_type}: {n_fz} of {len(store.slots)} parameter tensors frozen (vision tower: {a.freeze_vision_tower}, projector: {a.freeze_multi_modal_projector})", flush=True)
    eng = SFTEngine(cfg, store, SFTArgs(learning_rate=a.learning_rate, weight_decay=a.weight_decay, max_grad_norm=a.max_grad_norm,
                                        gradient_accumulation_steps=a.gradient_accumulation_steps, micro_batch_seqs=a.micro_batch_seqs, frozen=frozen,
                                        recompute="auto" if (a.gradient_checkpointing or group is not None) else "off"), group=group)
    from iadr1_amd import schedule
    if a.lr_scheduler_type not in schedule.SCHEDULES:
        raise ValueError(f"--lr_scheduler_type {a.lr_scheduler_type}: supported {schedule.SCHEDULES}")
    rows = load_sharegpt(a.dataset, a.dataset_dir, a.image_dir)
    bs, ga = a.per_device_train_batch_size, a.gradient_accumulation_steps
    # the step count comes from the GLOBAL row count and every rank holds ceil(n / world) rows per epoch in a seeded per-epoch order
    # (DistributedSampler semantics): all ranks run the same number of optimizer steps and issue the same collectives
    total = schedule.total_steps(len(rows), world, bs, ga, a.num_train_epochs, a.max_steps)
    sampler = schedule.RankSampler(len(rows), rank, world, seed=a.seed, shuffle=not a.disable_shuffling)
    os.makedirs(a.output_dir, exist_ok=True)
    log = open(os.path.join(a.output_dir, "trainer_log.jsonl"), "a") if rank == 0 else None
    pad = cfg.pad_token_id
    # resume (reference: llamafactory hparams/parser.py:332-354 picks the last checkpoint of an existing output_dir unless --overwrite_output_dir)
    ck = a.resume_from_checkpoint or (None if a.overwrite_output_dir else last_checkpoint(a.output_dir))
    start = 0
    if ck:
        st = load_training_state(store, ck)
        eng.opt_step, start = st["opt_step"], st["global_step"]
        print(f"resuming from {ck} at step {start}", flush=True)
    i, t0 = start * bs * ga, time.time()
    steps_per_epoch = max(1, schedule.total_steps(len(rows), world, bs, ga, 1.0, -1))
    window_loss, window_from = 0.0, start
    run_loss, history = 0.0, []
    for step in range(start, total):
        lr = schedule.lr_at(step, total, a.learning_rate, a.warmup_steps, a.lr_scheduler_type)
        eng.args.learning_rate = lr
        losses = []
        for k in range(ga):
            enc = [encode_example(proc, rows[sampler.index(i + j)], a.cutoff_len, a.train_on_prompt, a.mask_history, cfg.image_token_id, a.image_resolution, a.template, cfg)
                   for j in range(bs)]
            i += bs
            S = (max(len(e[0]) for e in enc) + 7) // 8 * 8  # pad_to_multiple_of=8 (sft/workflow.py:60), right padding
            ids = np.full((bs, S), pad, dtype=np.int64)
            mask = np.zeros((bs, S), dtype=np.int64)
            labels = np.full((bs, S), -100, dtype=np.int64)
            for r, (x, y, _, _) in enumerate(enc):
                ids[r, : len(x)], mask[r, : len(x)], labels[r, : len(y)] = x, 1, y
            grids = [tuple(g) for e in enc for g in e[3]]
            if cfg.is_llava:     # per-image crop stacks + original (height, width) of every image
                batch = {"input_ids": ids, "attention_mask": mask, "labels": labels, "pixel_values": [c for e in enc if e[2] is not None for c in e[2]], "image_sizes": grids,
                         "images_per_row": [len(e[3]) for e in enc]}
                if cfg.llava_family == "llava":
                    del batch["image_sizes"]       # LLaVA-1.5: one S x S crop per image, no sizes
            else:
                pv = torch.cat([e[2] for e in enc if e[2] is not None], 0)
                batch = {"input_ids": ids, "attention_mask": mask, "labels": labels, "pixel_values": pv, "image_grid_thw": grids, "images_per_row": [len(e[3]) for e in enc]}
            losses.append(eng.loss_and_grads(batch, last_micro_step=(k == ga - 1)))
        eng.optimizer_step()
        window_loss += float(np.mean(losses))
        run_loss += float(np.mean(losses))
        if (step + 1) % a.logging_steps == 0:
            # the record LLaMA-Factory's LogCallback writes (train/callbacks.py:279-318) from transformers.Trainer's log line: mean step loss since the last line
            # (4 places), the scheduler's NEXT learning rate, the epoch fraction, progress
            if log:
                el = time.time() - t0
                rec = {"current_steps": step + 1, "total_steps": total, "loss": round(window_loss / (step + 1 - window_from), 4),
                       "lr": schedule.lr_at(step + 1, total, a.learning_rate, a.warmup_steps, a.lr_scheduler_type), "epoch": round((step + 1) / steps_per_epoch, 2),
                       "percentage": round((step + 1) / total * 100, 2), "elapsed_time": round(el, 1), "remaining_time": round(el / (step + 1 - start) * (total - step - 1), 1)}
                log.write(json.dumps(rec) + "\n")
                log.flush()
                history.append({"epoch": rec["epoch"], "learning_rate": rec["lr"], "loss": rec["loss"], "step": step + 1})
            window_loss, window_from = 0.0, step + 1
        if rank == 0 and a.save_steps and (step + 1) % a.save_steps == 0:
            save_checkpoint(store, os.path.join(a.output_dir, f"checkpoint-{step + 1}"), json.load(open(os.path.join(a.model_name_or_path, "config.json"))))
            save_training_state(store, os.path.join(a.output_dir, f"checkpoint-{step + 1}"), eng.opt_step, step + 1)
    if rank == 0:
        save_checkpoint(store, a.output_dir, json.load(open(os.path.join(a.model_name_or_path, "config.json"))))
        proc.save_pretrained(a.output_dir)
        # what run_sft leaves beside the model (llamafactory/train/sft/workflow.py:108-110 -> transformers.Trainer.save_metrics / save_state): the run's metrics and
        # the log history (`--plot_loss` would draw training_loss.png from the latter; no plotting library here)
        done = max(1, total - start)
        runtime = time.time() - t0
        metrics = {"epoch": round(total / steps_per_epoch, 2), "train_loss": run_loss / done, "train_runtime": round(runtime, 4),
                   "train_samples_per_second": round(done * bs * ga * world / runtime, 3), "train_steps_per_second": round(done / runtime, 3)}
        for name in ("train_results.json", "all_results.json"):
            with open(os.path.join(a.output_dir, name), "w") as f:
                json.dump(metrics, f, indent=4, sort_keys=True)
        with open(os.path.join(a.output_dir, "trainer_state.json"), "w") as f:
            json.dump({"epoch": metrics["epoch"], "global_step": total, "max_steps": total, "logging_steps": a.logging_steps, "save_steps": a.save_steps,
                       "num_train_epochs": a.num_train_epochs, "train_batch_size": bs, "log_history": history + [{**metrics, "step": total}]}, f, indent=2)


if __name__ == "__main__":
    main()
