"""Evaluation harness of the IAD-R1 checkpoints on the MI355X engine.

Host side = the reference's multiple-choice protocol, restated: question / option parsing (`GPT4/gpt4v.py:123-169`),
answer extraction (`scripts/Inference/IAD-R1-Inference/vLLM_Qwen_detect_format.py:140-165`), per-dataset accuracy table
(`helper/summary.py:8-124`); pinned bit-exactly (strings, letters, CSV text) by `tests/golden/eval.json`, which
`tools/make_golden_eval.py` produced by calling those reference functions.  Device side = greedy decoding with the rollout engine
(`rollout.Rollout`, temperature 0 -- the reference's `SamplingParams(temperature=0.0, max_tokens=512, stop_token_ids=[eos])`)
instead of a vLLM server process.
"""
from __future__ import annotations

import json
import os
import re

import numpy as np

_ANSWER = re.compile(r"<answer>(.*?)</answer>")
LETTERS = ("A", "B", "C", "D", "E")


def parse_conversation(text_gt: dict):
    """First key that starts with 'conversation' -> ([{type, text, options}], [answer letters]); options are re-lettered A, B, ...
    in their stored order (the reference's shuffle is commented out)."""
    questions, answers = [], []
    for key, conv in text_gt.items():
        if not key.startswith("conversation"):
            continue
        for i, qa in enumerate(conv):
            items = list(qa["Options"].items())
            relettered = {chr(65 + j): val for j, (_, val) in enumerate(items)}
            lines = "".join(f"{chr(65 + j)}. {val}\n" for j, (_, val) in enumerate(items))
            new_key = [chr(65 + j) for j, (orig, _) in enumerate(items) if orig == qa["Answer"]]
            if not new_key:
                raise ValueError("Answer key not found after shuffling options.")
            questions.append({"type": "text", "text": f"Question {i + 1}: {qa['Question']} \n{lines}", "options": relettered})
            answers.append(new_key[-1])
        break
    return questions, answers


def get_ans(response_text, options=None):
    """Letter of the option named inside the first <answer>...</answer> (single line), 'E' when there is none / no match;
    the lower-cased answer string itself when `options` is None."""
    if not isinstance(response_text, str):
        return "E"
    m = _ANSWER.search(response_text)
    if m is None:
        return "E"
    ans = m.group(1).strip().lower()
    if options is None:
        return ans
    for key, val in options.items():
        if ans == val.lower().strip("."):
            return key
    for key, val in options.items():
        clean = val.lower().strip(".").strip()
        if ans in clean or clean in ans:
            return key
    return "E"


FAMILIES = ("qwen", "llava", "llava_1_5")


def build_messages(n_few_shot: int = 0, family: str = "qwen") -> list:
    """The reference's fixed detection prompt: optional normal templates, the test image, one question.  Three scripts, three wordings (pinned by
    tests/golden/eval.json `prompts`, captured from the reference's own `build_prompt` methods):
      qwen       vLLM_Qwen_detect_format.py:88-128       "Following is image of test sample:" only with templates; "... in the test image?"
      llava      vLLM_LLaVA_detect_format.py:90-127      (LLaVA-OneVision, LLaVA-1.6) the same structure; "... in the query image?"
      llava_1_5  vLLM_LLaVA_1_5_detect_format.py:90-126  "Following is image of test sample:" ALWAYS (also zero-shot); "... in the test image?" """
    if family not in FAMILIES:
        raise ValueError(f"family must be one of {FAMILIES}")
    parts = []
    if n_few_shot:
        parts.append({"type": "text", "text": f"Following is {n_few_shot} image of normal sample, which can be used as a template to compare the image being queried."})
        parts += [{"type": "image"}] * n_few_shot
    if n_few_shot or family == "llava_1_5":
        parts.append({"type": "text", "text": "Following is image of test sample:"})
    parts.append({"type": "image"})
    parts.append({"type": "text", "text": "Are there any defects in the query image?" if family == "llava" else "Are there any defects in the test image?"})
    return [{"role": "user", "content": parts}]


def template_kwargs(family: str = "qwen") -> dict:
    """Keyword arguments of the reference's apply_chat_template call: the Qwen script renders through the TOKENIZER with tokenize=False
    (vLLM_Qwen_detect_format.py:122-126), the LLaVA scripts through the PROCESSOR with its defaults (vLLM_LLaVA_detect_format.py:122-125)."""
    return {"tokenize": False, "add_generation_prompt": True} if family == "qwen" else {"add_generation_prompt": True}


def accuracy_table(all_answers: list, normal_flag: str = "good", show_overkill_miss: bool = False):
    """-> (pandas.DataFrame accuracy table [%], question_stats) exactly as the reference computes and saves it: 'Object Structure' /
    'Object Details' fold into 'Object Analysis'; 'Anomaly Detection' is the mean of normal and abnormal accuracy; an entry whose
    letters are not A-E is dropped -- and, like the reference (which removes from the list it is iterating), so is the entry after it."""
    import pandas as pd

    fold = lambda t: "Object Analysis" if t in ("Object Structure", "Object Details") else t
    entries = list(all_answers)
    datasets, types = [], []
    for a in entries:
        ds = a["image"].split("/")[0]
        if ds not in datasets:
            datasets.append(ds)
        if fold(a["question_type"]) not in types:
            types.append(fold(a["question_type"]))
    blank = lambda: {"total": 0, "correct": 0, "correct_answers": {}, "answers": {}}
    qstats = {ds: {t: blank() for t in types} for ds in datasets}
    det = {ds: {"normal": blank(), "abnormal": blank()} for ds in datasets}
    i = 0
    while i < len(entries):          # index walk over a list that shrinks underneath it
        a = entries[i]
        i += 1
        ds, qt = a["image"].split("/")[0], fold(a["question_type"])
        ga, ca = a["gpt_answer"], a["correct_answer"]
        if ca not in LETTERS or ga not in LETTERS:
            entries.remove(a)
            continue
        st = qstats[ds][qt]
        st["total"] += 1
        st["correct"] += int(ca == ga)
        if qt == "Anomaly Detection":
            side = det[ds]["normal" if normal_flag in a["image"] else "abnormal"]
            side["total"] += 1
            side["correct"] += int(ca == ga)
        st["answers"][ga] = st["answers"].get(ga, 0) + 1
        st["correct_answers"][ca] = st["correct_answers"].get(ca, 0) + 1
    ratio = lambda d: d["correct"] / d["total"] if d["total"] != 0 else 0
    df = pd.DataFrame(index=datasets)
    for ds in datasets:
        for t in types:
            df.at[ds, t] = ratio(qstats[ds][t]) * 100
            if t == "Anomaly Detection":
                df.at[ds, t] = (ratio(det[ds]["normal"]) + ratio(det[ds]["abnormal"])) / 2 * 100
    df["Average"] = df.mean(axis=1)
    if show_overkill_miss:
        for ds in datasets:
            df.at[ds, "Overkill"] = (1 - ratio(det[ds]["normal"])) * 100
            df.at[ds, "Miss"] = (1 - ratio(det[ds]["abnormal"])) * 100
    df.loc["Average"] = df.mean()
    return df, qstats


def write_accuracy(answers_json_path: str, normal_flag: str = "good", show_overkill_miss: bool = False):
    """answers_*.json -> answers_*_accuracy.csv next to it (same file name rule as the reference)."""
    with open(answers_json_path) as f:
        df, stats = accuracy_table(json.load(f), normal_flag, show_overkill_miss)
    df.to_csv(answers_json_path.replace(".json", "_accuracy.csv"))
    return df, stats


# ---------------------------------------------------------------------------------------------------------------------------
# device side
# ---------------------------------------------------------------------------------------------------------------------------
class GreedyGenerator:
    """Batched greedy decoding of a checkpoint with the rollout engine (what vLLM does in the reference's eval scripts)."""

    def __init__(self, cfg, store, max_new_tokens: int = 512):
        from .vlm import Engine
        if not getattr(store, "with_decode_pack", False):
            raise ValueError("GreedyGenerator needs a ParamStore built with with_decode_pack=True (trainer.load_checkpoint(..., with_decode_pack=True))")
        self.cfg, self.engine, self.max_new = cfg, Engine(store), max_new_tokens
        self._rollout = None

    def generate(self, batch: dict) -> np.ndarray:
        """batch: input_ids / attention_mask [B, P] (left padded), pixel_values, image_grid_thw, images_per_prompt.
        Returns completion ids [B, <= max_new] (pad after EOS)."""
        from .rollout import Rollout
        e, c = self.engine, self.cfg
        ids, mask = np.asarray(batch["input_ids"]), np.asarray(batch["attention_mask"])
        # Qwen-VL: image_grid_thw + patch rows; LLaVA families: image_sizes (+ crops) -- Engine.vision_inputs reads either (the trainer's own path)
        grids, plan_v, px, rows = e.vision_inputs(batch)
        img, _ = e.vision_forward(px, plan_v, save=False)
        gpr, off, k = [], [], 0
        for n in batch.get("images_per_prompt") or [1] * len(ids):
            gpr.append(grids[k: k + n])
            off.append([int(rows[j]) for j in range(k, k + n)])
            k += n
        plan = e.text_plan(ids, mask, gpr, off)
        B, P = ids.shape
        r = self._rollout
        if r is None or r.N != B or r.max_prompt < P:        # grow only: pool, block table and captured graph are sized for a prompt length
            grow = max(P, r.max_prompt if r is not None else 0)
            self._rollout = r = None
            self._rollout = r = Rollout(e, B, grow, self.max_new, max_prompts=B, use_graph=True)
        return r.generate(plan, img, 1, self.max_new, temperature=0.0, top_k=1, top_p=1.0, seed=0).cpu().numpy()


def encode_prompts(processor, prompts, images, family: str):
    """Prompt TEXT -> ids the way the reference's evaluation scripts get them: they hand the rendered prompt string to vLLM's `llm.generate`
    (vLLM_LLaVA_detect_format.py:330-340, vLLM_LLaVA_1_5_detect_format.py, vLLM_Qwen_detect_format.py:340-352), whose tokenizer call is `encode(prompt)` with
    the tokenizer's DEFAULT `add_special_tokens=True`.  For the Llama / Vicuna / Mistral tokenizers of LLaVA-1.5 / 1.6 that prepends `<s>` (the llava-hf chat
    templates emit no BOS themselves); the Qwen2 tokenizers of Qwen2(.5)-VL and LLaVA-OneVision define no BOS, so the flag changes nothing there.
    (The TRAINING path is different on purpose: REF sc_grpo_trainer.py:615 passes `add_special_tokens=False`; trainer.prepare_batch keeps that.)
    A chat template that renders `<s>` ITSELF would get a second BOS here -- and from vLLM's `encode(prompt)` in the reference just the same (vLLM only warns about
    it): the ids stay what the reference's engine would see, so nothing is stripped."""
    return processor(text=prompts, images=images, return_tensors="pt", padding=True, padding_side="left", add_special_tokens=(family != "qwen"))


def evaluate_dataset(generator: GreedyGenerator, processor, data_root: str, chat_ad: dict, few_shot_model: int = 0, batch_size: int = 4,
                     similar_template: bool = False, answers_json_path: str | None = None, existing: list | None = None, family: str = "qwen") -> list:
    """The reference's evaluation loop (vLLM_Qwen_detect_format.py:283-380, vLLM_LLaVA_detect_format.py:300-367, vLLM_LLaVA_1_5_detect_format.py): first
    question of every image, greedy answer, letter by `get_ans`, one entry per question; writes the answers json after every batch when a path is given.
    family: which script's prompt wording / template call / image mode (the LLaVA scripts convert every image to RGB, :85-87)."""
    from PIL import Image
    opener = (lambda p: Image.open(p)) if family == "qwen" else (lambda p: Image.open(p).convert("RGB"))
    render = processor.tokenizer if (family == "qwen" and hasattr(processor, "tokenizer") and getattr(processor.tokenizer, "chat_template", None)) else processor
    all_answers = list(existing or [])
    done = {a["image"] for a in all_answers}
    todo = [k for k in chat_ad if k not in done]
    for i in range(0, len(todo), batch_size):
        keys, prompts, images, per, metas = todo[i: i + batch_size], [], [], [], []
        for key in keys:
            text_gt = chat_ad[key]
            qs, ans = parse_conversation(text_gt)
            if not qs or not ans:
                continue
            shots = (text_gt["similar_templates"] if similar_template else text_gt["random_templates"])[:few_shot_model] if few_shot_model else []
            prompts.append(render.apply_chat_template(build_messages(len(shots), family), **template_kwargs(family)))
            ims = [opener(os.path.join(data_root, p)) for p in shots] + [opener(os.path.join(data_root, key))]
            images += ims
            per.append(len(ims))
            metas.append((key, qs[0:1], ans[0:1], text_gt))
        if not prompts:
            continue
        enc = encode_prompts(processor, prompts, images, family)
        batch = {"input_ids": enc["input_ids"].numpy(), "attention_mask": enc["attention_mask"].numpy(), "pixel_values": enc["pixel_values"], "images_per_prompt": per}
        if "image_sizes" in enc:            # LLaVA-OneVision / LLaVA-NeXT processors (trainer.prepare_batch reads the same keys)
            batch["image_sizes"] = np.asarray(enc["image_sizes"]).reshape(-1, 2).tolist()
        elif "image_grid_thw" in enc:       # Qwen2-VL / Qwen2.5-VL processors
            batch["image_grid_thw"] = np.asarray(enc["image_grid_thw"]).tolist()
        comp = generator.generate(batch)
        texts = processor.batch_decode(comp, skip_special_tokens=True)
        for (key, qs, ans, text_gt), response in zip(metas, texts):
            letter = get_ans(response, qs[0]["options"]) or response
            qtypes = [cv["type"] for cv in text_gt["conversation"]]
            for q, a, ga, qt in zip(qs, ans, [letter], qtypes):
                all_answers.append({"image": key, "question": q, "question_type": qt, "correct_answer": a, "gpt_answer": ga})
        if answers_json_path:
            with open(answers_json_path, "w") as f:
                json.dump(all_answers, f, indent=4)
    return all_answers


def detect_main(family: str, argv=None):
    """Shared body of scripts/Inference/IAD-R1-Inference/hip_{qwen,llava,llava_1_5}_detect_format.py: the reference script's flags (defaults per script:
    vLLM_Qwen_detect_format.py:254-266 -- 0-shot, name "Qwen"; vLLM_LLaVA*_detect_format.py:240-253 -- 1-shot, name "LlaVA", `--temperature` on the
    OneVision / 1.6 script only) and result files (`result/<name>/<test_dataset>/answers_<k>_shot_<model>_vllm.json` + `..._accuracy.csv`); decoding on the
    rollout engine instead of a vLLM process.  The checkpoint directory must hold the HF processor / tokenizer files, as for the reference."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", type=str, default="model_path")
    ap.add_argument("--few_shot_model", type=int, default=0 if family == "qwen" else 1)
    ap.add_argument("--reproduce", action="store_true")
    ap.add_argument("--similar_template", action="store_true")
    ap.add_argument("--record_history", action="store_true")
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--tensor_parallel_size", type=int, default=1, help="accepted for compatibility; one MI355X holds the model")
    ap.add_argument("--gpu_memory_utilization", type=float, default=0.9, help="accepted for compatibility")
    ap.add_argument("--step", type=int, default=500)
    ap.add_argument("--test_dataset", type=str, default="test_data")
    ap.add_argument("--name", type=str, default="Qwen" if family == "qwen" else "LlaVA")
    if family == "llava":
        ap.add_argument("--temperature", type=float, default=0.0)
    ap.add_argument("--data_path", type=str, default=os.environ.get("IADR1_TEST_DATA", "Industrial_test"))
    ap.add_argument("--json_path", type=str, default=None, help="default: data/Test/<test_dataset>_format.json")
    a = ap.parse_args(argv)
    if getattr(a, "temperature", 0.0) != 0.0:
        raise SystemExit("--temperature: only 0 (greedy -- what every reference Inference.sh runs) is built; vLLM's temperature > 0 with top_p = 1.0 samples the whole "
                         "vocabulary, the rollout's sampler keeps at most 64 candidates (top-k 50 is the training setting)")
    from transformers import AutoProcessor
    from .trainer import load_checkpoint
    cfg, store = load_checkpoint(a.model_path, "cuda", trainable=False, with_decode_pack=True)
    if (family == "qwen") == bool(cfg.is_llava) or (family == "llava_1_5") != (cfg.is_llava and cfg.llava_family == "llava"):
        raise SystemExit(f"{a.model_path}: this script evaluates the {family} checkpoints (config.json says otherwise)")
    processor = AutoProcessor.from_pretrained(a.model_path)
    if cfg.is_llava and hasattr(processor, "tokenizer"):
        processor.tokenizer.padding_side = "left"
    gen = GreedyGenerator(cfg, store, max_new_tokens=512)
    model_name = os.path.split(a.model_path.rstrip("/"))[-1] + ("_Similar_template" if a.similar_template else "")
    out_dir = f"result/{a.name}/{a.test_dataset}/"
    os.makedirs(out_dir, exist_ok=True)
    answers_path = f"{out_dir}answers_{a.few_shot_model}_shot_{model_name}_vllm.json"
    existing = json.load(open(answers_path)) if (os.path.exists(answers_path) and not a.reproduce) else []
    chat_ad = json.load(open(a.json_path or f"data/Test/{a.test_dataset}_format.json"))
    evaluate_dataset(gen, processor, a.data_path, chat_ad, a.few_shot_model, a.batch_size, a.similar_template, answers_path, existing, family=family)
    df, _ = write_accuracy(answers_path)
    print(df)
    return df
