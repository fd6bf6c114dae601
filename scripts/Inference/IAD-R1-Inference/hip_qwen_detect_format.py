#!/usr/bin/env python3
"""Drop-in for the reference's `scripts/Inference/IAD-R1-Inference/vLLM_Qwen_detect_format.py` (same flags, same result files:
`result/<name>/<test_dataset>/answers_<k>_shot_<model>_vllm.json` + `..._accuracy.csv`), decoding on the MI355X rollout engine
instead of a vLLM process.  Needs the checkpoint directory to hold the HF processor / tokenizer files (as the reference does)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", type=str, default="model_path")
    ap.add_argument("--few_shot_model", type=int, default=0)
    ap.add_argument("--reproduce", action="store_true")
    ap.add_argument("--similar_template", action="store_true")
    ap.add_argument("--record_history", action="store_true")
    ap.add_argument("--batch_size", type=int, default=4)
    ap.add_argument("--tensor_parallel_size", type=int, default=1, help="accepted for compatibility; one MI355X holds the model")
    ap.add_argument("--gpu_memory_utilization", type=float, default=0.9, help="accepted for compatibility")
    ap.add_argument("--step", type=int, default=500)
    ap.add_argument("--test_dataset", type=str, default="test_data")
    ap.add_argument("--name", type=str, default="Qwen")
    ap.add_argument("--data_path", type=str, default=os.environ.get("IADR1_TEST_DATA", "Industrial_test"))
    ap.add_argument("--json_path", type=str, default=None, help="default: data/Test/<test_dataset>_format.json")
    a = ap.parse_args()

    import iadr1_amd  # noqa: F401
    from transformers import AutoProcessor
    from iadr1_amd import evaluate
    from iadr1_amd.trainer import load_checkpoint

    cfg, store = load_checkpoint(a.model_path, "cuda", trainable=False, with_decode_pack=True)
    processor = AutoProcessor.from_pretrained(a.model_path)
    gen = evaluate.GreedyGenerator(cfg, store, max_new_tokens=512)
    model_name = os.path.split(a.model_path.rstrip("/"))[-1] + ("_Similar_template" if a.similar_template else "")
    out_dir = f"result/{a.name}/{a.test_dataset}/"
    os.makedirs(out_dir, exist_ok=True)
    answers_path = f"{out_dir}answers_{a.few_shot_model}_shot_{model_name}_vllm.json"
    existing = json.load(open(answers_path)) if (os.path.exists(answers_path) and not a.reproduce) else []
    chat_ad = json.load(open(a.json_path or f"data/Test/{a.test_dataset}_format.json"))
    evaluate.evaluate_dataset(gen, processor, a.data_path, chat_ad, a.few_shot_model, a.batch_size, a.similar_template, answers_path, existing)
    df, _ = evaluate.write_accuracy(answers_path)
    print(df)


if __name__ == "__main__":
    main()
