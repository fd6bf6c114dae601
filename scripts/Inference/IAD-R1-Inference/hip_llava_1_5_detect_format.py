#!/usr/bin/env python3
"""Drop-in for the reference's `scripts/Inference/IAD-R1-Inference/vLLM_LLaVA_1_5_detect_format.py` (same flags, same result files:
`result/<name>/<test_dataset>/answers_<k>_shot_<model>_vllm.json` + `..._accuracy.csv`), decoding on the MI355X rollout engine
instead of a vLLM process.  LLaVA-1.5: the reference's vLLM_LLaVA_1_5_Inference.sh."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))


def main():
    import iadr1_amd  # noqa: F401
    from iadr1_amd import evaluate
    return evaluate.detect_main("llava_1_5")


if __name__ == "__main__":
    main()
