#!/usr/bin/env python3
"""Drop-in for the reference's `scripts/Inference/IAD-R1-Inference/vLLM_LLaVA_detect_format.py` (same flags, same result files:
`result/<name>/<test_dataset>/answers_<k>_shot_<model>_vllm.json` + `..._accuracy.csv`), decoding on the MI355X rollout engine
instead of a vLLM process.  LLaVA-OneVision (0.5B / 7B) and LLaVA-1.6: the reference's vLLM_LLaVA_OneVision_SI_*_Inference.sh and vLLM_LLaVA_1_6_Inference.sh."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..")))


def main():
    import iadr1_amd  # noqa: F401
    from iadr1_amd import evaluate
    return evaluate.detect_main("llava")


if __name__ == "__main__":
    main()
