#!/usr/bin/env python3
"""Extracts the command-line flag list every reference launch script passes to its entry point
(/root/reference/scripts/train/{PA_SFT,SC_GRPO}/*.sh) into tests/golden/launch_flags.json, so that the CPU tests can replay
each list through this repo's two entry-point parsers.  Run here (build container) only; the fixture is data: script name,
entry point and the argv tokens after it, with $VARIABLES replaced by placeholder values of the right kind."""
import glob
import json
import os
import re
import shlex

REF = "/root/reference/scripts/train"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "launch_flags.json")
PLACE = {"PRETRAIN_MODEL_PATH": "/models/pretrain", "MODEL_PATH": "/models/pretrain", "OUTPUT_PATH": "/out/run", "OUTPUT_DIR": "/out/run", "DATASET": "Expert_AD_Stage_1",
         "IMAGE_DIR": "/data/Expert-AD", "IMAGE_PATH": "/data/Expert-AD", "DATA_PATH": "/data/train.json", "RUN_NAME": "run"}


def flags_of(path):
    txt = open(path).read()
    txt = re.sub(r"\\\n", " ", txt)
    env = dict(PLACE)
    for m in re.finditer(r"^export\s+(\w+)=(.*)$", txt, re.M):
        val = re.sub(r"\$\([^)]*\)", "STAMP", m.group(2).strip()).strip('"')       # $(date ...) / $(pwd) -> a fixed word
        env.setdefault(m.group(1), val)
    for line in txt.splitlines():
        if "train/stage_sft/train.py" in line or "train/stage_rl/grpo_ad.py" in line:
            entry = "train/stage_sft/train.py" if "stage_sft" in line else "train/stage_rl/grpo_ad.py"
            tail = line.split(entry, 1)[1]
            tail = re.split(r"\s2>&1|\s\|\s|\s>\s", tail)[0]
            tail = re.sub(r"\$\{?(\w+)\}?", lambda m: env.get(m.group(1), "x"), tail)
            return entry, shlex.split(tail)
    raise RuntimeError(path)


def main():
    out = {"meta": {"source": "scripts/train/{PA_SFT,SC_GRPO}/*.sh of Yanhui-Lee/IAD-R1", "generator": "tools/make_golden_launch_flags.py"}, "scripts": []}
    for sh in sorted(glob.glob(os.path.join(REF, "*", "*.sh"))):
        entry, argv = flags_of(sh)
        out["scripts"].append({"script": os.path.relpath(sh, "/root/reference"), "entry": entry, "argv": argv})
    json.dump(out, open(OUT, "w"), indent=1)
    print(len(out["scripts"]), "scripts ->", OUT)


if __name__ == "__main__":
    main()
