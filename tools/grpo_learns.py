#!/usr/bin/env python3
"""Does the SC-GRPO loop learn?  Tiny Qwen2.5-VL structure, reward = fraction of completion tokens with an id below half the vocabulary; prints the mean reward per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixture_util as fx
import iadr1_amd  # noqa
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
cfg = VLMConfig.from_dict(fx.TINY)
w = fx.make_weights(fx.TINY, 0)
pol, ref = ParamStore(cfg, "cuda", True), ParamStore(cfg, "cuda", False)
pol.load_named(w); ref.load_named(w)
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 3e-3
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=4096, max_completion_length=8, learning_rate=lr, beta=0.04, suppress_eos=True, seed=11))
grid = (1, 16, 12)
ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 3), fx.synth_prompt(grid, 9, fx.TINY, 4)], fx.TINY["pad_token_id"])
batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid, grid], fx.TINY, seed=3), "image_grid_thw": [grid, grid]}
reward = lambda comp: (np.asarray(comp) < 320).mean(1, keepdims=True).astype(np.float32)
hist = []
for step in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    m = eng.step(batch, reward)
    hist.append(m["reward"])
print(" ".join(f"{r:.2f}" for r in hist))
