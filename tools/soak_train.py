#!/usr/bin/env python3
"""Soak run of `SCGRPOTrainer.train()` on the tiny fixture model: ragged prompts (three image sizes, 3-40 text tokens), EOS live (the fixture's EOS id is sampled
often), gradient accumulation 2 with the batched rollout, the prefetch worker on, N optimizer steps.  Prints steps/s, the rollout counters and the allocator's
reserved memory at the start and the end (a leak or a per-step re-allocation shows as growth).  Usage: python tools/soak_train.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixture_util as fx
import iadr1_amd  # noqa
from iadr1_amd import rewards, rollout as ro
from iadr1_amd.params import VLMConfig
from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = VLMConfig.from_dict(fx.TINY)
grids = [(1, 16, 12), (1, 8, 8), (1, 12, 12)]
rs = np.random.RandomState(3)
n_rows = 64
parts = []
for i in range(n_rows):
    gr = grids[i % 3]
    ids, mask = fx.left_pad([fx.synth_prompt(gr, int(rs.randint(3, 41)), fx.TINY, 500 + i)], fx.TINY["pad_token_id"])
    parts.append({"input_ids": ids, "attention_mask": mask, "pixel_values": torch.from_numpy(fx.synth_pixel_values([gr], fx.TINY, seed=500 + i)), "image_grid_thw": [gr]})


class Proc:
    def apply_chat_template(self, conv, add_generation_prompt=True, tokenize=False):
        return "P"

    def __call__(self, text=None, images=None, **kw):
        return parts[images[0][1]]

    def batch_decode(self, ids, skip_special_tokens=True):
        return ["<think>a</think><location>top left</location><type>scratch</type><answer>%s</answer>" % ("yes" if int(r[0]) % 2 else "no") for r in np.asarray(ids)]


chat = [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "?"}]}]
rows = [{"prompt": chat, "image": [("synthetic", k)], "solution": "<think>gt</think><location>top left</location><type>scratch</type><answer>yes</answer>"} for k in range(n_rows)]
tr = SCGRPOTrainer((cfg, fx.make_weights(fx.TINY, 0)), [rewards.accuracy_reward, rewards.consistency_reward], processing_class=Proc(), train_dataset=rows,
                   args=GRPOConfig(output_dir="/tmp/iadr1_soak", num_generations=4, max_completion_length=48, max_prompt_length=4096, per_device_train_batch_size=1,
                                   gradient_accumulation_steps=2, max_steps=steps, logging_steps=max(1, steps // 10), save_steps=0, learning_rate=1e-5, temperature=1.0))
torch.cuda.synchronize()
r0 = torch.cuda.memory_reserved()
t0 = time.time()
hist = tr.train()
torch.cuda.synchronize()
dt = time.time() - t0
print(f"soak: {steps} optimizer steps in {dt:.1f} s ({steps / dt:.1f} steps/s); reserved {r0 / 2**20:.0f} -> {torch.cuda.memory_reserved() / 2**20:.0f} MiB; "
      f"rollout counters {ro.STATS}; last log {hist[-1] if hist else None}", flush=True)
assert all(np.isfinite(h["loss"]) for h in hist)
