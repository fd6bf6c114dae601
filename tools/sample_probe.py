#!/usr/bin/env python3
"""Sampler timing on the decode shape (64 rows x 151936 fp32 logits), rotating logits buffers."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
torch.manual_seed(0)
lg = [torch.randn(64, 151936, device=dev) * 3 for _ in range(6)]
out = torch.zeros(64, dtype=torch.int64, device=dev)
def run(i): ops.sample(lg[i % 6], 0.9, 50, 0.9, 1234, i, out=out)
for i in range(12): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(120): run(i)
e1.record(); torch.cuda.synchronize()
print(f"sample (stage 1 + stage 2): {e0.elapsed_time(e1) / 120 * 1e3:.1f} us per call; checksum {int(out.sum())}")
