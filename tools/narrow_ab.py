#!/usr/bin/env python3
"""The narrow decode projections (q|k|v-like and o) on weights that really come from HBM (rotating buffers), 64 rows, packed X: us per launch, back-to-back
(HIP events over 3 x NL launches; includes ~1 us of launch gap).  Round 4 measured the 32-column / 8-wave blocks against the shipped 16-column / 16-wave ones with
it: slower on every shape (profiles/EXPERIMENTS.md); the 7B q|k|v projection (288 tiles on 256 CUs) pays 8 us for its second round of blocks."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
def timeit(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(3):
        for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3
NL = 24
for name, N, K, ks in (("3b qkv-like", 2560, 2048, 1), ("3b o ks2", 2048, 2048, 2), ("3b o ks1", 2048, 2048, 1), ("7b qkv-like", 4608, 3584, 1), ("7b o ks1", 3584, 3584, 1), ("7b o ks2", 3584, 3584, 2)):
    x = ops.pack_act(torch.randn(64, K, device=dev).to(torch.bfloat16))
    w = [ops.pack_weight(torch.randn(N, K, device=dev).to(torch.bfloat16)) for _ in range(NL)]
    if ks == 1:
        y = torch.empty(64, N, dtype=torch.bfloat16, device=dev)
        us = timeit(lambda i: ops.gemm_skinny(x, w[i % NL], N, out=y), NL)
    else:
        part = torch.empty(ks, 64, N, dtype=torch.float32, device=dev)
        us = timeit(lambda i: ops.gemm_skinny(x, w[i % NL], N, out=part, ksplit=ks), NL)
    print(f"{name:12s} N {N} K {K} ks {ks}: {us:6.2f} us  {N * K * 2 / us / 1e6:5.2f} TB/s", flush=True)
    del w
