#!/usr/bin/env python3
"""Attention kernel timings on the 3B training shape (32 sequences x 768, GQA 16/2, D=128, causal)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
def bench(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B, S, Hq, Hkv, D = 32, 768, 16, 2, 128
T = B * S
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :Hq*D], qkv[:, Hq*D:(Hq+Hkv)*D], qkv[:, (Hq+Hkv)*D:]
seg = ops.Segments([b * S for b in range(B)], [(b + 1) * S for b in range(B)], dev)
o = torch.zeros(T, Hq * D, dtype=torch.bfloat16, device=dev)
fl = 4.0 * B * Hq * S * S * D / 2
t = bench(lambda: ops.attn_fwd(q, k, v, seg, Hq, Hkv, D, True, D ** -0.5, out=o))
print(f"fwd  {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TF")
_, lse = ops.attn_fwd(q, k, v, seg, Hq, Hkv, D, True, D ** -0.5, out=o)
do = torch.randn_like(o); dqkv = torch.zeros_like(qkv)
t = bench(lambda: ops.attn_bwd(q, k, v, o, do, lse, seg, Hq, Hkv, D, True, D ** -0.5, dqkv[:, :Hq*D], dqkv[:, Hq*D:(Hq+Hkv)*D], dqkv[:, (Hq+Hkv)*D:]))
print(f"bwd  {t*1e3:8.1f} us  {2.5*fl/t/1e9:7.1f} TF")
