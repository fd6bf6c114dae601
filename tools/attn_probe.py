#!/usr/bin/env python3
"""Attention kernel timings on the 3B SC-GRPO training shape in the shared-prefix layout (8 prompts x 512 + 64 completions x 256,
GQA 16/2, D=128) and, for comparison, in the repeated-rows layout (32 x 768)."""
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
Hq, Hkv, D = 16, 2, 128
def run(name, seg, T, fl):
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
    q, k, v = qkv[:, :Hq*D], qkv[:, Hq*D:(Hq+Hkv)*D], qkv[:, (Hq+Hkv)*D:]
    o = torch.zeros(T, Hq * D, dtype=torch.bfloat16, device=dev)
    t = bench(lambda: ops.attn_fwd(q, k, v, seg, Hq, Hkv, D, True, D ** -0.5, out=o))
    print(f"{name}: fwd  {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TF")
    _, lse = ops.attn_fwd(q, k, v, seg, Hq, Hkv, D, True, D ** -0.5, out=o)
    do = torch.randn_like(o); dqkv = torch.zeros_like(qkv)
    t = bench(lambda: ops.attn_bwd(q, k, v, o, do, lse, seg, Hq, Hkv, D, True, D ** -0.5, dqkv[:, :Hq*D], dqkv[:, Hq*D:(Hq+Hkv)*D], dqkv[:, (Hq+Hkv)*D:]))
    print(f"{name}: bwd  {t*1e3:8.1f} us  {2.5*fl/t/1e9:7.1f} TF", flush=True)
ng, P, G, C = 8, 512, 8, 256
n = ng * G
starts = [b * P for b in range(ng)] + [ng * P + r * C for r in range(n)]
ends = [b * P + P for b in range(ng)] + [ng * P + r * C + C for r in range(n)]
prefix = [[0, 0, ng + b * G, G] for b in range(ng)] + [[(r // G) * P, P, 0, 0] for r in range(n)]
fl = 4.0 * Hq * D * (ng * P * P / 2 + n * (C * P + C * C / 2))
run("shared-prefix 8x512 + 64x256", ops.Segments(starts, ends, dev, prefix=prefix), ng * P + n * C, fl)
B, S = 32, 768
run("repeated rows 32x768", ops.Segments([b * S for b in range(B)], [(b + 1) * S for b in range(B)], dev), B * S, 4.0 * B * Hq * S * S * D / 2)
