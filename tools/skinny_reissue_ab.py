#!/usr/bin/env python3
"""Round 6 probe (not product): the persistent decode GEMMs of a 3B layer on rotating weights (really from HBM), us per launch -- run once per library build
(IADR1_HIP_LIB: tools/build_variant.py reissue -DIADR1_PERS_REISSUE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import iadr1_amd  # noqa
from iadr1_amd import ops
dev = "cuda"
torch.cuda.set_stream(torch.cuda.Stream())
tag = os.path.basename(os.environ.get("IADR1_HIP_LIB", "product"))
def timeit(fn, n, reps=5):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3
NL = 24
for name, N, K, kw in [("gate|up (swiglu)", 22016, 2048, dict(swiglu=True)), ("down (8 K slices)", 2048, 11008, dict(ksplit=8)), ("lm_head", 151936, 2048, dict(out_dtype=torch.float32))]:
    nl = NL if N * K * 2 < 200e6 else 6
    Wp = []
    for _ in range(nl):
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        Wp.append(ops.pack_gateup(w) if kw.get("swiglu") else ops.pack_weight(w))
        del w
    X = ops.pack_act(torch.randn(64, K, device=dev).to(torch.bfloat16))
    if kw.get("swiglu"): out = ops.PackedAct(64, N // 2, dev)
    elif kw.get("ksplit", 1) > 1: out = torch.empty(kw["ksplit"], 64, N, dtype=torch.float32, device=dev)
    else: out = torch.empty(64, N, dtype=kw.get("out_dtype", torch.bfloat16), device=dev)
    kk = {k: v for k, v in kw.items() if k != "out_dtype"}
    t = timeit(lambda i: ops.gemm_skinny(X, Wp[i % nl], N, out=out, **kk), nl)
    mb = N * K * 2 / 1e6
    print(f"[{tag}] {name:18s} {mb:6.1f} MB: {t:6.2f} us ({mb / t:5.2f} TB/s)", flush=True)
    del Wp
