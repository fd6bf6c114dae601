#!/usr/bin/env python3
"""Micro-benchmark of the decode-step kernels on 3B shapes (M=64)."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
def bench(fn, iters=100, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 64
for name, N, K, ks in [("qkv", 2560, 2048, 1), ("o", 2048, 2048, 2), ("o_ks1", 2048, 2048, 1), ("gate_up", 22016, 2048, 1), ("down", 2048, 11008, 4), ("down_ks8", 2048, 11008, 8), ("down_ks16", 2048, 11008, 16), ("down_ks32", 2048, 11008, 32), ("lm_head", 151936, 2048, 1)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    if ks > 1:
        out = torch.empty(ks, M, N, dtype=torch.float32, device=dev)
    else:
        out = torch.empty(M, N, dtype=torch.float32 if name == "lm_head" else torch.bfloat16, device=dev)
    wp = ops.pack_weight(w)
    us = bench(lambda: ops.gemm_skinny(x, wp, N, out=out, ksplit=ks))
    print(f"{name:10s} N={N:6d} K={K:5d} ks={ks} {us:8.1f} us  {N*K*2/us/1e6:7.2f} TB/s", flush=True)
H = 2048
x = torch.randn(M, H, device=dev).to(torch.bfloat16); w = torch.ones(H, device=dev, dtype=torch.bfloat16); res = x.clone(); y = torch.empty_like(x)
print("rmsnorm bf16 in", bench(lambda: ops.rmsnorm_fwd(x, w, 1e-6, res=res, res_out=res, out=y)))
p4 = torch.randn(4, M, H, device=dev)
print("rmsnorm 4 slabs", bench(lambda: ops.rmsnorm_fwd(None, w, 1e-6, res=res, res_out=res, x32=p4, out=y)))
gu = torch.randn(M, 22016, device=dev).to(torch.bfloat16); a = torch.empty(M, 11008, dtype=torch.bfloat16, device=dev)
print("swiglu", bench(lambda: ops.swiglu_fwd(gu, out=a)))
lg = torch.randn(M, 151936, device=dev)
print("sample", bench(lambda: ops.sample(lg, 0.9, 50, 0.9, 1, 0)))
