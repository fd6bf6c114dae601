#!/usr/bin/env python3
"""Weight-gradient GEMM forms on the step's shapes: iadr1_gemm_tn_acc_bf16 (row-major dY / X, transpose reads out of LDS) against gemm_nt on transposed copies,
with and without the two transposes in the timed region.  python tools/gemm_tn_probe.py [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import iadr1_amd  # noqa
from iadr1_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
dev = torch.device("cuda", 0)
shapes = [("dW_gate|up", 22016, 2048), ("dW_down", 2048, 11008), ("dW_q|k|v", 2560, 2048), ("dW_o", 2048, 2048)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, N, K in shapes:
    dy = torch.randn(T, N, device=dev).to(torch.bfloat16)
    x = torch.randn(T, K, device=dev).to(torch.bfloat16)
    out = torch.zeros(N, K, device=dev)
    dyT, xT = ops.transpose(dy, pad_rows_to=8), ops.transpose(x, pad_rows_to=8)
    fl = 2.0 * N * K * T
    t_tn = timed(lambda: ops.gemm_tn_acc(dy, x, out))
    t_nt = timed(lambda: ops.gemm_nt(dyT, xT, out=out, accumulate=True))
    t_tr = timed(lambda: (ops.transpose(dy, pad_rows_to=8), ops.transpose(x, pad_rows_to=8)))
    print(f"{name:<12} [{N} x {K}] over T = {T}: TN {t_tn*1e3:8.1f} us ({fl/t_tn/1e9:7.1f} TF/s) | NT on transposed copies {t_nt*1e3:8.1f} us ({fl/t_nt/1e9:7.1f} TF/s) "
          f"+ the two transposes {t_tr*1e3:7.1f} us = {1e3*(t_nt+t_tr):8.1f} us", flush=True)
