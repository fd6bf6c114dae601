#!/usr/bin/env python3
"""Fixed (prologue + epilogue) cost of gemm_nt per output mode: time at K and 2K on the same M, N; t = fixed + slope * K."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K), mode in [((20480, 22016, 2048), "bf16"), ((20480, 2048, 2048), "bf16"), ((4096, 151936, 2048), "f32"), ((22016, 2048, 10240), "acc"), ((2048, 11008, 10240), "acc"), ((2560, 2048, 10240), "acc")]:
    ts = []
    for k in (K, 2 * K):
        a = torch.randn(M, k, device=dev).to(torch.bfloat16); b = torch.randn(N, k, device=dev).to(torch.bfloat16)
        out = torch.zeros(M, N, dtype=torch.bfloat16 if mode == "bf16" else torch.float32, device=dev)
        ts.append(timeit(lambda: ops.gemm_nt(a, b, out=out, accumulate=(mode == "acc"))))
        del a, b, out
    slope = (ts[1] - ts[0]) / K
    fixed = ts[0] - slope * K
    print(f"M={M:6d} N={N:6d} {mode:4s} K={K:6d}: {ts[0]:8.1f} us  2K: {ts[1]:8.1f} us  fixed {fixed:7.1f} us ({100*fixed/ts[0]:4.1f}% at K)  main-loop rate {2.0*M*N/slope/1e6:7.1f} TF", flush=True)
