#!/usr/bin/env python3
"""gemm_nt on given [M, N, K] shapes under the tile selected by IADR1_GEMM_TILE (0 = the launcher's rule, 128, 256): python tools/gemm_tile_probe.py M,N,K[,acc] ...
(one process per setting: the launcher reads the switch once)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for spec in sys.argv[1:]:
    parts = spec.split(",")
    M, N, K = (int(v) for v in parts[:3])
    acc = len(parts) > 3
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.float32 if acc else torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm_nt(a, b, out=out, accumulate=acc))
    print(f"tile={os.environ.get('IADR1_GEMM_TILE', '0'):>3s} M={M:6d} N={N:6d} K={K:6d} {'acc' if acc else 'bf16'} {t*1e3:8.1f} us {2.0*M*N*K/t/1e9:7.1f} TF", flush=True)
    del a, b, out
