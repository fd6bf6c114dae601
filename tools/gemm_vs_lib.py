#!/usr/bin/env python3
"""gemm_nt (hand-written) vs the vendor library GEMM behind torch.matmul (hipBLASLt / rocBLAS) on the hot shapes of the SC-GRPO step.
Random data, 20 calls each after warm-up; the library is only a yardstick here, the product never calls it."""
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
    return e0.elapsed_time(e1) / n
for M, N, K in [(20480, 22016, 2048), (20480, 2048, 11008), (20480, 2048, 22016), (20480, 2048, 2048), (20480, 2560, 2048), (22016, 2048, 20480), (4096, 151936, 2048)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t0 = timeit(lambda: ops.gemm_nt(a, b, out=out))
    t1 = timeit(lambda: torch.matmul(a, b.t(), out=out))
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:6d} K={K:6d}  gemm_nt {t0*1e3:8.1f} us {fl/t0/1e9:7.1f} TF | library {t1*1e3:8.1f} us {fl/t1/1e9:7.1f} TF", flush=True)
    del a, b, out
