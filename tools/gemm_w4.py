#!/usr/bin/env python3
"""Probe driver: the 4-wave 128x128-wave-tile GEMM (tools/probe/gemm_w4.hip) vs the product gemm_nt on the hot shapes."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libgemm_w4.so"))
dev = "cuda"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N, K in [(1024, 512, 256), (20480, 22016, 2048), (20480, 2048, 11008), (20480, 2048, 2048), (4096, 151808, 2048)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); out2 = torch.empty_like(out)
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    run = lambda: L.run_w4(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(out2.data_ptr()), M, N, K, st())
    assert run() == 0
    ops.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    err = (out2.float() - out.float()).abs().max().item() / out.float().abs().max().item()
    t0, t1 = timeit(lambda: ops.gemm_nt(a, b, out=out)), timeit(run)
    fl = 2.0 * M * N * K
    print(f"M={M:6d} N={N:6d} K={K:6d}  gemm_nt {t0*1e3:8.1f} us {fl/t0/1e9:7.1f} TF | w4 {t1*1e3:8.1f} us {fl/t1/1e9:7.1f} TF  relerr vs gemm_nt {err:.1e}", flush=True)
    del a, b, out, out2
