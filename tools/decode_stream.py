#!/usr/bin/env python3
"""Probe driver (NOT product): M=64 decode weight stream with weights that really come from HBM (rotating buffers)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libdecode_stream.so"))
dev = "cuda"
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(3):
        for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3
wide_names = {0: "wide NB8 W8 d2 Wonly", 1: "wide NB8 W8 d2 W+X", 2: "wide NB8 W8 d2 +mfma", 3: "wide NB8 W8 d2 full", 4: "wide NB4 W8 d2 Wonly", 5: "wide NB4 W8 d2 full",
              6: "wide NB4 W8 d4 Wonly", 7: "wide NB4 W8 d4 full", 8: "wide NB2 W8 d4 Wonly", 9: "wide NB2 W8 d4 full", 10: "wide NB4 W4 d4 Wonly", 11: "wide NB4 W4 d4 full",
              20: "pkx NB8 W8 d2 W+X", 21: "pkx NB8 W8 d2 +mfma", 22: "pkx NB8 W8 d2 full", 23: "pkx NB4 W8 d2 W+X", 24: "pkx NB4 W8 d2 +mfma", 25: "pkx NB4 W8 d2 full",
              26: "pkx NB4 W8 d1 full", 27: "pkx NB4 W8 d4 full", 28: "pkx NB2 W8 d2 full", 29: "pkx NB2 W8 d4 full", 30: "pkx NB4 W4 d2 full", 31: "pkx NB4 W4 d4 full", 32: "pkx NB8 W4 d2 full", 33: "pkx NB8 W8 d1 full",
              40: "pkx pers W8 T2 g256", 41: "pkx pers W8 T1 g256", 42: "pkx pers W8 T2 g512", 43: "pkx pers W16 T1 g256", 44: "pkx pers W16 T2 g256", 45: "pkx pers W8 T4 g256", 46: "pkx pers W8 T1 g512", 47: "pkx pers W16 T1 g512", 48: "pkx pers W4 T2 g256", 49: "pkx pers W4 T4 g256", 50: "pkx pers W4 T1 g256",
              12: "ldsx W4 T1 KC256 d4", 13: "ldsx W4 T1 KC256 d8", 14: "ldsx W8 T1 KC256 d8", 15: "ldsx W4 T2 KC256 d4", 16: "ldsx W2 T2 KC256 d4", 17: "ldsx W4 T1 KC512 d8", 18: "ldsx W2 T1 KC256 d8"}
only = [int(a) for a in sys.argv[1:]]
for N, K, NL in [(22016, 2048, 24)]:
    Ws = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(NL)]
    Wp = [ops.pack_weight(w) for w in Ws]
    X = torch.randn(64, K, device=dev).to(torch.bfloat16)
    out = torch.zeros(64, N, device=dev)
    # X in MFMA B-fragment order: Xp[k/32][row group i][lane = lm + 16 lq][8] = X[i*16 + lm][k32*32 + lq*8 ..]
    Xpk = X.view(4, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous()
    flag = torch.zeros(4096, dtype=torch.int32, device=dev)
    nbytes = N * K * 2
    print(f"== N={N} K={K}: {nbytes/1e6:.1f} MB per call, {NL} rotating buffers")
    if not only:
        for v, nm in enumerate(["pure U4 nt g2048", "pure U8 nt g2048", "pure U8 plain g2048", "pure U8 nt g1024", "pure U16 nt g512"]):
            us = timeit(lambda i: L.run_pure(v, ctypes.c_void_p(Wp[i % NL].data_ptr()), ctypes.c_longlong(nbytes), ctypes.c_void_p(flag.data_ptr()), st()), NL)
            print(f"  {nm:24s} {us:8.1f} us {nbytes/us/1e6:6.2f} TB/s", flush=True)
        obf = torch.empty(64, N, dtype=torch.bfloat16, device=dev)
        us = timeit(lambda i: ops.gemm_skinny(X, Wp[i % NL], N, out=obf), NL)
        print(f"  {'product gemm_skinny':24s} {us:8.1f} us {nbytes/us/1e6:6.2f} TB/s", flush=True)
    ref = X.float() @ Ws[0].float().t()
    for v, nm in wide_names.items():
        if only and v not in only: continue
        cols = {0: 128, 1: 128, 2: 128, 3: 128, 4: 64, 5: 64, 6: 64, 7: 64, 8: 32, 9: 32, 10: 64, 11: 64, 40: 32, 41: 16, 42: 32, 43: 16, 44: 32, 45: 64, 46: 16, 47: 16, 48: 32, 49: 64, 50: 16, 20: 128, 21: 128, 22: 128, 23: 64, 24: 64, 25: 64, 26: 64, 27: 64, 28: 32, 29: 32, 30: 64, 31: 64, 32: 128, 33: 128, 12: 64, 13: 64, 14: 128, 15: 128, 16: 64, 17: 64, 18: 32}[v]
        if N % cols: continue
        out.zero_()
        Xa = Xpk if nm.startswith("pkx") else X
        L.run_wide(v, ctypes.c_void_p(Wp[0].data_ptr()), ctypes.c_void_p(Xa.data_ptr()), ctypes.c_void_p(out.data_ptr()), N, K, st())
        torch.cuda.synchronize()
        err = ""
        if "full" in nm or "ldsx" in nm or "pers" in nm:
            err = f"relerr {((out - ref).abs().max() / ref.abs().max()).item():.1e}"
        us = timeit(lambda i: L.run_wide(v, ctypes.c_void_p(Wp[i % NL].data_ptr()), ctypes.c_void_p(Xa.data_ptr()), ctypes.c_void_p(out.data_ptr()), N, K, st()), NL)
        print(f"  {v:2d} {nm:22s} {us:8.1f} us {nbytes/us/1e6:6.2f} TB/s  blocks {N//cols:5d} {err}", flush=True)

# ---- MALL experiment: does a plain read of the weights just before the GEMM (a prefetch into the 256 MB memory-side cache)
# make the GEMM itself faster?  t(prefetch + gemm) - t(prefetch) vs t(gemm)
if not only or 99 in only:
    N, K, NL = 22016, 2048, 24
    Ws = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(NL)]
    Wp = [ops.pack_weight(w) for w in Ws]
    X = torch.randn(64, K, device=dev).to(torch.bfloat16)
    Xpk = X.view(4, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous()
    out = torch.zeros(64, N, device=dev)
    flag = torch.zeros(4096, dtype=torch.int32, device=dev)
    nbytes = N * K * 2
    pre = lambda i, v=2: L.run_pure(v, ctypes.c_void_p(Wp[i % NL].data_ptr()), ctypes.c_longlong(nbytes), ctypes.c_void_p(flag.data_ptr()), st())
    gem = lambda i: L.run_wide(25, ctypes.c_void_p(Wp[i % NL].data_ptr()), ctypes.c_void_p(Xpk.data_ptr()), ctypes.c_void_p(out.data_ptr()), N, K, st())
    t_pre = timeit(lambda i: pre(i), NL)
    t_gem = timeit(lambda i: gem(i), NL)
    t_both = timeit(lambda i: (pre(i), gem(i)), NL)
    t_pre_nt = timeit(lambda i: pre(i, 1), NL)
    t_both_nt = timeit(lambda i: (pre(i, 1), gem(i)), NL)
    t_gem2 = timeit(lambda i: (gem(i), gem(i)), NL)
    print(f"== MALL: prefetch(plain) {t_pre:.1f} us | gemm {t_gem:.1f} | prefetch+gemm {t_both:.1f} -> gemm after prefetch {t_both - t_pre:.1f} us")
    print(f"         prefetch(nt) {t_pre_nt:.1f} us | prefetch(nt)+gemm {t_both_nt:.1f} -> gemm after nt prefetch {t_both_nt - t_pre_nt:.1f} us | gemm twice on same W {t_gem2:.1f}")
