#!/usr/bin/env python3
"""GPU-box probe: (1) GEMM throughput of iadr1_gemm_nt_bf16 on the hot-path shapes (vs torch.matmul as a
yardstick only), (2) skinny GEMM bandwidth, (3) ds_read_b64_tr_b16 lane mapping.  Writes gpurun_out/probe.json."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iadr1_amd  # noqa
from iadr1_amd import ops

dev = "cuda"
res = {"device": torch.cuda.get_device_name(0)}


def bench(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (24576, 2560, 2048), (24576, 2048, 2048), (24576, 22016, 2048), (24576, 2048, 11008),
          (4096, 151936, 2048), (8192, 3840, 1280), (8192, 6912, 1280), (2048, 2048, 24576), (22016, 2048, 24576)]
gem = []
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    if M * N * 2 > 6e9:
        continue
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = bench(lambda: ops.gemm_nt(a, b, out=out))
    t2 = bench(lambda: torch.matmul(a, b.t(), out=out))
    fl = 2.0 * M * N * K
    gem.append({"M": M, "N": N, "K": K, "iadr1_TF": fl / t / 1e12, "torch_TF": fl / t2 / 1e12})
    print(gem[-1], flush=True)
    del a, b, out
res["gemm"] = gem

sk = []
for M, N, K in []:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    t = bench(lambda: ops.gemm_skinny(x, w, out=y), iters=50)
    t2 = bench(lambda: torch.matmul(x, w.t()), iters=50)
    sk.append({"M": M, "N": N, "K": K, "us": t * 1e6, "GBps": N * K * 2 / t / 1e9, "torch_us": t2 * 1e6})
    print(sk[-1], flush=True)
res["skinny"] = sk

try:
    L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libtr_probe.so"))
    out = torch.zeros(4 * 64 * 4, dtype=torch.int16, device=dev)
    rc = L.run_tr_probe(ctypes.c_void_p(out.data_ptr()))
    res["tr_probe_rc"] = rc
    res["tr_probe"] = out.view(4, 64, 4).cpu().tolist()
except Exception as e:  # probe only
    res["tr_probe_error"] = repr(e)

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"))
print("probe done")
