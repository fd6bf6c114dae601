#!/usr/bin/env python3
"""Round 6 probe (not product): what does a RESIDENT weight-prefetch launch (polling only: its mark never arrives) cost the side stream's GEMMs?
Side stream = CUs [0, 64); the prefetcher on a third CU-masked stream, (a) inside the side stream's CUs ([0, P)), (b) on CUs of its own ([64, 64 + P)), with
n blocks; GEMM chain = 20 x [2048 x 22016 x 2048] (one gate|up of a 32-step chunk) and 4 x [20480 x 22016 x 2048].   python tools/pf_residency_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd import ops, hip
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream())
K = 2048
W = (torch.randn(22016, K, device=dev) * 0.02).to(torch.bfloat16)
a_s = torch.randn(2048, K, device=dev).to(torch.bfloat16); c_s = torch.empty(2048, 22016, dtype=torch.bfloat16, device=dev)
a_b = torch.randn(20480, K, device=dev).to(torch.bfloat16); c_b = torch.empty(20480, 22016, dtype=torch.bfloat16, device=dev)
segs = ops.h2d(np.asarray([[[W.data_ptr(), 1 << 20]]], dtype=np.int64), dev)
mark = torch.zeros(1, dtype=torch.int32, device=dev)
side = hip.cu_mask_stream(0, 64)


def timed(body, resident=None, n_blocks=0, ms=250):
    torch.cuda.synchronize()
    if resident is not None:
        with torch.cuda.stream(resident):
            ops.hip.call("weight_prefetch", segs, 1, 1, mark, 1, 1, None, 0, 0, 0, n_blocks, ms, None)
    with torch.cuda.stream(side):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); body(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


small = lambda: [ops.gemm_nt(a_s, W, out=c_s) for _ in range(20)]
big = lambda: [ops.gemm_nt(a_b, W, out=c_b) for _ in range(4)]
timed(small), timed(big)
t_s, t_b = timed(small), timed(big)
print(f"side stream alone: 20 small GEMMs {t_s:.2f} ms, 4 big GEMMs {t_b:.2f} ms", flush=True)
for where, first in (("inside the side stream's CUs", 0), ("on CUs of its own", 64)):
    for P, nb in ((16, 8), (16, 16), (16, 32), (8, 8), (32, 32)):
        st = hip.cu_mask_stream(first, P)
        r_s = timed(small, st, nb) / t_s
        r_b = timed(big, st, nb) / t_b
        print(f"resident prefetcher {where}: mask {P} CUs, {nb} blocks -> small GEMMs {r_s:.2f}x, big GEMMs {r_b:.2f}x", flush=True)
# the same on torch's ordinary stream as the prefetcher's queue (no CU mask: blocks land anywhere)
st = torch.cuda.Stream()
print(f"resident prefetcher on an ordinary stream, 16 blocks -> small {timed(small, st, 16) / t_s:.2f}x, big {timed(big, st, 16) / t_b:.2f}x", flush=True)
