#!/usr/bin/env python3
"""Is a CU-masked stream slow ON ITS OWN depending on which hardware queue it got?  Creates masked streams one after the other (192 of 256 CUs each) and times a
chain of 400 dependent 64-row RMSNorm launches and a chain of 200 skinny-GEMM launches on each, nothing else running; `plain` = an ordinary torch stream.
python tools/masked_queue_probe.py [n_streams]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import iadr1_amd  # noqa
from iadr1_amd import hip, ops

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
K = 2048
x = torch.zeros(64, K, dtype=torch.bfloat16, device=dev)
g = torch.ones(K, dtype=torch.bfloat16, device=dev)
y = torch.empty_like(x)
w = torch.randn(11008 * 2, K, device=dev).to(torch.bfloat16)
wp = ops.pack_weight(w) if hasattr(ops, "pack_weight") else None


def chain(stream):
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(20):
            ops.hip.call("rmsnorm_fwd", x, None, 0, None, None, None, g, y, None, 64, K, K, K, K, 1e-6, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(400):
            ops.hip.call("rmsnorm_fwd", x, None, 0, None, None, None, g, y, None, 64, K, K, K, K, 1e-6, None)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 400 * 1e3


print(f"plain stream: {chain(torch.cuda.Stream()):.2f} us per launch")
print(f"plain stream: {chain(torch.cuda.Stream()):.2f} us per launch")
for i in range(n):
    s = hip.cu_mask_stream(64, 192)
    print(f"masked stream #{i} (handle {int(s.cuda_stream):#x}): {chain(s):.2f} us per launch, again {chain(s):.2f}")
