#!/usr/bin/env python3
"""Product skinny-GEMM timings on the three big decode streams of the 3B model with weights that really come from HBM (rotating
buffers): gate|up (fused SwiGLU, packed in/out), down (split-K 8 slabs), lm_head (fp32).  Run once per IADR1_SKINNY_PERS value."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
def timeit(fn, n):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(3):
        for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3
H, I, V, NL = 2048, 11008, 151936, int(os.environ.get("NL", "16"))   # NL=1: the same 90 MB every launch (memory-side cache hits)
x = ops.pack_act(torch.randn(64, H, device=dev).to(torch.bfloat16))
gu = [ops.pack_gateup(torch.randn(2 * I, H, device=dev).to(torch.bfloat16)) for _ in range(NL)]
a = ops.PackedAct(64, I, dev)
us = timeit(lambda i: ops.gemm_skinny(x, gu[i % NL], 2 * I, swiglu=True, out=a), NL)
print(f"PERS={os.environ.get('IADR1_SKINNY_PERS','1')} gate|up swiglu {us:7.1f} us  {2*I*H*2/us/1e6:5.2f} TB/s", flush=True)
del gu
dn = [ops.pack_weight(torch.randn(H, I, device=dev).to(torch.bfloat16)) for _ in range(2 * NL)]
ain = ops.pack_act(torch.randn(64, I, device=dev).to(torch.bfloat16))
part = torch.empty(8, 64, H, dtype=torch.float32, device=dev)
us = timeit(lambda i: ops.gemm_skinny(ain, dn[i % (2 * NL)], H, out=part, ksplit=8), 2 * NL)
print(f"PERS={os.environ.get('IADR1_SKINNY_PERS','1')} down ks8        {us:7.1f} us  {I*H*2/us/1e6:5.2f} TB/s", flush=True)
del dn
if NL < 3:
    sys.exit(0)
lm = [ops.pack_weight(torch.randn(V, H, device=dev).to(torch.bfloat16)) for _ in range(3)]
lg = torch.empty(64, V, dtype=torch.float32, device=dev)
us = timeit(lambda i: ops.gemm_skinny(x, lm[i % 3], V, out=lg), 3)
print(f"PERS={os.environ.get('IADR1_SKINNY_PERS','1')} lm_head        {us:7.1f} us  {V*H*2/us/1e6:5.2f} TB/s", flush=True)
