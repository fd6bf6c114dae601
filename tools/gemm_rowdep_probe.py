#!/usr/bin/env python3
"""Are the training GEMMs row-independent ACROSS launch shapes?  Rows [0, 128) of A through gemm_nt / gemm_swiglu / linear_logprob at several M: the launcher picks
the 128^2 or the 256^2 kernel (and the fused or un-fused SwiGLU) by M, and the chunked reference pass (iadr1_amd/overlap.py) is bit-equal to the one-shot pass
only if those choices do not change a row's bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iadr1_amd  # noqa
from iadr1_amd import ops
dev = "cuda"
torch.manual_seed(0)
for (K, N) in ((2048, 2560), (2048, 2048), (2048, 22016), (11008, 2048)):
    A = (torch.randn(4096, K, device=dev) * 0.5).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    ref = None
    for M in (128, 256, 512, 1024, 2048, 4096):
        C = ops.gemm_nt(A[:M], W)[:128].clone()
        if ref is None:
            ref = C
        print(f"gemm_nt K={K} N={N} M={M}: rows[0:128] equal to M=128: {torch.equal(C, ref)}  max|d|={float((C.float() - ref.float()).abs().max()):.3e}")
K, I = 2048, 11008
A = (torch.randn(4096, K, device=dev) * 0.5).to(torch.bfloat16)
W = (torch.randn(2 * I, K, device=dev) * 0.02).to(torch.bfloat16)
ref = None
for M in (128, 256, 512, 1024, 2048, 4096):
    _, a = ops.gemm_swiglu(A[:M], W, keep_gu=False)
    a = a[:128].clone()
    ref = a if ref is None else ref
    print(f"gemm_swiglu M={M}: equal to M=128: {torch.equal(a, ref)} max|d|={float((a.float() - ref.float()).abs().max()):.3e}")
