#!/usr/bin/env python3
"""Launch the dominant GEMM shapes of the SC-GRPO step a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
for M, N, K in [(24576, 22016, 2048), (24576, 2048, 11008), (22016, 2048, 24576)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB: evicts the 256 MiB Infinity Cache between launches
    for _ in range(3):
        flush.zero_()
        ops.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    del a, b, out, flush
