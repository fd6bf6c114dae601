#!/usr/bin/env python3
"""Round 6 probe (not product): gemm_nt_256 on v_mfma_f32_16x16x32_bf16 (the product build) vs v_mfma_f32_32x32x16_bf16 (a probe build of the library:
python tools/build_variant.py mfma32 -DIADR1_PROBE_MFMA32, selected with IADR1_HIP_LIB=iad-r1_amd/lib/variants/libiadr1_hip_mfma32.so; run this once per library).  Correctness of the selected form against an fp32 matmul (bf16 store, fp32 store, fp32 accumulate, a ragged edge shape), then
the seven hot shapes of profiles/r05_gemm_band.txt, alternating, on random data (DVFS: MI355X_MICROARCH.md).   python tools/gemm_mfma32_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import iadr1_amd  # noqa
from iadr1_amd import ops
dev = "cuda"
tag = "32x32x16" if "mfma32" in os.environ.get("IADR1_HIP_LIB", "") else "16x16x32"
torch.manual_seed(0)
for (M, N, K) in [(4096, 4096, 512), (4096, 12288, 2048), (4000, 4100, 2048), (2048, 2048, 4096)]:      # (>= 192 tiles of 256^2, M and N >= 512: the 256^2 kernel, except the last)
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    want = a.float() @ b.float().t()
    got = ops.gemm_nt(a, b, bias=bias).float()
    e0 = float((got - (want + bias.float())).abs().max() / want.abs().max())
    got32 = ops.gemm_nt(a, b, out_dtype=torch.float32)
    e1 = float((got32 - want).abs().max() / want.abs().max())
    acc = torch.ones(M, N, dtype=torch.float32, device=dev)
    ops.gemm_nt(a, b, out=acc, accumulate=True)
    e2 = float((acc - 1 - want).abs().max() / want.abs().max())
    print(f"[{tag}] M={M} N={N} K={K}: rel err bf16+bias {e0:.2e}  fp32 {e1:.2e}  fp32 accumulate {e2:.2e}", flush=True)
    assert e0 < 1e-2 and e1 < 1e-4 and e2 < 1e-4
shapes = [(20480, 22016, 2048, "bf16"), (20480, 2048, 11008, "bf16"), (20480, 2048, 2048, "bf16"), (20480, 2560, 2048, "bf16"), (22016, 2048, 20480, "acc"), (2048, 11008, 20480, "acc"),
          (12288, 22016, 2048, "bf16")]
for rep in range(2):
    for (M, N, K, mode) in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
        out = torch.zeros(M, N, dtype=torch.float32 if mode == "acc" else torch.bfloat16, device=dev)
        f = (lambda: ops.gemm_nt(a, b, out=out, accumulate=True)) if mode == "acc" else (lambda: ops.gemm_nt(a, b, out=out))
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"[{tag}] M={M:6d} N={N:6d} K={K:6d} {mode:4s} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
        del a, b, out
