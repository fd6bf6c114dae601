#!/usr/bin/env python3
"""Where the microseconds of the latency-bound decode kernels go: run with IADR1_HIP_LIB pointing at a -DIADR1_STAMPS build
(tools/build_variant.py stamps -DIADR1_STAMPS).  Wave 0 of every block stamps the 100 MHz wall clock at fixed program points; this script
launches each decode kernel of the 3B shapes in a loop (rotating weights so they come from HBM) and prints, per kernel, the distribution over
blocks of:   first stamp - earliest first stamp (dispatch skew), and the stage durations between consecutive stamps; plus the HIP-event time
per launch.  Output: text table (profiles/r02_decode_stamps.txt)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd import hip, ops
dev = "cuda"
L = hip.lib()


def stamps(unit):
    buf = (ctypes.c_ulonglong * (8 * 4096))()
    fn = getattr(L, "iadr1_debug_stamps_" + unit)
    fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
    assert fn(buf) == 0
    return np.frombuffer(buf, dtype=np.uint64).reshape(8, 4096).astype(np.int64)


def report(name, unit, slots, nblocks, labels, us):
    s = stamps(unit)[:, :nblocks]
    t0 = s[slots[0]].min()
    rows = [f"{name}: {us:6.2f} us per launch (HIP events, back-to-back); {nblocks} blocks; stamps of the LAST launch, 10 ns ticks -> us"]
    first = (s[slots[0]] - t0) / 100.0
    rows.append(f"   block start after the first block's start: median {np.median(first):5.2f}  p90 {np.percentile(first, 90):5.2f}  max {first.max():5.2f}")
    for a, b, lab in zip(slots[:-1], slots[1:], labels):
        d = (s[b] - s[a]) / 100.0
        rows.append(f"   {lab:38s} median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f}")
    end = (s[slots[-1]].max() - t0) / 100.0
    rows.append(f"   first block start -> last block end: {end:5.2f} us")
    print("\n".join(rows), flush=True)


def timeit(fn, n, reps=5):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


M, H, I, NL = 64, 2048, 11008, 12
Hq, Hkv, D = 16, 2, 128
x = ops.pack_act(torch.randn(M, H, device=dev).to(torch.bfloat16))
# ---- q|k|v projection + rotary + cache append (narrow kernel, out_mode 4) -------------------------------------------------------------------
qkv_w = [torch.randn((Hq + 2 * Hkv) * D, H, device=dev).to(torch.bfloat16) * 0.02 for _ in range(NL)]
qkv_b = torch.zeros((Hq + 2 * Hkv) * D, device=dev, dtype=torch.bfloat16)
pk = [ops.pack_qkv_rope(w, qkv_b, Hq, Hkv, D) for w in qkv_w]
npages = 64 * 26 + 8
kc = torch.zeros(npages, Hkv, 32, D, dtype=torch.bfloat16, device=dev); vc = torch.zeros(npages, Hkv, D, 32, dtype=torch.bfloat16, device=dev)
slot = (torch.arange(M, device=dev, dtype=torch.int64) * 26 + 20) * 32 + 5
cos = torch.ones(M, D // 2, device=dev); sin = torch.zeros(M, D // 2, device=dev)
q_out = torch.empty(M, (Hq + 2 * Hkv) * D, dtype=torch.bfloat16, device=dev)
us = timeit(lambda i: ops.gemm_qkv_rope_kv(x, pk[i % NL][0], pk[i % NL][1], q_out, cos, sin, slot, kc, vc, Hq, Hkv, D), NL)
report("qkv+rope+kv (gemm_skinny_kernel<1,16>, 10.5 MB)", "gemm", [0, 1, 2, 3], 160, ["entry -> MFMAs done (loads + MFMA)", "LDS exchange + barrier", "epilogue (reduce, rope, stores)"], us)
# ---- o projection (narrow kernel, split-K 2) ----------------------------------------------------------------------------------------------------
ow = [ops.pack_weight(torch.randn(H, Hq * D, device=dev).to(torch.bfloat16) * 0.02) for _ in range(NL)]
part = torch.empty(2, M, H, dtype=torch.float32, device=dev)
us = timeit(lambda i: ops.gemm_skinny(x, ow[i % NL], H, out=part, ksplit=2), NL)
report("o projection (gemm_skinny_kernel<1,16>, ksplit 2, 8.4 MB)", "gemm", [0, 1, 2], 256, ["entry -> MFMAs done (loads + MFMA)", "LDS exchange + barrier"], us)
# ---- row RMSNorm summing 8 slabs ----------------------------------------------------------------------------------------------------------------
p8 = torch.randn(8, M, H, device=dev); res = torch.randn(M, H, device=dev).to(torch.bfloat16); w = torch.ones(H, device=dev, dtype=torch.bfloat16)
y = ops.PackedAct(M, H, dev)
us = timeit(lambda i: ops.rmsnorm_fwd(None, w, 1e-6, res=res, res_out=res, x32=p8, out=y), 4)
report("RMSNorm row kernel, 8 slabs", "norm", [0, 1, 2, 3], 64, ["entry -> row summed (slab + residual loads)", "block reduction", "scale + stores"], us)
p2 = torch.randn(2, M, H, device=dev)
us = timeit(lambda i: ops.rmsnorm_fwd(None, w, 1e-6, res=res, res_out=res, x32=p2, out=y), 4)
report("RMSNorm row kernel, 2 slabs", "norm", [0, 1, 2, 3], 64, ["entry -> row summed (slab + residual loads)", "block reduction", "scale + stores"], us)
# ---- paged decode attention, context 640 ------------------------------------------------------------------------------------------------------------
bt = torch.zeros(M, 26, dtype=torch.int32, device=dev)
for s_ in range(M):
    bt[s_, :26] = torch.arange(26, dtype=torch.int32) + s_ * 26 + 1
ctx = torch.full((M,), 640, dtype=torch.int32, device=dev)
kcs = [torch.randn(npages, Hkv, 32, D, device=dev).to(torch.bfloat16) for _ in range(4)]
vcs = [torch.randn(npages, Hkv, D, 32, device=dev).to(torch.bfloat16) for _ in range(4)]
o = ops.PackedAct(M, Hq * D, dev)
us = timeit(lambda i: ops.attn_decode(q_out[:, : Hq * D], kcs[i % 4], vcs[i % 4], bt, ctx, Hq, Hkv, D, D**-0.5, out=o), 4)
report("attn_decode (ctx 640, 42 MB of K/V)", "attn", [0, 1, 2, 3], 128, ["entry -> pages done (table, K/V loads, MFMA)", "LDS exchange + barrier", "combine + stores"], us)
# ---- persistent gate|up with SwiGLU ---------------------------------------------------------------------------------------------------------------------
gu = [ops.pack_gateup(torch.randn(2 * I, H, device=dev).to(torch.bfloat16)) for _ in range(NL)]
a = ops.PackedAct(M, I, dev)
us = timeit(lambda i: ops.gemm_skinny(x, gu[i % NL], 2 * I, swiglu=True, out=a), NL)
report("gate|up + SwiGLU (gemm_skinny_pers_kernel<8,8>, 90 MB)", "gemm", [4, 5, 6], 256, ["entry -> first group's MFMAs done", "remaining groups + epilogues"], us)
