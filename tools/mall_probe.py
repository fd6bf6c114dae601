#!/usr/bin/env python3
"""Round 6 probe (not product): what would the decode step's weight-streaming launches gain if their weights were already in the 256 MB memory-side cache?
The product gemm_skinny launches of a 3B decoder layer (M = 64, decode-packed X) on (a) rotating weight buffers (24 x: really from HBM), (b) one buffer again and
again (whatever the cache hierarchy keeps of a 10 - 90 MB matrix), (c) rotating buffers, each read by a plain / non-temporal read kernel just before (t(read + gemm) -
t(read)): the upper bound for any prefetcher.    python tools/mall_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import iadr1_amd  # noqa
from iadr1_amd import ops, hip
dev = "cuda"
torch.cuda.set_stream(torch.cuda.Stream())
NL = 24


def timeit(fn, n, reps=3):
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(n):
            fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


mark = torch.zeros(1, dtype=torch.int32, device=dev)
status = torch.zeros(4, dtype=torch.int32, device=dev)


def reader(w, nt, blocks=256):
    """the prefetch kernel as a one-shot read of one tensor: mark == first == last, lead 0, one unit"""
    seg = ops.h2d(__import__("numpy").asarray([[[w.data_ptr(), w.numel() * w.element_size()]]], dtype="int64"), w.device)
    return lambda: hip.call("weight_prefetch", seg, 1, 1, mark, 0, 0, None, 0, 0, nt, blocks, 1000, status)


shapes = [("gate|up (swiglu)", 22016, 2048, dict(swiglu=True)), ("down (8 K slices)", 2048, 11008, dict(ksplit=8)), ("q|k|v-like", 2560, 2048, {}), ("o (2 K slices)", 2048, 2048, dict(ksplit=2)),
          ("lm_head", 151936, 2048, dict(out_dtype=torch.float32))]
for name, N, K, kw in shapes:
    nl = NL if N * K * 2 < 200e6 else 6
    Ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nl)]
    Wp = [ops.pack_gateup(w) if kw.get("swiglu") else ops.pack_weight(w) for w in Ws]
    del Ws
    X = ops.pack_act((torch.randn(64, K, device=dev)).to(torch.bfloat16))
    if kw.get("swiglu"):
        out = ops.PackedAct(64, N // 2, dev)
    elif kw.get("ksplit", 1) > 1:
        out = torch.empty(kw["ksplit"], 64, N, dtype=torch.float32, device=dev)
    else:
        out = torch.empty(64, N, dtype=kw.get("out_dtype", torch.bfloat16), device=dev)
    kk = {k: v for k, v in kw.items() if k != "out_dtype"}
    gem = lambda i: ops.gemm_skinny(X, Wp[i % nl], N, out=out, **kk)
    mb = N * K * 2 / 1e6
    t_cold = timeit(gem, nl)
    t_same = timeit(lambda i: gem(0), nl)
    line = f"{name:18s} {mb:6.1f} MB: rotating {t_cold:6.1f} us ({mb / t_cold:5.2f} TB/s) | same buffer {t_same:6.1f} us ({mb / t_same:5.2f} TB/s)"
    for nt in (0, 1):
        rd = [reader(w, nt) for w in Wp]
        t_rd = timeit(lambda i: rd[i % nl](), nl)
        t_both = timeit(lambda i: (rd[i % nl](), gem(i)), nl)
        line += f" | after a {'nt' if nt else 'plain'} read ({t_rd:5.1f} us, {mb / t_rd:4.2f} TB/s): {t_both - t_rd:6.1f} us"
    print(line, flush=True)
    del Wp
# how fast do few CUs read?  (the prefetcher's share of the device)
W = torch.empty(8, 128 << 20, dtype=torch.uint8, device=dev)       # eight 128 MiB slices in turn: every read comes from HBM
for cus in (8, 16, 32, 64, 128, 256):
    st = hip.cu_mask_stream(0, cus) if cus < 256 else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        for nt in (0, 1):
            rd = [reader(W[j], nt, blocks=cus) for j in range(8)]
            t = timeit(lambda i: rd[i % 8](), 8)
            print(f"read kernel on {cus:3d} CUs ({'nt' if nt else 'plain'}): 128 MiB in {t:7.1f} us = {134.2 / t:5.2f} TB/s = {134.2e3 / t / cus:6.1f} GB/s per CU", flush=True)
