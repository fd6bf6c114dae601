#!/usr/bin/env python3
"""Which pairs of HIP streams really run concurrently?  (round 5, tools/overlap_trace.py: with the shadow pass on one stream and the decode replay on another, the
decode queue's dependent kernels were each dispatched ~50 us late, or not at all, for as long as the other queue had a big grid in dispatch -- even on disjoint CU
masks.)  Stream A runs a chain of small dependent kernels (RMSNorm of 64 rows, the decode step's smallest kernel), stream B a chain of big-grid GEMMs; for each of
several candidate B streams the A chain is timed alone and next to B.  python tools/stream_pair_probe.py [n_candidates]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iadr1_amd  # noqa
from iadr1_amd import ops
dev = "cuda"
n_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = 2048
x = torch.randn(64, H, device=dev).to(torch.bfloat16)
g = torch.ones(H, device=dev, dtype=torch.bfloat16)
y = torch.empty_like(x)
A2 = torch.randn(2048, H, device=dev).to(torch.bfloat16)
W = (torch.randn(22016, H, device=dev) * 0.02).to(torch.bfloat16)
C = torch.empty(2048, 22016, dtype=torch.bfloat16, device=dev)


def small_chain(n=400):
    for _ in range(n):
        ops.hip.call("rmsnorm_fwd", x, None, 0, None, None, None, g, y, None, 64, H, H, H, H, 1e-6, None)


def big_chain(n=60):
    for _ in range(n):
        ops.gemm_nt(A2, W, out=C)


def timed_small(sa, sb=None):
    torch.cuda.synchronize()
    if sb is not None:
        with torch.cuda.stream(sb):
            big_chain()
    with torch.cuda.stream(sa):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        small_chain()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


use_graph = os.environ.get("PROBE_GRAPH", "1") == "1"
sa = torch.cuda.Stream()
cands = [torch.cuda.Stream() for _ in range(n_cand)]
if use_graph:           # the A chain as a captured graph, like the decode step
    with torch.cuda.stream(sa):
        small_chain(8)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        small_chain(400)

    def timed_small(sa, sb=None):       # noqa: F811
        torch.cuda.synchronize()
        if sb is not None:
            with torch.cuda.stream(sb):
                big_chain()
        with torch.cuda.stream(sa):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

for rep in range(2):
    alone = timed_small(sa)
    print(f"A alone: {alone:.2f} ms for 400 dependent 64-row RMSNorm launches ({alone / 400 * 1e3:.1f} us each), graph={use_graph}, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
    for i, sb in enumerate(cands):
        t = timed_small(sa, sb)
        print(f"  next to big GEMMs on candidate stream {i} (ptr {sb.cuda_stream:#x}): {t:.2f} ms ({t / alone:.2f}x)")

# ---- the same with disjoint CU masks: A on 192 CUs, B candidates on the other 64
from iadr1_amd import hip
if os.environ.get("PROBE_MASKED", "1") == "1":
    main = torch.cuda.Stream()
    torch.cuda.set_stream(main)          # (masked streams are blocking streams: stay off the null stream)
    sa_m = hip.cu_mask_stream(64, 192)
    alone = timed_small(sa_m)
    print(f"masked: A alone on 192 CUs: {alone:.2f} ms")
    for i in range(n_cand):
        sb_m = hip.cu_mask_stream(0, 64)
        t = timed_small(sa_m, sb_m)
        t2 = timed_small(sa_m, sb_m)
        print(f"  next to big GEMMs on a fresh 64-CU stream #{i}: {t:.2f} / {t2:.2f} ms ({t / alone:.2f}x)")
