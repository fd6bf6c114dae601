#!/usr/bin/env python3
"""Second feasibility probe (round 5): WHY does the decode replay stall completely while the shadow pass's kernels run (tools/overlap_trace.py), when a
pre-enqueued stream of GEMMs next to it only slows it down (tools/overlap_probe.py)?  A worker thread starts enqueuing on a second stream `delay` ms AFTER the
rollout has begun (so the decode stream's ring already holds ~60 replays), with a selectable kernel mix:
    gemm      the four projections of a layer at M rows (what overlap_probe.py ran)
    norm      + the two RMSNorms and the rotary kernel
    attn      + the training attention over a [prompt ++ completion] batch
    h2d       + one pinned host-to-device copy per layer
Usage: python tools/overlap_probe2.py gemm norm attn h2d [--rows 2048] [--delay 100]"""
import argparse, os, sys, time, dataclasses, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd import hip, ops
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("mixes", nargs="+")
ap.add_argument("--rows", type=int, default=2048)
ap.add_argument("--delay", type=float, default=100.0)
ap.add_argument("--layers", type=int, default=300)
ap.add_argument("--pre", type=int, default=0, help="1: enqueue the second stream BEFORE the rollout (overlap_probe.py's order)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=36, v_depth=2, v_fullatt=(1,))
pol = ParamStore(cfg, dev, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, dev, trainable=False); ref.copy_from(pol)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=256, micro_batch_seqs=64, suppress_eos=True))
batch = bench.synth_batch(cfg, 8, 512, seed=5)
batch["pixel_values"] = batch["pixel_values"].to(dev)
H, I, QW = cfg.hidden_size, cfg.intermediate_size, cfg.qkv_width
Hq, Hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
M = a.rows
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
w = lambda n: ref.w("layers.0." + n)
qkv = torch.randn(M, QW, device=dev).to(torch.bfloat16)
o_in = torch.randn(M, Hq * D, device=dev).to(torch.bfloat16)
ab = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
h = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
act = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
cos = torch.rand(M, D // 2, device=dev); sin = torch.rand(M, D // 2, device=dev)
seg = ops.Segments(list(range(0, M, 256)), list(range(256, M + 256, 256)), dev)
host = torch.zeros(1024, dtype=torch.int32).pin_memory()
devbuf = torch.zeros(1024, dtype=torch.int32, device=dev)


def layer(mix):
    if mix in ("norm", "attn", "h2d", "all"):
        ops.hip.call("rmsnorm_fwd", x, None, 0, None, None, None, w("ln1"), h, None, M, H, H, H, H, 1e-6, None)
    if mix in ("h2d", "all"):
        devbuf.copy_(host, non_blocking=True)
    ops.gemm_nt(x, w("qkv.w"), bias=w("qkv.b"), out=qkv)
    if mix in ("norm", "attn", "all"):
        ops.rope_(qkv, cos, sin, Hq + Hkv, D)
    if mix in ("attn", "all"):
        ops.attn_fwd(qkv[:, :Hq * D], qkv[:, Hq * D: Hq * D + Hkv * D], qkv[:, Hq * D + Hkv * D:], seg, Hq, Hkv, D, True, D ** -0.5, out=o_in, want_lse=False)
    ops.gemm_nt(o_in, w("o.w"), out=ab)
    if mix in ("norm", "attn", "h2d", "all"):
        ops.hip.call("rmsnorm_fwd", ab, None, 0, None, x, x, w("ln2"), h, None, M, H, H, H, H, 1e-6, None)
    ops.gemm_swiglu(x, w("gu.w"), gu_out=None, a_out=act, keep_gu=False)
    ops.gemm_nt(act, w("down.w"), out=ab)


main = torch.cuda.Stream()
torch.cuda.set_stream(main)
gstream = torch.cuda.Stream()
eng.rollout(batch)      # capture
torch.cuda.synchronize()
for mix in ["none"] + a.mixes:
    evs = []

    def worker():
        torch.cuda.set_device(0)
        if not a.pre:
            time.sleep(a.delay * 1e-3)
        with torch.cuda.stream(gstream):
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(a.layers):
                layer(mix)
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            evs.append((e0, e1))

    eng._rollout.decode_events = []
    th = None
    if mix != "none":
        th = threading.Thread(target=worker)
        th.start()
        if a.pre:
            th.join()
    t0 = time.perf_counter()
    eng.rollout(batch)
    if th is not None:
        th.join()
    torch.cuda.synchronize()
    e0, e1, n, _ = eng._rollout.decode_events[-1]
    out = dict(mix=mix, rows=M, pre=a.pre, decode_ms_per_step=round(e0.elapsed_time(e1) / n, 4))
    if evs:
        g0, g1 = evs[0]
        out.update(second_stream_ms=round(g0.elapsed_time(g1), 1), ms_per_layer=round(g0.elapsed_time(g1) / a.layers, 3), started_after_decode_start_ms=round(e0.elapsed_time(g0), 1),
                   ended_before_decode_end_ms=round(g1.elapsed_time(e1), 1))
    print(out, flush=True)
