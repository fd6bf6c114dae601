#!/usr/bin/env python3
"""Round-5 feasibility probe of co-scheduling: the hipGraph decode replay (latency-bound, MFMA idle) next to a stream of training-shaped GEMMs
(MFMA-bound, HBM mostly idle) on ONE MI355X, separated by CU masks (include/iadr1_hip.h iadr1_stream_create_cu_mask).

For every configuration `D:G[:m]` on the command line (D = CUs the decode stream owns, G = CUs of the GEMM stream; G = 0: no GEMM stream; `m` = 0 leaves
the decode stream unmasked, i.e. only the GEMM side is confined) it reports the decode step time (HIP events around the replay loop) and the rate of the
GEMM stream inside that window (one decoder layer's four projections at M = chunk rows: q|k|v, o, gate|up + SwiGLU, down).

    python tools/overlap_probe.py 256:0 192:0 192:64 160:96 256:64:0 [--trace 1] [--rows 4096] [--layers 36]
"""
import argparse, os, sys, time, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd import hip, ops
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="+")
ap.add_argument("--layers", type=int, default=36)
ap.add_argument("--steps", type=int, default=255)
ap.add_argument("--trace", type=int, default=0)
ap.add_argument("--rows", type=int, default=4096, help="token rows of a teacher-forced chunk (64 sequences x 64 steps)")
ap.add_argument("--gemm-layers", type=int, default=700)
a = ap.parse_args()
dev = torch.device("cuda", 0)
NCU = torch.cuda.get_device_properties(dev).multi_processor_count
base = VLMConfig.qwen25vl_3b()
cfg = dataclasses.replace(base, num_hidden_layers=a.layers, v_depth=2, v_fullatt=(1,))
pol = ParamStore(cfg, dev, trainable=True)
pol.init_random(seed=0)
ref = ParamStore(cfg, dev, trainable=False)
ref.copy_from(pol)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=a.steps + 1, micro_batch_seqs=64, suppress_eos=True))
batch = bench.synth_batch(cfg, 8, 512, seed=5)
batch["pixel_values"] = batch["pixel_values"].to(dev)

H, I, QW = cfg.hidden_size, cfg.intermediate_size, cfg.qkv_width
M = a.rows
x = torch.randn(M, H, device=dev).to(torch.bfloat16)
P = ref
w = lambda n: P.w("layers.0." + n)
qkv = torch.empty(M, QW, dtype=torch.bfloat16, device=dev)
o_in = torch.randn(M, cfg.num_attention_heads * cfg.head_dim, device=dev).to(torch.bfloat16)
ab = torch.empty(M, H, dtype=torch.bfloat16, device=dev)
act = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
flop_layer = 2.0 * M * (QW * H + H * o_in.shape[1] + 2 * I * H + H * I)


def gemm_layer():
    ops.gemm_nt(x, w("qkv.w"), bias=w("qkv.b"), out=qkv)
    ops.gemm_nt(o_in, w("o.w"), out=ab)
    ops.gemm_swiglu(x, w("gu.w"), gu_out=None, a_out=act, keep_gu=False)
    ops.gemm_nt(act, w("down.w"), out=ab)


def run(dc, gc, mask_decode=True, prio=0):
    hip.set_decode_cus(dc if mask_decode else 0)
    if eng._rollout is not None:
        eng._rollout.graph = None
    dstream = hip.cu_mask_stream(NCU - dc, dc) if (mask_decode and dc < NCU) else torch.cuda.Stream(priority=-1 if prio else 0)
    gstream = (hip.cu_mask_stream(0, gc) if gc < NCU else torch.cuda.Stream()) if gc else None
    out = None
    for rep in range(2):          # rep 0 captures the graph for this CU count
        torch.cuda.synchronize()
        evs = []
        t_ref = torch.cuda.Event(enable_timing=True)
        t_ref.record()
        if gstream is not None and rep == 1:
            gstream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(gstream):
                for _ in range(a.gemm_layers):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gemm_layer()
                    e1.record()
                    evs.append((e0, e1))
        dstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(dstream):
            carry = {} if a.trace else None
            vis = eng.vision_policy(batch, save=bool(a.trace))
            eng._rollout and setattr(eng._rollout, "decode_events", [])
            t0 = time.perf_counter()
            eng.rollout(batch, vis=vis, train_carry=carry)
            if eng._rollout.decode_events is None:
                eng._rollout.decode_events = []
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if rep == 0:
            continue
        ev = eng._rollout.decode_events
        e0, e1, n, _ = ev[-1]
        d0, d1 = t_ref.elapsed_time(e0), t_ref.elapsed_time(e1)
        ms = (d1 - d0) / n
        inside = [(g0, g1) for g0, g1 in evs if t_ref.elapsed_time(g0) >= d0 and t_ref.elapsed_time(g1) <= d1]
        if inside:
            span = t_ref.elapsed_time(inside[-1][1]) - t_ref.elapsed_time(inside[0][0])
            tf = len(inside) * flop_layer / (span * 1e-3) / 1e12
            per = span / len(inside)
        else:
            tf, per = 0.0, 0.0
        out = dict(decode_cus=dc, gemm_cus=gc, mask_decode=mask_decode, decode_prio=prio, decode_ms_per_step=round(ms, 4), rollout_wall_ms=round(wall * 1e3, 1),
                   gemm_layers_inside=len(inside), gemm_ms_per_layer=round(per, 3), gemm_tflops=round(tf, 1))
    print(out, flush=True)
    return out


# GEMM stream alone on G CUs (no decode) for reference
def gemm_alone(gc):
    gstream = hip.cu_mask_stream(0, gc) if gc < NCU else torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(gstream):
        for _ in range(5):
            gemm_layer()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 60
        for _ in range(n):
            gemm_layer()
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(dict(gemm_alone_cus=gc, ms_per_layer=round(ms, 3), tflops=round(flop_layer / ms / 1e9, 1)), flush=True)


# everything on non-null streams: hipExtStreamCreateWithCUMask streams are BLOCKING streams (hipStreamDefault), i.e. any operation on the null stream -- an event
# record, a wait_stream against it -- waits for all their queued work and holds back what they queue afterwards
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
seen = set()
for c in a.configs:
    parts = [int(z) for z in c.split(":")]
    dc, gc = parts[0], parts[1]
    md = (parts[2] != 0) if len(parts) > 2 else True
    pr = parts[3] if len(parts) > 3 else 0
    if gc and gc not in seen:
        seen.add(gc)
        gemm_alone(gc)
    run(dc, gc, md, pr)
