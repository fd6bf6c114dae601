#!/usr/bin/env python3
"""Round 6 probe (not product): the decode step next to the persistent weight prefetcher (iadr1_weight_prefetch, iad-r1_amd/wprefetch.py).
Each configuration D:P[:lead[:nt[:what]]] = decode replays on a stream masked to D CUs (256 = torch's plain stream), the prefetcher on P other CUs
(0 = none) picked on a different dispatch pipe (overlap.pick_concurrent_stream), `lead` layers ahead, nt = non-temporal prefetch loads, what = mlp | all | gu.
    python tools/wprefetch_probe.py 256:0 224:0 224:32 224:32:1 240:16 ..."""
import argparse, os, sys, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd import hip, overlap
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
from iadr1_amd.wprefetch import WeightPrefetcher
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--trace", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("cfgs", nargs="*", default=["256:0", "224:0", "224:32"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream())
cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), v_depth=2, v_fullatt=(1,))
pol = ParamStore(cfg, dev, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, dev, trainable=False); ref.copy_from(pol)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=256, micro_batch_seqs=64, suppress_eos=True))
batch = bench.synth_batch(cfg, 8, 512, seed=5)
batch["pixel_values"] = batch["pixel_values"].to(dev)
NCU = torch.cuda.get_device_properties(dev).multi_processor_count
WHAT = {"mlp": (("gu", "down"), ()), "gu": (("gu",), ()), "all": (("gu", "down"), ("qkv", "o")), "down": (("down",), ()), "qkvo": ((), ("qkv", "o"))}
for spec in a.cfgs:
    f = spec.split(":")
    D, P = int(f[0]), int(f[1])
    lead = int(f[2]) if len(f) > 2 else 0
    nt = int(f[3]) if len(f) > 3 else 0
    what = f[4] if len(f) > 4 else "mlp"
    eng._rollout = None
    if D >= NCU:
        hip.set_decode_cus(0)
        split, anchor = {}, torch.cuda.current_stream()
    else:
        hip.set_decode_cus(D)
        ds = hip.cu_mask_stream(NCU - D, D)
        split, anchor = {"decode_cus": D, "decode_stream": ds}, ds
    eng._cu_split = lambda N=None, split=split: split
    pf = None
    ratio = 0.0
    if P > 0:
        st, ratio = overlap.pick_concurrent_stream(anchor, lambda: hip.cu_mask_stream(0, P), ref.w("layers.0.gu.w"))
        if st is None:
            print(f"{spec}: no pipe-clean stream pair found (best ratio {ratio:.2f}); skipped", flush=True)
            continue
        pf = WeightPrefetcher(eng.pol, st, P, what=WHAT[what][0], next_what=WHAT[what][1], lead=lead, nt=bool(nt))
    for rep in range(a.reps):
        carry = {} if a.trace else None
        vis = eng.vision_policy(batch, save=bool(a.trace))
        if eng._rollout is not None:
            eng._rollout.decode_events = []
            eng._rollout.wprefetch = pf
        eng.rollout(batch, vis=vis, train_carry=carry)
        torch.cuda.synchronize()
        if rep == 0:
            eng._rollout.decode_events = []
            eng._rollout.wprefetch = pf
            continue
        e0, e1, n_, _ = eng._rollout.decode_events[-1]
        rp = pf.report() if pf is not None else {}
        print(f"{spec:>16s}: decode {e0.elapsed_time(e1) / n_:.4f} ms per step (decode CUs {D}, prefetch CUs {P}, lead {lead}, nt {nt}, {what}; pair ratio {ratio:.2f}) {rp}", flush=True)
