#!/usr/bin/env python3
"""Decode step time on a CU-masked stream, nothing else running (round 5): is the 10 % the masked decode stream costs (2.8 -> 3.1 ms) the missing CUs, or the
masked queue as such?  For each CU count: a fresh Rollout sized for that many CUs (split-K choices, persistent grids), the decode replays on a stream masked to
them (256 = a mask with every bit set; `plain` = torch's current stream).  python tools/decode_mask_probe.py [--trace 1]"""
import argparse, os, sys, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd import hip
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
from iadr1_amd.rollout import Rollout
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--trace", type=int, default=1)
ap.add_argument("cus", nargs="*", default=["plain", "256", "224", "192"])
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
for c in a.cus:
    eng._rollout = None
    if c == "plain":
        hip.set_decode_cus(0)
        split = {}
    elif c.startswith("x"):       # whole XCDs: x6 = every CU of XCDs 0..5
        nx = int(c[1:])
        cus = hip.xcd_cus(range(nx))
        hip.set_decode_cus(len(cus))
        split = {"decode_cus": len(cus), "decode_stream": hip.cu_mask_stream(0, 0, cus=cus)}
    else:
        n = int(c)
        hip.set_decode_cus(0 if n == NCU else n)
        split = {"decode_cus": 0 if n == NCU else n, "decode_stream": hip.cu_mask_stream(NCU - n, n)}
    eng._cu_split = lambda N=None, split=split: split
    for rep in range(3):
        carry = {} if a.trace else None
        vis = eng.vision_policy(batch, save=bool(a.trace))
        if eng._rollout is not None:
            eng._rollout.decode_events = []
        eng.rollout(batch, vis=vis, train_carry=carry)
        torch.cuda.synchronize()
        if rep == 0:
            eng._rollout.decode_events = []
            continue
        e0, e1, n_, _ = eng._rollout.decode_events[-1]
        print(f"decode on {c} CUs: {e0.elapsed_time(e1) / n_:.4f} ms per step (trace {a.trace}, ks_o {eng._rollout.ks_o}, ks_down {eng._rollout.ks_down})", flush=True)
