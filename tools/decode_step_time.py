#!/usr/bin/env python3
"""Time of ONE decode step of the rollout (hipGraph replay, 64 sequences, 3B shapes, context 512 + t) without any training around it:
builds random-init policy weights, runs the prefill, replays the captured decode graph and reports ms per step (HIP events on the replay
stream) for each value of the A/B environment switches given on the command line (NAME=v1,v2 ...; each combination in a fresh process is
NOT needed: the kernels read their switches per launch unless noted).  Usage: python tools/decode_step_time.py [--layers N] [--steps K]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import dataclasses
import iadr1_amd  # noqa
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="3b", choices=["3b", "7b"])
ap.add_argument("--layers", type=int, default=0, help="0 = the model's own depth")
ap.add_argument("--group", type=int, default=8)
ap.add_argument("--prompts", type=int, default=8)
ap.add_argument("--steps", type=int, default=255)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--trace", type=int, default=0, help="1: with the training-arena side outputs (needs memory for the arena)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
base = VLMConfig.qwen25vl_3b() if a.model == "3b" else VLMConfig.qwen25vl_7b()
a.layers = a.layers or base.num_hidden_layers
cfg = dataclasses.replace(base, num_hidden_layers=a.layers, v_depth=2, v_fullatt=(1,))
pol = ParamStore(cfg, dev, trainable=True)
pol.init_random(seed=0)
ref = ParamStore(cfg, dev, trainable=False)
ref.copy_from(pol)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=a.group, max_prompt_length=512, max_completion_length=a.steps + 1, micro_batch_seqs=64, suppress_eos=True))
batch = bench.synth_batch(cfg, a.prompts, 512, seed=5)
batch["pixel_values"] = batch["pixel_values"].to(dev)
for rep in range(a.reps + 1):
    carry = {} if a.trace else None
    vis = eng.vision_policy(batch, save=bool(a.trace))
    if eng._rollout is not None:
        eng._rollout.decode_events = []
    t0 = time.perf_counter()
    eng.rollout(batch, vis=vis, train_carry=carry)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if rep == 0:
        eng._rollout.decode_events = []
        continue
    ev = eng._rollout.decode_events
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in ev)
    n = sum(k for _, _, k, _ in ev)
    print(f"decode step {ms / n:.4f} ms  ({n} steps, rollout wall {wall * 1e3:.1f} ms, layers {a.layers}, trace {a.trace})", flush=True)
