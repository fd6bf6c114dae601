#!/usr/bin/env python3
"""How far can the host run ahead of the GPU in the decode loop?  After one rollout (graph captured) replays the decode graph K times WITHOUT synchronising and
reports the host time of the loop next to the GPU time: a host that is held inside hipGraphLaunch until the previous launch of the same executable graph has
drained has no lead, and any other launches it has to make (the shadow pass of iadr1_amd/overlap.py) stall the decode queue."""
import os, sys, time, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
import bench
dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 36
cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=L, v_depth=2, v_fullatt=(1,))
pol = ParamStore(cfg, dev, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, dev, trainable=False); ref.copy_from(pol)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=256, micro_batch_seqs=64, suppress_eos=True))
batch = bench.synth_batch(cfg, 8, 512, seed=5)
batch["pixel_values"] = batch["pixel_values"].to(dev)
eng.rollout(batch)
r = eng._rollout
r.step.zero_(); r.ctx_len.fill_(512); r.pos.fill_(511)
torch.cuda.synchronize()
for K in (8, 32, 64):
    r.step.zero_(); r.ctx_len.fill_(512)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    stamps = []
    for _ in range(K):
        r.graph.replay()
        stamps.append(time.perf_counter() - t0)
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"K={K}: host loop {t_host * 1e3:.2f} ms ({t_host / K * 1e3:.3f} ms per replay), GPU {e0.elapsed_time(e1):.2f} ms ({e0.elapsed_time(e1) / K:.3f} per step), until sync {t_all * 1e3:.2f} ms; "
          f"host stamps of the first replays (ms): {[round(s * 1e3, 2) for s in stamps[:6]]}")
