#!/usr/bin/env python3
"""Host vs GPU timeline of one SC-GRPO step at the bench shape (3B, 8 prompts x G 8, P 512, C 256): for every phase the host time at which it was ENTERED / LEFT
(enqueue side) and the GPU time at which the stream reached those points (HIP events), both in ms after the start of the step.  Where `gpu_enter` ~ `host_enter`
the GPU had caught up with the host: it was idle, waiting for launches.  python tools/step_timeline.py [--layers 36]"""
import argparse, os, sys, time, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd import vlm, sc_grpo, rollout as ro_mod
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=36)
ap.add_argument("--steps", type=int, default=2)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = VLMConfig.qwen25vl_3b()
if a.layers != cfg.num_hidden_layers:
    cfg = dataclasses.replace(cfg, num_hidden_layers=a.layers)
pol = ParamStore(cfg, dev, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, dev, trainable=False); ref.copy_from(pol)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=256, micro_batch_seqs=64, suppress_eos=True))
batch = bench.synth_batch(cfg, 8, 512, seed=5)
batch["pixel_values"] = batch["pixel_values"].to(dev)
rew = lambda comp: np.zeros((len(comp), 2), dtype=np.float32)
marks = []
t0 = [0.0, None]


def wrap(obj, name, label=None):
    orig = getattr(obj, name)
    label = label or name

    def w(*args, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter(); e0.record()
        r = orig(*args, **kw)
        e1.record(); h1 = time.perf_counter()
        marks.append((label, h0, e0, h1, e1))
        return r
    setattr(obj, name, w)


wrap(eng, "vision_policy"); wrap(eng, "rollout"); wrap(eng, "loss_and_grads"); wrap(eng, "optimizer_step")
wrap(eng.pol, "logprobs", "policy.logprobs"); wrap(eng.pol, "logprobs_backward", "policy.logprobs_backward"); wrap(eng.pol, "text_backward", "policy.text_backward")
wrap(eng.pol, "vision_backward", "policy.vision_backward"); wrap(eng.pol, "text_context_from_trace", "policy.context_from_trace")
wrap(eng.pol, "text_plan_shared", "host: text_plan_shared")
for _ in range(2):
    eng.step(batch, rew)
torch.cuda.synchronize()
for it in range(a.steps):
    marks.clear()
    main = eng.__dict__.get("_main_stream") or torch.cuda.current_stream()       # the engine's own stream (co-scheduling) -- NOT a fresh one: the caching allocator's blocks belong to streams
    with torch.cuda.stream(main):
        ref_ev = torch.cuda.Event(enable_timing=True)
        h_ref = time.perf_counter(); ref_ev.record()
        eng.step(batch, rew)
        end_ev = torch.cuda.Event(enable_timing=True); end_ev.record()
    torch.cuda.synchronize()
    h_end = time.perf_counter()
    print(f"--- step {it}: host {1e3 * (h_end - h_ref):.1f} ms, gpu {ref_ev.elapsed_time(end_ev):.1f} ms; shadowed {eng.last_step_shadowed}")
    print(f"{'phase':<32} {'host_enter':>10} {'gpu_enter':>10} {'host_exit':>10} {'gpu_exit':>10}   gpu_lag_at_enter")
    for label, h0, e0, h1, e1 in sorted(marks, key=lambda m: m[1]):
        he, hx = 1e3 * (h0 - h_ref), 1e3 * (h1 - h_ref)
        ge, gx = ref_ev.elapsed_time(e0), ref_ev.elapsed_time(e1)
        print(f"{label:<32} {he:>10.1f} {ge:>10.1f} {hx:>10.1f} {gx:>10.1f}   {ge - he:>8.1f}")
