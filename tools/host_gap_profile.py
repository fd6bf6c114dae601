#!/usr/bin/env python3
"""Host time between the end of the rollout (device sync: the sampled tokens come to the host) and the first kernel of the reference pass: cProfile of
SCGRPOEngine.loss_and_grads(backward=False) up to its first text_forward, on the 3B bench shapes with 2 layers (host work does not depend on the depth)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cProfile, pstats, bench, iadr1_amd, dataclasses
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
from iadr1_amd import vlm
DEV = torch.device("cuda", 0)
cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
pol = ParamStore(cfg, DEV, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, DEV, trainable=False); ref.copy_from(pol)
batch = bench.synth_batch(cfg, 8, 512, seed=5)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=256, micro_batch_seqs=64, suppress_eos=True))
rew = lambda comp: np.zeros((len(comp), 2), dtype=np.float32)
for _ in range(2): eng.step(batch, rew)
torch.cuda.synchronize()
marks = {}
orig = vlm.Engine.text_forward
def tf(self, *a, **k):
    marks.setdefault("first_forward", time.perf_counter())
    return orig(self, *a, **k)
vlm.Engine.text_forward = tf
vis = eng.vision_policy(batch, save=True)
comp = eng.rollout(batch, vis=vis, train_carry=None)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable()
eng.loss_and_grads(batch, comp, rew(comp), backward=False, vis=vis)
pr.disable(); torch.cuda.synchronize()
print(f"rollout end -> first text_forward call: {1e3 * (marks['first_forward'] - t0):.2f} ms of host time")
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
