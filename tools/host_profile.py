#!/usr/bin/env python3
"""cProfile of one SC-GRPO step on the host (where does Python spend time while the GPU may be idle?).  GPU box."""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.argv = ["bench.py"]
import numpy as np, torch
import bench
import iadr1_amd
from iadr1_amd import rewards
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
dev = torch.device("cuda", 0)
cfg = VLMConfig.qwen25vl_3b()
pol = ParamStore(cfg, dev, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, dev, trainable=False); ref.copy_from(pol)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=256, micro_batch_seqs=64, suppress_eos=True))
N = 64
texts = [bench.CANNED[i % len(bench.CANNED)] for i in range(N)]
def reward_fn(ids):
    comps = [[{"role": "assistant", "content": t}] for t in texts]
    return np.stack([rewards.accuracy_reward(comps, [bench.SOLUTION] * N), rewards.consistency_reward(comps, [bench.SOLUTION] * N)], 1).astype(np.float32)
batches = []
for i in range(3):
    b = bench.synth_batch(cfg, 8, 512, seed=i); b["pixel_values"] = b["pixel_values"].to(dev); batches.append(b)
eng.step(batches[0], reward_fn); eng.step(batches[1], reward_fn)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
eng.step(batches[2], reward_fn)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_callees("text_plan_shared|text_plan|vision_plan|shared_logit_rows|Segments"); print(s.getvalue()[:6000])
