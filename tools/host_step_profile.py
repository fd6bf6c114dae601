import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cProfile, pstats, bench, iadr1_amd
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
import dataclasses
DEV = torch.device("cuda", 0)
cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
pol = ParamStore(cfg, DEV, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, DEV, trainable=False); ref.copy_from(pol)
batch = bench.synth_batch(cfg, 8, 512, seed=5)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=4, micro_batch_seqs=64, suppress_eos=True))
rew = lambda comp: np.zeros((len(comp), 2), dtype=np.float32)
for _ in range(2): eng.step(batch, rew)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): eng.step(batch, rew)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
