import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import dataclasses
import iadr1_amd, bench
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
DEV = torch.device("cuda", 0)
cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
pol = ParamStore(cfg, DEV, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, DEV, trainable=False); ref.copy_from(pol)
G, C, Bp = 8, 32, 2
batch = bench.synth_batch(cfg, Bp, 512, seed=5)
rew = lambda comp: np.random.RandomState(0).rand(len(comp), 2).astype(np.float32)
res = {}
for mode in (False, True):
    pol.grad.zero_()
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=512, max_completion_length=C, micro_batch_seqs=Bp * G, suppress_eos=True, reuse_decode=mode, seed=3))
    m = eng.step(batch, rew, do_optimizer_step=False)
    torch.cuda.synchronize()
    res[mode] = (m, pol.grad.clone())
    print(mode, m)
a, b = res[False][1], res[True][1]
print("grad rel err", float((a - b).norm() / a.norm()), "cos", float((a @ b) / (a.norm() * b.norm())))
for mode in (False, True):
    g = res[mode][1]
    print(mode, "nan count", int(torch.isnan(g).sum()), "of", g.numel())
# where do the two arenas differ?  compare saved activations of layer 0/1 between a normal forward and the traced one
