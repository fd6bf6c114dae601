#!/usr/bin/env python3
"""Which device allocations (hipMalloc through torch's caching allocator) happen in steps 2, 3, ... of the 3B SC-GRPO bench step: new segments per step with size and
the stream they belong to.  python tools/alloc_steps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd.params import ParamStore, VLMConfig
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
import bench
dev = torch.device("cuda", 0)
cfg = VLMConfig.qwen25vl_3b()
pol = ParamStore(cfg, dev, trainable=True); pol.init_random(seed=0)
ref = ParamStore(cfg, dev, trainable=False); ref.copy_from(pol)
eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=512, max_completion_length=256, micro_batch_seqs=64, suppress_eos=True))
batch = bench.synth_batch(cfg, 8, 512, seed=5)
rew = lambda comp: np.zeros((len(comp), 2), dtype=np.float32)
seen = set()
for it in range(5):
    n0 = torch.cuda.memory_stats()["num_device_alloc"]
    eng.step(batch, rew)
    torch.cuda.synchronize()
    n1 = torch.cuda.memory_stats()["num_device_alloc"]
    segs = {(s["address"], s["total_size"], s["stream"]) for s in torch.cuda.memory_snapshot()}
    new = sorted(segs - seen, key=lambda z: -z[1])
    seen |= segs
    print(f"step {it}: {n1 - n0} device allocations; new segments (MB, stream): {[(round(sz / 2**20, 1), hex(st)) for _, sz, st in new][:24]}", flush=True)
print("streams: main", hex(eng.__dict__.get('_main_stream').cuda_stream) if eng.__dict__.get('_main_stream') else None, "side", hex(eng._shadow.stream.cuda_stream) if eng._shadow else None,
      "decode", hex(eng._rollout.decode_stream.cuda_stream) if eng._rollout.decode_stream else None, "wgrad", hex(eng.pol.wgrad_stream.cuda_stream))
