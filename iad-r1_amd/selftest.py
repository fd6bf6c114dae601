"""smoke(): one tiny SC-GRPO micro-step (teacher-forced completions + a short hipGraph rollout) of the HIP path
on cuda:0, checked against the CPU oracle.  The oracle is imported here ONLY as the checker."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def smoke() -> None:
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import fixture_util as fx
    from oracle import qwen25vl as oq
    from oracle import sc_grpo as og

    from . import hip, rewards
    from .params import ParamStore, VLMConfig
    from .sc_grpo import GRPOArgs, SCGRPOEngine

    assert torch.cuda.is_available(), "smoke() needs an MI355X (cuda:0); the HIP path has no CPU fallback"
    print("libiadr1_hip", hip.version(), "on", torch.cuda.get_device_name(0))
    cfg_d = fx.TINY
    cfg = VLMConfig.from_dict(cfg_d)
    G, C, seed = 4, 10, 21
    w_ref = fx.make_weights(cfg_d, 0)
    w_pol = fx.perturb_weights(w_ref, 1)
    pol, ref = ParamStore(cfg, "cuda:0", True), ParamStore(cfg, "cuda:0", False)
    pol.load_named(w_pol)
    ref.load_named(w_ref)
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C))
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, cfg_d, seed)], cfg_d["pad_token_id"])
    px = fx.synth_pixel_values([grid], cfg_d, seed=seed)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": px, "image_grid_thw": [grid]}
    comps = fx.synth_completions(G, C, cfg_d, seed + 100, {1: 6, 3: 0})
    texts = ["<think>a</think><location>upper left</location><type>scratch</type><answer>yes</answer>", "<think>b</think><answer>no</answer>", "junk",
             "<think>c</think><location>top left</location><type>surface scratch</type><answer>yes</answer>"]
    sol = ["<think>gt</think><location>top left</location><type>scratch</type><answer>yes</answer>"] * G
    wrapped = [[{"role": "assistant", "content": t}] for t in texts]
    rew = np.stack([rewards.accuracy_reward(wrapped, sol), rewards.consistency_reward(wrapped, sol)], 1).astype(np.float32)
    out = eng.loss_and_grads(batch, comps, rew)
    # checker
    o_pol = oq.Qwen25VLOracle(cfg_d, w_pol, requires_grad=True)
    o_ref = oq.Qwen25VLOracle(cfg_d, w_ref)
    ref_out = og.sc_grpo_step(o_pol, o_ref, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(px), [grid], comps,
                              torch.from_numpy(rew), G, 0.04, cfg_d["eos_token_id"], cfg_d["pad_token_id"])
    m = ref_out["completion_mask"].bool().numpy()
    dlp = np.abs(out["logps"].cpu().numpy()[m] - ref_out["logps"].detach().numpy()[m]).max()
    dloss = abs(out["metrics"]["loss"] - float(ref_out["loss"]))
    print(f"smoke: loss hip={out['metrics']['loss']:.6f} oracle={float(ref_out['loss']):.6f} |dlogp|max={dlp:.4f}")
    assert dlp < 0.06 and dloss < 1e-3, (dlp, dloss)
    eng.optimizer_step()
    toks = eng.rollout(batch, greedy=True)
    want = o_ref.__class__(cfg_d, pol.export_named()).greedy_generate(torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(px), [grid], C)[:, ids.shape[1]:]
    agree = float((torch.from_numpy(toks[:1, :C]) == want).float().mean())
    print(f"smoke: greedy rollout agreement with the oracle on updated weights = {agree:.2f}")
    assert agree >= 0.8, agree  # bf16 device path vs fp32 oracle: a near-tie may flip late tokens; the bit-exact check lives in tests/
    print("smoke OK")
