"""Model-level parity of the HIP engine (through the C ABI) on the tiny Qwen2.5-VL-shaped workload:
vs the golden vectors captured from the reference (tests/golden/*) and vs the CPU oracle.  GPU only.
Tolerances are for bf16 storage / fp32 accumulate against an fp32 reference, stated per check."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import fixture_util as fx  # noqa: E402
import iadr1_amd  # noqa: E402,F401
from iadr1_amd.params import ParamStore, VLMConfig  # noqa: E402
from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine  # noqa: E402
from iadr1_amd.sft import SFTArgs, SFTEngine  # noqa: E402
from iadr1_amd.vlm import Engine  # noqa: E402

DEV = "cuda"
CFG = VLMConfig.from_dict(fx.TINY)


def store(weights, trainable):
    s = ParamStore(CFG, DEV, trainable=trainable)
    s.load_named(weights)
    return s


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_param_store_roundtrip():
    w = fx.make_weights(fx.TINY, 0)
    s = store(w, trainable=False)
    back = s.export_named()
    for k, v in w.items():
        assert np.array_equal(back[k].numpy(), v.reshape(back[k].shape)), k


def test_forward_matches_reference_golden(golden_dir):
    g = load(golden_dir, "logps_padded.npz")
    e = Engine(store(fx.make_weights(fx.TINY, 0), False))
    grids = [tuple(int(z) for z in r) for r in g["image_grid_thw"]]
    vp = e.vision_plan(grids)
    img, _ = e.vision_forward(torch.from_numpy(g["pixel_values"]).to(DEV), vp, save=False)
    assert relerr(img.float().cpu().numpy(), g["image_embeds"]) < 3e-2          # bf16 ViT vs fp32 reference
    ids, mask = g["input_ids"], g["attention_mask"]
    rows = np.cumsum([0] + [t * h * w // 4 for t, h, w in grids])
    plan = e.text_plan(ids, mask, [[gr] for gr in grids], [[int(r)] for r in rows[:-1]])
    pos_ref = g["position_ids"]
    hf, _ = e.text_forward(plan, img, save=False)
    B, S = ids.shape
    keep = mask.astype(bool)
    assert relerr(hf.float().cpu().numpy().reshape(B, S, -1)[keep], g["hidden_last"][keep]) < 3e-2
    sel = np.arange(B * S).reshape(B, S)[:, :-1].reshape(-1)
    tgt = ids[:, 1:].reshape(-1)
    lp, _ = e.logprobs(hf, torch.from_numpy(sel).to(DEV), torch.from_numpy(tgt).to(DEV), save=False)
    valid = (mask[:, 1:] * mask[:, :-1]).astype(bool)
    got = lp.cpu().numpy().reshape(B, S - 1)
    assert np.abs(got[valid] - g["per_token_logps"][valid]).max() < 0.06          # log-prob units
    assert pos_ref.shape[0] == 3


@pytest.mark.parametrize("name,mb", [("sc_grpo_g4.npz", 16), ("sc_grpo_g8.npz", 16), ("sc_grpo_g8.npz", 3), ("sc_grpo_g8_far.npz", 16), ("sc_grpo_trunc.npz", 16),
                                     ("sc_grpo_7b_like.npz", 16), ("sc_grpo_7b_like.npz", 3), ("sc_grpo_qwen2vl.npz", 16), ("sc_grpo_qwen2vl.npz", 3)])
def test_sc_grpo_step_matches_reference_golden(golden_dir, name, mb):
    """HIP engine vs the reference's own compute_loss (tests/golden/sc_grpo_*.npz).  g4 / g8: policy close to the frozen reference (KL ~ 3e-3,
    loss ~ 1e-4); g8_far: policy far from it (KL ~ 0.2, loss ~ 8e-3), where loss and KL tolerances are RELATIVE; trunc: the reference's left
    truncation of the prompt (max_prompt_length = P - 2, REF:630-634); 7b_like: the reference's compute_loss on TINY7 -- untied lm_head and 7 query heads
    per kv head, the structure of BASELINE config 4 (Qwen2.5-VL-7B, 28:4) and of config 5's decoder -- forward AND backward, in both layouts
    (mb = 16: shared prefix, mb = 3: repeated rows, micro-batches cutting the group); qwen2vl: the reference's compute_loss on TINY_Q2 (Qwen2-VL: LayerNorm /
    QuickGELU vision tower without windows, the reference's SC_GRPO_Qwen_Instruct_2_VL.sh), incl. the gradients of its LayerNorm biases."""
    g = load(golden_dir, name)
    meta = json.loads(str(g["meta"]))
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    cfg_d = getattr(fx, meta.get("config", "fixture_util.TINY").split(".")[-1])
    cfg = VLMConfig.from_dict(cfg_d)
    w_ref = fx.make_weights(cfg_d, 0)
    pol, ref = ParamStore(cfg, DEV, trainable=True), ParamStore(cfg, DEV, trainable=False)
    pol.load_named(fx.perturb_weights(w_ref, 1, scale=meta.get("perturb_scale", 0.02)))
    ref.load_named(w_ref)
    mpl = meta["max_prompt_length"] if meta.get("truncate") else 4096
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=mpl, max_completion_length=C, beta=0.04, micro_batch_seqs=mb))
    grid = tuple(meta["grid"])
    ids, mask = fx.left_pad([fx.synth_prompt(grid, meta["n_text"], cfg_d, seed)], cfg_d["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], cfg_d, seed=seed), "image_grid_thw": [grid]}
    eos_rows = {int(k): v for k, v in meta["eos_rows"].items()}
    comps = fx.synth_completions(G, C, cfg_d, seed + 100, eos_rows)
    out = eng.loss_and_grads(batch, comps, g["rewards_per_func"])
    assert np.array_equal(out["completion_mask"], g["completion_mask"])             # integer work: bit-exact
    assert np.array_equal(out["ids"], g["prompt_completion_ids"])
    assert np.array_equal(out["mask"], g["attention_mask"])
    if meta.get("truncate"):
        assert out["ids"].shape[1] == ids.shape[1] - meta["truncate"] + C           # the prompt really lost its first tokens
    m = g["completion_mask"].astype(bool)
    dlp = np.abs(out["logps"].cpu().numpy()[m] - g["per_token_logps"][m]).max()
    dlr = np.abs(out["ref_logps"].cpu().numpy()[m] - g["ref_per_token_logps"][m]).max()
    # bf16 storage vs the fp32 reference: 0.06 log-prob units on TINY (hidden 256, logits within +-5); TINY7 (hidden 896, logits within +-10, log-probs down
    # to -12.8) measured 0.065 / 0.071 -> 0.10 there, the same 1 % of the log-prob range
    tol_lp = 0.06 if cfg.hidden_size <= 256 else 0.10
    assert dlp < tol_lp and dlr < tol_lp, (dlp, dlr)
    np.testing.assert_allclose(out["advantages"].numpy(), g["advantages"], rtol=1e-5, atol=1e-6)
    mt = out["metrics"]
    assert mt["completion_length"] == float(g["metric_completion_length"])
    assert abs(mt["reward"] - float(g["metric_reward"])) < 1e-6 and abs(mt["reward_std"] - float(g["metric_reward_std"])) < 1e-5
    # loss = beta * KL - mean(A-term): the advantage part is exact (exp(p - p.detach()) == 1), the KL part carries the bf16 log-prob noise.  k3 KL is
    # quadratic in (ref - policy): with |noise| ~ 0.02 on a difference of ~ 0.07 (g4 / g8) the relative error of a row's KL can reach tens of percent,
    # on the far-policy goldens (difference ~ 0.6) a few percent.  Tolerances: relative on KL, and on the loss the same error scaled by beta.
    gl, gk = float(g["loss"]), float(g["metric_kl"])
    dk, dl = abs(mt["kl"] - gk), abs(mt["loss"] - gl)
    print(f"[parity] {name} mb={mb}: loss hip={mt['loss']:.6e} ref={gl:.6e} (d={dl:.2e})  kl hip={mt['kl']:.6e} ref={gk:.6e} (d={dk:.2e}, {100 * dk / gk:.1f}%)  |dlogp|max={dlp:.4f}/{dlr:.4f}")
    rel_kl = 0.10 if gk > 0.05 else 0.25
    assert dk <= rel_kl * gk, (mt["kl"], gk)
    assert dl <= 0.04 * rel_kl * gk + 2e-6, (mt["loss"], gl)              # beta = 0.04; never looser than the north star's 1e-3
    assert dl < 1e-3
    grads = pol.export_named(source="grad")
    names = [str(n) for n in g["grad_norm_names"]]
    for n, ref_norm in zip(names, g["grad_norms"]):
        if (n == "lm_head.weight" and cfg.tie_word_embeddings) or ref_norm < 1e-9:
            continue
        got = float(grads[n].norm())
        assert abs(got - ref_norm) <= 0.08 * ref_norm + 1e-7, (n, got, ref_norm)
    for k in g.files:
        if k.startswith("grad::"):
            a, b = grads[k[6:]].numpy().reshape(-1).astype(np.float64), g[k].reshape(-1).astype(np.float64)
            cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
            assert cos > 0.99, (k, cos)


def test_micro_batching_does_not_change_gradients(golden_dir):
    g = load(golden_dir, "sc_grpo_g8.npz")
    meta = json.loads(str(g["meta"]))
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    w_ref = fx.make_weights(fx.TINY, 0)
    grid = tuple(meta["grid"])
    ids, mask = fx.left_pad([fx.synth_prompt(grid, meta["n_text"], fx.TINY, seed)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=seed), "image_grid_thw": [grid]}
    comps = fx.synth_completions(G, C, fx.TINY, seed + 100, {int(k): v for k, v in meta["eos_rows"].items()})
    flat = []
    for mb in (8, 2):
        pol, ref = store(fx.perturb_weights(w_ref, 1), True), store(w_ref, False)
        eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=mb))
        eng.loss_and_grads(batch, comps, g["rewards_per_func"])
        flat.append(pol.grad.clone())
    a, b = flat[0].double(), flat[1].double()
    assert float((a - b).norm() / b.norm()) < 2e-2


def test_a_step_is_bit_reproducible(golden_dir):
    """Round 5: the backward path has no float atomics left -- norm-gain / bias gradients are two-stage ordered sums, the embedding gradient an
    ordered CSR scatter, dK/dV and the split-K weight gradients were ordered already -- so two runs of the same SC-GRPO step from the same state leave the SAME BITS in
    the whole gradient buffer (decoder, vision tower, embedding), and an optimizer step the same parameters."""
    g = load(golden_dir, "sc_grpo_g8.npz")
    meta = json.loads(str(g["meta"]))
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    w_ref = fx.make_weights(fx.TINY, 0)
    grid = tuple(meta["grid"])
    ids, mask = fx.left_pad([fx.synth_prompt(grid, meta["n_text"], fx.TINY, seed)], fx.TINY["pad_token_id"])
    comps = fx.synth_completions(G, C, fx.TINY, seed + 100, {int(k): v for k, v in meta["eos_rows"].items()})
    runs = []
    for rep in range(2):
        pol, ref = store(fx.perturb_weights(w_ref, 1), True), store(w_ref, False)
        eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=16, learning_rate=1e-3))
        batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=seed), "image_grid_thw": [grid]}
        eng.loss_and_grads(batch, comps, g["rewards_per_func"])
        grad = pol.grad.clone()
        eng.optimizer_step()
        torch.cuda.synchronize()
        runs.append((grad, pol.flat.clone()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert float(runs[0][0].abs().sum()) > 0


@pytest.mark.parametrize("share", [True, False])
def test_gradient_checkpointing_recomputes_the_same_gradients(golden_dir, share):
    """`--gradient_checkpointing` (every reference SC-GRPO script): with recomputation forced (GRPOArgs.recompute = "on") the decoder keeps only the rows
    entering each layer and rebuilds a layer's activations in backward with the forward's own kernels -- log-probs and loss are bit-identical to the run that
    kept them, the gradients equal to fp32 rounding (the norm-gain / embedding gradients are summed with fp32 atomics, whose order changes from run to run;
    a recomputed ln1 normalises the stored bf16 row instead of re-adding the branch); "auto" on a tiny model keeps the activations (nothing to save); the
    per-layer arena is not allocated by the checkpointed run."""
    g = load(golden_dir, "sc_grpo_g8_far.npz")
    meta = json.loads(str(g["meta"]))
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    w_ref = fx.make_weights(fx.TINY, 0)
    grid = tuple(meta["grid"])
    ids, mask = fx.left_pad([fx.synth_prompt(grid, meta["n_text"], fx.TINY, seed)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=seed), "image_grid_thw": [grid]}
    comps = fx.synth_completions(G, C, fx.TINY, seed + 100, {int(k): v for k, v in meta["eos_rows"].items()})
    res = {}
    for mode in ("off", "on", "auto"):
        pol, ref = store(fx.perturb_weights(w_ref, 1, scale=0.25), True), store(w_ref, False)
        eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=16 if share else 3, recompute=mode))
        out = eng.loss_and_grads(batch, comps, g["rewards_per_func"])
        res[mode] = (out["logps"].clone(), out["metrics"]["loss"], pol.grad.clone(), "act_save" in eng.pol._ws, "ckpt_x_in" in eng.pol._ws)
    assert res["off"][3] and not res["off"][4] and res["on"][4] and not res["on"][3] and res["auto"][3] and not res["auto"][4]
    for mode in ("on", "auto"):
        assert torch.equal(res[mode][0], res["off"][0]) and res[mode][1] == res["off"][1]
        a_, b_ = res[mode][2].double(), res["off"][2].double()
        dmax, cosv = float((a_ - b_).abs().max()), float((a_ @ b_) / (a_.norm() * b_.norm()))
        assert dmax <= 2e-4 * float(b_.abs().max()) and cosv > 1 - 1e-7, (mode, dmax, float(b_.abs().max()), cosv)
    assert float(res["off"][2].abs().max()) > 0


def test_shared_prefix_layout_equals_repeated_prompt_rows():
    """SC-GRPO loss + gradients with every prompt computed once per group (shared-prefix attention) vs the reference's layout of
    G full [P+C] rows per prompt: two left-padded prompts of different length / image size, ragged completions with EOS."""
    G, C = 4, 9
    grids = [(1, 16, 12), (1, 8, 8)]
    ids, mask = fx.left_pad([fx.synth_prompt(grids[0], 6, fx.TINY, 3), fx.synth_prompt(grids[1], 21, fx.TINY, 4)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values(grids, fx.TINY, seed=5), "image_grid_thw": grids}
    comps = fx.synth_completions(2 * G, C, fx.TINY, 77, {1: 3, 2: 0, 6: 8})
    rewards = np.random.RandomState(0).rand(2 * G, 2).astype(np.float32)
    w_ref = fx.make_weights(fx.TINY, 0)
    res = []
    for share in (True, False):
        pol, ref = store(fx.perturb_weights(w_ref, 1), True), store(w_ref, False)
        eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=2 * G, share_prefix=share))
        out = eng.loss_and_grads(batch, comps, rewards)
        res.append((out, pol.grad.clone()))
    (o1, g1), (o0, g0) = res
    m = o0["completion_mask"].astype(bool)
    assert np.abs(o1["logps"].cpu().numpy()[m] - o0["logps"].cpu().numpy()[m]).max() < 0.03      # bf16 hidden states, different tile orders
    assert np.abs(o1["ref_logps"].cpu().numpy()[m] - o0["ref_logps"].cpu().numpy()[m]).max() < 0.03
    assert abs(o1["metrics"]["loss"] - o0["metrics"]["loss"]) < 1e-3
    a, b = g1.double(), g0.double()
    assert float((a - b).norm() / b.norm()) < 2e-2
    cos = float((a @ b) / (a.norm() * b.norm()))
    assert cos > 0.999, cos


@pytest.mark.parametrize("use_graph", [False, True])
def test_greedy_rollout_token_ids_bit_exact(golden_dir, use_graph):
    g = load(golden_dir, "greedy.npz")
    meta = json.loads(str(g["meta"]))
    w = fx.make_weights(fx.TINY, 0)
    pol, ref = store(w, True), store(w, False)
    new = meta["new_tokens"]
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=2, max_prompt_length=4096, max_completion_length=new, use_hip_graph=use_graph, suppress_eos=False))
    grids = [tuple(x) for x in meta["grids"]]
    batch = {"input_ids": g["prompt_ids"], "attention_mask": g["prompt_mask"], "pixel_values": fx.synth_pixel_values(grids, fx.TINY, seed=meta["seed"]), "image_grid_thw": grids}
    eng._rollout = None
    toks = eng.rollout(batch, greedy=True)
    P = g["prompt_ids"].shape[1]
    want = g["sequences"][:, P:]
    assert float(g["margin"].min()) > 0.05   # the reference's own top-2 margin: no near-ties in this fixture
    for b in range(2):
        for gi in range(2):
            assert toks[b * 2 + gi, :new].tolist() == want[b].tolist(), (b, gi, toks[b * 2 + gi].tolist(), want[b].tolist())


def test_one_rollout_for_all_micro_batches_of_an_optimizer_step():
    """GRPOConfig.batch_rollouts (default): training_step rolls the prompts of ALL gradient-accumulation micro-batches out in one group rollout -- the policy is the
    same for all of them (the reference pushes its weights to vLLM once per optimizer step, REF:637-641) and a decode step costs the same for 4 or 8 sequences.
    (1) `combine_batches` + one greedy rollout gives bit-identical token ids to the per-micro-batch greedy rollouts (ragged prompt lengths, two image sizes: left
    padding is outside every attention segment).  (2) the trainer runs ONE rollout per optimizer step with the switch on and accum many with it off; the step's
    logged metrics are finite and the completions of micro-batch k are scored against micro-batch k's solutions."""
    from iadr1_amd import rewards, rollout as ro
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer, combine_batches, trim_completions
    w = fx.make_weights(fx.TINY, 0)
    pol, ref = store(w, True), store(w, False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=12, suppress_eos=False))
    grids = [(1, 16, 12), (1, 8, 8), (1, 12, 12)]
    parts = []
    for i, (gr, n) in enumerate(zip(grids, [5, 17, 9])):
        ids, mask = fx.left_pad([fx.synth_prompt(gr, n, fx.TINY, 177 + i)], fx.TINY["pad_token_id"])
        parts.append({"input_ids": ids, "attention_mask": mask, "pixel_values": torch.from_numpy(fx.synth_pixel_values([gr], fx.TINY, seed=177 + i)), "image_grid_thw": [gr]})
    both = combine_batches(parts, CFG.pad_token_id)
    assert both["input_ids"].shape == (3, max(p["input_ids"].shape[1] for p in parts)) and both["image_grid_thw"] == grids and both["images_per_prompt"] == [1, 1, 1]
    assert both["pixel_values"].shape[0] == sum(p["pixel_values"].shape[0] for p in parts)
    together = eng.rollout(both, greedy=True)
    for k, p in enumerate(parts):
        alone = eng.rollout(p, greedy=True)
        assert np.array_equal(together[4 * k: 4 * k + 4], alone), k
    t = np.array([[7, 8, CFG.eos_token_id, 0, 0, 0], [7, CFG.eos_token_id, 0, 0, 0, 0]])
    assert trim_completions(t, CFG.eos_token_id).shape == (2, 3) and trim_completions(np.full((2, 6), 9), CFG.eos_token_id).shape == (2, 6)

    class Proc:                      # the processor call returns the prepared prompt of the row it is asked for
        def apply_chat_template(self, conv, add_generation_prompt=True, tokenize=False):
            return "P"

        def __call__(self, text=None, images=None, **kw):
            return parts[images[0][1]]

        def batch_decode(self, ids, skip_special_tokens=True):
            return ["<think>a</think><location>top left</location><type>scratch</type><answer>yes</answer>"] * len(ids)
    chat = [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "?"}]}]
    rows = lambda k: [{"prompt": chat, "image": [("synthetic", k)], "solution": "<think>gt</think><location>top left</location><type>scratch</type><answer>yes</answer>"}]
    for on in (True, False):
        tr = SCGRPOTrainer((CFG, w), [rewards.accuracy_reward, rewards.consistency_reward], processing_class=Proc(), train_dataset=None,
                           args=GRPOConfig(output_dir="/tmp/iadr1_batch_rollouts", num_generations=4, max_completion_length=12, max_prompt_length=4096, per_device_train_batch_size=1,
                                           gradient_accumulation_steps=3, save_steps=0, batch_rollouts=on))
        n0 = ro.STATS["rollouts"]
        losses = tr.training_step([rows(0), rows(1), rows(2)])
        assert ro.STATS["rollouts"] - n0 == (1 if on else 3)
        assert len(losses) == 3 and all(np.isfinite(l) for l in losses) and len(tr._metrics["reward"]) == 3


def test_rollout_stops_soon_after_every_sequence_has_finished():
    """EOS live: the decode loop ends once all sequences have finished, without draining the queue -- `decode_advance` keeps an all-finished flag, the host reads a copy
    that is three polls (12 steps) old.  One prompt x G greedy rows are identical, so with the EOS id set to the token greedy decoding emits at step 6 every row ends
    there: the loop must run far fewer than max_new steps, the tokens up to and including the EOS equal the unstopped run's, everything after is pad (REF:680-683)."""
    import dataclasses
    from iadr1_amd import rollout as ro
    w = fx.make_weights(fx.TINY, 0)
    pol, ref = store(w, True), store(w, False)
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 91)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=91), "image_grid_thw": [grid]}
    C = 120
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=C, suppress_eos=True))
    free = eng.rollout(batch, greedy=True)
    k = next(j for j in range(4, 40) if free[0, j] not in free[0, :j].tolist())          # first occurrence of that token: the EOS position is unambiguous
    cfg2 = dataclasses.replace(CFG, eos_token_id=int(free[0, k]))
    pol2, ref2 = ParamStore(cfg2, DEV, True), ParamStore(cfg2, DEV, False)
    pol2.load_named(w)
    ref2.load_named(w)
    eng2 = SCGRPOEngine(cfg2, pol2, ref2, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=C, suppress_eos=False))
    n0 = ro.STATS["decode_steps"]
    got = eng2.rollout(batch, greedy=True)
    ran = ro.STATS["decode_steps"] - n0
    assert ran <= k + 1 + 4 * 3 + 4 and ran < C - 1, (ran, k)                           # k + 1 steps to reach the EOS, then at most LAG x POLL (+ one poll period) more
    assert np.array_equal(got[:, : k + 1], free[:, : k + 1]) and (got[:, k + 1:] == cfg2.pad_token_id).all()


def test_sampled_rollout_is_reproducible_and_in_vocab():
    w = fx.make_weights(fx.TINY, 0)
    pol, ref = store(w, True), store(w, False)
    grids = [(1, 16, 12), (1, 8, 8)]
    rows = [fx.synth_prompt(gr, n, fx.TINY, 77 + i) for i, (gr, n) in enumerate(zip(grids, [5, 17]))]
    ids, mask = fx.left_pad(rows, fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values(grids, fx.TINY, seed=77), "image_grid_thw": grids}
    outs = []
    for _ in range(2):
        eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=16, seed=5))
        outs.append(eng.rollout(batch))
    assert np.array_equal(outs[0], outs[1])
    assert outs[0].min() >= 0 and outs[0].max() < fx.TINY["text"]["vocab_size"]
    assert len({tuple(r) for r in outs[0][:4].tolist()}) > 1   # the G samples of one prompt differ


def test_sft_loss_curve_matches_reference_golden(golden_dir):
    g = load(golden_dir, "sft.npz")
    meta = json.loads(str(g["meta"]))
    p = store(fx.make_weights(fx.TINY, 0), True)
    eng = SFTEngine(CFG, p, SFTArgs(learning_rate=meta["lr"], weight_decay=meta["wd"], max_grad_norm=0.0))
    batch = {k: g[k] for k in ("input_ids", "attention_mask", "labels", "pixel_values")}
    batch["image_grid_thw"] = [tuple(int(z) for z in r) for r in g["image_grid_thw"]]
    losses = []
    for _ in range(3):
        losses.append(eng.loss_and_grads(batch))
        eng.optimizer_step()
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-2, atol=2e-2)


class _FakeProcessor:
    """Stands in for AutoProcessor (no tokenizer files offline): fixed prompt tensors, canned decode strings."""

    def __init__(self, batch, texts):
        self.batch, self.texts = batch, texts

    def apply_chat_template(self, conv, add_generation_prompt=True, tokenize=False):
        return "PROMPT"

    def __call__(self, text=None, images=None, **kw):
        return {k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in self.batch.items()}

    def batch_decode(self, ids, skip_special_tokens=True):
        return list(self.texts[: len(ids)])


def test_trainer_api_runs_two_optimizer_steps():
    from iadr1_amd import rewards
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 3)], fx.TINY["pad_token_id"])
    batch = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "pixel_values": torch.from_numpy(fx.synth_pixel_values([grid], fx.TINY, seed=3)),
             "image_grid_thw": torch.tensor([grid])}
    texts = ["<think>a</think><location>upper left</location><type>scratch</type><answer>yes</answer>", "<think>b</think><answer>no</answer>", "junk",
             "<think>c</think><location>top left</location><type>surface scratch</type><answer>yes</answer>"]
    rows = [{"prompt": [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "q"}]}], "image": [object()],
             "solution": "<think>gt</think><location>top left</location><type>scratch</type><answer>yes</answer>"}] * 4
    cfgT = GRPOConfig(output_dir="/tmp/iadr1_trainer_test", num_generations=4, max_completion_length=8, max_prompt_length=4096, learning_rate=1e-3,
                      per_device_train_batch_size=1, gradient_accumulation_steps=2, max_steps=2, logging_steps=1, save_steps=0)
    tr = SCGRPOTrainer((CFG, fx.make_weights(fx.TINY, 0)), [rewards.accuracy_reward, rewards.consistency_reward], args=cfgT, train_dataset=rows,
                       processing_class=_FakeProcessor(batch, texts))
    before = tr.policy.flat.clone()
    hist = tr.train()
    assert len(hist) == 2 and {"loss", "grad_norm", "learning_rate", "reward", "reward_std", "kl", "completion_length", "rewards/accuracy_reward", "rewards/consistency_reward"} <= set(hist[-1])
    assert all(np.isfinite(h["grad_norm"]) and h["grad_norm"] > 0 for h in hist)
    assert not torch.equal(before, tr.policy.flat)            # parameters moved
    assert torch.equal(tr.ref.flat, before)                   # the frozen reference did not
    with pytest.raises(ValueError, match="does not support returning outputs"):
        tr.compute_loss(None, rows[:1], return_outputs=True)
    tr.save_model("/tmp/iadr1_trainer_test/final")
    from safetensors import safe_open
    with safe_open("/tmp/iadr1_trainer_test/final/model.safetensors", framework="pt") as sf:
        assert "model.layers.0.self_attn.q_proj.weight" in sf.keys() and "visual.blocks.0.attn.qkv.weight" in sf.keys()


def test_trainer_prefetch_is_bit_identical_to_inline_preparation():
    """The host input pipeline (iadr1_amd.prefetch: micro-batch k+1 prepared on a worker thread -- the REAL offline Qwen2-VL processor on uint8 images here --
    and uploaded through pinned memory on a copy stream while the GPU runs k) against the reference's form, the processor call inside compute_loss
    (REF:600-625): same sampler order, same batches -> every logged metric and the trained parameters are bit-identical."""
    from iadr1_amd import rewards
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer
    proc = fx.local_qwen2vl_processor(max_pixels=480000, min_pixels=3136)
    cfg_d = dict(fx.TINY, image_token_id=5, vision_start_token_id=3, vision_end_token_id=4, eos_token_id=2, pad_token_id=0)      # the local tokenizer's ids
    cfg = VLMConfig.from_dict(cfg_d)
    img = lambda: {"type": "image"}
    rows = [{"prompt": [{"role": "user", "content": [img(), {"type": "text", "text": f"Is there any defect {i}?"}]}], "image": [fx.synth_pil_image(224 + 28 * (i % 3), 196, 10 + i)],
             "solution": "<think>gt</think><location>top left</location><type>scratch</type><answer>yes</answer>"} for i in range(6)]
    out = {}
    for prefetch in (False, True):
        cfgT = GRPOConfig(output_dir="/tmp/iadr1_prefetch_test", num_generations=4, max_completion_length=6, max_prompt_length=4096, learning_rate=1e-3,
                          per_device_train_batch_size=1, gradient_accumulation_steps=2, max_steps=3, logging_steps=1, save_steps=0, prefetch_batches=prefetch)
        tr = SCGRPOTrainer((cfg, fx.make_weights(cfg_d, 0)), [rewards.accuracy_reward, rewards.consistency_reward], args=cfgT, train_dataset=rows, processing_class=proc)
        hist = tr.train()
        assert tr.prefetch_used == prefetch and tr._prefetcher is None      # train() stops its worker on the way out
        out[prefetch] = ([{k: v for k, v in h.items() if k != "elapsed_s"} for h in hist], tr.policy.flat.clone())
    assert out[False][0] == out[True][0], (out[False][0], out[True][0])
    assert torch.equal(out[False][1], out[True][1])


def _ddp_worker(rank, world, port, q, hook, wire="fp32"):
    import os as _os
    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), IADR1_REDUCE_DTYPE=wire)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # 2 ranks share the one GPU of the test box: gloo moves the CUDA buffers
    pol, ref = store(fx.make_weights(fx.TINY, 0), True), store(fx.make_weights(fx.TINY, 0), False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=6, learning_rate=1e-3, micro_batch_seqs=2), group=dist.group.WORLD)
    if not hook:
        eng.reducer.layer_ready = lambda i: None   # everything is exchanged in finish(): the buffer still holds LOCAL gradients after backward
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 50 + rank)], fx.TINY["pad_token_id"])     # each rank its own prompt
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=50 + rank), "image_grid_thw": [grid]}
    comps = fx.synth_completions(4, 6, fx.TINY, 70 + rank)
    rew = np.array([[1.0, 0.0], [0.5, 1.0], [2.0, 1.0], [0.0, 0.0]], dtype=np.float32) * (1 + rank)
    eng.loss_and_grads(batch, comps, rew, last_micro_step=True)
    local = pol.grad.clone()
    eng.optimizer_step()
    q.put((rank, local.cpu().numpy(), pol.flat.float().cpu().numpy()))
    dist.destroy_process_group()


def _run_ddp(hook, wire="fp32"):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, hook, wire)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_ddp_two_ranks_match_single_process_average():
    """world_size 2 (gloo, both ranks on the test GPU).  (1) With the gradient exchange done in one go after backward,
    both ranks end bit-identical and equal to a single process that averages the two local gradients itself.
    (2) With per-layer buckets launched from the backward hook (overlap path) the ranks are again bit-identical and
    land on the same parameters."""
    (_, g0, w0), (_, g1, w1) = _run_ddp(hook=False)
    assert np.array_equal(w0, w1)                                   # replicas stay bit-identical
    pol, ref = store(fx.make_weights(fx.TINY, 0), True), store(fx.make_weights(fx.TINY, 0), False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=6, learning_rate=1e-3))
    pol.grad.copy_(torch.from_numpy(g0 + g1).to(DEV))               # same fp32 sum the all-reduce forms
    eng.accum = 2                                                   # scale 1/2 == 1/world
    eng.optimizer_step()
    assert np.array_equal(pol.flat.float().cpu().numpy(), w0)
    (_, _, h0), (_, _, h1) = _run_ddp(hook=True)
    assert np.array_equal(h0, h1)
    # round 5: no float atomics are left in the backward path (ordered two-stage reductions), so a rank's local gradients are the same bits in every run and the
    # bucketed exchange from the backward hook lands on EXACTLY the parameters of the one-shot exchange (it used to be allowed 4e-3 of atomics-order noise)
    assert np.array_equal(h0, w0)
    # the default wire type (bf16 buckets, what DeepSpeed ZeRO-3 moves for a bf16 model): replicas still bit-identical, same update up to the bf16 rounding of the
    # exchanged gradients (Adam's first step is ~ lr * sign(g): 1e-3 per flipped sign)
    (_, _, b0), (_, _, b1) = _run_ddp(hook=True, wire="bf16")
    assert np.array_equal(b0, b1)
    assert np.abs(b0 - w0).max() <= 4e-3


def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2 --steps 1 --model tiny`: the N > 1 path of the bench END TO END on this one-GPU box -- respawn under torch.distributed.run, one
    process per rank, WORLD_SIZE asserted, sharded prompts, the bucketed gradient exchange launched from the backward hook, barrier + max-over-ranks timing, ONE JSON
    line from rank 0 with `grad_exchange.{bytes_on_wire, n_buckets, exposed_ms}`.  Both ranks share cuda:0 and the exchange runs over gloo (RCCL refuses two ranks
    on one device: IADR1_BENCH_SHARE_GPU / IADR1_BENCH_BACKEND, test-only switches); the RCCL leg itself runs in test_rccl_exchange_path_on_one_rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(IADR1_BENCH_SHARE_GPU="1", IADR1_BENCH_BACKEND="gloo", IADR1_QUIET="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--model", "tiny", "--prompts", "2", "--group", "4",
                        "--prompt-len", "300", "--gen-len", "8", "--no-cpu-baseline", "--no-repeated-rows-leg"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["rccl_ranks"] == 2 and len(rec["per_rank_ms_per_step"]) == 2
    st = rec["per_rank_step_structure"]                                  # one step structure for the whole job (overlap.agree_across_ranks, Engine.check_ddp_headroom)
    assert len(st) == 2 and st[0]["co_scheduled"] == st[1]["co_scheduled"] and st[0]["recompute_forced"] == st[1]["recompute_forced"], st
    assert rec["config"]["gradient_checkpointing"] == "auto"            # the data-parallel default (static budget, Engine.recompute_wanted)
    ge = rec["config"]["grad_exchange"]
    assert ge["n_buckets"] >= 1 and ge["bytes_on_wire"] > 0 and ge["exposed_ms"] is not None and ge["exposed_ms"] >= 0.0
    assert abs(rec["value"] - 2 * 8 * 1 / (rec["ms_per_step"] * 1e-3)) < 1e-6 * rec["value"]       # whole-job samples / max-over-ranks time


def _rccl_one_rank_worker(port, q, algo="all_reduce"):
    import os as _os
    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", IADR1_FORCE_REDUCE="1", IADR1_REDUCE_ALGO=algo)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    pol, ref = store(fx.make_weights(fx.TINY, 0), True), store(fx.make_weights(fx.TINY, 0), False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=6, learning_rate=1e-3, micro_batch_seqs=2))
    assert eng.reducer.active and eng.reducer.stream is not None and eng.reducer.algo == algo
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 50)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=50), "image_grid_thw": [grid]}
    comps = fx.synth_completions(4, 6, fx.TINY, 70)
    rew = np.array([[1.0, 0.0], [0.5, 1.0], [2.0, 1.0], [0.0, 0.0]], dtype=np.float32)
    eng.loss_and_grads(batch, comps, rew, last_micro_step=True)
    eng.optimizer_step()
    torch.cuda.synchronize()
    q.put(pol.flat.float().cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["all_reduce", "rs_ag"])
def test_rccl_exchange_path_on_one_rank(algo):
    """The RCCL leg of the data-parallel exchange (per-layer buckets from the backward hook on a side stream, remainder in finish()) run in a
    one-rank group on the test GPU: it must leave the step where a step without any process group leaves it.  algo: one all-reduce per bucket, or its two
    halves issued explicitly (IADR1_REDUCE_ALGO=rs_ag: in-place reduce_scatter_tensor + all_gather_into_tensor on the padded bucket)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_rccl_one_rank_worker, args=(port, q, algo))
    pr.start()
    w_rccl = q.get(timeout=300)
    pr.join(timeout=60)
    assert pr.exitcode == 0
    pol, ref = store(fx.make_weights(fx.TINY, 0), True), store(fx.make_weights(fx.TINY, 0), False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=6, learning_rate=1e-3, micro_batch_seqs=2))
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 50)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=50), "image_grid_thw": [grid]}
    eng.loss_and_grads(batch, fx.synth_completions(4, 6, fx.TINY, 70), np.array([[1.0, 0.0], [0.5, 1.0], [2.0, 1.0], [0.0, 0.0]], dtype=np.float32), last_micro_step=True)
    eng.optimizer_step()
    # bf16 wire: the exchanged gradients are rounded to bf16 on the way (Adam's first step is ~ lr * sign(g)); the local gradients themselves are bit-reproducible
    assert np.abs(pol.flat.float().cpu().numpy() - w_rccl).max() <= 4e-3


def _rccl_two_rank_worker(rank, port, q, algo, wire):
    import os as _os
    _os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", IADR1_REDUCE_ALGO=algo, IADR1_REDUCE_DTYPE=wire, HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = f"cuda:{rank}"
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", rank))
    mk = lambda tr: (lambda st: (st.load_named(fx.make_weights(fx.TINY, 0)), st)[1])(ParamStore(CFG, dev, trainable=tr))
    pol, ref = mk(True), mk(False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=6, learning_rate=1e-3, micro_batch_seqs=2), group=dist.group.WORLD)
    assert eng.reducer.active and eng.reducer.world == 2 and eng.reducer.algo == algo
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 50 + rank)], fx.TINY["pad_token_id"])     # each rank its own prompt
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=50 + rank), "image_grid_thw": [grid]}
    rew = np.array([[1.0, 0.0], [0.5, 1.0], [2.0, 1.0], [0.0, 0.0]], dtype=np.float32) * (1 + rank)
    eng.loss_and_grads(batch, fx.synth_completions(4, 6, fx.TINY, 70 + rank), rew, last_micro_step=True)
    local = pol.grad.clone()
    eng.optimizer_step()
    torch.cuda.synchronize()
    q.put((rank, local.cpu().numpy(), pol.flat.float().cpu().numpy(), eng.reducer.exposed_ms(), eng.reducer.last_bytes_on_wire, eng.reducer.last_n_buckets))
    dist.destroy_process_group()


@pytest.mark.parametrize("algo,wire", [("all_reduce", "bf16"), ("rs_ag", "bf16"), ("rs_ag", "fp32")])
def test_rccl_two_ranks_on_two_gpus(algo, wire):
    """First contact with a REAL multi-rank RCCL group (VERDICT r4 #6a): two processes, one per GPU, the bucketed gradient exchange over RCCL from the backward hook
    on the side stream, both collectives.  Skipped where the box has one GPU (this pool's test boxes); the first multi-GPU node that runs `pytest -m gpu` runs it.
    Checked: the replicas' parameters are bit-identical after the step; the update equals a single process that sums the two local gradients itself (fp32 wire: bit
    for bit; bf16 wire: to the bf16 rounding of the exchanged gradients); `exposed_ms` is reported and the wire volume is the model's gradient bytes."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_two_rank_worker, args=(r, port, q, algo, wire)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, w0, ex0, nb0, k0), (_, g1, w1, ex1, nb1, k1) = res
    assert np.array_equal(w0, w1)                                   # replicas stay bit-identical
    assert ex0 is not None and ex0 >= 0.0 and nb0 == nb1 and k0 == k1 >= 1
    assert nb0 >= g0.size * (2 if wire == "bf16" else 4)             # every gradient element crossed the wire once (+ shard padding for rs_ag)
    pol, ref = store(fx.make_weights(fx.TINY, 0), True), store(fx.make_weights(fx.TINY, 0), False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=6, learning_rate=1e-3))
    pol.grad.copy_(torch.from_numpy(g0 + g1).to(DEV))
    eng.accum = 2
    eng.optimizer_step()
    ref_w = pol.flat.float().cpu().numpy()
    if wire == "fp32":
        assert np.array_equal(ref_w, w0)
    else:
        assert np.abs(ref_w - w0).max() <= 4e-3                      # Adam's first step is ~lr * sign(g): a bf16-rounded gradient flips a sign only at |g| ~ 0


def test_7b_like_config_forward_and_rollout(golden_dir):
    """Untied lm_head, GQA group 7 (Qwen2.5-VL-7B structure): log-probs vs the reference golden, and the decode path
    (skinny GEMMs with N=768 / group-7 paged attention) agrees with the training-kernel forward on greedy tokens."""
    g = load(golden_dir, "logps_7b_like.npz")
    cfg7 = VLMConfig.from_dict(fx.TINY7)
    w = fx.make_weights(fx.TINY7, 0)
    pol = ParamStore(cfg7, DEV, trainable=True)
    pol.load_named(w)
    e = Engine(pol)
    ids, mask = g["input_ids"], g["attention_mask"]
    grids = [tuple(int(z) for z in r) for r in g["image_grid_thw"]]
    px = fx.synth_pixel_values(grids, fx.TINY7, seed=91)
    img, _ = e.vision_forward(torch.from_numpy(px).to(DEV), e.vision_plan(grids), save=False)
    rows = np.cumsum([0] + [t * h * w_ // 4 for t, h, w_ in grids])
    plan = e.text_plan(ids, mask, [[gr] for gr in grids], [[int(r)] for r in rows[:-1]])
    hf, _ = e.text_forward(plan, img, save=False)
    B, S = ids.shape
    sel = np.arange(B * S).reshape(B, S)[:, :-1].reshape(-1)
    lp, _ = e.logprobs(hf, torch.from_numpy(sel).to(DEV), torch.from_numpy(ids[:, 1:].reshape(-1)).to(DEV), save=False)
    valid = (mask[:, 1:] * mask[:, :-1]).astype(bool)
    # bf16 storage vs the fp32 reference; wider hidden (896) and logit scale ~ +-10 here: |dlogp| < 0.15 (1.5 % of the range)
    assert np.abs(lp.cpu().numpy().reshape(B, S - 1)[valid] - g["per_token_logps"][valid]).max() < 0.15
    # greedy rollout (graph) == teacher-forced argmax of the training-kernel forward on the same weights
    ref = ParamStore(cfg7, DEV, trainable=False)
    ref.load_named(w)
    eng = SCGRPOEngine(cfg7, pol, ref, GRPOArgs(num_generations=2, max_prompt_length=4096, max_completion_length=6))
    P = S - 5
    batch = {"input_ids": ids[:, :P], "attention_mask": mask[:, :P], "pixel_values": px, "image_grid_thw": grids}
    toks = eng.rollout(batch, greedy=True)
    full = np.concatenate([np.repeat(ids[:, :P], 2, 0), toks], 1)
    fmask = np.concatenate([np.repeat(mask[:, :P], 2, 0), np.ones_like(toks)], 1)
    plan2 = e.text_plan(full, fmask, [[gr] for gr in grids for _ in range(2)], [[int(r)] for r in rows[:-1] for _ in range(2)])
    hf2, _ = e.text_forward(plan2, img, save=False)
    S2 = full.shape[1]
    rsel = (np.arange(4)[:, None] * S2 + np.arange(P - 1, S2 - 1)[None, :]).reshape(-1)
    lg = e.logits_rows(hf2, torch.from_numpy(rsel).to(DEV)).float().cpu().numpy().reshape(4, 6, -1)
    top2 = np.sort(lg, -1)[..., -2:]
    agree = (lg.argmax(-1) == toks) | ((top2[..., 1] - top2[..., 0]) < 0.05)   # decode and prefill kernels may split a near-tie
    assert agree.all()


def test_qwen2vl_variant_matches_reference_golden(golden_dir):
    """Qwen2-VL structure (BASELINE config 1: PA-SFT on 4 samples): LayerNorm / QuickGELU ViT, no windows.  Image embeds,
    logps and the 3-step AdamW loss curve vs a tiny HF Qwen2VLForConditionalGeneration (tests/golden/qwen2vl_sft.npz)."""
    g = load(golden_dir, "qwen2vl_sft.npz")
    meta = json.loads(str(g["meta"]))
    cfg = VLMConfig.from_dict(fx.TINY_Q2)
    w = fx.make_weights(fx.TINY_Q2, 0)
    p = ParamStore(cfg, DEV, trainable=True)
    p.load_named(w)
    back = p.export_named()
    for k, v in w.items():
        assert np.array_equal(back[k].numpy(), v.reshape(back[k].shape)), k
    e = Engine(p)
    grids = [tuple(int(z) for z in r) for r in g["image_grid_thw"]]
    img, _ = e.vision_forward(torch.from_numpy(g["pixel_values"]).to(DEV), e.vision_plan(grids), save=False)
    assert relerr(img.float().cpu().numpy(), g["image_embeds"]) < 3e-2          # bf16 ViT vs fp32 reference
    ids, mask = g["input_ids"], g["attention_mask"]
    rows = np.cumsum([0] + [t * h * w_ // 4 for t, h, w_ in grids])
    plan = e.text_plan(ids, mask, [[gr] for gr in grids], [[int(r)] for r in rows[:-1]])
    hf, _ = e.text_forward(plan, img, save=False)
    B, S = ids.shape
    valid = (mask[:, 1:] * mask[:, :-1]).astype(bool)
    rr = (np.arange(B)[:, None] * S + np.arange(S - 1)[None, :])[valid]
    lp, _ = e.logprobs(hf, torch.from_numpy(rr).to(DEV), torch.from_numpy(ids[:, 1:][valid].astype(np.int64)).to(DEV), save=False)
    err = np.abs(lp.cpu().numpy() - g["per_token_logps"][valid]).max()
    assert err < 0.08, err                                                         # bf16 logits, |logp| ~ 6.5
    eng = SFTEngine(cfg, p, SFTArgs(learning_rate=meta["lr"], weight_decay=meta["wd"], max_grad_norm=0.0))
    batch = {k: g[k] for k in ("input_ids", "attention_mask", "labels", "pixel_values")}
    batch["image_grid_thw"] = grids
    losses = []
    for _ in range(3):
        losses.append(eng.loss_and_grads(batch))
        eng.optimizer_step()
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-2, atol=2e-2)


def test_pa_sft_default_freezes_vision_tower_and_projector_like_the_reference(golden_dir):
    """The reference's PA-SFT of a registered composite family (Qwen2-VL = BASELINE config 1) runs with LLaMA-Factory's defaults: vision tower and projector
    frozen (tests/golden/sft_freeze.json from its own get_forbidden_modules).  3-step AdamW curve, gradient norms and selected tensors after the steps vs a tiny HF
    Qwen2VLForConditionalGeneration with exactly that trainable set (tests/golden/qwen2vl_sft_frozen.npz); frozen tensors must come back bit-identical."""
    from iadr1_amd.sft import frozen_parameter_rule
    g0, g = load(golden_dir, "qwen2vl_sft.npz"), load(golden_dir, "qwen2vl_sft_frozen.npz")
    meta = json.loads(str(g["meta"]))
    cfg = VLMConfig.from_dict(fx.TINY_Q2)
    w = fx.make_weights(fx.TINY_Q2, 0)
    p = ParamStore(cfg, DEV, trainable=True)
    p.load_named(w)
    rule = frozen_parameter_rule("qwen2_vl")
    eng = SFTEngine(cfg, p, SFTArgs(learning_rate=meta["lr"], weight_decay=meta["wd"], max_grad_norm=0.0, frozen=rule))
    assert eng.vision_grads == "none" and all(rule(n) == n.startswith("visual.") for n in p.slots)
    batch = {k: g0[k] for k in ("input_ids", "attention_mask", "labels", "pixel_values")}
    batch["image_grid_thw"] = [tuple(int(z) for z in r) for r in g0["image_grid_thw"]]
    losses, norms = [], []
    for _ in range(3):
        losses.append(eng.loss_and_grads(batch))
        eng.optimizer_step()
        norms.append(eng.grad_norm())
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=8e-2)
    back = p.export_named()
    for k, v in w.items():
        moved = not np.array_equal(back[k].numpy(), v.reshape(back[k].shape))
        assert moved == (not k.startswith("visual.")), k                        # frozen: untouched, not even by weight decay; everything else moved
    for k in (f for f in g.files if f.startswith("after::")):
        want = g[k]
        got = back[k[7:]].numpy().reshape(want.shape)
        # three Adam steps move an element by <= 3 lr; where the bf16 gradient of a near-zero element has the other sign than the fp32 one, twice that apart
        # (+ the bf16 rounding of the exported parameter)
        assert np.abs(got - want).max() <= 2 * 3 * meta["lr"] + np.abs(want).max() * 2.0 ** -8, k
    # both unfrozen: the whole model trains and this is the unfrozen golden's curve
    p2 = ParamStore(cfg, DEV, trainable=True)
    p2.load_named(w)
    assert frozen_parameter_rule("qwen2_vl", False, False) is None and frozen_parameter_rule("qwen2_5_vl") is None and frozen_parameter_rule("llava_onevision") is None
    # projector trained, tower frozen: the vision backward runs, the tower's gradients are discarded
    eng3 = SFTEngine(cfg, p2, SFTArgs(learning_rate=meta["lr"], weight_decay=meta["wd"], max_grad_norm=1.0, frozen=frozen_parameter_rule("qwen2_vl", True, False)))
    assert eng3.vision_grads == "all"
    eng3.loss_and_grads(batch)
    eng3.optimizer_step()
    back3 = p2.export_named()
    for k, v in w.items():
        moved = not np.array_equal(back3[k].numpy(), v.reshape(back3[k].shape))
        fz = k.startswith("visual.patch_embed") or k.startswith("visual.blocks.")
        assert not (fz and moved), k
        assert moved or fz or v.ndim < 2, k          # (one step of 1e-3 on a gain near 1 stays inside its bf16 rounding interval: matrices must move)


def test_pa_sft_20_step_loss_curves_match_the_reference(golden_dir):
    """North star: "loss curve matching reference".  20 AdamW steps at lr 5e-5 (the scripts use 1e-5 / 2e-5), weight decay 0.1, HF's parameter groups, against the
    tiny HF models' own curves (tests/golden/sft.npz: Qwen2.5-VL, everything trained; qwen2vl_sft_frozen.npz: Qwen2-VL = BASELINE config 1 with the reference's
    trainable set), TF:loss/loss_utils.py:32-71 + llamafactory/train/sft/trainer.py:92-107.
    The curve compared is `losses20_bf16w`: the HF model stepped the way the reference's `--bf16` run steps it -- bf16 parameters in the forward, fp32 master copy
    under AdamW (tools/make_golden.py::curve_bf16_weights).  At these learning rates an Adam step is smaller than half a bf16 spacing of a weight, so the bf16 copy
    moves in stair steps and the curve differs from the pure-fp32 one (`losses20`, printed for reference) by up to 0.5 -- a property of the precision the reference
    trains in, reproduced here to a few 1e-3.  Bound: 1e-2 max / 4e-3 mean ABSOLUTE on a loss that runs 6.7 -> 1.1 / 2.5 (measured 5.5e-3 / 2.3e-3: bf16
    activations -- per-token log-prob noise ~1e-2 averaged over the 16-32 label tokens -- and the step at which a weight crosses a rounding boundary); the fp32
    oracle holds both golden curves to 1e-3 (tests/test_oracle_model.py)."""
    from iadr1_amd.sft import frozen_parameter_rule
    for name, cfg_d, rule, batch_file in (("sft.npz", fx.TINY, None, "sft.npz"), ("qwen2vl_sft_frozen.npz", fx.TINY_Q2, frozen_parameter_rule("qwen2_vl"), "qwen2vl_sft.npz")):
        g, g0 = load(golden_dir, name), load(golden_dir, batch_file)
        meta = json.loads(str(g["meta"]))
        cfg = VLMConfig.from_dict(cfg_d)
        p = ParamStore(cfg, DEV, trainable=True)
        p.load_named(fx.make_weights(cfg_d, 0))
        eng = SFTEngine(cfg, p, SFTArgs(learning_rate=meta["lr20"], weight_decay=meta["wd"], max_grad_norm=0.0, frozen=rule))
        batch = {k: g0[k] for k in ("input_ids", "attention_mask", "labels", "pixel_values")}
        batch["image_grid_thw"] = [tuple(int(z) for z in r) for r in g0["image_grid_thw"]]
        losses = []
        for _ in range(len(g["losses20_bf16w"])):
            losses.append(eng.loss_and_grads(batch))
            eng.optimizer_step()
        d, d32 = np.abs(np.array(losses) - g["losses20_bf16w"]), np.abs(np.array(losses) - g["losses20"])
        print(f"[sft curve] {name}: max |dloss| over 20 steps vs the bf16-weight reference curve = {d.max():.2e} at step {int(d.argmax())} (mean {d.mean():.2e}); "
              f"vs the pure-fp32 curve {d32.max():.2e}; loss {g['losses20_bf16w'][0]:.3f} -> {g['losses20_bf16w'][-1]:.3f}, hip last {losses[-1]:.4f}")
        # measured on MI355X: 5.5e-3 max / 2.3e-3 mean (Qwen2.5-VL), in the steep part of the curve -- which step a master weight crosses a bf16 rounding boundary
        # differs between two bf16 implementations by a step or so; bound: 1e-2 max, 4e-3 mean
        assert len(losses) == 20 and d.max() < 1e-2 and d.mean() < 4e-3, (name, d.max(), d.mean(), losses)


def test_eval_harness_greedy_generator_matches_hf_generate(golden_dir):
    """iadr1_amd.evaluate.GreedyGenerator (the eval scripts' decoding path, G = 1) reproduces HF `generate(do_sample=False)` token ids."""
    from iadr1_amd.evaluate import GreedyGenerator
    g = load(golden_dir, "greedy.npz")
    meta = json.loads(str(g["meta"]))
    grids = [tuple(x) for x in meta["grids"]]
    frozen = ParamStore(CFG, DEV, trainable=False, with_decode_pack=True)
    frozen.load_named(fx.make_weights(fx.TINY, 0))
    gen = GreedyGenerator(CFG, frozen, max_new_tokens=meta["new_tokens"])
    gen.cfg.eos_token_id, keep = -1, gen.cfg.eos_token_id       # the golden was generated with eos disabled (min_new_tokens)
    try:
        toks = gen.generate({"input_ids": g["prompt_ids"], "attention_mask": g["prompt_mask"], "pixel_values": fx.synth_pixel_values(grids, fx.TINY, seed=meta["seed"]), "image_grid_thw": grids})
    finally:
        gen.cfg.eos_token_id = keep
    P = g["prompt_ids"].shape[1]
    assert toks.tolist() == g["sequences"][:, P:].tolist()


@pytest.mark.parametrize("model", ["3b", "7b"])
def test_full_width_3b_shapes_shared_prefix_and_rollout_properties(model):
    """BASELINE-size widths (Qwen2.5-VL-3B hidden 2048 / 16:2 heads / MLP 11008 / vocab 151936 / ViT 1280, 448x448 image, P = 512; "7b": Qwen2.5-VL-7B --
    BASELINE config 4 -- hidden 3584 / 28:4 heads / MLP 18944 / untied 152064-token head, whose decode step runs the one-shot skinny kernels) on a
    depth-reduced model (2 decoder layers, 2 ViT blocks), random-init weights -- size-independent properties of the SC-GRPO step:
    (1) policy == reference  =>  KL is exactly 0 and the two log-prob tensors are bit-identical (same kernels, same inputs);
    (2) the shared-prefix layout and the reference's repeated-row layout give the same log-probs and gradients;
    (3) the hipGraph rollout and the eager rollout produce the same token ids (greedy)."""
    import dataclasses
    sys_path_bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import sys
    sys.path.insert(0, sys_path_bench)
    import bench
    cfg = dataclasses.replace(VLMConfig.qwen25vl_3b() if model == "3b" else VLMConfig.qwen25vl_7b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    pol = ParamStore(cfg, DEV, trainable=True)
    pol.init_random(seed=0)
    ref = ParamStore(cfg, DEV, trainable=False)
    ref.copy_from(pol)
    G, C, Bp = 8, 48, 2
    batch = bench.synth_batch(cfg, Bp, 512, seed=5)
    res = {}
    for share in (True, False):
        pol.grad.zero_()
        eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=512, max_completion_length=C, micro_batch_seqs=Bp * G, share_prefix=share, suppress_eos=True))
        if share:
            toks = {g: eng.rollout(batch, greedy=True) for g in (True,)}[True]
            eng2 = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=512, max_completion_length=C, use_hip_graph=False, suppress_eos=True))
            assert np.array_equal(toks, eng2.rollout(batch, greedy=True))                         # (3)
            assert toks.shape == (Bp * G, C) and (toks >= 0).all() and (toks < cfg.vocab_size).all()
            # decode kernels (persistent / fused skinny GEMMs, paged attention) vs training kernels on the same weights: the logits the last
            # decode step produced == teacher-forced logits of [prompt | tokens] at that position
            dec = eng._rollout.logits.float().cpu().numpy()
            ids_p, mask_p = np.asarray(batch["input_ids"]), np.asarray(batch["attention_mask"])
            full = np.concatenate([np.repeat(ids_p, G, 0), toks[:, : C - 1]], 1)
            fmask = np.ones_like(full)
            e_pol = eng.pol
            vis = eng.vision_policy(batch, save=False)
            gpr, off = eng._per_row_images(batch, vis["grids"], vis["rows"])
            plan = e_pol.text_plan(full, fmask, [gpr[r // G] for r in range(Bp * G)], [off[r // G] for r in range(Bp * G)])
            hf, _ = e_pol.text_forward(plan, vis["img"], save=False)
            S2 = full.shape[1]
            tf = e_pol.logits_rows(hf, torch.arange(Bp * G, device=DEV) * S2 + (S2 - 1)).float().cpu().numpy()
            assert np.abs(dec - tf).max() < 0.03 * np.abs(tf).max(), (np.abs(dec - tf).max(), np.abs(tf).max())
        # distinct completions for the gradient comparison: the G greedy completions of a prompt are identical, and with zero-sum group
        # advantages their exact gradient is 0 (both layouts then return rounding noise only)
        comp = np.random.RandomState(3).randint(1000, 100000, (Bp * G, C))
        rewards = np.random.RandomState(1).rand(Bp * G, 2).astype(np.float32)
        out = eng.loss_and_grads(batch, comp, rewards)
        res[share] = (out, pol.grad.clone())
    (o1, g1), (o0, g0) = res[True], res[False]
    for o in (o1, o0):
        assert torch.equal(o["logps"], o["ref_logps"]) and float(o["kl"].abs().max()) == 0.0        # (1)
    assert float((o1["logps"] - o0["logps"]).abs().max()) < 0.05                                      # (2) bf16 hidden states, different tile orders
    a, b = g1.double(), g0.double()
    assert float((a @ b) / (a.norm() * b.norm())) > 0.999
    assert torch.isfinite(g1).all() and torch.isfinite(g0).all()


def test_full_depth_3b_sc_grpo_step_vs_oracle():
    """The UNREDUCED Qwen2.5-VL-3B (36 decoder layers, 32 ViT blocks, 151 936-token vocabulary; BASELINE configs 2 / 3) against the fp32 CPU oracle on
    the same weights: what bf16 storage costs over the real depth (every other oracle comparison runs <= 4 layers).  1 prompt x G = 4, one 8 x 8-patch
    image (16 image tokens) + 64 text tokens, C = 48, policy = reference x (1 + 2 % element-wise noise), EOS inside one completion.  Checked: per-token
    log-probs of both models, KL and loss relative, gradients of five named tensors (two of them at the bottom of the decoder stack / in the ViT).
    REF sc_grpo_trainer.py:116-137 (model), :384-514 (log-probs), :746-798 (loss).
    The yardstick for the log-probs is the reference's OWN precision: the same oracle run in bf16 (what `--bf16` makes the reference compute, torch CPU kernels)
    against the fp32 oracle.  The HIP path must not be further from fp32 than 1.5x that (two bf16 implementations with different summation orders).
    Measured on MI355X: at 5 % noise (KL 1.14) HIP |dlogp| max 0.263 / mean 0.104, bf16 oracle max 0.247 / mean 0.124, KL within 3.0 %; at 2 % noise
    (KL 0.38) HIP 0.252 / 0.109, bf16 oracle 0.339 / 0.151, KL within 6.7 % -- bf16 storage over 36 + 32 layers costs the HIP path what it costs the
    reference; gradient cosines 0.994-0.998, gradient norms within 1.2 %."""
    import time
    from oracle import qwen25vl as oq
    from oracle import sc_grpo as og
    t0 = time.time()
    cfg = VLMConfig.qwen25vl_3b()
    d3 = {"text": {"vocab_size": 151936, "hidden_size": 2048, "intermediate_size": 11008, "num_hidden_layers": 36, "num_attention_heads": 16, "num_key_value_heads": 2,
                   "rms_norm_eps": 1e-6, "rope_theta": 1e6, "mrope_section": [16, 24, 24]},
          "vision": {"depth": 32, "hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16, "in_channels": 3, "patch_size": 14, "spatial_merge_size": 2,
                     "temporal_patch_size": 2, "window_size": 112, "out_hidden_size": 2048, "fullatt_block_indexes": [7, 15, 23, 31]},
          "image_token_id": cfg.image_token_id, "video_token_id": 151656, "vision_start_token_id": cfg.vision_start_token_id, "vision_end_token_id": cfg.vision_end_token_id,
          "eos_token_id": cfg.eos_token_id, "pad_token_id": cfg.pad_token_id, "tie_word_embeddings": True}
    pol, ref = ParamStore(cfg, DEV, trainable=True), ParamStore(cfg, DEV, trainable=False)
    ref.init_random(seed=0)
    ref.w("embed").mul_(2.0)            # logits std ~ 1.8 instead of 0.9: a distribution with real structure over the vocabulary
    ref.finalize()
    pol.flat.copy_(ref.flat)
    gen = torch.Generator(device=DEV).manual_seed(7)
    for lo in range(0, pol.flat.numel(), 1 << 28):
        v = pol.flat[lo: lo + (1 << 28)]
        v.copy_((v.float() * (1.0 + 0.02 * torch.randn(v.shape, generator=gen, device=DEV))).to(torch.bfloat16))     # zero padding / zero biases stay zero
    pol.finalize()
    G, C = 4, 48        # (round 5: 150 scored tokens instead of 78 -- the k3 estimate of ~80 tokens scattered by 2.5 - 7.3 % from build to build)
    grid = (1, 8, 8)
    rs = np.random.RandomState(5)
    row = rs.randint(1000, 150000, 3).tolist() + [cfg.vision_start_token_id] + [cfg.image_token_id] * 16 + [cfg.vision_end_token_id] + rs.randint(1000, 150000, 64).tolist()
    ids = np.array([row], dtype=np.int64)
    mask = np.ones_like(ids)
    px = rs.standard_normal((64, cfg.patch_dim)).astype(np.float32)
    comps = [rs.randint(1000, 150000, C).tolist(), rs.randint(1000, 150000, 5).tolist() + [cfg.eos_token_id], rs.randint(1000, 150000, C).tolist(), rs.randint(1000, 150000, C).tolist()]
    rew = np.array([[1.0, 0.5], [0.2, 0.0], [0.0, 1.0], [2.0, 0.5]], dtype=np.float32)
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, beta=0.04, micro_batch_seqs=G))
    out = eng.loss_and_grads({"input_ids": ids, "attention_mask": mask, "pixel_values": torch.from_numpy(px), "image_grid_thw": [grid]}, comps, rew)
    torch.cuda.synchronize()
    t1 = time.time()
    names = ["model.norm.weight", "model.layers.35.post_attention_layernorm.weight", "model.layers.0.input_layernorm.weight", "model.layers.17.self_attn.k_proj.bias",
             "visual.merger.ln_q.weight", "visual.blocks.0.norm1.weight"]
    grads = pol.export_named(source="grad")
    grads = {n: grads[n].numpy().reshape(-1).astype(np.float64) for n in names}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    o_pol = oq.Qwen25VLOracle(d3, pol.export_named(), requires_grad=set(names), copy=False)
    o_ref = oq.Qwen25VLOracle(d3, ref.export_named(), copy=False)
    t2 = time.time()
    want = og.sc_grpo_step(o_pol, o_ref, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(px), [grid], comps, torch.from_numpy(rew), G, 0.04,
                           cfg.eos_token_id, cfg.pad_token_id)
    want["loss"].backward()
    t3 = time.time()
    # the reference's precision: bf16 parameters and activations (log-softmax in fp32 as REF:509), same inputs
    ids_t = torch.cat([torch.from_numpy(ids).repeat(G, 1), og.right_pad(comps, cfg.pad_token_id)], 1)
    mask_t = torch.cat([torch.from_numpy(mask).repeat(G, 1), want["completion_mask"].long()], 1)
    with torch.no_grad():
        o16 = oq.Qwen25VLOracle(d3, {k: t.detach() for k, t in o_pol.w.items() if not (k == "lm_head.weight")}, dtype=torch.bfloat16)
        lp16 = o16.per_token_logps(ids_t, mask_t, torch.from_numpy(px).repeat(G, 1), [grid] * G)[:, ids.shape[1] - 1:].float().numpy()
        del o16
        o16 = oq.Qwen25VLOracle(d3, {k: t.detach() for k, t in o_ref.w.items() if not (k == "lm_head.weight")}, dtype=torch.bfloat16)
        lr16 = o16.per_token_logps(ids_t, mask_t, torch.from_numpy(px).repeat(G, 1), [grid] * G)[:, ids.shape[1] - 1:].float().numpy()
        del o16
    cm = want["completion_mask"].float()
    kl16 = float(og.grpo_loss(torch.from_numpy(lp16), torch.from_numpy(lr16), want["advantages"], cm, 0.04)[2])      # the k3 estimate the bf16 reference would log
    t4 = time.time()
    m = want["completion_mask"].bool().numpy()
    assert np.array_equal(out["completion_mask"], want["completion_mask"].numpy()) and int(m.sum()) == 3 * C + 6
    # error of the per-token DIFFERENCE ref - policy (what the k3 estimator reads): the two models' bf16 errors partly cancel when they are correlated
    e_p, e_r = out["logps"].cpu().numpy()[m] - want["logps"].detach().numpy()[m], out["ref_logps"].cpu().numpy()[m] - want["ref_logps"].numpy()[m]
    e_p16, e_r16 = lp16[m] - want["logps"].detach().numpy()[m], lr16[m] - want["ref_logps"].numpy()[m]
    diag = (f"err(ref - pol): HIP std {np.std(e_r - e_p):.4f} corr(e_pol, e_ref) {np.corrcoef(e_p, e_r)[0, 1]:.3f} | bf16 oracle std {np.std(e_r16 - e_p16):.4f} corr {np.corrcoef(e_p16, e_r16)[0, 1]:.3f}"
            f" | true |ref - pol| mean {np.abs(want['ref_logps'].numpy()[m] - want['logps'].detach().numpy()[m]).mean():.3f}")
    dlp = np.abs(out["logps"].cpu().numpy()[m] - want["logps"].detach().numpy()[m]).max()
    dlr = np.abs(out["ref_logps"].cpu().numpy()[m] - want["ref_logps"].numpy()[m]).max()
    d16 = np.abs(lp16[m] - want["logps"].detach().numpy()[m]).max()                 # bf16 oracle vs fp32 oracle (policy weights)
    dmean = np.abs(out["logps"].cpu().numpy()[m] - want["logps"].detach().numpy()[m]).mean()
    d16mean = np.abs(lp16[m] - want["logps"].detach().numpy()[m]).mean()
    wl, wk = float(want["loss"].detach()), float(want["metrics"]["kl"])
    mt = out["metrics"]
    dk, dl, dk16 = abs(mt["kl"] - wk), abs(mt["loss"] - wl), abs(kl16 - wk)
    cos = {}
    for n in names:
        a, b = grads[n], o_pol.w[n].grad.numpy().reshape(-1).astype(np.float64)
        cos[n] = (float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)), float(np.linalg.norm(a) / (np.linalg.norm(b) + 1e-30)))
    print(f"[full depth 3B] |dlogp|max policy={dlp:.4f} (mean {dmean:.4f}) ref={dlr:.4f}; the bf16 ORACLE vs fp32: max {d16:.4f} mean {d16mean:.4f}  logp range [{want['logps'].min().item():.2f}, {want['logps'].max().item():.2f}]  "
          f"kl hip={mt['kl']:.5e} oracle={wk:.5e} ({100 * dk / wk:.2f}%; the bf16 oracle: {kl16:.5e}, {100 * dk16 / wk:.2f}%)  loss hip={mt['loss']:.6e} oracle={wl:.6e} (d={dl:.2e})  "
          f"grad (cos, norm ratio)={ {k: (round(c, 4), round(r, 3)) for k, (c, r) in cos.items()} }  {diag}  "
          f"seconds: hip {t1 - t0:.0f}, export {t2 - t1:.0f}, oracle fp32 {t3 - t2:.0f}, bf16 {t4 - t3:.0f}", flush=True)
    assert dlp <= 1.5 * d16 + 0.02 and dlr <= 1.5 * d16 + 0.02 and dmean <= 1.5 * d16mean + 0.005, (dlp, dlr, d16, dmean, d16mean)
    tol_k = max(0.10 * wk, 1.5 * dk16)          # relative 10 %, or what the reference's own bf16 arithmetic does to the k3 estimate
    assert dk <= tol_k and dl <= 0.04 * tol_k + 2e-6 and dl < 1e-3, (mt, wk, wl, kl16)
    for n, (c, r) in cos.items():
        assert c > 0.97 and 0.85 < r < 1.15, (n, c, r)


def test_two_images_per_prompt_one_shot_template_vs_oracle():
    """The reference's 1-shot prompts carry TWO images (a normal template + the query image, REF:train/stage_rl/grpo_ad.py:92-116,
    `--single_img 0`).  Log-probs of the SC-GRPO passes (shared-prefix layout) vs the fp32 oracle on such prompts, plus greedy rollout
    agreement; two prompts with different image sizes and left padding."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import qwen25vl as oq
    T = fx.TINY
    rs = np.random.RandomState(11)

    def prompt(g1, g2, n_text):
        img = lambda g: [T["vision_start_token_id"]] + [T["image_token_id"]] * fx.n_image_tokens(g, T) + [T["vision_end_token_id"]]
        return rs.randint(3, T["vision_start_token_id"], 4).tolist() + img(g1) + rs.randint(3, T["vision_start_token_id"], 3).tolist() + img(g2) + rs.randint(3, T["vision_start_token_id"], n_text).tolist()

    grids = [(1, 8, 8), (1, 16, 12), (1, 12, 8), (1, 8, 12)]
    ids, mask = fx.left_pad([prompt(grids[0], grids[1], 5), prompt(grids[2], grids[3], 11)], T["pad_token_id"])
    pv = fx.synth_pixel_values(grids, T, seed=21)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": pv, "image_grid_thw": grids, "images_per_prompt": [2, 2]}
    G, C = 4, 7
    comps = fx.synth_completions(2 * G, C, T, 5, {3: 2})
    w = fx.make_weights(T, 0)
    pol, ref = store(fx.perturb_weights(w, 1), True), store(w, False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=2 * G))
    out = eng.loss_and_grads(batch, comps, np.random.RandomState(0).rand(2 * G, 2).astype(np.float32))
    # oracle: the reference's repeated-row layout, every row with its two images
    m = oq.Qwen25VLOracle(T, fx.perturb_weights(w, 1))
    full_ids, full_mask = torch.from_numpy(out["ids"]), torch.from_numpy(out["mask"])
    pv_rows = torch.cat([torch.from_numpy(pv[: 64 + 192])] * G + [torch.from_numpy(pv[64 + 192:])] * G)      # patches per prompt: 64+192, 96+96
    grids_rows = [grids[0], grids[1]] * G + [grids[2], grids[3]] * G
    with torch.no_grad():
        lp = m.per_token_logps(full_ids, full_mask, pv_rows, grids_rows)
    P = ids.shape[1]
    cm = out["completion_mask"].astype(bool)
    assert np.abs(out["logps"].cpu().numpy()[cm] - lp.numpy()[:, P - 1:][cm]).max() < 0.06
    toks = eng.rollout(batch, greedy=True)
    assert toks.shape[0] == 2 * G and all((toks[b * G] == toks[b * G + 1]).all() for b in range(2))


def test_prefill_reused_as_prompt_part_of_the_policy_forward():
    """Two-phase policy forward: the rollout's prefill (saved in the training arena) is the prompt region of the shared-prefix batch, the
    forward after the rollout runs the completion rows only.  Same completions, same loss, same gradients as the one-shot forward."""
    G, C = 4, 9
    grids = [(1, 16, 12), (1, 8, 8)]
    ids, mask = fx.left_pad([fx.synth_prompt(grids[0], 6, fx.TINY, 3), fx.synth_prompt(grids[1], 21, fx.TINY, 4)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values(grids, fx.TINY, seed=5), "image_grid_thw": grids}
    w = fx.make_weights(fx.TINY, 0)
    reward_fn = lambda comp: np.stack([(comp[:, 0] % 3).astype(np.float32), (comp[:, 1] % 2).astype(np.float32)], 1)
    res = []
    for reuse in (True, False):
        pol, ref = store(fx.perturb_weights(w, 1), True), store(w, False)
        eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=2 * G, reuse_prefill=reuse, suppress_eos=True))
        m = eng.step(batch, reward_fn, do_optimizer_step=False)
        res.append((m, pol.grad.clone()))
    (m1, g1), (m0, g0) = res
    assert abs(m1["loss"] - m0["loss"]) < 1e-6 and m1["reward"] == m0["reward"] and abs(m1["kl"] - m0["kl"]) < 1e-6
    assert float((g1 - g0).abs().max()) <= 1e-6 * float(g0.abs().max()) + 1e-12


def test_training_state_resume(tmp_path):
    """Save after 2 SFT steps, load into a store initialised differently: master weights, Adam moments and bf16 parameters come back bit for bit and the
    optimizer's step counter travels in trainer_state.json; 2 more steps then land where 4 straight steps land (up to the atomics-order noise of the
    gradients, see the two-rank test); a state written for another layout is refused."""
    from iadr1_amd.sft import SFTArgs, SFTEngine
    from iadr1_amd.trainer import last_checkpoint, load_training_state, save_training_state

    def batch(seed):
        grid = (1, 16, 12)
        ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, seed)], fx.TINY["pad_token_id"])
        ids, mask = np.asarray(ids)[:, ::1], np.asarray(mask)
        # right-aligned prompt of the fixture works as an SFT row: supervise the last 6 positions
        labels = np.where(np.arange(ids.shape[1])[None, :] >= ids.shape[1] - 6, ids, -100)
        return {"input_ids": ids, "attention_mask": mask, "labels": labels, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=seed), "image_grid_thw": [grid]}

    def run(eng, lo, hi):
        for st in range(lo, hi):
            eng.args.learning_rate = 1e-3 * (st + 1)
            eng.loss_and_grads(batch(100 + st))
            eng.optimizer_step()

    s4 = store(fx.make_weights(fx.TINY, 0), True)
    e4 = SFTEngine(CFG, s4, SFTArgs(learning_rate=1e-3))
    run(e4, 0, 4)
    s2 = store(fx.make_weights(fx.TINY, 0), True)
    e2 = SFTEngine(CFG, s2, SFTArgs(learning_rate=1e-3))
    run(e2, 0, 2)
    ck = str(tmp_path / "checkpoint-2")
    save_training_state(s2, ck, e2.opt_step, 2)
    assert last_checkpoint(str(tmp_path)) == ck
    s3 = store(fx.make_weights(fx.TINY, 1), True)          # different weights: everything must come from the checkpoint
    e3 = SFTEngine(CFG, s3, SFTArgs(learning_rate=1e-3))
    state = load_training_state(s3, ck)
    assert state["global_step"] == 2 and state["opt_step"] == 2
    e3.opt_step = state["opt_step"]
    for name in ("master", "m", "v", "flat", "flat_t"):
        assert torch.equal(getattr(s3, name), getattr(s2, name)), name
    run(e3, 2, 4)
    run(e2, 2, 4)                                            # the run that never stopped
    assert (s3.master - s4.master).abs().max().item() <= 4e-3 and (s2.master - s4.master).abs().max().item() <= 4e-3
    import json as _json
    st_path = os.path.join(ck, "trainer_state.json")
    bad = _json.load(open(st_path)); bad["layout"] = "0" * 40
    _json.dump(bad, open(st_path, "w"))
    with pytest.raises(ValueError, match="different parameter layout"):
        load_training_state(s3, ck)


def test_sc_grpo_on_the_qwen2vl_structure():
    """SC-GRPO on the Qwen2-VL structure (the reference's SC_GRPO_Qwen_Instruct_2_VL.sh: LayerNorm / QuickGELU ViT without windows in front of the
    same decoder): policy == reference => per-token KL exactly 0 and a loss of exactly 0 with zero-mean advantages over identical log-probs is not
    required -- what is: finite loss, gradients on both towers, graph and eager rollouts identical, and the greedy tokens are the arg-max of the
    training kernels' logits at every generated position (decode path vs training path on the same weights)."""
    cfg = VLMConfig.from_dict(fx.TINY_Q2)
    w = fx.make_weights(fx.TINY_Q2, 0)
    pol = ParamStore(cfg, DEV, trainable=True); pol.load_named(w)
    ref = ParamStore(cfg, DEV, trainable=False); ref.load_named(w)
    grids = [(1, 16, 12), (1, 8, 8)]
    rows = [fx.synth_prompt(gr, n, fx.TINY_Q2, 31 + i) for i, (gr, n) in enumerate(zip(grids, [7, 12]))]
    ids, mask = fx.left_pad(rows, fx.TINY_Q2["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values(grids, fx.TINY_Q2, seed=31), "image_grid_thw": grids}
    G, C = 4, 10
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, suppress_eos=True))
    toks = eng.rollout(batch, greedy=True)
    eager = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, suppress_eos=True, use_hip_graph=False))
    assert np.array_equal(toks, eager.rollout(batch, greedy=True)) and toks.shape == (2 * G, C)
    assert all(np.array_equal(toks[b * G], toks[b * G + g]) for b in range(2) for g in range(G))       # greedy: the G rows of a prompt agree
    # teacher-forced training forward over [prompt | greedy tokens]: its arg-max reproduces the tokens (top-2 margin permitting)
    e = eng.pol
    full = np.concatenate([np.repeat(ids, G, 0), toks], 1)
    fmask = np.concatenate([np.repeat(mask, G, 0), np.ones_like(toks)], 1)
    img, _ = e.vision_forward(torch.from_numpy(batch["pixel_values"]).to(DEV), e.vision_plan(grids), save=False)
    off = np.cumsum([0] + [t * h * w_ // 4 for t, h, w_ in grids])
    plan = e.text_plan(full, fmask, [[grids[r // G]] for r in range(2 * G)], [[int(off[r // G])] for r in range(2 * G)])
    hf, _ = e.text_forward(plan, img, save=False)
    S, P = full.shape[1], ids.shape[1]
    rr = (np.arange(2 * G)[:, None] * S + np.arange(P - 1, S - 1)[None, :]).reshape(-1)
    logits = (hf[torch.from_numpy(rr).to(DEV)].float() @ pol.w("embed").float().t() if cfg.tie_word_embeddings else hf[torch.from_numpy(rr).to(DEV)].float() @ pol.w("lm_head").float().t())
    top2 = logits.topk(2, -1)
    sure = (top2.values[:, 0] - top2.values[:, 1]) > 0.05
    assert sure.float().mean() > 0.5 and torch.equal(top2.indices[:, 0][sure].cpu(), torch.from_numpy(toks.reshape(-1))[sure.cpu()])
    # one full step with distinct sampled completions: finite loss, both towers receive gradient, KL == 0 exactly because policy == reference
    comps = fx.synth_completions(2 * G, C, fx.TINY_Q2, 5)
    rew = np.random.RandomState(0).rand(2 * G, 2).astype(np.float32)
    out = eng.loss_and_grads(batch, comps, rew)
    assert np.isfinite(out["metrics"]["loss"]) and out["metrics"]["kl"] == 0.0 and torch.equal(out["logps"], out["ref_logps"])
    assert float(pol.g("visual.blocks.0.fc1.w").abs().max()) > 0 and float(pol.g("layers.0.qkv.w").abs().max()) > 0
    eng.optimizer_step()
    assert np.isfinite(eng.grad_norm()) and eng.grad_norm() > 0


def test_decode_steps_fill_the_training_arena():
    """Rollout -> training hand-over at BASELINE widths (depth-reduced 3B shapes): with reuse_decode the decode steps write the completion rows of the
    policy's activation arena (side outputs of the decode kernels) and no policy forward runs over the completions before backward.  Against the step
    that does run that forward: same tokens and rewards, loss and KL within the bf16 noise of decode-vs-training kernels, gradient cosine > 0.999."""
    import dataclasses
    import sys
    if os.environ.get("IADR1_SKINNY_PERS", "1") == "0" or os.environ.get("IADR1_DECODE_PACKED", "1") == "0":
        pytest.skip("the side outputs live in the persistent / fused decode kernels")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    pol = ParamStore(cfg, DEV, trainable=True)
    pol.init_random(seed=0)
    ref = ParamStore(cfg, DEV, trainable=False)
    ref.copy_from(pol)
    G, C, Bp = 8, 32, 2
    batch = bench.synth_batch(cfg, Bp, 512, seed=5)
    seen = {}

    def rew(comp):
        seen.setdefault("comp", []).append(np.asarray(comp).copy())
        return np.random.RandomState(0).rand(len(comp), 2).astype(np.float32)

    res = {}
    for mode in (False, True):
        pol.grad.zero_()
        eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=512, max_completion_length=C, micro_batch_seqs=Bp * G, suppress_eos=True,
                                                   reuse_decode=mode, reuse_prefill=True, share_prefix=True, seed=3))
        m = eng.step(batch, rew, do_optimizer_step=False)
        torch.cuda.synchronize()
        assert eng.last_step_traced == mode
        res[mode] = (m, pol.grad.clone())
    assert np.array_equal(seen["comp"][0], seen["comp"][1])                       # the same rollout either way
    (m0, g0), (m1, g1) = res[False], res[True]
    assert m0["kl"] == 0.0 and 0.0 <= m1["kl"] < 1e-3                              # policy == reference: exactly 0 on the training kernels, ~1e-5 through the decode kernels
    assert abs(m0["loss"] - m1["loss"]) < 1e-4 and m0["reward"] == m1["reward"]
    assert bool(torch.isfinite(g1).all())
    cos = float((g0 @ g1) / (g0.norm() * g1.norm()))
    assert cos > 0.999 and float((g0 - g1).norm() / g0.norm()) < 0.03, cos


def _oracle_cfg_dict(cfg):
    """VLMConfig -> the nested dict form the oracle takes."""
    return {"text": {"vocab_size": cfg.vocab_size, "hidden_size": cfg.hidden_size, "intermediate_size": cfg.intermediate_size, "num_hidden_layers": cfg.num_hidden_layers,
                     "num_attention_heads": cfg.num_attention_heads, "num_key_value_heads": cfg.num_key_value_heads, "rms_norm_eps": cfg.rms_norm_eps,
                     "rope_theta": cfg.rope_theta, "mrope_section": list(cfg.mrope_section)},
            "vision": {"depth": cfg.v_depth, "hidden_size": cfg.v_hidden, "intermediate_size": cfg.v_inter, "num_heads": cfg.v_heads, "in_channels": cfg.v_in_channels,
                       "patch_size": cfg.v_patch, "spatial_merge_size": cfg.v_merge, "temporal_patch_size": cfg.v_temporal, "window_size": cfg.v_window,
                       "out_hidden_size": cfg.hidden_size, "fullatt_block_indexes": list(cfg.v_fullatt)},
            "image_token_id": cfg.image_token_id, "video_token_id": cfg.image_token_id + 1, "vision_start_token_id": cfg.vision_start_token_id,
            "vision_end_token_id": cfg.vision_end_token_id, "eos_token_id": cfg.eos_token_id, "pad_token_id": cfg.pad_token_id, "tie_word_embeddings": cfg.tie_word_embeddings}


@pytest.mark.parametrize("overlap_cus", ["0", "64"])
def test_api_step_with_rollout_handover_matches_the_oracle(monkeypatch, overlap_cus):
    """The path bench.py times and SCGRPOTrainer.compute_loss runs (SCGRPOEngine.step with the rollout's prefill as the prompt part of the policy
    forward and the decode steps filling the completion rows of the training arena) against the CPU oracle, at BASELINE widths (Qwen2.5-VL-3B
    hidden 2048 / 16:2 heads / MLP 11008 / vocabulary 151936 / ViT 1280; 2 decoder layers, 2 ViT blocks), policy != reference, EOS ENABLED with
    ragged completion lengths, two left-padded prompts of different length.  The tokens are whatever the device sampled; the oracle then runs
    the reference's arithmetic (oracle.sc_grpo.sc_grpo_step, fp32) on exactly those tokens and rewards.

    overlap_cus (VERDICT r5 #3a): "0" = the SEQUENTIAL step (reference pass after the rollout); "64" = the CO-SCHEDULED step the bench times by default -- the frozen
    reference's chunked pass, the policy's REBUILT gate|up / SwiGLU rows and its lm_head log-probs on a 64-CU side stream under the decode replays on the other CUs
    -- forced at this small shape (2 prompts x G 8 = 16 sequences: one 256-row tile per time block, C = 32 = two time blocks).  Same tolerances both ways, INCLUDING
    the gradient cosines: the backward of the co-scheduled step reads rows rebuilt by the training GEMM, and this holds them to the oracle directly.  A box without a
    clean stream pair SKIPS the co-scheduled case with that reason (a forced split raises; nothing passes on a fallback)."""
    import dataclasses
    import sys
    monkeypatch.setenv("IADR1_OVERLAP_CUS", overlap_cus)
    if os.environ.get("IADR1_SKINNY_PERS", "1") == "0" or os.environ.get("IADR1_DECODE_PACKED", "1") == "0":
        pytest.skip("the side outputs live in the persistent / fused decode kernels")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import qwen25vl as oq
    from oracle import sc_grpo as og
    cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    ref = ParamStore(cfg, DEV, trainable=False)
    ref.init_random(seed=0)
    w_ref = {k: v.float().numpy() for k, v in ref.export_named().items()}
    w_pol = fx.perturb_weights(w_ref, 1, scale=0.25)                  # a policy well away from the reference: KL ~ 0.1, so relative tolerances bind
    pol = ParamStore(cfg, DEV, trainable=True)
    pol.load_named(w_pol)
    G, C, Bp = 8, 32, 2
    cd = _oracle_cfg_dict(cfg)
    grids = [(1, 16, 16), (1, 16, 12)]
    rows = [fx.synth_prompt(grids[0], 37, cd, 5), fx.synth_prompt(grids[1], 21, cd, 6)]
    ids, mask = fx.left_pad(rows, cfg.pad_token_id)
    px = fx.synth_pixel_values(grids, cd, seed=5)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": px, "image_grid_thw": grids}
    args = lambda: GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=Bp * G, reuse_prefill=True, reuse_decode=True, share_prefix=True,
                            seed=11, beta=0.04)
    reward_fn = lambda comp: np.stack([(np.asarray(comp)[:, 0] % 5).astype(np.float32), (np.asarray(comp)[:, 1] % 3).astype(np.float32) * 0.5], 1)
    # pass 1 (EOS can practically not be sampled from 151936 tokens): find out what the sampler draws, then declare the token that ends the most rows
    # early to be EOS -- sampling is a pure function of (seed, step, row, logits), so pass 2 draws the same tokens up to each row's first EOS
    eng0 = SCGRPOEngine(cfg, pol, ref, args())
    try:
        comp0 = eng0.rollout(batch, vis=None)
    except RuntimeError as exc:
        if overlap_cus != "0" and "no stream pair" in str(exc):
            pytest.skip(f"co-scheduled case NOT RUN on this box: {exc}")
        raise
    best, best_rows = None, []
    for tok in np.unique(comp0[:, 2: C - 2]):
        hit = [r for r in range(Bp * G) if tok in comp0[r, 2: C - 2] and tok not in comp0[r, :2]]
        if len(hit) > len(best_rows):
            best, best_rows = int(tok), hit
    assert best is not None
    cfg.eos_token_id = best                     # one config object is shared by the stores, the engines and the rollout
    del eng0
    eng = SCGRPOEngine(cfg, pol, ref, args())
    out = eng.step(batch, reward_fn, do_optimizer_step=False, return_outputs=True)
    torch.cuda.synchronize()
    assert eng.last_step_traced                  # no policy forward ran over the completions: backward read what the decode steps wrote
    if overlap_cus == "0":
        assert not eng.last_step_shadowed and eng._rollout.decode_cus == 0
    else:                                        # the co-scheduled structure really ran: side-stream reference pass, rebuilt policy mlp rows, side-stream lm_head
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        assert eng.last_step_shadowed and eng._rollout.decode_cus == ncu - int(overlap_cus) and eng._rollout.trace["mlp_on_shadow"] and eng.shadow_policy_head is not None
    print(f"[parity] api step: IADR1_OVERLAP_CUS={overlap_cus} -> {'CO-SCHEDULED' if eng.last_step_shadowed else 'sequential'} step")
    comp = out["completion_ids"]
    lens = out["completion_mask"].sum(1)
    assert (lens < C).any() and (lens == C).any(), lens           # ragged: some rows stopped at EOS, some ran to the end
    for r in range(Bp * G):
        n = int(lens[r])
        assert np.array_equal(comp[r, :n], comp0[r, :n]) and (n == C or (comp[r, n - 1] == best and (comp[r, n:] == cfg.pad_token_id).all()))
    # ---- the oracle on the same tokens -------------------------------------------------------------------------------------------------
    cd = _oracle_cfg_dict(cfg)
    o_pol, o_ref = oq.Qwen25VLOracle(cd, w_pol, requires_grad=True), oq.Qwen25VLOracle(cd, w_ref)
    rew = torch.from_numpy(reward_fn(comp))
    want = og.sc_grpo_step(o_pol, o_ref, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(px), grids, [list(map(int, comp[r])) for r in range(Bp * G)], rew, G, 0.04, cfg.eos_token_id, cfg.pad_token_id, interleaved=True)
    assert np.array_equal(want["completion_mask"].numpy(), out["completion_mask"]) and np.array_equal(want["ids"].numpy(), out["ids"])
    m = out["completion_mask"].astype(bool)
    dlp = np.abs(out["logps"].cpu().numpy()[m] - want["logps"].detach().numpy()[m]).max()
    dlr = np.abs(out["ref_logps"].cpu().numpy()[m] - want["ref_logps"].numpy()[m]).max()
    np.testing.assert_allclose(out["advantages"].numpy(), want["advantages"].numpy(), rtol=1e-5, atol=1e-6)
    mt = out["metrics"]
    wl, wk = float(want["loss"].detach()), float(want["metrics"]["kl"])
    print(f"[parity] api step vs oracle: loss hip={mt['loss']:.6e} oracle={wl:.6e}  kl hip={mt['kl']:.6e} oracle={wk:.6e} ({100 * abs(mt['kl'] - wk) / wk:.1f}%)  "
          f"|dlogp|max pol={dlp:.4f} ref={dlr:.4f}  lens={lens.tolist()} eos={best}")
    assert dlp < 0.06 and dlr < 0.06, (dlp, dlr)
    assert abs(mt["kl"] - wk) <= 0.10 * wk, (mt["kl"], wk)
    assert abs(mt["loss"] - wl) <= 0.04 * 0.10 * wk + 2e-6, (mt["loss"], wl)
    assert mt["completion_length"] == want["metrics"]["completion_length"] and abs(mt["reward"] - want["metrics"]["reward"]) < 1e-6
    want["loss"].backward()
    grads = pol.export_named(source="grad")
    og_ = dict(o_pol.parameters())
    for n in ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.k_proj.bias", "model.layers.1.mlp.down_proj.weight", "model.layers.1.mlp.gate_proj.weight",
              "model.layers.1.self_attn.o_proj.weight", "model.norm.weight", "visual.blocks.1.attn.qkv.weight", "visual.blocks.0.mlp.down_proj.weight", "visual.merger.mlp.2.weight",
              "model.embed_tokens.weight"):
        a, b = grads[n].numpy().reshape(-1).astype(np.float64), og_[n].grad.numpy().reshape(-1).astype(np.float64)
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
        ratio = float(np.linalg.norm(a) / (np.linalg.norm(b) + 1e-30))
        assert cos > 0.99 and 0.9 < ratio < 1.1, (n, cos, ratio)


# ------------------------------------------------------------------------------------------------------------------------------------
# LLaVA-OneVision branch (BASELINE.json config 5)
# ------------------------------------------------------------------------------------------------------------------------------------
CFG_OV = VLMConfig.from_dict(fx.TINY_OV)


def _ov_store(weights, trainable):
    s_ = ParamStore(CFG_OV, DEV, trainable=trainable)
    s_.load_named(weights)
    return s_


@pytest.mark.parametrize("cfg_name,golden", [("TINY_OV", "llava_ov.npz"), ("TINY_OV64", "llava_ov_hd64.npz")])
def test_llava_onevision_forward_matches_hf_golden(golden_dir, cfg_name, golden):
    """(second case: 64-wide decoder heads -- LLaVA-OneVision-0.5B's Qwen2-0.5B structure -- stored zero-padded to the 128 the decoder kernels run at, rotary
    halves at [0, 32) and [64, 96): VLMConfig.head_dim)  SigLIP tower (head width 72 run zero-padded to 80), projector, any-resolution packing (one image shrunk by the bilinear interpolation) and the
    Qwen2 decoder with 1-D rotary positions on the HIP kernels vs a tiny HF LlavaOnevisionForConditionalGeneration (tests/golden/llava_ov.npz)."""
    g = load(golden_dir, golden)
    meta = json.loads(str(g["meta"]))
    cfg_d = getattr(fx, cfg_name)
    w = fx.make_weights_ov(cfg_d, 0)
    st = ParamStore(VLMConfig.from_dict(cfg_d), DEV, trainable=False)
    st.load_named(w)
    assert st.cfg.head_dim == 128 and st.cfg.head_dim_real == (64 if cfg_name == "TINY_OV64" else 128)
    back = st.export_named()
    for k, v in w.items():
        assert np.array_equal(back[k].numpy(), v.reshape(back[k].shape)), k           # checkpoint names <-> padded fused layout round trip
    e = Engine(st)
    sizes = [tuple(x) for x in meta["sizes"]]
    batch = {"input_ids": g["input_ids"], "attention_mask": g["attention_mask"], "pixel_values": fx.synth_crops(meta["crops"], cfg_d, meta["seed"]), "image_sizes": sizes}
    grids, plan_v, px, rows = e.vision_inputs(batch)
    assert plan_v.lens == g["feature_lens"].tolist()
    img, _ = e.vision_forward(px, plan_v, save=False)
    assert relerr(img.float().cpu().numpy(), g["image_features"]) < 3e-2          # bf16 tower vs fp32 reference
    ids, mask = g["input_ids"], g["attention_mask"]
    plan = e.text_plan(ids, mask, [[s_] for s_ in sizes], [[int(r)] for r in rows[:-1]])
    hf, _ = e.text_forward(plan, img, save=False)
    B, S = ids.shape
    valid = (mask[:, 1:] * mask[:, :-1]).astype(bool)
    rr = (np.arange(B)[:, None] * S + np.arange(S - 1)[None, :])[valid]
    lp, _ = e.logprobs(hf, torch.from_numpy(rr).to(DEV), torch.from_numpy(ids[:, 1:][valid].astype(np.int64)).to(DEV), save=False)
    err = np.abs(lp.cpu().numpy() - g["per_token_logps"][valid]).max()
    assert err < 0.08, err


@pytest.mark.parametrize("share,golden", [(True, "sc_grpo_llava_ov.npz"), (False, "sc_grpo_llava_ov.npz"), (True, "sc_grpo_llava_ov_g7.npz"), (False, "sc_grpo_llava_ov_g7.npz")])
def test_llava_onevision_sc_grpo_matches_reference_golden(golden_dir, share, golden):
    """The reference's compute_loss on its llava branch (tests/golden/sc_grpo_llava_ov.npz, model id containing "llava_ov"): two of the four rows end
    early, so `_ensure_left_padding_data` (REF:516-567) rotates them and their loss terms are the log-probs of EARLIER tokens; the engine reproduces
    that by index arithmetic on the host (GRPOArgs.llava_rotate_right_padded_rows), in the shared-prefix and in the repeated-rows layout.
    sc_grpo_llava_ov_g7.npz: the same through the reference on TINY_OV7 -- 7 query heads per kv head, the decoder geometry of LLaVA-OneVision-7B (BASELINE config 5):
    forward AND backward of the SC-GRPO step at GQA group 7 on the SigLIP / any-resolution path."""
    g = load(golden_dir, golden)
    meta = json.loads(str(g["meta"]))
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    cfg_d = getattr(fx, (meta.get("config") or "fixture_util.TINY_OV").split(".")[-1] or "TINY_OV")
    CFG_OV = VLMConfig.from_dict(cfg_d)

    def _ov_store(weights, trainable):
        s_ = ParamStore(CFG_OV, DEV, trainable=trainable)
        s_.load_named(weights)
        return s_
    w_ref = fx.make_weights_ov(cfg_d, 0)
    pol, ref = _ov_store(fx.perturb_weights(w_ref, 1, scale=meta["perturb_scale"]), True), _ov_store(w_ref, False)
    eng = SCGRPOEngine(CFG_OV, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, beta=0.04, micro_batch_seqs=16, share_prefix=share))
    P = g["prompt_completion_ids"].shape[1] - C
    batch = {"input_ids": g["prompt_completion_ids"][:1, :P], "attention_mask": g["attention_mask"][:1, :P], "pixel_values": fx.synth_crops(meta["crops"], cfg_d, seed),
             "image_sizes": [tuple(x) for x in meta["sizes"]]}
    comps = fx.synth_completions(G, C, cfg_d, seed + 100, {int(k): v for k, v in meta["eos_rows"].items()})
    out = eng.loss_and_grads(batch, comps, g["rewards_per_func"])
    assert np.array_equal(out["ids"], g["prompt_completion_ids"]) and np.array_equal(out["completion_mask"], g["completion_mask"])
    m = g["completion_mask"].astype(bool)
    dlp = np.abs(out["logps"].cpu().numpy()[m] - g["per_token_logps"][m]).max()
    dlr = np.abs(out["ref_logps"].cpu().numpy()[m] - g["ref_per_token_logps"][m]).max()
    gl, gk = float(g["loss"]), float(g["metric_kl"])
    mt = out["metrics"]
    print(f"[parity] llava-ov {golden} share={share}: loss hip={mt['loss']:.6e} ref={gl:.6e}  kl hip={mt['kl']:.6e} ref={gk:.6e} ({100 * abs(mt['kl'] - gk) / gk:.1f}%)  |dlogp|max={dlp:.4f}/{dlr:.4f}")
    tol_lp = 0.08 if CFG_OV.hidden_size <= 256 else 0.12           # hidden 896: log-probs down to -13 (1 % of the range, as for TINY7)
    assert dlp < tol_lp and dlr < tol_lp, (dlp, dlr)
    np.testing.assert_allclose(out["advantages"].numpy(), g["advantages"], rtol=1e-5, atol=1e-6)
    assert abs(mt["kl"] - gk) <= 0.10 * gk and abs(mt["loss"] - gl) <= 0.04 * 0.10 * gk + 2e-6
    grads = pol.export_named(source="grad")
    for n, ref_norm in zip([str(n) for n in g["grad_norm_names"]], g["grad_norms"]):
        # (a key bias shifts every score of a query by the same amount: its exact gradient is 0, the reference's value is fp32 rounding noise ~1e-9 and the bf16
        # path's ~1e-4 -- norms below 1e-6 of the largest are not compared)
        if n == "language_model.lm_head.weight" or ref_norm < 1e-6 * float(np.max(g["grad_norms"])):
            continue
        got = float(grads[n].norm())
        assert abs(got - ref_norm) <= 0.10 * ref_norm + 1e-7, (n, got, ref_norm)
    for k in g.files:
        if k.startswith("grad::"):
            a, b = grads[k[6:]].numpy().reshape(-1).astype(np.float64), g[k].reshape(-1).astype(np.float64)
            cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
            assert cos > 0.99, (k, cos)
    # without the rotation the two early-ended rows are scored on their own completion tokens: a different loss (the quirk is exercised)
    pol2 = _ov_store(fx.perturb_weights(w_ref, 1, scale=meta["perturb_scale"]), True)
    eng2 = SCGRPOEngine(CFG_OV, pol2, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, beta=0.04, micro_batch_seqs=16, share_prefix=share,
                                                    llava_rotate_right_padded_rows=False))
    out2 = eng2.loss_and_grads(batch, comps, g["rewards_per_func"], backward=False)
    assert abs(out2["metrics"]["loss"] - gl) > 1e-4


def test_llava_onevision_rollout_and_step():
    """Group rollout on the LLaVA-OneVision structure: hipGraph and eager rollouts agree, the greedy tokens are the arg-max of the training kernels'
    logits (decode path vs training path on the same weights), and one full SC-GRPO step (trainer-level engine.step) moves the parameters of both
    towers with finite loss; policy == reference gives KL exactly 0."""
    w = fx.make_weights_ov(fx.TINY_OV, 0)
    pol, ref = _ov_store(w, True), _ov_store(w, False)
    sizes = [(80, 100), (120, 100)]
    v = fx.TINY_OV["vision"]
    from iadr1_amd import llava_ov as lo
    rs = np.random.RandomState(7)
    rows, ncrops = [], 0
    for sz, nt in zip(sizes, (5, 12)):
        n_img = lo.num_image_tokens(sz, fx.TINY_OV["image_grid_pinpoints"], v["image_size"], v["image_size"] // v["patch_size"], 9)
        rows.append(rs.randint(3, 600, 3).tolist() + [CFG_OV.image_token_id] * n_img + rs.randint(3, 600, nt).tolist())
        ncrops += lo.num_crops(sz, fx.TINY_OV["image_grid_pinpoints"], v["image_size"])
    ids, mask = fx.left_pad(rows, CFG_OV.pad_token_id)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_crops(ncrops, fx.TINY_OV, 9), "image_sizes": sizes}
    G, C = 4, 8
    eng = SCGRPOEngine(CFG_OV, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, suppress_eos=True))
    toks = eng.rollout(batch, greedy=True)
    eager = SCGRPOEngine(CFG_OV, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, suppress_eos=True, use_hip_graph=False))
    assert np.array_equal(toks, eager.rollout(batch, greedy=True)) and toks.shape == (2 * G, C)
    e = eng.pol
    full = np.concatenate([np.repeat(ids, G, 0), toks], 1)
    fmask = np.concatenate([np.repeat(mask, G, 0), np.ones_like(toks)], 1)
    grids, plan_v, px, off = e.vision_inputs(batch)
    img, _ = e.vision_forward(px, plan_v, save=False)
    plan = e.text_plan(full, fmask, [[sizes[r // G]] for r in range(2 * G)], [[int(off[r // G])] for r in range(2 * G)])
    hf, _ = e.text_forward(plan, img, save=False)
    S, P = full.shape[1], ids.shape[1]
    rr = (np.arange(2 * G)[:, None] * S + np.arange(P - 1, S - 1)[None, :]).reshape(-1)
    logits = hf[torch.from_numpy(rr).to(DEV)].float() @ pol.w("lm_head").float().t()
    top2 = logits.topk(2, -1)
    sure = (top2.values[:, 0] - top2.values[:, 1]) > 0.05
    assert sure.float().mean() > 0.5 and torch.equal(top2.indices[:, 0][sure].cpu(), torch.from_numpy(toks.reshape(-1))[sure.cpu()])
    before = pol.flat.clone()
    eng2 = SCGRPOEngine(CFG_OV, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, learning_rate=1e-3, seed=3))
    mt = eng2.step(batch, lambda comp: np.stack([(np.asarray(comp)[:, 0] % 5).astype(np.float32), (np.asarray(comp)[:, 1] % 3).astype(np.float32)], 1))
    assert np.isfinite(mt["loss"]) and mt["kl"] == 0.0 and np.isfinite(eng2.grad_norm()) and eng2.grad_norm() > 0
    moved = (pol.flat != before)
    sl = pol.slots["visual.blocks.0.fc1.w"]
    assert bool(moved[sl.offset: sl.offset + 100].any()) and bool(moved[pol.slots["layers.0.qkv.w"].offset: pol.slots["layers.0.qkv.w"].offset + 100].any())
    # the zero padding of the vision heads (72 -> 80) and of the patch columns (588 -> 592) is still exactly zero after an optimizer step
    d, dp, nh = CFG_OV.v_hidden // CFG_OV.v_heads, CFG_OV.v_head_pad, CFG_OV.v_heads
    assert float(pol.w("visual.blocks.0.qkv.w").view(3 * nh, dp, -1)[:, d:].abs().max()) == 0.0 and float(pol.w("visual.patch_embed")[:, CFG_OV.patch_dim:].abs().max()) == 0.0
    assert float(pol.w("visual.blocks.1.proj.w").view(-1, nh, dp)[:, :, d:].abs().max()) == 0.0


def _ov_hf_config(d):
    """config.json of a LLaVA-OneVision checkpoint (transformers 4.51.3 layout, the reference's pin) for the tiny structure."""
    t, v = d["text"], d["vision"]
    return {"model_type": "llava_onevision", "architectures": ["LlavaOnevisionForConditionalGeneration"], "image_token_index": d["image_token_id"],
            "image_grid_pinpoints": [list(p) for p in d["image_grid_pinpoints"]], "vision_aspect_ratio": f"anyres_max_{d['anyres_max']}", "vision_feature_layer": -1,
            "vision_feature_select_strategy": "full", "tie_word_embeddings": d["tie_word_embeddings"],
            "text_config": {"model_type": "qwen2", "vocab_size": t["vocab_size"], "hidden_size": t["hidden_size"], "intermediate_size": t["intermediate_size"],
                            "num_hidden_layers": t["num_hidden_layers"], "num_attention_heads": t["num_attention_heads"], "num_key_value_heads": t["num_key_value_heads"],
                            "rms_norm_eps": t["rms_norm_eps"], "rope_theta": t["rope_theta"], "eos_token_id": d["eos_token_id"], "pad_token_id": d["pad_token_id"]},
            "vision_config": {"model_type": "siglip_vision_model", "num_hidden_layers": v["depth"], "hidden_size": v["hidden_size"], "intermediate_size": v["intermediate_size"],
                              "num_attention_heads": v["num_heads"], "num_channels": v["in_channels"], "patch_size": v["patch_size"], "image_size": v["image_size"],
                              "layer_norm_eps": v["layer_norm_eps"]}}


def test_llava_onevision_through_the_trainer_api_and_checkpoint_roundtrip(tmp_path):
    """BASELINE config 5 end to end at the reference's API: a LLaVA-OneVision checkpoint DIRECTORY (config.json + safetensors in the 4.51.3 names) is
    what `SCGRPOTrainer(model=<path>)` loads (REF:124-132), the prompt tensors come from a real transformers LlavaOnevisionProcessor (any-resolution
    crops + `<image>` expansion, tests/fixture_util.py::local_llava_ov_processor) through prepare_batch, two images of different crop counts (one shrunk
    by the interpolation) in one batch, one optimizer step, save_model, and the saved directory loads back bit-identical (bf16)."""
    from iadr1_amd import rewards
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer, load_checkpoint, save_checkpoint
    proc = fx.local_llava_ov_processor()
    proc.save_pretrained = lambda *a_, **k_: None          # (the offline stand-in for the video processor cannot be serialised; the model files are what is checked)
    d = dict(fx.TINY_OV, image_token_id=proc.tokenizer.convert_tokens_to_ids("<image>"), eos_token_id=proc.tokenizer.eos_token_id, pad_token_id=proc.tokenizer.pad_token_id)
    cfg = VLMConfig.from_dict(d)
    src = str(tmp_path / "llava-ov-tiny")
    s0 = ParamStore(cfg, DEV, trainable=False)
    w0 = fx.make_weights_ov(d, 0)
    s0.load_named(w0)
    save_checkpoint(s0, src, _ov_hf_config(d))
    import dataclasses
    cfg = dataclasses.replace(cfg, vision_start_token_id=-1, vision_end_token_id=-1)       # Qwen-VL special tokens: not part of a LLaVA-OneVision config.json
    cfg_l, s1 = load_checkpoint(src, DEV, trainable=False)
    assert cfg_l == cfg
    back = s1.export_named()
    assert set(back) == set(w0)
    for k, v_ in w0.items():
        assert np.array_equal(back[k].float().numpy().reshape(-1), v_.reshape(-1)), k
    q = lambda n: [{"type": "image"}, {"type": "text", "text": "Is there any defect?" + " more" * n}]
    sizes = [(80, 100), (150, 400)]
    rows = [{"prompt": [{"role": "user", "content": q(3 * i)}], "image": [fx.synth_pil_image(w_, h_, 20 + i)],
             "solution": "<think>gt</think><location>top left</location><type>scratch</type><answer>yes</answer>"} for i, (h_, w_) in enumerate(sizes)]
    out_dir = str(tmp_path / "run")
    def text_checksum_reward(prompts, completions, **kw):       # a reward plugin that separates the random completions (the shipped ones score 0 on noise)
        return [float(sum(map(ord, c[0]["content"])) % 7) for c in completions]

    tr = SCGRPOTrainer(src, [rewards.accuracy_reward, rewards.consistency_reward, text_checksum_reward],
                       args=GRPOConfig(output_dir=out_dir, num_generations=4, max_completion_length=8, max_prompt_length=None, learning_rate=1e-3, per_device_train_batch_size=2,
                                       max_steps=1, logging_steps=1, save_steps=0, shuffle=False),
                       train_dataset=rows, processing_class=proc)
    assert tr.cfg.is_llava and proc.tokenizer.padding_side == "left"
    before = tr.policy.flat.clone()
    hist = tr.train()
    assert len(hist) == 1 and np.isfinite(hist[0]["loss"]) and np.isfinite(hist[0]["grad_norm"]) and hist[0]["grad_norm"] > 0 and hist[0]["kl"] == 0.0
    assert not torch.equal(before, tr.policy.flat) and torch.equal(tr.ref.flat, before)
    tr.save_model(out_dir + "/final")
    cfg2, s2 = load_checkpoint(out_dir + "/final", DEV, trainable=False)
    assert cfg2 == cfg
    a, b = tr.policy.export_named(), s2.export_named()
    assert set(a) == set(b) and all(torch.equal(a[k].to(torch.bfloat16), b[k].to(torch.bfloat16)) for k in a)
    assert "vision_tower.vision_model.encoder.layers.0.self_attn.q_proj.weight" in b and "language_model.model.layers.0.self_attn.q_proj.weight" in b and "image_newline" in b


def test_fp8_decode_weights_rollout_and_step():
    """Opt-in FP8 weight stream of the rollout (BASELINE config 5 "fp8 weights"): gate|up, down and lm_head of the decode step read e4m3 packs with
    per-row scales, everything the loss differentiates stays bf16.  Stated tolerance of the rollout's logits against the bf16 decode on the same
    inputs: relative L2 error <= 8 % (measured 5.0 %: the format's 3 mantissa bits through two layers and the head); the step runs the policy's training forward
    itself (no hand-over of FP8-computed activations) and moves the parameters with finite loss; policy == reference still gives KL exactly 0."""
    w = fx.make_weights(fx.TINY, 0)
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 3)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=3), "image_grid_thw": [grid]}
    G = 4
    logits = {}
    for kind in ("bf16", "fp8"):
        pol = ParamStore(CFG, DEV, trainable=True, decode_weights=kind)
        pol.load_named(w)
        ref = store(w, False)
        assert pol.decode_fp8 == (kind == "fp8")
        eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=2, suppress_eos=True, learning_rate=1e-3, seed=5))
        toks = eng.rollout(batch, greedy=True)              # token 0 from the (bf16) prefill logits, then ONE decode step on the same inputs in both runs
        logits[kind] = (toks.copy(), eng._rollout.logits.float().cpu().clone())
    assert np.array_equal(logits["bf16"][0][:, 0], logits["fp8"][0][:, 0])
    a, b = logits["bf16"][1], logits["fp8"][1]
    rel = float((a - b).norm() / a.norm())
    print(f"[fp8 decode weights] relative L2 error of the decode logits vs bf16 weights: {rel:.4f}")
    assert 1e-4 < rel < 0.08, rel
    before = pol.flat.clone()
    eng8 = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=8, learning_rate=1e-3, seed=7))
    mt = eng8.step(batch, lambda comp: np.stack([(np.asarray(comp)[:, 0] % 5).astype(np.float32), (np.asarray(comp)[:, 1] % 3).astype(np.float32)], 1))
    assert np.isfinite(mt["loss"]) and mt["kl"] == 0.0 and not eng8.last_step_traced and not torch.equal(before, pol.flat)
    assert eng8.grad_norm() > 0


def test_llava_onevision_64_wide_heads_train_and_roll_out():
    """LLaVA-OneVision-0.5B's decoder structure (64-wide heads, stored padded to 128): one SC-GRPO step moves the parameters with finite loss and KL exactly 0 for
    policy == reference; the padding of every head (q|k|v rows, o columns outside the real dims' slots) is still exactly zero after the optimizer step; the greedy
    rollout's tokens are the arg-max of the training kernels' logits; the checkpoint written afterwards has the model's own shapes."""
    cfg = VLMConfig.from_dict(fx.TINY_OV64)
    w = fx.make_weights_ov(fx.TINY_OV64, 0)
    pol, ref = ParamStore(cfg, DEV, trainable=True), ParamStore(cfg, DEV, trainable=False)
    pol.load_named(w)
    ref.load_named(w)
    from iadr1_amd import llava_ov as lo
    v = fx.TINY_OV64["vision"]
    sizes, rs, rows, ncrops = [(80, 100), (120, 100)], np.random.RandomState(11), [], 0
    for sz, nt in zip(sizes, (5, 12)):
        n_img = lo.num_image_tokens(sz, fx.TINY_OV64["image_grid_pinpoints"], v["image_size"], v["image_size"] // v["patch_size"], 9)
        rows.append(rs.randint(3, 600, 3).tolist() + [cfg.image_token_id] * n_img + rs.randint(3, 600, nt).tolist())
        ncrops += lo.num_crops(sz, fx.TINY_OV64["image_grid_pinpoints"], v["image_size"])
    ids, mask = fx.left_pad(rows, cfg.pad_token_id)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_crops(ncrops, fx.TINY_OV64, 9), "image_sizes": sizes}
    G, C = 4, 8
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, suppress_eos=True, learning_rate=1e-3, seed=3))
    toks = eng.rollout(batch, greedy=True)
    e = eng.pol
    full, fmask = np.concatenate([np.repeat(ids, G, 0), toks], 1), np.concatenate([np.repeat(mask, G, 0), np.ones_like(toks)], 1)
    grids, plan_v, px, off = e.vision_inputs(batch)
    img, _ = e.vision_forward(px, plan_v, save=False)
    plan = e.text_plan(full, fmask, [[sizes[r // G]] for r in range(2 * G)], [[int(off[r // G])] for r in range(2 * G)])
    hf, _ = e.text_forward(plan, img, save=False)
    S, P = full.shape[1], ids.shape[1]
    rr = (np.arange(2 * G)[:, None] * S + np.arange(P - 1, S - 1)[None, :]).reshape(-1)
    logits = hf[torch.from_numpy(rr).to(DEV)].float() @ pol.w("lm_head").float().t()
    top2 = logits.topk(2, -1)
    sure = (top2.values[:, 0] - top2.values[:, 1]) > 0.05
    assert sure.float().mean() > 0.5 and torch.equal(top2.indices[:, 0][sure].cpu(), torch.from_numpy(toks.reshape(-1))[sure.cpu()])
    before = pol.flat.clone()
    mt = eng.step(batch, lambda comp: np.stack([(np.asarray(comp)[:, 0] % 5).astype(np.float32), (np.asarray(comp)[:, 1] % 3).astype(np.float32)], 1))
    assert np.isfinite(mt["loss"]) and mt["kl"] == 0.0 and eng.grad_norm() > 0 and not torch.equal(before, pol.flat)
    keep = torch.zeros(128, dtype=torch.bool, device=DEV)
    keep[torch.from_numpy(cfg.head_slots).to(DEV)] = True
    for i in range(cfg.num_hidden_layers):
        assert float(pol.w(f"layers.{i}.qkv.w").view(-1, 128, cfg.hidden_size)[:, ~keep].abs().max()) == 0.0
        assert float(pol.w(f"layers.{i}.qkv.b").view(-1, 128)[:, ~keep].abs().max()) == 0.0
        assert float(pol.w(f"layers.{i}.o.w").view(cfg.hidden_size, -1, 128)[:, :, ~keep].abs().max()) == 0.0
    out = pol.export_named()
    assert tuple(out["language_model.model.layers.0.self_attn.q_proj.weight"].shape) == (256, 256) and tuple(out["language_model.model.layers.0.self_attn.k_proj.bias"].shape) == (128,)
    assert tuple(out["language_model.model.layers.1.self_attn.o_proj.weight"].shape) == (256, 256)


# ------------------------------------------------------------------------------------------------------------------------------------
# LLaVA-1.5 / LLaVA-NeXT branches of the reference's model switch (REF sc_grpo_trainer.py:130-135)
# ------------------------------------------------------------------------------------------------------------------------------------
_LLAVA = {"llava": ("TINY_LLAVA15", "llava15.npz", "sc_grpo_llava15.npz"), "llava_next": ("TINY_LLAVA_NEXT", "llava_next.npz", "sc_grpo_llava_next.npz")}


def _llava_store(cfg_d, weights, trainable):
    s_ = ParamStore(VLMConfig.from_dict(cfg_d), DEV, trainable=trainable)
    s_.load_named(weights)
    return s_


def _llava_batch(cfg_d, g, meta, ids, mask):
    b = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_crops(meta["crops"], cfg_d, meta["seed"])}
    if cfg_d["family"] == "llava_next":
        b["image_sizes"] = [tuple(x) for x in meta["sizes"]]
    return b


@pytest.mark.parametrize("family", ["llava", "llava_next"])
def test_llava15_and_next_forward_matches_hf_golden(golden_dir, family):
    """CLIP tower (64-wide heads zero-padded to 80, class token assembled / dropped through the sparse row maps, pre-LayerNorm, QuickGELU, the blocks up to the feature
    layer), projector, NeXT's packing without the shrink step, bias-free LLaMA (MHA) / Mistral (GQA) decoder on the HIP kernels vs tiny HF models; the checkpoint names
    round-trip including the tensors this path never touches (the last CLIP block, post_layernorm)."""
    cfg_name, gname, _ = _LLAVA[family]
    cfg_d = getattr(fx, cfg_name)
    g = load(golden_dir, gname)
    meta = json.loads(str(g["meta"]))
    w = fx.make_weights_llava(cfg_d, 0)
    st = _llava_store(cfg_d, w, False)
    assert st.cfg.v_arch == "clip" and st.cfg.v_run_depth == cfg_d["vision"]["depth"] - 1 and not st.cfg.qkv_bias
    back = st.export_named()
    assert set(back) == set(w)
    for k, v_ in w.items():
        assert np.array_equal(back[k].numpy().reshape(-1), v_.reshape(-1)), k
    e = Engine(st)
    ids, mask = g["input_ids"], g["attention_mask"]
    batch = _llava_batch(cfg_d, g, meta, ids, mask)
    sizes, plan_v, px, rows = e.vision_inputs(batch)
    assert plan_v.lens == g["feature_lens"].tolist()
    img, _ = e.vision_forward(px, plan_v, save=False)
    assert relerr(img.float().cpu().numpy(), g["image_features"]) < 3e-2
    plan = e.text_plan(ids, mask, [[s_] for s_ in sizes], [[int(r)] for r in rows[:-1]])
    hf, _ = e.text_forward(plan, img, save=False)
    B, S = ids.shape
    valid = (mask[:, 1:] * mask[:, :-1]).astype(bool)
    rr = (np.arange(B)[:, None] * S + np.arange(S - 1)[None, :])[valid]
    lp, _ = e.logprobs(hf, torch.from_numpy(rr).to(DEV), torch.from_numpy(ids[:, 1:][valid].astype(np.int64)).to(DEV), save=False)
    err = np.abs(lp.cpu().numpy() - g["per_token_logps"][valid]).max()
    assert err < 0.08, err


@pytest.mark.parametrize("family", ["llava", "llava_next"])
def test_llava15_and_next_sc_grpo_matches_reference_golden(golden_dir, family):
    """The reference's compute_loss under its llava_1_5 / llava_next model ids (both pass through `_ensure_left_padding_data`) vs the HIP engine: log-probs, advantages,
    loss / KL relative, gradient norms and directions incl. the class embedding and the pre-LayerNorm; the bias-free decoder's fused q|k|v bias receives no gradient."""
    cfg_name, _, gname = _LLAVA[family]
    cfg_d = getattr(fx, cfg_name)
    g = load(golden_dir, gname)
    meta = json.loads(str(g["meta"]))
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    w_ref = fx.make_weights_llava(cfg_d, 0)
    pol, ref = _llava_store(cfg_d, fx.perturb_weights(w_ref, 1, scale=meta["perturb_scale"]), True), _llava_store(cfg_d, w_ref, False)
    eng = SCGRPOEngine(pol.cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, beta=0.04, micro_batch_seqs=16))
    P = g["prompt_completion_ids"].shape[1] - C
    batch = _llava_batch(cfg_d, g, meta, g["prompt_completion_ids"][:1, :P], g["attention_mask"][:1, :P])
    comps = fx.synth_completions(G, C, cfg_d, seed + 100, {int(k): v for k, v in meta["eos_rows"].items()})
    out = eng.loss_and_grads(batch, comps, g["rewards_per_func"])
    assert np.array_equal(out["ids"], g["prompt_completion_ids"]) and np.array_equal(out["completion_mask"], g["completion_mask"])
    m = g["completion_mask"].astype(bool)
    dlp = np.abs(out["logps"].cpu().numpy()[m] - g["per_token_logps"][m]).max()
    dlr = np.abs(out["ref_logps"].cpu().numpy()[m] - g["ref_per_token_logps"][m]).max()
    gl, gk = float(g["loss"]), float(g["metric_kl"])
    mt = out["metrics"]
    print(f"[parity] {family}: loss hip={mt['loss']:.6e} ref={gl:.6e}  kl hip={mt['kl']:.6e} ref={gk:.6e} ({100 * abs(mt['kl'] - gk) / gk:.1f}%)  |dlogp|max={dlp:.4f}/{dlr:.4f}")
    assert dlp < 0.08 and dlr < 0.08, (dlp, dlr)
    np.testing.assert_allclose(out["advantages"].numpy(), g["advantages"], rtol=1e-5, atol=1e-6)
    assert abs(mt["kl"] - gk) <= 0.10 * gk and abs(mt["loss"] - gl) <= 0.04 * 0.10 * gk + 2e-6
    grads = pol.export_named(source="grad")
    for n, ref_norm in zip([str(n) for n in g["grad_norm_names"]], g["grad_norms"]):
        if n == "language_model.lm_head.weight" or ref_norm < 1e-7 or n not in grads:      # (a key bias shifts every score of a row alike: its gradient is 0 up to rounding)
            continue
        got = float(grads[n].norm())
        assert abs(got - ref_norm) <= 0.10 * ref_norm + 1e-7, (n, got, ref_norm)
    for k in g.files:
        if k.startswith("grad::"):
            a, b = grads[k[6:]].numpy().reshape(-1).astype(np.float64), g[k].reshape(-1).astype(np.float64)
            cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
            assert cos > 0.99, (k, cos)
    assert float(pol.g("layers.0.qkv.b").abs().max()) == 0.0 and float(pol.w("layers.1.qkv.b").abs().max()) == 0.0


@pytest.mark.parametrize("family", ["llava", "llava_next"])
def test_llava15_and_next_rollout_and_step(family):
    """Group rollout (hipGraph == eager) and one full SC-GRPO step on both structures: finite loss, KL exactly 0 for policy == reference, both towers move."""
    cfg_name = _LLAVA[family][0]
    cfg_d = getattr(fx, cfg_name)
    w = fx.make_weights_llava(cfg_d, 0)
    pol, ref = _llava_store(cfg_d, w, True), _llava_store(cfg_d, w, False)
    cfg = pol.cfg
    e0 = Engine(ref)
    sizes = [(56, 56), (56, 56)] if family == "llava" else [(80, 100), (150, 60)]
    rs, rows = np.random.RandomState(7), []
    for sz, nt in zip(sizes, (5, 12)):
        rows.append(rs.randint(3, 600, 3).tolist() + [cfg.image_token_id] * e0.n_image_tokens(sz) + rs.randint(3, 600, nt).tolist())
    ids, mask = fx.left_pad(rows, cfg.pad_token_id)
    from iadr1_amd import llava_ov as lo
    ncrops = 2 if family == "llava" else sum(lo.num_crops(s_, cfg_d["image_grid_pinpoints"], cfg_d["vision"]["image_size"]) for s_ in sizes)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_crops(ncrops, cfg_d, 9)}
    if family == "llava_next":
        batch["image_sizes"] = sizes
    G, C = 4, 8
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, suppress_eos=True, learning_rate=1e-3, seed=3))
    toks = eng.rollout(batch, greedy=True)
    eager = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, suppress_eos=True, use_hip_graph=False))
    assert np.array_equal(toks, eager.rollout(batch, greedy=True)) and toks.shape == (2 * G, C)
    before = pol.flat.clone()
    mt = eng.step(batch, lambda comp: np.stack([(np.asarray(comp)[:, 0] % 5).astype(np.float32), (np.asarray(comp)[:, 1] % 3).astype(np.float32)], 1))
    assert np.isfinite(mt["loss"]) and mt["kl"] == 0.0 and eng.grad_norm() > 0
    moved = pol.flat != before
    for name in ("visual.blocks.0.fc1.w", "visual.cls", "layers.0.qkv.w"):
        sl = pol.slots[name]
        assert bool(moved[sl.offset: sl.offset + min(100, int(np.prod(sl.shape)))].any()), name
    assert float(pol.w("layers.0.qkv.b").abs().max()) == 0.0


def test_rollout_regrows_for_longer_prompts():
    """Prompt lengths change from batch to batch (image sizes of the any-resolution families, text lengths): a prompt longer than any seen so far -- including one
    whose last partial page plus the completion needs one block-table column more -- rebuilds the KV pool / block table / graph instead of overrunning them."""
    w = fx.make_weights(fx.TINY, 0)
    pol, ref = store(w, True), store(w, False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=4, max_prompt_length=4096, max_completion_length=8, suppress_eos=True))
    grid = (1, 16, 12)                       # 48 image tokens
    last = None
    for n_text in (9, 40, 47, 48, 100, 20):  # padded prompt lengths on both sides of page boundaries, then a shorter one (no rebuild)
        ids, mask = fx.left_pad([fx.synth_prompt(grid, n_text, fx.TINY, 3)], fx.TINY["pad_token_id"])
        batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid], fx.TINY, seed=3), "image_grid_thw": [grid]}
        toks = eng.rollout(batch, greedy=True)
        assert toks.shape == (4, 8) and eng._rollout.max_prompt >= ids.shape[1]
        last = eng._rollout
    assert eng._rollout is last and last.max_prompt >= 100


def test_sc_grpo_loop_learns_the_rewarded_behaviour():
    """The whole loop as a learner (rollout -> rewards -> group advantages -> clipped-ratio-free GRPO gradient with the KL term -> AdamW -> updated weights in the next
    rollout): reward = fraction of completion tokens with an id in the lower half of the vocabulary.  From 0.29 at the start, the mean reward of the sampled groups
    reaches > 0.9 within 10 optimizer steps while the KL to the frozen reference grows from exactly 0 (tools/grpo_learns.py prints the curve)."""
    w = fx.make_weights(fx.TINY, 0)
    pol, ref = store(w, True), store(w, False)
    eng = SCGRPOEngine(CFG, pol, ref, GRPOArgs(num_generations=8, max_prompt_length=4096, max_completion_length=8, learning_rate=3e-3, beta=0.04, suppress_eos=True, seed=11))
    grid = (1, 16, 12)
    ids, mask = fx.left_pad([fx.synth_prompt(grid, 9, fx.TINY, 3), fx.synth_prompt(grid, 9, fx.TINY, 4)], fx.TINY["pad_token_id"])
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values([grid, grid], fx.TINY, seed=3), "image_grid_thw": [grid, grid]}
    reward = lambda comp: (np.asarray(comp) < fx.TINY["text"]["vocab_size"] // 2).mean(1, keepdims=True).astype(np.float32)
    hist = [eng.step(batch, reward) for _ in range(10)]
    rewards, kls = [h["reward"] for h in hist], [h["kl"] for h in hist]
    assert rewards[0] < 0.5 and min(rewards[-3:]) > 0.9, rewards
    assert kls[0] == 0.0 and kls[-1] > 1e-3, kls


def test_trainer_level_traced_path_with_gradient_accumulation():
    """The fast path at the reference's API with the trainer's own loop: `SCGRPOTrainer.train()` at the 3B widths (2 layers), gradient_accumulation_steps = 2 -- two
    compute_loss calls per optimizer step, each with its own rollout whose prefill / decode steps fill the SAME training arena (captured graph, arena pointers
    frozen) -- against the same trainer with the hand-over switched off (the policy forward runs after the rollout).  The first optimizer step of both runs sees the same
    tokens (same seeds, same decode kernels): same rewards, gradient norms within 1 %, parameters after the step pointing the same way (AdamW's first update is
    +-lr per element, so near-zero gradients may flip); the second step runs on (sampling diverges with the weights), and the hand-over really was active in all four
    micro-steps."""
    import dataclasses
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer
    cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    G, C, Bp = 4, 12, 2
    b = bench.synth_batch(cfg, Bp, 300, seed=5)
    batch = {"input_ids": torch.from_numpy(b["input_ids"]), "attention_mask": torch.from_numpy(b["attention_mask"]), "pixel_values": b["pixel_values"], "image_grid_thw": torch.tensor(b["image_grid_thw"])}
    rows = [{"prompt": [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "q"}]}], "image": [object()], "solution": "s"}] * (Bp * 4)

    def token_reward(prompts, completions, **kw):
        return [float(sum(map(ord, c[0]["content"])) % 5) for c in completions]

    class Proc(_FakeProcessor):
        def batch_decode(self, ids, skip_special_tokens=True):
            return [" ".join(str(int(t) % 97) for t in row) for row in np.asarray(ids)]      # text that depends on the sampled ids -> rewards that separate the group

    finals, traced = {}, {}
    for reuse in (True, False):
        pol = ParamStore(cfg, DEV, trainable=True)
        pol.init_random(seed=0)
        tr = SCGRPOTrainer((cfg, pol), [token_reward], args=GRPOConfig(output_dir="/tmp/iadr1_traced_test", num_generations=G, max_completion_length=C, max_prompt_length=None,
                                                                        per_device_train_batch_size=Bp, gradient_accumulation_steps=2, learning_rate=1e-4, max_steps=2, logging_steps=1, save_steps=0,
                                                                        shuffle=False, micro_batch_seqs=64, seed=7, batch_rollouts=False),
                           train_dataset=rows, processing_class=Proc(batch, None))
        tr.engine.args.reuse_decode = reuse
        tr.engine.args.suppress_eos = True
        seen = []
        orig = tr.engine.step

        def step(*a, _o=orig, _s=seen, _e=tr.engine, **k):
            out = _o(*a, **k)
            _s.append(bool(_e.last_step_traced))
            return out
        tr.engine.step = step
        first = {}
        orig_opt = tr.engine.optimizer_step

        def opt(_o=orig_opt, _p=pol, _f=first):
            _o()
            _f.setdefault("flat", _p.flat.float().clone())
        tr.engine.optimizer_step = opt
        hist = tr.train()
        assert len(hist) == 2 and all(np.isfinite(h["loss"]) and h["grad_norm"] > 0 for h in hist)
        finals[reuse], traced[reuse] = (first["flat"], hist[0]), seen
        del tr, pol
    assert traced[True] == [True] * 4 and traced[False] == [False] * 4                     # 2 optimizer steps x 2 micro-steps, hand-over active in every one
    (a, ha), (b_, hb) = finals[True], finals[False]
    assert ha["reward"] == hb["reward"] and ha["reward_std"] == hb["reward_std"] and abs(ha["grad_norm"] - hb["grad_norm"]) <= 0.01 * hb["grad_norm"]
    a, b_ = a.double(), b_.double()
    assert float((a - b_).abs().max()) <= 2.5e-4 and float((a @ b_) / (a.norm() * b_.norm())) > 0.9999


# ------------------------------------------------------------------------------------------------------------------------------------
# Round 5: the frozen reference's teacher-forced pass co-scheduled with the rollout (iadr1_amd/overlap.py)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("steps,cus", [(16, -1), (32, -1), (16, 64), (32, 96)])
def test_chunked_reference_pass_is_bit_equal_to_the_one_shot_pass(monkeypatch, steps, cus):
    """SCGRPOEngine.step with the reference's pass running UNDER the rollout, `steps` decode steps' worth of rows at a time on a second stream (time-blocked
    completion rows, iadr1_attn_fwd_chunk; cus > 0: that stream confined to `cus` CUs, the decode replays on the others, gated by the device step counter --
    IADR1_OVERLAP_CUS; -1: two ordinary streams), against the same step with the reference's one-shot pass after the rollout (the default): BASELINE widths
    (3B: hidden 2048, 16:2 heads, MLP 11008, vocabulary 151936; 2 layers), policy != reference, two left-padded prompts of different length, EOS live with
    ragged completion lengths.  Same tokens; reference log-probs BIT-equal on every scored position; loss / KL / gradients therefore equal too."""
    import dataclasses
    cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    ref = ParamStore(cfg, DEV, trainable=False)
    ref.init_random(seed=0)
    w_ref = {k: v.float().numpy() for k, v in ref.export_named().items()}
    pol = ParamStore(cfg, DEV, trainable=True)
    pol.load_named(fx.perturb_weights(w_ref, 1, scale=0.25))
    G, C, Bp = 4, 48, 2
    cd = _oracle_cfg_dict(cfg)
    grids = [(1, 16, 16), (1, 16, 12)]
    rows = [fx.synth_prompt(grids[0], 37, cd, 5), fx.synth_prompt(grids[1], 21, cd, 6)]
    ids, mask = fx.left_pad(rows, cfg.pad_token_id)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values(grids, cd, seed=5), "image_grid_thw": grids}
    args = lambda: GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=Bp * G, seed=11, beta=0.04)
    reward_fn = lambda comp: np.stack([(np.asarray(comp)[:, 0] % 5).astype(np.float32), (np.asarray(comp)[:, 1] % 3).astype(np.float32) * 0.5], 1)
    comp0 = SCGRPOEngine(cfg, pol, ref, args()).rollout(batch, vis=None)
    best, best_rows = None, []
    for tok in np.unique(comp0[:, 2: C - 2]):        # declare the token that ends the most rows early to be EOS (sampling is a pure function of seed / step / row / logits)
        hit = [r for r in range(Bp * G) if tok in comp0[r, 2: C - 2] and tok not in comp0[r, :2]]
        if len(hit) > len(best_rows):
            best, best_rows = int(tok), hit
    cfg.eos_token_id = best
    monkeypatch.setenv("IADR1_OVERLAP_STEPS", str(steps))
    if cus > 0:       # a rollout sized for fewer CUs splits the o projection's K differently (fp32 partial sums in another order -> other samples): same split both ways
        monkeypatch.setenv("IADR1_DECODE_KS", "1,8")
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("IADR1_OVERLAP_CUS", str(cus) if mode == "1" else "0")
        pol.grad.zero_()
        eng = SCGRPOEngine(cfg, pol, ref, args())
        out = eng.step(batch, reward_fn, do_optimizer_step=False, return_outputs=True)
        torch.cuda.synchronize()
        assert eng.last_step_shadowed == (mode == "1")
        res[mode] = (out, pol.grad.clone())
    (o1, g1), (o0, g0) = res["1"], res["0"]
    assert np.array_equal(o1["completion_ids"], o0["completion_ids"]) and np.array_equal(o1["completion_mask"], o0["completion_mask"])
    m = torch.from_numpy(o1["completion_mask"].astype(bool)).to(DEV)
    lens = o1["completion_mask"].sum(1)
    assert (lens < C).any() and (lens == C).any(), lens
    a, b = o1["ref_logps"][m], o0["ref_logps"][m]
    assert bool(torch.isfinite(a).all())
    print(f"[parity] chunked vs one-shot reference pass: max |d| = {float((a - b).abs().max()):.3e} over {int(m.sum())} scored tokens, lens {lens.tolist()}")
    assert torch.equal(a, b)
    assert torch.equal(o1["logps"][m], o0["logps"][m])
    assert o1["metrics"]["kl"] == o0["metrics"]["kl"] and o1["metrics"]["loss"] == o0["metrics"]["loss"]
    cos = float((g0 @ g1) / (g0.norm() * g1.norm()))
    assert cos > 0.99999, cos            # (the forward bits are equal; the backward's reductions are ordered two-stage sums since round 5 -- no float atomics --, the margin is kept for the two-stream launch order of the weight-gradient GEMMs' fp32 accumulation into shared buffers)


@pytest.mark.parametrize("steps,cus,eos_live", [(32, -1, False), (16, 64, False), (32, -1, True)])
def test_policy_mlp_rows_rebuilt_under_the_rollout(monkeypatch, steps, cus, eos_live):
    """Overlap mode, IADR1_OVERLAP_GU=1 (its default): the decode step's gate|up kernel stores nothing for the training hand-over; the policy's gate|up and SwiGLU
    rows of the completion tokens are rebuilt on the side stream, chunk by chunk, from the `h2` rows the decode steps did store (iadr1_gemm_swiglu_rows_bf16).
    (1) After a rollout every such row of the arena is BIT-equal to iadr1_gemm_swiglu_bf16 of its `h2` row; (2) a whole step against the same step with the stores
    (IADR1_OVERLAP_GU=0): same tokens, log-probs of both models, KL and loss bit-equal (the forward never reads those rows), gradients equal to bf16 rounding of the
    two kernels' gate|up sums (training-GEMM vs decode-kernel summation order: cosine > 0.9999).  The policy's lm_head log-probs come from the side stream as well
    (IADR1_OVERLAP_HEAD, same fused launch: bit-equal).  3B widths, 2 layers, 2 prompts x G 8 (16 sequences x one time block = one 256-row tile), C = 48 (chunks of
    32 + 16 or 16 + 16 + 16).  eos_live: ragged completions, some rows cut short (positions the side stream never scores are masked; their lse stays +1e30)."""
    import dataclasses
    from iadr1_amd import ops, overlap
    cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    ref = ParamStore(cfg, DEV, trainable=False)
    ref.init_random(seed=0)
    w_ref = {k: v.float().numpy() for k, v in ref.export_named().items()}
    pol = ParamStore(cfg, DEV, trainable=True)
    pol.load_named(fx.perturb_weights(w_ref, 1, scale=0.25))
    G, C, Bp = 8, 48, 2
    cd = _oracle_cfg_dict(cfg)
    grids = [(1, 16, 16), (1, 16, 12)]
    rows = [fx.synth_prompt(grids[0], 37, cd, 5), fx.synth_prompt(grids[1], 21, cd, 6)]
    ids, mask = fx.left_pad(rows, cfg.pad_token_id)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values(grids, cd, seed=5), "image_grid_thw": grids}
    args = lambda: GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=Bp * G, seed=11, beta=0.04, suppress_eos=not eos_live)
    reward_fn = lambda comp: np.stack([(np.asarray(comp)[:, 0] % 5).astype(np.float32), (np.asarray(comp)[:, 1] % 3).astype(np.float32) * 0.5], 1)
    monkeypatch.setenv("IADR1_OVERLAP_STEPS", str(steps))
    monkeypatch.setenv("IADR1_OVERLAP_CUS", str(cus))
    if cus > 0:
        monkeypatch.setenv("IADR1_DECODE_KS", "1,8")
    if eos_live:      # declare the token that ends the most rows early to be EOS (sampling is a pure function of seed / step / row / logits)
        monkeypatch.setenv("IADR1_OVERLAP_CUS", "0")
        comp0 = SCGRPOEngine(cfg, pol, ref, args()).rollout(batch, vis=None)
        monkeypatch.setenv("IADR1_OVERLAP_CUS", str(cus))
        best, best_rows = None, []
        for tok in np.unique(comp0[:, 2: C - 2]):
            hit = [r for r in range(Bp * G) if tok in comp0[r, 2: C - 2] and tok not in comp0[r, :2]]
            if len(hit) > len(best_rows):
                best, best_rows = int(tok), hit
        cfg.eos_token_id = best
    # (1) the rows themselves
    monkeypatch.setenv("IADR1_OVERLAP_GU", "1")
    eng = SCGRPOEngine(cfg, pol, ref, args())
    before = overlap.STATS["policy_mlp_rows"]

    def roll():
        vis = eng.vision_policy(batch, save=True)
        carry = {}
        eng.rollout(batch, vis=vis, train_carry=carry, shadow_ref=True)
        torch.cuda.synchronize()
        assert carry.get("traced") and eng.shadow_logps is not None

    if cus > 0:
        main = torch.cuda.Stream()
        main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(main):
            roll()
    else:
        roll()
    N = Bp * G
    assert overlap.STATS["policy_mlp_rows"] - before == N * C and eng.shadow_policy_head is not None
    tr = eng._rollout.trace
    assert tr["mlp_on_shadow"]
    T0 = ids.shape[0] * ids.shape[1]
    arena_rows = (T0 + torch.arange(N, device=DEV)[:, None] * C + torch.arange(C, device=DEV)[None, :]).reshape(-1)       # 768 = 3 x 256 rows
    I = cfg.intermediate_size
    for i in range(cfg.num_hidden_layers):
        h2 = tr["h2"][i][arena_rows].contiguous()
        gu_x, a_x = torch.empty(N * C, 2 * I, dtype=torch.bfloat16, device=DEV), torch.empty(N * C, I, dtype=torch.bfloat16, device=DEV)
        ops.gemm_swiglu_fused(h2, pol.w(f"layers.{i}.gu.w"), gu_x, a_x)
        assert torch.equal(tr["gu"][i][arena_rows].view(torch.int16), gu_x.view(torch.int16)) and torch.equal(tr["a"][i][arena_rows].view(torch.int16), a_x.view(torch.int16))
        assert float(gu_x.float().abs().max()) > 0
    del eng
    # (2) the step
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("IADR1_OVERLAP_GU", mode)
        pol.grad.zero_()
        eng = SCGRPOEngine(cfg, pol, ref, args())
        out = eng.step(batch, reward_fn, do_optimizer_step=False, return_outputs=True)
        torch.cuda.synchronize()
        assert eng.last_step_shadowed and eng.last_step_traced and eng._rollout.trace["mlp_on_shadow"] == (mode == "1")
        res[mode] = (out, pol.grad.clone())
        del eng
    (o1, g1), (o0, g0) = res["1"], res["0"]
    assert np.array_equal(o1["completion_ids"], o0["completion_ids"]) and np.array_equal(o1["completion_mask"], o0["completion_mask"])
    m = torch.from_numpy(o1["completion_mask"].astype(bool)).to(DEV)
    if eos_live:
        lens = o1["completion_mask"].sum(1)
        assert (lens < C).any() and (lens == C).any(), lens
    assert torch.equal(o1["ref_logps"][m], o0["ref_logps"][m]) and torch.equal(o1["logps"][m], o0["logps"][m])
    assert o1["metrics"]["kl"] == o0["metrics"]["kl"] and o1["metrics"]["loss"] == o0["metrics"]["loss"]
    assert bool(torch.isfinite(g1).all())
    cos = float((g0.double() @ g1.double()) / (g0.double().norm() * g1.double().norm()))
    rel = float((g0 - g1).norm() / g0.norm())
    print(f"[parity] policy mlp rows rebuilt on the side stream vs stored by the decode step: gradient cosine {cos:.7f}, relative difference {rel:.3e}")
    assert cos > 0.9999 and rel < 1.5e-2, (cos, rel)


def test_weight_prefetcher_follows_the_decode_step_and_changes_no_token():
    """The opt-in persistent weight prefetcher (iadr1_weight_prefetch, iad-r1_amd/wprefetch.py; REF:637-683 is the loop it runs under): attached to a rollout it is
    paced by the progress marks the first kernel of every decoder layer stores -- it sees the (C - 1) x layers units and reads most of them, no timeout --,
    it returns when the rollout stores its sequence number in the stop word (twice in a row: the second launch starts from a clean progress word), and the tokens are
    those of the rollout without it (it only reads).  3B widths, 2 layers, 16 sequences, C = 24; the prefetcher on an ordinary second stream."""
    import dataclasses
    from iadr1_amd.wprefetch import WeightPrefetcher
    cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    pol = ParamStore(cfg, DEV, trainable=True)
    pol.init_random(seed=0)
    ref = ParamStore(cfg, DEV, trainable=False)
    ref.copy_from(pol)
    G, C, Bp = 8, 24, 2
    cd = _oracle_cfg_dict(cfg)
    grids = [(1, 16, 16), (1, 16, 12)]
    rows = [fx.synth_prompt(grids[0], 37, cd, 5), fx.synth_prompt(grids[1], 21, cd, 6)]
    ids, mask = fx.left_pad(rows, cfg.pad_token_id)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values(grids, cd, seed=5), "image_grid_thw": grids}
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=Bp * G, seed=11, suppress_eos=True))
    main = torch.cuda.Stream()
    with torch.cuda.stream(main):
        want = eng.rollout(batch)
        pf = WeightPrefetcher(eng.pol, torch.cuda.Stream(), 8, what=(), next_what=("qkv", "o"), lead=0)
        eng._rollout.wprefetch = pf
        for _ in range(2):
            got = eng.rollout(batch)
            rep = pf.report()
            assert np.array_equal(got, want)
            n_units = (C - 1) * cfg.num_hidden_layers      # (a unit the decode step has passed is skipped or dropped: block 0 may see a few at this 2-layer depth)
            assert not rep["timed_out"] and rep["units_read"] + rep["units_stale"] >= 0.8 * n_units and rep["units_read"] >= 0.5 * n_units, rep
        assert pf.launched == 2 and int(eng._rollout.pf_stop.item()) == 2 and int(eng._rollout.mark.item()) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("full_depth", [False, True])
def test_co_scheduling_is_on_by_default_at_the_bench_shape_and_changes_no_forward_bit(monkeypatch, full_depth):
    """IADR1_OVERLAP_CUS unset (= auto): at the benchmark's shape class (hidden <= 2048, 8 prompts x G 8 = 64 sequences, C a multiple of 64) SCGRPOEngine.step
    co-schedules the reference pass on 64 CUs with the decode replays on the other 192 and rebuilds the policy's mlp rows there; against IADR1_OVERLAP_CUS=0 (the
    reference pass after the rollout): same tokens, BIT-equal log-probs of both models, KL and loss; gradients equal to the bf16 rounding of the rebuilt gate|up
    rows.  Small shapes stay un-co-scheduled (overlap.auto_applies).  3B widths; 2 layers, and the UNREDUCED model (36 decoder + 32 vision layers: the chunked
    attention, the time-blocked rows and the side-stream lm_head through the whole depth; policy and reference independently initialised)."""
    import dataclasses
    from iadr1_amd import overlap
    cfg = VLMConfig.qwen25vl_3b() if full_depth else dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    assert overlap.auto_applies(cfg, 64, 256) and not overlap.auto_applies(cfg, 8, 512) and not overlap.auto_applies(VLMConfig.qwen25vl_7b(), 64, 256)
    ref = ParamStore(cfg, DEV, trainable=False)
    ref.init_random(seed=0)
    pol = ParamStore(cfg, DEV, trainable=True)
    if full_depth:
        pol.init_random(seed=1)
    else:
        w_ref = {k: v.float().numpy() for k, v in ref.export_named().items()}
        pol.load_named(fx.perturb_weights(w_ref, 1, scale=0.25))
    G, C, Bp = 8, 128, 8
    cd = _oracle_cfg_dict(cfg)
    grids = [(1, 16, 16), (1, 16, 12)] * 4
    rows = [fx.synth_prompt(grids[k], 21 + 3 * k, cd, 5 + k) for k in range(Bp)]
    ids, mask = fx.left_pad(rows, cfg.pad_token_id)
    batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": fx.synth_pixel_values(grids, cd, seed=5), "image_grid_thw": grids}
    args = lambda: GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, micro_batch_seqs=Bp * G, seed=11, beta=0.04, suppress_eos=True)
    reward_fn = lambda comp: np.stack([(np.asarray(comp)[:, 0] % 5).astype(np.float32), (np.asarray(comp)[:, 1] % 3).astype(np.float32) * 0.5], 1)
    monkeypatch.setenv("IADR1_DECODE_KS", "1,8")       # (a rollout sized for 192 CUs splits the o projection's K differently: same split both ways)
    res = {}
    for mode in ("auto", "0"):
        if mode == "auto":
            monkeypatch.delenv("IADR1_OVERLAP_CUS", raising=False)
        else:
            monkeypatch.setenv("IADR1_OVERLAP_CUS", mode)
        pol.grad.zero_()
        eng = SCGRPOEngine(cfg, pol, ref, args())
        out = eng.step(batch, reward_fn, do_optimizer_step=False, return_outputs=True)
        torch.cuda.synchronize()
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        if mode == "auto":
            if ncu != 256:
                pytest.skip(f"the default split (192 + 64 CUs) is defined for the 256-CU MI355X; this device has {ncu}")
            if not eng.last_step_shadowed and overlap.cu_split(torch.device("cuda", 0), overlap.AUTO_CUS, ref.w("layers.0.gu.w")) is None:
                pytest.skip("co-scheduling did NOT run: no stream pair on separate dispatch pipes was found on this box (the engine fell back to the sequential step)")
            # on an MI355X with a clean stream pair the default step IS the co-scheduled one (VERDICT r5 #3b: this test cannot pass on the fallback branch)
            assert eng.last_step_shadowed and eng._rollout.decode_cus == ncu - overlap.AUTO_CUS and eng._rollout.trace["mlp_on_shadow"]
            print(f"[parity] default step at the bench shape class: CO-SCHEDULED (decode on {eng._rollout.decode_cus} CUs, side stream on {overlap.AUTO_CUS})")
        else:
            assert not eng.last_step_shadowed and eng._rollout.decode_cus == 0
        res[mode] = (out, pol.grad.clone())
        del eng
    (o1, g1), (o0, g0) = res["auto"], res["0"]
    assert np.array_equal(o1["completion_ids"], o0["completion_ids"])
    assert torch.equal(o1["ref_logps"], o0["ref_logps"]) and torch.equal(o1["logps"], o0["logps"])
    assert o1["metrics"]["kl"] == o0["metrics"]["kl"] and o1["metrics"]["loss"] == o0["metrics"]["loss"]
    step_ = 1 << 27                      # (chunked: torch's dot takes at most 2^31 - 1 elements, the full model has 3.75 G)
    dot = lambda a, b: sum(float((a[k: k + step_].double() * b[k: k + step_].double()).sum()) for k in range(0, a.numel(), step_))
    cos = dot(g0, g1) / (dot(g0, g0) * dot(g1, g1)) ** 0.5
    print(f"[parity] co-scheduled vs sequential step ({'36 + 32' if full_depth else '2 + 2'} layers): forward bits equal, gradient cosine {cos:.7f}")
    assert cos > (0.9995 if full_depth else 0.9999), cos       # (measured 0.99983 through 36 layers of rebuilt gate|up rows, 0.99999 through 2)


def test_weight_gradient_forms_leave_the_same_bits(monkeypatch):
    """The three ways a weight gradient is formed -- transposed copies + the NT kernel (IADR1_GEMM_TN=0), the TN kernel straight from row-major dY / X for the split-K
    shapes (1) or for every shape the 256 x 256 kernel takes (2; what the PA-SFT engine runs) -- leave the SAME BITS in the whole gradient buffer of a PA-SFT step at the
    3B widths (2 layers, 8 rows x ~600 tokens = 4.9 k token rows: q|k|v, o, down through the split-K form, gate|up and the 151 936-row lm_head through the plain one)."""
    import dataclasses
    from iadr1_amd import ops
    cfg = dataclasses.replace(VLMConfig.qwen25vl_3b(), num_hidden_layers=2, v_depth=2, v_fullatt=(1,))
    cd = _oracle_cfg_dict(cfg)
    grids = [(1, 16, 16), (1, 16, 12), (1, 16, 16), (1, 12, 16)] * 2
    rows = [fx.synth_prompt(grids[k], 520 + 7 * k, cd, 5 + k) for k in range(8)]
    ids, mask = fx.left_pad(rows, cfg.pad_token_id)
    ids, mask = np.asarray(ids), np.asarray(mask)
    assert ids.size >= 4608
    labels = np.where((np.arange(ids.shape[1])[None, :] >= ids.shape[1] - 120) & (mask != 0), ids, -100)
    batch = {"input_ids": ids, "attention_mask": mask, "labels": labels, "pixel_values": fx.synth_pixel_values(grids, cd, seed=5), "image_grid_thw": grids}
    p = ParamStore(cfg, DEV, trainable=True)
    p.init_random(seed=3)
    grads, calls = {}, {}
    orig = ops.hip.call
    for mode in ("0", "1", "2"):
        monkeypatch.setattr(ops, "_GEMM_TN", mode)
        seen = []
        ops.hip.call = lambda name, *a, _s=seen: (_s.append(name), orig(name, *a))[1]
        try:
            p.grad.zero_()
            eng = SFTEngine(cfg, p, SFTArgs(learning_rate=1e-5, weight_decay=0.0, max_grad_norm=0.0))
            eng.loss_and_grads(batch)
            torch.cuda.synchronize()
        finally:
            ops.hip.call = orig
        grads[mode], calls[mode] = p.grad.clone(), seen
    assert "gemm_tn_acc_bf16" not in calls["0"] and calls["1"].count("gemm_tn_acc_bf16") > 0 and calls["2"].count("gemm_tn_acc_bf16") > calls["1"].count("gemm_tn_acc_bf16")
    assert calls["2"].count("transpose_bf16") < calls["1"].count("transpose_bf16") < calls["0"].count("transpose_bf16")
    assert float(grads["0"].abs().sum()) > 0
    assert torch.equal(grads["0"], grads["1"]) and torch.equal(grads["0"], grads["2"])


def test_full_size_3b_parity_at_the_headline_shape():
    """Driver-witnessed parity AT THE BENCHMARK'S SHAPE (VERDICT r4 #2a): the unreduced Qwen2.5-VL-3B (36 + 32 layers, 151 936-token head), one prompt of 448 x 448
    image + 512 positions, G = 8 completions of 256 tokens sampled by the engine's own hipGraph rollout (one row cut by EOS), policy = reference x (1 + 2 % noise):
    `SCGRPOEngine.loss_and_grads` on the GPU against `oracle.sc_grpo.sc_grpo_step` in fp32 on the host, forward quantities only (per-token log-probs of both models,
    KL, loss, completion mask, advantages; the backward at this size is the builder-run record profiles/r04_full_size_parity.json: 4 more minutes and 115 GB of host
    memory), with the same oracle in bf16 -- the precision `--bf16` gives the reference -- as the yardstick.  bench.full_size_parity is the code of
    `python bench.py --cpu-full-step --check --check-forward-only`.  ~3.5 minutes of host time.
    Stated tolerances (fp32 oracle = truth): |dlogp| max <= 1.5 x the bf16 oracle's + 0.02 and mean <= 1.5 x + 0.005 (measured r04: 0.264 / 0.055 vs 0.282 / 0.057);
    KL within max(5 %, 1.5 x the bf16 oracle's error) (measured 2.7 % vs 3.5 %); loss within beta x that KL tolerance (the loss is beta x KL - mean advantage term:
    RELATIVE to the KL, at |loss| ~ 7e-3); greedy token ids: >= 19 of 24 equal, every disagreement at an oracle top-2 gap below 4 x the measured spread of the
    decode kernels' top-2-gap error.
    Round 6 (VERDICT r5 #3c): the BACKWARD runs here too -- the oracle's autograd through the unreduced model on the host (~4 more minutes, ~115 GB of host memory):
    six named gradients from the head (final norm) to the first ViT block, cosine > 0.99 and norm within 3 % of the fp32 oracle's (builder-run record of round 4:
    0.9977 - 0.9992, norms within 0.5 %).  IADR1_TEST_FULL_BACKWARD=0 keeps the forward-only form (a host with < 160 GB of memory)."""
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    with_backward = os.environ.get("IADR1_TEST_FULL_BACKWARD", "1") != "0"
    a = argparse.Namespace(model="3b", prompt_len=512, gen_len=256, group=8, check_forward_only=not with_backward, check_noise=0.02, check_greedy_tokens=24)
    rec = bench.full_size_parity(a)
    print(f"[full-size parity, forward{' + backward' if with_backward else ''}] " + json.dumps(rec), flush=True)
    if with_backward:
        gr = rec["hip_vs_fp32_oracle"]["gradients"]
        assert len(gr) == 6 and "skipped" not in gr, gr
        for n, v in gr.items():
            assert v["cosine"] > 0.99 and 0.97 < v["norm_ratio"] < 1.03, (n, v)
    h, b, g = rec["hip_vs_fp32_oracle"], rec["bf16_oracle_vs_fp32_oracle"], rec["greedy_ids"]
    assert rec["shape"]["scored_tokens"] >= 7 * 256 and "error" not in b, (rec["shape"], b)
    for k in ("policy", "ref"):
        assert h[f"dlogp_{k}_max"] <= 1.5 * b[f"dlogp_{k}_max"] + 0.02 and h[f"dlogp_{k}_mean"] <= 1.5 * b[f"dlogp_{k}_mean"] + 0.005, (k, h, b)
    tol_k = max(0.05, 1.5 * b["kl_rel_err"])
    assert h["kl_rel_err"] <= tol_k, (h["kl_rel_err"], b["kl_rel_err"])
    assert h["loss_abs_err"] <= 0.04 * tol_k * h["kl_oracle"] + 2e-6, (h["loss_abs_err"], h["kl_oracle"])
    assert h["metrics_hip"]["completion_length"] == h["metrics_oracle"]["completion_length"] and abs(h["metrics_hip"]["reward"] - h["metrics_oracle"]["reward"]) < 1e-6
    le = g["logit_error_vs_fp32_oracle"]
    spread = le["hip_decode_kernels"]["top2_gap_error_std"]
    assert g["agree"] >= 19 and all(x < max(4 * spread, 0.05) for x in g["oracle_top2_logit_gap_at_disagreements"]), g
    # neither device path is further from fp32 than 1.5 x the reference's own precision, and the two device paths agree with each other at least as well
    for path in ("hip_decode_kernels", "hip_training_kernels"):
        assert le[path]["top2_gap_error_std"] <= 1.5 * le["bf16_oracle"]["top2_gap_error_std"] + 0.01 and le[path]["top50_logit_error_rms"] <= 1.5 * le["bf16_oracle"]["top50_logit_error_rms"] + 0.005, le
