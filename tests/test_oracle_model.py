"""Pins the CPU oracle (oracle/) against the golden vectors captured from the reference
(tools/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import fixture_util as fx
from oracle import qwen25vl as oq
from oracle import sc_grpo as og

torch.set_num_threads(8)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_window_and_position_indices(golden_dir):
    obj = json.load(open(os.path.join(golden_dir, "vision_index.json")))
    for c in obj["window"]:
        grids = [tuple(g) for g in c["grid_thw"]]
        wi, cu = oq.vision_window_index(grids)
        assert wi.tolist() == c["window_index"]
        assert cu == c["cu_window_seqlens"]
        assert oq.vision_position_ids(grids).tolist() == c["position_ids"]
        assert oq.vision_cu_seqlens(grids) == c["cu_seqlens"]
    for c in obj["rope_index"]:
        pos, d = oq.mrope_position_ids(torch.tensor(c["input_ids"]), torch.tensor(c["attention_mask"]), [tuple(g) for g in c["grid_thw"]], fx.TINY["image_token_id"])
        assert pos.tolist() == c["position_ids"]
        assert d.tolist() == c["rope_deltas"]


def test_forward_left_padded_batch(golden_dir):
    g = _load(golden_dir, "logps_padded.npz")
    m = oq.Qwen25VLOracle(fx.TINY, fx.make_weights(fx.TINY, 0))
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    pv = torch.from_numpy(g["pixel_values"])
    grids = [tuple(int(z) for z in r) for r in g["image_grid_thw"]]
    with torch.no_grad():
        merged, last = m.visual(pv, grids, return_last_hidden=True)
        np.testing.assert_allclose(last.numpy(), g["vit_last_hidden"], rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(merged.numpy(), g["image_embeds"], rtol=2e-4, atol=2e-4)
        pos, _ = oq.mrope_position_ids(ids, mask, grids, fx.TINY["image_token_id"])
        assert np.array_equal(pos.numpy(), g["position_ids"])
        x = m.embed(ids, merged)
        np.testing.assert_allclose(x.numpy(), g["hidden_0"], rtol=2e-4, atol=2e-4)
        h, hs = m.text_model(x, mask, pos, return_hidden=True)
        keep = mask.bool().numpy()
        np.testing.assert_allclose(hs[1].numpy()[keep], g["hidden_1"][keep], rtol=5e-4, atol=5e-4)
        np.testing.assert_allclose(h.numpy()[keep], g["hidden_last"][keep], rtol=5e-4, atol=5e-4)
        logits = h @ m.w["lm_head.weight"].t()
        np.testing.assert_allclose(logits.numpy()[keep], g["logits"][keep], rtol=1e-3, atol=1e-3)
        lp = m.per_token_logps(ids, mask, pv, grids)
    # left-pad rows: positions whose *target* is real and whose query is real
    valid = (mask[:, 1:] * mask[:, :-1]).bool().numpy()
    np.testing.assert_allclose(lp.numpy()[valid], g["per_token_logps"][valid], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("name", ["sc_grpo_g4.npz", "sc_grpo_g8.npz", "sc_grpo_g8_far.npz", "sc_grpo_trunc.npz", "sc_grpo_7b_like.npz", "sc_grpo_qwen2vl.npz"])
def test_sc_grpo_compute_loss(golden_dir, name):
    g = _load(golden_dir, name)
    meta = json.loads(str(g["meta"]))
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    cfg = getattr(fx, meta.get("config", "fixture_util.TINY").split(".")[-1])      # 7b_like: TINY7 (untied lm_head, GQA group 7)
    w_ref = fx.make_weights(cfg, 0)
    pol = oq.Qwen25VLOracle(cfg, fx.perturb_weights(w_ref, 1, scale=meta.get("perturb_scale", 0.02)), requires_grad=True)
    ref = oq.Qwen25VLOracle(cfg, w_ref)
    grid = tuple(meta["grid"])
    rows = [fx.synth_prompt(grid, meta["n_text"], cfg, seed)]
    ids, mask = fx.left_pad(rows, cfg["pad_token_id"])
    pv = fx.synth_pixel_values([grid], cfg, seed=seed)
    eos_rows = {int(k): v for k, v in meta["eos_rows"].items()}
    comps = fx.synth_completions(G, C, cfg, seed + 100, eos_rows)
    assert np.array_equal(og.right_pad(comps, cfg["pad_token_id"]).numpy(), g["completion_ids"])
    out = og.sc_grpo_step(pol, ref, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(pv), [grid], comps,
                          torch.from_numpy(g["rewards_per_func"]), G, 0.04, cfg["eos_token_id"], cfg["pad_token_id"],
                          max_prompt_length=meta.get("max_prompt_length") if meta.get("truncate") else None)
    if meta.get("truncate"):
        assert out["ids"].shape[1] == ids.shape[1] - meta["truncate"] + C          # the left truncation really cut the prompt
    assert np.array_equal(out["completion_mask"].numpy(), g["completion_mask"])
    assert np.array_equal(out["ids"].numpy(), g["prompt_completion_ids"])
    assert np.array_equal(out["mask"].numpy(), g["attention_mask"])
    np.testing.assert_allclose(out["logps"].detach().numpy(), g["per_token_logps"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(out["ref_logps"].numpy(), g["ref_per_token_logps"], rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(out["kl"].detach().numpy(), g["per_token_kl"], rtol=2e-2, atol=2e-5)
    np.testing.assert_allclose(out["advantages"].numpy(), g["advantages"], rtol=1e-5, atol=1e-6)
    assert abs(out["loss"].item() - float(g["loss"])) < 2e-6
    assert abs(out["metrics"]["kl"] - float(g["metric_kl"])) < 2e-6
    assert out["metrics"]["completion_length"] == float(g["metric_completion_length"])
    assert abs(out["metrics"]["reward"] - float(g["metric_reward"])) < 1e-6
    assert abs(out["metrics"]["reward_std"] - float(g["metric_reward_std"])) < 1e-5
    out["loss"].backward()
    grads = dict(pol.parameters())
    names = [str(n) for n in g["grad_norm_names"]]
    for n, ref_norm in zip(names, g["grad_norms"]):
        if n == "lm_head.weight" and cfg["tie_word_embeddings"]:
            continue
        got = float(grads[n].grad.norm())
        assert abs(got - ref_norm) <= 2e-3 * ref_norm + 1e-7, (n, got, ref_norm)
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(grads[k[6:]].grad.numpy(), g[k], rtol=5e-3, atol=1e-6 + 5e-3 * np.abs(g[k]).max())


def test_greedy_rollout_token_ids(golden_dir):
    g = _load(golden_dir, "greedy.npz")
    meta = json.loads(str(g["meta"]))
    cfg = fx.TINY
    m = oq.Qwen25VLOracle(cfg, fx.make_weights(cfg, 0))
    grids = [tuple(x) for x in meta["grids"]]
    pv = torch.from_numpy(fx.synth_pixel_values(grids, cfg, seed=meta["seed"]))
    seqs = m.greedy_generate(torch.from_numpy(g["prompt_ids"]), torch.from_numpy(g["prompt_mask"]), pv, grids, meta["new_tokens"])
    assert np.array_equal(seqs.numpy(), g["sequences"])  # bit-exact token ids
    # the key/value-cached form (what bench.py's cpu_baseline leg times) produces the same ids, with and without EOS handling
    cached = m.greedy_generate_cached(torch.from_numpy(g["prompt_ids"]), torch.from_numpy(g["prompt_mask"]), pv, grids, meta["new_tokens"])
    assert np.array_equal(cached.numpy(), g["sequences"])
    eos = int(g["sequences"][0, g["prompt_ids"].shape[1] + 2])
    a = m.greedy_generate(torch.from_numpy(g["prompt_ids"]), torch.from_numpy(g["prompt_mask"]), pv, grids, meta["new_tokens"], eos_token_id=eos, pad_token_id=2)
    b = m.greedy_generate_cached(torch.from_numpy(g["prompt_ids"]), torch.from_numpy(g["prompt_mask"]), pv, grids, meta["new_tokens"], eos_token_id=eos, pad_token_id=2)
    P = g["prompt_ids"].shape[1]
    first = int(np.flatnonzero(a.numpy()[0, P:] == eos)[0])
    assert np.array_equal(a.numpy()[:, : P + first + 1], b.numpy()[:, : P + first + 1])      # up to the EOS the two loops agree (after it the
    # full-recompute loop feeds pads back into the context while the cached one does not attend differently: only row 0's prefix is compared)


def _hf_groups(golden_dir, model_type, params):
    """(decay, no_decay) lists of the oracle's parameters as transformers.Trainer forms them for the family's HF model (tests/golden/sft_freeze.json:
    decay_parameters = Trainer.get_decay_parameter_names; names brought back to the classic checkpoint naming the fixtures use)."""
    d = json.load(open(os.path.join(golden_dir, "sft_freeze.json")))
    def norm(n):
        n = n[len("model."):] if n.startswith("model.visual.") else n
        return "model." + n[len("model.language_model."):] if n.startswith("model.language_model.") else n
    names = {norm(n) for n in d["decay_parameters"][model_type]}
    assert {norm(n) for n in d["parameters"][model_type]} == set(params), sorted(set(params) ^ {norm(n) for n in d["parameters"][model_type]})[:6]
    return [p for n, p in params.items() if n in names], [p for n, p in params.items() if n not in names]


def test_sft_loss_curve(golden_dir):
    g = _load(golden_dir, "sft.npz")
    meta = json.loads(str(g["meta"]))
    cfg = fx.TINY
    m = oq.Qwen25VLOracle(cfg, fx.make_weights(cfg, 0), requires_grad=True)
    grids = [tuple(int(z) for z in r) for r in g["image_grid_thw"]]
    ids, mask, labels = (torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels"))
    pv = torch.from_numpy(g["pixel_values"])
    params = dict(m.parameters())
    decay, no_decay = _hf_groups(golden_dir, "qwen2_5_vl", params)
    assert any(p.ndim < 2 for p in decay)          # the vision tower's RMSNorm gains decay under HF's name rule
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": meta["wd"]}, {"params": no_decay, "weight_decay": 0.0}], lr=meta["lr"])
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = m.sft_loss(ids, mask, labels, pv, grids)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4, atol=2e-4)


def test_sft_20_step_loss_curves(golden_dir):
    """The north star's "loss curve": 20 AdamW steps at lr 5e-5 (tests/golden/sft.npz `losses20`: Qwen2.5-VL structure, everything trained;
    qwen2vl_sft_frozen.npz `losses20`: Qwen2-VL with the reference's PA-SFT trainable set) -- the fp32 oracle follows the HF model to 1e-3 over all 20 steps."""
    for name, cfg, mt, batch_file in (("sft.npz", fx.TINY, "qwen2_5_vl", "sft.npz"), ("qwen2vl_sft_frozen.npz", fx.TINY_Q2, "qwen2_vl", "qwen2vl_sft.npz")):
        g, g0 = _load(golden_dir, name), _load(golden_dir, batch_file)
        meta = json.loads(str(g["meta"]))
        m = oq.Qwen25VLOracle(cfg, fx.make_weights(cfg, 0), requires_grad=True)
        params = dict(m.parameters())
        frozen = {n[len("model."):] if n.startswith("model.visual.") else n for n in meta.get("frozen_hf_names", [])}
        for k in frozen:
            params[k].requires_grad_(False)
        decay, no_decay = _hf_groups(golden_dir, mt, params)
        keep = {id(p) for k, p in params.items() if k not in frozen}
        opt = torch.optim.AdamW([{"params": [p for p in decay if id(p) in keep], "weight_decay": meta["wd"]},
                                 {"params": [p for p in no_decay if id(p) in keep], "weight_decay": 0.0}], lr=meta["lr20"])
        grids = [tuple(int(z) for z in r) for r in g0["image_grid_thw"]]
        ids, mask, labels = (torch.from_numpy(g0[k]) for k in ("input_ids", "attention_mask", "labels"))
        pv = torch.from_numpy(g0["pixel_values"])
        losses = []
        for _ in range(len(g["losses20"])):
            opt.zero_grad()
            loss = m.sft_loss(ids, mask, labels, pv, grids)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        dev = np.abs(np.array(losses) - g["losses20"]).max()
        print(f"[sft curve, oracle] {name}: max |dloss| over 20 steps = {dev:.2e} (loss {g['losses20'][0]:.3f} -> {g['losses20'][-1]:.3f})")
        assert len(losses) == 20 and dev < 1e-3, (name, dev)
        # the reference's mixed precision (bf16 parameters in the forward, fp32 master under AdamW: tools/make_golden.py::curve_bf16_weights), same oracle
        m = oq.Qwen25VLOracle(cfg, fx.make_weights(cfg, 0), requires_grad=True)
        params = dict(m.parameters())
        for k in frozen:
            params[k].requires_grad_(False)
        train = {k: p for k, p in params.items() if k not in frozen}
        master = {k: p.detach().clone().requires_grad_(True) for k, p in train.items()}
        decay, no_decay = _hf_groups(golden_dir, mt, params)
        dec_ids = {id(p) for p in decay}
        opt = torch.optim.AdamW([{"params": [master[k] for k, p in train.items() if id(p) in dec_ids], "weight_decay": meta["wd"]},
                                 {"params": [master[k] for k, p in train.items() if id(p) not in dec_ids], "weight_decay": 0.0}], lr=meta["lr20"])
        losses = []
        for _ in range(20):
            with torch.no_grad():
                for k, p in train.items():
                    p.copy_(master[k].to(torch.bfloat16).float())
                    p.grad = None
            loss = m.sft_loss(ids, mask, labels, pv, grids)
            loss.backward()
            for k, p in train.items():
                master[k].grad = p.grad.detach().clone()
            opt.step()
            losses.append(loss.item())
        dev = np.abs(np.array(losses) - g["losses20_bf16w"]).max()
        print(f"[sft curve, oracle, bf16 weights + fp32 master] {name}: max |dloss| = {dev:.2e}; vs the fp32 curve {np.abs(g['losses20_bf16w'] - g['losses20']).max():.2e}")
        assert dev < 1e-3, (name, dev)


def test_forward_7b_like_config(golden_dir):
    """Untied lm_head + GQA group of 7 (the structural deltas of Qwen2.5-VL-7B, BASELINE config 4)."""
    g = _load(golden_dir, "logps_7b_like.npz")
    m = oq.Qwen25VLOracle(fx.TINY7, fx.make_weights(fx.TINY7, 0))
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    grids = [tuple(int(z) for z in r) for r in g["image_grid_thw"]]
    pv = torch.from_numpy(fx.synth_pixel_values(grids, fx.TINY7, seed=91))
    with torch.no_grad():
        lp = m.per_token_logps(ids, mask, pv, grids)
    valid = (mask[:, 1:] * mask[:, :-1]).bool().numpy()
    np.testing.assert_allclose(lp.numpy()[valid], g["per_token_logps"][valid], rtol=1e-4, atol=3e-4)


def test_qwen2vl_variant(golden_dir):
    """Qwen2-VL structure (BASELINE config 1: Qwen2-VL-2B PA-SFT on 4 samples): LayerNorm / QuickGELU vision tower without
    window attention.  Golden from a tiny HF Qwen2VLForConditionalGeneration: image embeds, logps, 3-step loss curve."""
    g = _load(golden_dir, "qwen2vl_sft.npz")
    meta = json.loads(str(g["meta"]))
    cfg = fx.TINY_Q2
    m = oq.Qwen25VLOracle(cfg, fx.make_weights(cfg, 0), requires_grad=True)
    grids = [tuple(int(z) for z in r) for r in g["image_grid_thw"]]
    ids, mask, labels = (torch.from_numpy(g[k]) for k in ("input_ids", "attention_mask", "labels"))
    pv = torch.from_numpy(g["pixel_values"])
    with torch.no_grad():
        np.testing.assert_allclose(m.visual(pv, grids).numpy(), g["image_embeds"], rtol=1e-4, atol=2e-4)
        lp = m.per_token_logps(ids, mask, pv, grids)
    valid = (mask[:, 1:] * mask[:, :-1]).bool().numpy()
    np.testing.assert_allclose(lp.numpy()[valid], g["per_token_logps"][valid], rtol=1e-4, atol=3e-4)
    params = dict(m.parameters())
    decay, no_decay = _hf_groups(golden_dir, "qwen2_vl", params)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": meta["wd"]}, {"params": no_decay, "weight_decay": 0.0}], lr=meta["lr"])
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = m.sft_loss(ids, mask, labels, pv, grids)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4, atol=2e-4)


def test_qwen2vl_pa_sft_with_the_reference_trainable_set(golden_dir):
    """PA-SFT as the reference runs it for a registered composite family: vision tower + projector frozen (LLaMA-Factory defaults; tests/golden/sft_freeze.json).
    Oracle 3-step AdamW curve and gradient norms over exactly the golden's trainable tensors vs the tiny HF model (tests/golden/qwen2vl_sft_frozen.npz)."""
    g0, g = _load(golden_dir, "qwen2vl_sft.npz"), _load(golden_dir, "qwen2vl_sft_frozen.npz")
    meta = json.loads(str(g["meta"]))
    cfg = fx.TINY_Q2
    m = oq.Qwen25VLOracle(cfg, fx.make_weights(cfg, 0), requires_grad=True)
    frozen = {n[len("model."):] if n.startswith("model.visual.") else n for n in meta["frozen_hf_names"]}
    params = {k: p for k, p in dict(m.parameters()).items()}
    assert frozen and frozen <= set(params) and all(k.startswith("visual.") for k in frozen) and all(k in frozen for k in params if k.startswith("visual."))
    train = {k: p for k, p in params.items() if k not in frozen}
    for k in frozen:
        params[k].requires_grad_(False)
    grids = [tuple(int(z) for z in r) for r in g0["image_grid_thw"]]
    ids, mask, labels = (torch.from_numpy(g0[k]) for k in ("input_ids", "attention_mask", "labels"))
    pv = torch.from_numpy(g0["pixel_values"])
    decay, no_decay = _hf_groups(golden_dir, "qwen2_vl", params)
    keep = {id(p) for p in train.values()}
    opt = torch.optim.AdamW([{"params": [p for p in decay if id(p) in keep], "weight_decay": meta["wd"]},
                             {"params": [p for p in no_decay if id(p) in keep], "weight_decay": 0.0}], lr=meta["lr"])
    losses, norms = [], []
    for _ in range(3):
        opt.zero_grad()
        loss = m.sft_loss(ids, mask, labels, pv, grids)
        loss.backward()
        norms.append(float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in train.values()))))
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3)
    for k in (f for f in g.files if f.startswith("after::")):
        np.testing.assert_allclose(params[k[7:]].detach().numpy().reshape(g[k].shape), g[k], rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("cfg_name,golden", [("TINY_OV", "llava_ov.npz"), ("TINY_OV64", "llava_ov_hd64.npz")])
def test_llava_onevision_forward(golden_dir, cfg_name, golden):
    """(second case: 64-wide decoder heads, the Qwen2-0.5B structure of LLaVA-OneVision-0.5B)  oracle/llava_ov.py (SigLIP tower, projector, any-resolution packing incl. the bilinear shrink, Qwen2 decoder) vs a tiny HF
    LlavaOnevisionForConditionalGeneration (tests/golden/llava_ov.npz), and the product's host-side packing plan (iadr1_amd.llava_ov) vs both."""
    from oracle import llava_ov as oo
    import iadr1_amd  # noqa: F401
    from iadr1_amd import llava_ov as lo
    g = _load(golden_dir, golden)
    meta = json.loads(str(g["meta"]))
    cfg = getattr(fx, cfg_name)
    m = oo.LlavaOVOracle(cfg, fx.make_weights_ov(cfg, 0))
    sizes = [tuple(s) for s in meta["sizes"]]
    pv = torch.from_numpy(fx.synth_crops(meta["crops"], cfg, meta["seed"]))
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    with torch.no_grad():
        feats = m.visual(pv, sizes)
        np.testing.assert_allclose(feats.numpy(), g["image_features"], rtol=1e-3, atol=2e-4)
        lg = m.logits(ids, mask, pv, sizes)
        np.testing.assert_allclose(lg[:, -1].numpy(), g["logits_last"], rtol=1e-3, atol=1e-3)
        lp = m.per_token_logps(ids, mask, pv, sizes)
    valid = (mask[:, 1:] * mask[:, :-1]).bool().numpy()
    np.testing.assert_allclose(lp.numpy()[valid], g["per_token_logps"][valid], rtol=1e-4, atol=3e-4)
    # host plan of the product: token counts per image, crop counts, and the sparse map itself applied to the oracle's projector rows
    v = cfg["vision"]
    side = v["image_size"] // v["patch_size"]
    plan = lo.pack_plan(sizes, cfg["image_grid_pinpoints"], v["image_size"], side, cfg["anyres_max"])
    assert plan["lens"] == g["feature_lens"].tolist() and sum(plan["crops"]) == meta["crops"]
    assert [int((ids[b] == cfg["image_token_id"]).sum()) for b in range(2)] == plan["lens"]
    with torch.no_grad():
        src = torch.cat([m.project(m.tower(pv)).reshape(-1, cfg["text"]["hidden_size"]), m.w["image_newline"][None]], 0)
    T = len(plan["ptr"]) - 1
    rows = np.repeat(np.arange(T), np.diff(plan["ptr"]))
    packed = torch.zeros(T, src.shape[1]).index_add_(0, torch.from_numpy(rows), src[torch.from_numpy(plan["idx"]).long()] * torch.from_numpy(plan["w"])[:, None])
    np.testing.assert_allclose(packed.numpy(), g["image_features"], rtol=1e-3, atol=2e-4)
    tp = lo.transpose_plan(plan)                                  # the backward map is the exact transpose
    dense = np.zeros((T, plan["n_src"])); dense[rows, plan["idx"]] += plan["w"]
    rows_t = np.repeat(np.arange(plan["n_src"]), np.diff(tp["ptr"]))
    dense_t = np.zeros((plan["n_src"], T)); dense_t[rows_t, tp["idx"]] += tp["w"]
    assert np.array_equal(dense.T, dense_t)


@pytest.mark.parametrize("golden", ["sc_grpo_llava_ov.npz", "sc_grpo_llava_ov_g7.npz"])
def test_llava_onevision_sc_grpo_compute_loss(golden_dir, golden):
    """The reference's compute_loss on its llava branch (model id containing "llava_ov": `_ensure_left_padding_data` rotates the rows whose completion
    ended early, REF:502-504,516-567) vs oracle.sc_grpo.sc_grpo_step(rotate_right_padded_rows=True).  _g7: the 7:1 head geometry of LLaVA-OneVision-7B's
    decoder (BASELINE config 5), fixture TINY_OV7."""
    from oracle import llava_ov as oo
    g = _load(golden_dir, golden)
    meta = json.loads(str(g["meta"]))
    cfg = getattr(fx, (meta.get("config") or "fixture_util.TINY_OV").split(".")[-1] or "TINY_OV")
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    w_ref = fx.make_weights_ov(cfg, 0)
    pol = oo.LlavaOVOracle(cfg, fx.perturb_weights(w_ref, 1, scale=meta["perturb_scale"]), requires_grad=True)
    ref = oo.LlavaOVOracle(cfg, w_ref)
    sizes = [tuple(s) for s in meta["sizes"]]
    P = g["prompt_completion_ids"].shape[1] - C
    ids, mask = g["prompt_completion_ids"][:1, :P], g["attention_mask"][:1, :P]
    pv = torch.from_numpy(fx.synth_crops(meta["crops"], cfg, seed))
    comps = fx.synth_completions(G, C, cfg, seed + 100, {int(k): v for k, v in meta["eos_rows"].items()})
    out = og.sc_grpo_step(pol, ref, torch.from_numpy(ids), torch.from_numpy(mask), pv, sizes, comps, torch.from_numpy(g["rewards_per_func"]), G, 0.04,
                          cfg["eos_token_id"], cfg["pad_token_id"], rotate_right_padded_rows=True)
    assert np.array_equal(out["ids"].numpy(), g["prompt_completion_ids"]) and np.array_equal(out["completion_mask"].numpy(), g["completion_mask"])
    np.testing.assert_allclose(out["logps"].detach().numpy(), g["per_token_logps"], rtol=1e-4, atol=3e-4)
    np.testing.assert_allclose(out["ref_logps"].numpy(), g["ref_per_token_logps"], rtol=1e-4, atol=3e-4)
    np.testing.assert_allclose(out["advantages"].numpy(), g["advantages"], rtol=1e-5, atol=1e-6)
    assert abs(out["loss"].item() - float(g["loss"])) < 5e-6 and abs(out["metrics"]["kl"] - float(g["metric_kl"])) < 5e-5
    # without the rotation the rows that ended early differ: the quirk is really exercised by this fixture
    plain = og.sc_grpo_step(oo.LlavaOVOracle(cfg, fx.perturb_weights(w_ref, 1, scale=meta["perturb_scale"])), ref, torch.from_numpy(ids), torch.from_numpy(mask), pv, sizes, comps,
                            torch.from_numpy(g["rewards_per_func"]), G, 0.04, cfg["eos_token_id"], cfg["pad_token_id"])
    assert abs(plain["loss"].item() - float(g["loss"])) > 1e-4
    out["loss"].backward()
    grads = dict(pol.parameters())
    for n, ref_norm in zip([str(n) for n in g["grad_norm_names"]], g["grad_norms"]):
        if ref_norm > 1e-9:
            assert abs(float(grads[n].grad.norm()) - ref_norm) <= 3e-3 * ref_norm + 1e-7, n


_LLAVA_CFG = {"llava": ("TINY_LLAVA15", "llava15.npz", "sc_grpo_llava15.npz"), "llava_next": ("TINY_LLAVA_NEXT", "llava_next.npz", "sc_grpo_llava_next.npz")}


@pytest.mark.parametrize("family", ["llava", "llava_next"])
def test_llava15_and_next_forward(golden_dir, family):
    """oracle.llava_ov.LlavaOracle (CLIP tower with class token / pre-LayerNorm / QuickGELU / feature layer -2 without the class token, projector, NeXT's
    any-resolution packing, LLaMA / Mistral decoder without q/k/v biases) vs tiny HF LlavaForConditionalGeneration / LlavaNextForConditionalGeneration."""
    from oracle import llava_ov as oo
    cfg_name, gname, _ = _LLAVA_CFG[family]
    cfg = getattr(fx, cfg_name)
    g = _load(golden_dir, gname)
    meta = json.loads(str(g["meta"]))
    m = oo.LlavaOracle(cfg, fx.make_weights_llava(cfg, 0))
    sizes = [tuple(s) for s in meta["sizes"]]
    pv = torch.from_numpy(fx.synth_crops(meta["crops"], cfg, meta["seed"]))
    ids, mask = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["attention_mask"])
    with torch.no_grad():
        np.testing.assert_allclose(m.visual(pv, sizes).numpy(), g["image_features"], rtol=1e-3, atol=2e-4)
        np.testing.assert_allclose(m.logits(ids, mask, pv, sizes)[:, -1].numpy(), g["logits_last"], rtol=1e-3, atol=1e-3)
        lp = m.per_token_logps(ids, mask, pv, sizes)
    valid = (mask[:, 1:] * mask[:, :-1]).bool().numpy()
    np.testing.assert_allclose(lp.numpy()[valid], g["per_token_logps"][valid], rtol=1e-4, atol=3e-4)
    assert {k for k, _ in m.parameters()} == set(fx.param_shapes_llava(cfg))            # the zero q/k/v biases of the restatement are not parameters
    if family == "llava_next":      # the product's host plan (no shrink step) reserves the same number of tokens per image
        import iadr1_amd  # noqa: F401
        from iadr1_amd import llava_ov as lo
        v = cfg["vision"]
        plan = lo.pack_plan(sizes, cfg["image_grid_pinpoints"], v["image_size"], v["image_size"] // v["patch_size"], None)
        assert plan["lens"] == g["feature_lens"].tolist() and sum(plan["crops"]) == meta["crops"]


@pytest.mark.parametrize("family", ["llava", "llava_next"])
def test_llava15_and_next_sc_grpo_compute_loss(golden_dir, family):
    """The reference's compute_loss under the model ids its switch routes to LLaVA-1.5 / LLaVA-NeXT (REF:130-135; `"llava" in model_id` also sends them through
    `_ensure_left_padding_data`, REF:502-504) vs oracle.sc_grpo.sc_grpo_step(rotate_right_padded_rows=True)."""
    from oracle import llava_ov as oo
    cfg_name, _, gname = _LLAVA_CFG[family]
    cfg = getattr(fx, cfg_name)
    g = _load(golden_dir, gname)
    meta = json.loads(str(g["meta"]))
    G, C, seed = meta["G"], meta["C"], meta["seed"]
    w_ref = fx.make_weights_llava(cfg, 0)
    pol = oo.LlavaOracle(cfg, fx.perturb_weights(w_ref, 1, scale=meta["perturb_scale"]), requires_grad=True)
    ref = oo.LlavaOracle(cfg, w_ref)
    sizes = [tuple(s) for s in meta["sizes"]]
    P = g["prompt_completion_ids"].shape[1] - C
    ids, mask = g["prompt_completion_ids"][:1, :P], g["attention_mask"][:1, :P]
    pv = torch.from_numpy(fx.synth_crops(meta["crops"], cfg, seed))
    comps = fx.synth_completions(G, C, cfg, seed + 100, {int(k): v for k, v in meta["eos_rows"].items()})
    out = og.sc_grpo_step(pol, ref, torch.from_numpy(ids), torch.from_numpy(mask), pv, sizes, comps, torch.from_numpy(g["rewards_per_func"]), G, 0.04,
                          cfg["eos_token_id"], cfg["pad_token_id"], rotate_right_padded_rows=True)
    assert np.array_equal(out["ids"].numpy(), g["prompt_completion_ids"]) and np.array_equal(out["completion_mask"].numpy(), g["completion_mask"])
    np.testing.assert_allclose(out["logps"].detach().numpy(), g["per_token_logps"], rtol=1e-4, atol=3e-4)
    np.testing.assert_allclose(out["ref_logps"].numpy(), g["ref_per_token_logps"], rtol=1e-4, atol=3e-4)
    assert abs(out["loss"].item() - float(g["loss"])) < 5e-6 and abs(out["metrics"]["kl"] - float(g["metric_kl"])) < 5e-5
    out["loss"].backward()
    grads = dict(pol.parameters())
    for n, ref_norm in zip([str(n) for n in g["grad_norm_names"]], g["grad_norms"]):
        if ref_norm > 1e-9:
            assert abs(float(grads[n].grad.norm()) - ref_norm) <= 3e-3 * ref_norm + 1e-7, n


def test_sampler_candidates_match_transformers_warpers(golden_dir):
    """Third-party pin of the rollout sampler's filter (REF sc_grpo_trainer.py:353-358: temperature, top_k=50, top_p=0.9): the kept index set and
    the renormalised probabilities of oracle.sampler.candidates against transformers' TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper
    on fixed logits (tools/make_golden_sampler.py).  Sets bit-exact, probabilities to 2e-6 (fp32 softmax)."""
    from oracle import sampler as osamp
    g = np.load(os.path.join(golden_dir, "sampler_hf.npz"))
    for i in range(len(g["vocab"])):
        t, k, p = g["settings"][i]
        lg = g["logits"][i, : int(g["vocab"][i])]
        ids, w = osamp.candidates(lg, float(t), int(k), float(p))
        n = int(g["count"][i])
        want_ids, want_p = g["ids"][i, :n], g["probs"][i, :n]
        assert set(ids.tolist()) == set(want_ids.tolist()), f"row {i}: candidate set differs from transformers'"
        pr = dict(zip(ids.tolist(), (w / w.sum(dtype=np.float32)).tolist()))
        assert max(abs(pr[int(a)] - float(b)) for a, b in zip(want_ids, want_p)) < 2e-6, f"row {i}"
