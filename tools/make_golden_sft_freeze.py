#!/usr/bin/env python3
"""tests/golden/sft_freeze.json + qwen2vl_sft_frozen.npz: which parameters the reference's PA-SFT leaves trainable.
The launch scripts pass `--finetuning_type full` and no freeze flag, so LLaMA-Factory's defaults apply (hparams/finetuning_args.py:416-427: freeze_vision_tower = True,
freeze_multi_modal_projector = True, train_mm_proj_only = False): `_setup_full_tuning` (model/adapter.py:39-55) clears requires_grad of every parameter whose name
contains one of `get_forbidden_modules(config, finetuning_args)` (model/model_utils/visual.py:153-171), and that set is empty for model types the vendored
LLaMA-Factory has not registered (visual.py:236-288: qwen2_vl, llava, llava_next are; qwen2_5_vl and llava_onevision are not).
Runs the reference's own get_forbidden_modules (build container only; peft / trl stubbed at import) on the model types of the launch scripts and on the HF parameter
names of the tiny fixture models, and a 3-step AdamW curve of a tiny Qwen2-VL with exactly that trainable set."""
import importlib.machinery, json, os, sys, types

import numpy as np
import torch
from transformers import Trainer  # noqa: F401  (before the peft stub below: transformers probes for the real package)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m


stub("peft", __version__="0.0")
stub("peft.utils", SAFETENSORS_WEIGHTS_NAME="a", WEIGHTS_NAME="b")
stub("peft.tuners")
stub("peft.tuners.lora", LoraLayer=object)
stub("trl", __version__="0.0")
sys.path.insert(0, "/root/reference/train/stage_sft")
import llamafactory  # noqa: E402
# llamafactory/model/__init__.py pulls in its loader, which needs a transformers class this container's version no longer has; the freezing rule lives in a leaf
# module, so the package object is supplied here and only that leaf (and what it imports) is executed
_pkg = types.ModuleType("llamafactory.model")
_pkg.__path__ = [os.path.join(os.path.dirname(llamafactory.__file__), "model")]
_pkg.__spec__ = importlib.machinery.ModuleSpec("llamafactory.model", None, is_package=True)
sys.modules["llamafactory.model"] = _pkg
from llamafactory.model.model_utils.visual import get_forbidden_modules  # noqa: E402

sys.path = [p for p in sys.path if p != "/root/reference/train/stage_sft"]
import fixture_util as fx  # noqa: E402
import make_golden as mg  # noqa: E402
import make_golden_llava as mgl  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FLAGS = [dict(freeze_vision_tower=True, freeze_multi_modal_projector=True, train_mm_proj_only=False),      # the defaults = what every launch script runs
         dict(freeze_vision_tower=False, freeze_multi_modal_projector=True, train_mm_proj_only=False),
         dict(freeze_vision_tower=True, freeze_multi_modal_projector=False, train_mm_proj_only=False),
         dict(freeze_vision_tower=False, freeze_multi_modal_projector=False, train_mm_proj_only=False),
         dict(freeze_vision_tower=False, freeze_multi_modal_projector=True, train_mm_proj_only=True)]


def finetuning_args(**kw):
    a = types.SimpleNamespace(**kw)
    # FinetuningArguments.__post_init__ (hparams/finetuning_args.py:458-459)
    a.freeze_vision_tower = a.freeze_vision_tower or a.train_mm_proj_only
    a.freeze_multi_modal_projector = a.freeze_multi_modal_projector and not a.train_mm_proj_only
    return a


def frozen_names(model_type, names, **kw):
    forb = get_forbidden_modules(types.SimpleNamespace(model_type=model_type), finetuning_args(**kw))
    return sorted(forb), [n for n in names if any(f in n for f in forb)]      # adapter.py:49-55


def main():
    torch.manual_seed(0)
    models = {
        "qwen2_vl": mg.build_hf_model(fx.TINY_Q2, fx.make_weights(fx.TINY_Q2, seed=0)),
        "qwen2_5_vl": mg.build_hf_model(fx.TINY, fx.make_weights(fx.TINY, seed=0)),
        "llava_onevision": mgl.build_hf(fx.TINY_OV, fx.make_weights_ov(fx.TINY_OV, seed=0)),
        "llava": mgl.build_hf_llava(fx.TINY_LLAVA15, fx.make_weights_llava(fx.TINY_LLAVA15, seed=0)),
        "llava_next": mgl.build_hf_llava(fx.TINY_LLAVA_NEXT, fx.make_weights_llava(fx.TINY_LLAVA_NEXT, seed=0)),
    }
    cases = []
    for mt, model in models.items():
        if model is None:
            continue
        names = [n for n, _ in model.named_parameters()]
        for fl in FLAGS:
            forb, frozen = frozen_names(mt, names, **fl)
            cases.append({"model_type": mt, "flags": fl, "forbidden_modules": forb, "n_parameters": len(names), "frozen": frozen})
    # the optimizer's parameter groups: transformers.Trainer.get_decay_parameter_names of the installed version (the weight-decay group; everything else gets 0) --
    # LLaMA-Factory's CustomSeq2SeqTrainer and trl's GRPOTrainer both fall through to Trainer.create_optimizer
    decay = {mt: sorted(Trainer.get_decay_parameter_names(Trainer.__new__(Trainer), model)) for mt, model in models.items()}
    allp = {mt: [n for n, _ in model.named_parameters()] for mt, model in models.items()}
    json.dump({"meta": {**mg.meta(), "source": "llamafactory/model/model_utils/visual.py:153-171,236-288 + model/adapter.py:39-55 through tools/make_golden_sft_freeze.py; "
                                              "decay_parameters: transformers.Trainer.get_decay_parameter_names on the tiny fixture models"},
               "cases": cases, "decay_parameters": decay, "parameters": allp}, open(os.path.join(OUT, "sft_freeze.json"), "w"), indent=0)
    for mt in decay:
        nd = [n for n in allp[mt] if n not in set(decay[mt])]
        print(mt, "decay", len(decay[mt]), "no-decay", len(nd), "| 1-D decayed:", [n for n, p in models[mt].named_parameters() if p.ndim < 2 and n in set(decay[mt])][:6])
    for c in cases:
        print(c["model_type"], c["flags"], c["forbidden_modules"], "frozen", None if c["frozen"] is None else len(c["frozen"]), "of", c["n_parameters"])

    # the 3-step curve of BASELINE config 1's structure (Qwen2-VL) with the default trainable set: same batch as qwen2vl_sft.npz
    cfg = fx.TINY_Q2
    model = models["qwen2_vl"].train()
    _, frozen = frozen_names("qwen2_vl", [n for n, _ in model.named_parameters()], **FLAGS[0])
    for n, p in model.named_parameters():
        if n in frozen:
            p.requires_grad_(False)
    g0 = np.load(os.path.join(OUT, "qwen2vl_sft.npz"))
    ids, mask, labels = (torch.from_numpy(g0[k]) for k in ("input_ids", "attention_mask", "labels"))
    inputs = dict(input_ids=ids, attention_mask=mask, pixel_values=torch.from_numpy(g0["pixel_values"]), image_grid_thw=torch.from_numpy(g0["image_grid_thw"]),
                  mm_token_type_ids=(ids == cfg["image_token_id"]).int(), labels=labels)
    dnames = set(Trainer.get_decay_parameter_names(Trainer.__new__(Trainer), model))
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if p.requires_grad:
            (decay if n in dnames else no_decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.1}, {"params": no_decay, "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses, gnorms = [], []
    for _ in range(3):
        opt.zero_grad()
        loss = model(**inputs).loss
        loss.backward()
        gnorms.append(float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None))))
        opt.step()
        losses.append(loss.item())
    moved = {n: float((p.detach() - before[n]).abs().max()) for n, p in model.named_parameters()}
    assert all(moved[n] == 0.0 for n in frozen) and all(v > 0 for n, v in moved.items() if n not in frozen)
    inv = {mg.hf_name(k): k for k in fx.param_shapes(cfg)}
    after = {inv[n]: p.detach().numpy() for n, p in model.named_parameters() if n in inv and inv[n] in ("model.norm.weight", "model.layers.1.self_attn.k_proj.bias", "visual.merger.mlp.2.bias", "visual.blocks.0.attn.qkv.bias")}
    # 20-step curve with the same trainable set from the same start (lr as tools/make_golden.py LR20)
    model = mg.build_hf_model_qwen2vl(cfg, fx.make_weights(cfg, seed=0)).train()
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if n in frozen:
            p.requires_grad_(False)
        else:
            (decay if n in dnames else no_decay).append(p)
    opt = torch.optim.AdamW([{"params": decay, "weight_decay": 0.1}, {"params": no_decay, "weight_decay": 0.0}], lr=mg.LR20, betas=(0.9, 0.999), eps=1e-8)
    losses20 = []
    for _ in range(20):
        opt.zero_grad()
        loss = model(**inputs).loss
        loss.backward()
        opt.step()
        losses20.append(loss.item())
    def groups(mdl):
        dn = set(Trainer.get_decay_parameter_names(Trainer.__new__(Trainer), mdl))
        return [p for n, p in mdl.named_parameters() if n in dn], [p for n, p in mdl.named_parameters() if n not in dn]
    losses20_bf16w = mg.curve_bf16_weights(mg.build_hf_model_qwen2vl(cfg, fx.make_weights(cfg, seed=0)).train(), inputs, groups, trainable=lambda n: n not in frozen)
    np.savez_compressed(os.path.join(OUT, "qwen2vl_sft_frozen.npz"),
                        meta=json.dumps({**mg.meta(), "batch": "qwen2vl_sft.npz", "lr": 1e-3, "wd": 0.1, "lr20": mg.LR20, "flags": FLAGS[0], "frozen_hf_names": frozen}),
                        losses=np.array(losses, dtype=np.float64), grad_norms=np.array(gnorms, dtype=np.float64), losses20=np.array(losses20, dtype=np.float64),
                        losses20_bf16w=np.array(losses20_bf16w, dtype=np.float64), **{"after::" + k: v for k, v in after.items()})
    print("  bf16 weights + fp32 master", [round(x, 4) for x in losses20_bf16w])
    print("qwen2vl_sft_frozen.npz: losses", losses, "grad norms", gnorms, "frozen", len(frozen), "tensors; kept", sorted(after), "\n  20 steps", [round(x, 4) for x in losses20])


if __name__ == "__main__":
    main()
