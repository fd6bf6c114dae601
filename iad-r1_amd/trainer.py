"""`SCGRPOTrainer`: the drop-in trainer API of the reference's SC-GRPO stage
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:72-90 constructor, :586 compute_loss, :821 log;
used by /root/reference/train/stage_rl/grpo_ad.py:188-207) on top of the HIP engine.

What is kept: constructor signature and error behaviour, the reward-function plugin signature
`f(prompts=, completions=, current_step=, **dataset_columns) -> list[float]` (REF:773-781), the metric keys
(`completion_length`, `rewards/<name>`, `reward`, `reward_std`, `kl`; REF:801-827), `train()`, `save_model()`.
What is replaced: transformers.Trainer + DeepSpeed ZeRO-3 + the vLLM side GPU -> one process per MI355X with the
whole model resident, in-process hipGraph rollout, RCCL all-reduce of a flat gradient buffer.
Host-side tokenisation / image patching stays with the HF processor (CPU plumbing, REF:184-227, 600-622).
"""
from __future__ import annotations

import json
import math
import os
import threading
import time
from collections import defaultdict
from dataclasses import dataclass, field
from typing import Any, Callable, Optional, Union

import numpy as np
import torch

from . import schedule
from .params import ParamStore, VLMConfig
from .sc_grpo import GRPOArgs, SCGRPOEngine


@dataclass
class GRPOConfig:
    """Flag surface of the reference's `GRPOConfig` (train/stage_rl/configs.py:24-42 over trl's GRPOConfig) that
    the SC-GRPO scripts set (scripts/train/SC_GRPO/*.sh:40-63) or whose defaults matter.  Unknown / unused
    reference flags are accepted by the CLI and ignored."""
    output_dir: str = "outputs"
    per_device_train_batch_size: int = 8     # (transformers TrainingArguments; every launch script passes 1)
    gradient_accumulation_steps: int = 1
    num_generations: int = 8
    max_prompt_length: Optional[int] = 512
    max_completion_length: int = 256
    beta: float = 0.04
    temperature: float = 0.9
    learning_rate: float = 1e-6
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0
    lr_scheduler_type: str = "linear"
    warmup_steps: int = 0
    num_train_epochs: float = 3.0       # transformers TrainingArguments' default, like logging_steps / save_steps = 500 below; every SC_GRPO_*.sh passes
                                        # --num_train_epochs 1 --logging_steps 1 --save_steps 100 --per_device_train_batch_size 1 (tests/golden/launch_flags.json)
    max_steps: int = -1
    logging_steps: int = 500
    save_steps: int = 500
    seed: int = 42
    bf16: bool = True
    gradient_checkpointing: bool = False   # True (every reference script): decoder activations are recomputed in backward WHEN they would not fit comfortably
                                           # in HBM (vlm.Engine.recompute_wanted: "auto"); 3B at the bench shape keeps them, 7B at 20 480 token rows recomputes
    model_init_kwargs: Optional[dict] = None
    micro_batch_seqs: int = 64      # sequences per reference / policy pass; >= batch x group lets the rollout double as the policy's training forward
    shuffle: bool = True            # seeded per-epoch permutation (HF Trainer's RandomSampler / DistributedSampler)
    # one group rollout for ALL micro-batches of an optimizer step (the policy does not change between them: the reference syncs its vLLM weights once per
    # optimizer step, REF:637-641): a decode step streams the whole model whatever the number of sequences, so gradient_accumulation_steps rollouts of
    # B x G sequences cost that many times one rollout of accum x B x G -- the launch scripts' B = 1, G = 4, accum 2 (training_step; IADR1_BATCH_ROLLOUTS=0 = per micro-batch).
    # What it trades: the micro-batches then arrive with their completions already rolled out, so the rollout -> training hand-over is off (the policy forward over
    # the completions runs before each backward) and the vision tower runs once for the combined rollout and once per micro-batch; the sampling seed is per optimizer
    # step, not per micro-batch, so token streams differ between the two modes.  Applies only while the step's sequences fit one 64-row decode tile.
    batch_rollouts: bool = os.environ.get("IADR1_BATCH_ROLLOUTS", "1") != "0"
    prefetch_batches: bool = os.environ.get("IADR1_PREFETCH", "1") != "0"    # prepare micro-batch k+1 on a worker thread while the GPU runs k (iadr1_amd.prefetch)
    run_name: Optional[str] = None
    report_to: Any = None
    push_to_hub: bool = False
    eval_strategy: str = "no"


def load_checkpoint(path: str, device, trainable: bool, with_decode_pack=None):
    """HF Qwen2.5-VL / Qwen2-VL checkpoint directory (config.json + *.safetensors) -> (VLMConfig, ParamStore).
    with_decode_pack=True also builds the decode-packed weights a frozen copy needs to generate (evaluation)."""
    from safetensors import safe_open

    with open(os.path.join(path, "config.json")) as f:
        cfg = VLMConfig.from_hf_config(json.load(f))
    store = ParamStore(cfg, device, trainable=trainable, with_decode_pack=with_decode_pack)
    sd = {}
    files = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    for fn in files:
        with safe_open(os.path.join(path, fn), framework="pt") as sf:
            for k in sf.keys():
                name = k
                # transformers>=5 layout -> classic checkpoint names (the LLaVA-OneVision store normalises its own names: ParamStore._llava_canonical)
                if not cfg.is_llava:
                    if name.startswith("model.visual."):
                        name = name[len("model."):]
                    elif name.startswith("model.language_model."):
                        name = "model." + name[len("model.language_model."):]
                sd[name] = sf.get_tensor(k)
    store.load_named(sd)
    return cfg, store


def save_checkpoint(store: ParamStore, path: str, hf_config: Optional[dict] = None):
    """Writes HF-layout safetensors (names as the reference's vLLM eval scripts expect) + config.json."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    sd = {k: v.to(torch.bfloat16).contiguous() for k, v in store.export_named().items()}
    save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    if hf_config is not None:
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(hf_config, f, indent=2)


def average_over_ranks(values: dict) -> dict:
    """Mean over the ranks of a data-parallel job of a small {name: float} dict (sorted keys, one all-reduce); identity in a single process.
    The second (and last) collective of the path next to the gradient exchange."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or not values:
        return dict(values)
    keys = sorted(values)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(values[k]) for k in keys], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    t /= dist.get_world_size()
    return {k: float(v) for k, v in zip(keys, t.tolist())}


def _layout_digest(store: ParamStore) -> str:
    import hashlib
    h = hashlib.sha1()
    for name in sorted(store.slots):
        sl = store.slots[name]
        h.update(f"{name}:{sl.offset}:{tuple(sl.shape)};".encode())
    h.update(str(store.n_total).encode())
    return h.hexdigest()


def save_training_state(store: ParamStore, path: str, opt_step: int, global_step: int, extra: Optional[dict] = None):
    """What a run needs to continue bit-exactly, next to the HF-layout weights of `save_checkpoint`: the fp32 master weights and both Adam moments
    (flat, in the ParamStore layout) as optimizer.safetensors, and trainer_state.json (global_step, the optimizer's step counter -- bias correction and
    the rollout seed derive from it -- and the layout digest the tensors are only valid for).  The reference gets this from transformers.Trainer
    (_save_checkpoint TF:trainer.py:3079: optimizer.pt, scheduler.pt, trainer_state.json); the LR schedule here is a pure function of global_step."""
    from safetensors.torch import save_file

    os.makedirs(path, exist_ok=True)
    save_file({"master": store.master.detach().cpu(), "exp_avg": store.m.detach().cpu(), "exp_avg_sq": store.v.detach().cpu()},
              os.path.join(path, "optimizer.safetensors"), metadata={"format": "pt", "layout": _layout_digest(store)})
    with open(os.path.join(path, "trainer_state.json"), "w") as f:
        json.dump({"global_step": int(global_step), "opt_step": int(opt_step), "layout": _layout_digest(store), "n_total": int(store.n_total), **(extra or {})}, f, indent=2)


def load_training_state(store: ParamStore, path: str) -> dict:
    """Inverse of save_training_state on a ParamStore of the same configuration: restores master weights / moments, re-derives the bf16 parameters and
    their transposed / decode-packed copies from the masters, returns the trainer_state.json dict."""
    from safetensors import safe_open

    with open(os.path.join(path, "trainer_state.json")) as f:
        state = json.load(f)
    if state.get("layout") != _layout_digest(store) or state.get("n_total") != store.n_total:
        raise ValueError(f"{path}: optimizer state was written for a different parameter layout")
    with safe_open(os.path.join(path, "optimizer.safetensors"), framework="pt") as sf:
        store.master.copy_(sf.get_tensor("master"))
        store.m.copy_(sf.get_tensor("exp_avg"))
        store.v.copy_(sf.get_tensor("exp_avg_sq"))
    store.flat.copy_(store.master.to(torch.bfloat16))      # round-to-nearest-even, as the optimizer kernel writes them
    store.refresh_shadows()
    return state


def last_checkpoint(output_dir: str) -> Optional[str]:
    """Highest-numbered <output_dir>/checkpoint-N that holds a training state (the reference's auto-resume rule, llamafactory hparams/parser.py:332-354
    -> transformers.trainer_utils.get_last_checkpoint)."""
    best = None
    if os.path.isdir(output_dir):
        for d in os.listdir(output_dir):
            if d.startswith("checkpoint-") and d[11:].isdigit() and os.path.exists(os.path.join(output_dir, d, "trainer_state.json")):
                if best is None or int(d[11:]) > int(best[11:]):
                    best = d
    return os.path.join(output_dir, best) if best else None


def maybe_apply_chat_template(example: dict, processing_class) -> str:
    """trl.data_utils.maybe_apply_chat_template for the prompt-only rows SC-GRPO feeds it (REF:600 -> trl/trl/data_utils.py:172-227 -> :71-169):
    a conversational prompt (list of {"role", "content"} messages) is rendered with the processor's chat template and the generation prompt
    appended, untokenised; a plain-string prompt passes through unchanged."""
    p = example["prompt"]
    if isinstance(p, list) and p and isinstance(p[0], dict) and "role" in p[0] and "content" in p[0]:       # trl is_conversational
        return processing_class.apply_chat_template(p, add_generation_prompt=True, tokenize=False)
    return p


def prepare_batch(processing_class, inputs: list[dict]) -> dict:
    """Dataset rows -> the prompt tensors of one micro-batch, the reference's host-side steps REF:600-622: chat template per example, PIL open of every
    image path, ONE processor call with left padding and no extra special tokens.  Pure host code (no device access)."""
    from PIL import Image
    prompts_text = [maybe_apply_chat_template(ex, processing_class) for ex in inputs]
    images, per_prompt = [], []
    for ex in inputs:
        im = ex.get("image")
        im = im if isinstance(im, list) else ([im] if im is not None else [])
        per_prompt.append(len(im))
        images += [Image.open(i) if isinstance(i, str) else i for i in im]
    enc = processing_class(text=prompts_text, images=images if images else None, return_tensors="pt", padding=True, padding_side="left", add_special_tokens=False)
    out = {"input_ids": np.asarray(enc["input_ids"]), "attention_mask": np.asarray(enc["attention_mask"]), "pixel_values": enc["pixel_values"], "images_per_prompt": per_prompt}
    if "image_sizes" in enc:            # LLaVA-OneVision / LLaVA-NeXT processors: crops [images, max crops, 3, S, S] + original (height, width) per image
        out["image_sizes"] = np.asarray(enc["image_sizes"]).reshape(-1, 2).tolist()
    elif "image_grid_thw" in enc:       # Qwen2-VL / Qwen2.5-VL processors: patch rows + (t, h, w) grids
        grid = enc["image_grid_thw"]
        out["image_grid_thw"] = grid.tolist() if hasattr(grid, "tolist") else [tuple(g) for g in grid]
    # (LLaVA-1.5 processor: one resized crop per image [images, 3, S, S], nothing else)
    return out


def trim_completions(toks: np.ndarray, eos_token_id: int) -> np.ndarray:
    """[N, C] completion ids (pad after the first EOS) cut to the longest completion of these rows -- what the reference's right-padding of the vLLM outputs
    yields (REF:680-683: `pad(completion_ids)` to the longest of the micro-batch); the columns dropped are masked for every row."""
    is_eos = toks == eos_token_id
    n = np.where(is_eos.any(1), is_eos.argmax(1) + 1, toks.shape[1])
    return toks[:, : max(1, int(n.max()))]


def combine_batches(batches: list[dict], pad_token_id: int) -> Optional[dict]:
    """The prompts of several micro-batches as ONE batch (for one group rollout over all of them, SCGRPOTrainer.training_step): ids / masks left-padded to the
    longest prompt, image tensors and their grids / sizes concatenated in prompt order.  None when the pixel tensors cannot be concatenated (the any-resolution
    LLaVA processors pad the crop dimension per call)."""
    to_np = lambda x: x.cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
    ids = [to_np(b["input_ids"]) for b in batches]
    masks = [to_np(b["attention_mask"]) for b in batches]
    S = max(i.shape[1] for i in ids)
    padl = lambda a, v: np.concatenate([np.full((a.shape[0], S - a.shape[1]), v, dtype=a.dtype), a], 1)
    px = [b["pixel_values"] for b in batches]
    if any(p is None for p in px) or len({tuple(p.shape[1:]) for p in px}) != 1 or len({p.device for p in px if isinstance(p, torch.Tensor)}) > 1:
        return None
    out = {"input_ids": np.concatenate([padl(i, pad_token_id) for i in ids], 0), "attention_mask": np.concatenate([padl(m, 0) for m in masks], 0),
           "pixel_values": torch.cat([p if isinstance(p, torch.Tensor) else torch.as_tensor(p) for p in px], 0),
           "images_per_prompt": [n for b, i in zip(batches, ids) for n in (b.get("images_per_prompt") or [1] * i.shape[0])]}
    for key in ("image_grid_thw", "image_sizes"):
        if any(key in b for b in batches):
            if not all(key in b for b in batches):
                return None
            rows = [to_np(b[key]) for b in batches]
            out[key] = [tuple(int(z) for z in g) for r in rows for g in r.reshape(-1, r.shape[-1])]
    return out


def resolve_reward_funcs(reward_funcs, reward_processing_classes, model_init_kwargs: Optional[dict] = None, device=None):
    """The reference's reward-function set-up (REF:228-262, 303-305): a string is a sequence-classification checkpoint (`num_labels=1`), a torch
    module is a reward MODEL that needs a tokenizer (default: the one of its checkpoint; pad token defaults to EOS; the model's `pad_token_id` is set to
    the tokenizer's, because such models score the last non-pad token), everything else is the callable plug-in API of SURVEY section 8(b).3.
    Reward models are the caller's torch modules: they run through torch on `device`, in eval mode, outside the HIP hot path."""
    funcs = list(reward_funcs) if isinstance(reward_funcs, (list, tuple)) else [reward_funcs]
    for i, f in enumerate(funcs):
        if isinstance(f, str):
            from transformers import AutoModelForSequenceClassification
            funcs[i] = AutoModelForSequenceClassification.from_pretrained(f, num_labels=1, **(model_init_kwargs or {}))
    if reward_processing_classes is None:
        classes = [None] * len(funcs)
    elif not isinstance(reward_processing_classes, list):
        classes = [reward_processing_classes]
    else:
        classes = list(reward_processing_classes)
        if len(classes) != len(funcs):
            raise ValueError("The number of reward processing classes must match the number of reward functions.")  # REF:246-247
    if len(classes) != len(funcs):       # (a single class with several functions: the reference's zip() silently drops the rest; refuse instead)
        raise ValueError("The number of reward processing classes must match the number of reward functions.")
    for i, (c, f) in enumerate(zip(classes, funcs)):
        if isinstance(f, torch.nn.Module):
            if c is None:
                from transformers import AutoTokenizer
                c = AutoTokenizer.from_pretrained(f.config._name_or_path)
            if c.pad_token_id is None:
                c.pad_token = c.eos_token
            f.config.pad_token_id = c.pad_token_id
            classes[i] = c
            f.eval()
            if device is not None:
                f.to(device)
        elif not callable(f):
            raise ValueError(f"reward function {i}: expected a callable, a torch module (reward model) or a checkpoint path, got {type(f).__name__}")
    return funcs, classes


def reward_func_name(f) -> str:
    """metric key of a reward function (REF:804-809)"""
    return f.config._name_or_path.split("/")[-1] if isinstance(f, torch.nn.Module) else f.__name__


def reward_model_scores(model, tokenizer, prompts: list, completions: list, conversational: bool) -> np.ndarray:
    """REF:760-772: the reward model reads prompt + completion as one text (conversational rows: the tokenizer's chat template over the prompt's messages +
    the assistant turn, trl/trl/data_utils.py:71-169 "messages" case), right-padded, no special tokens added; reward = logits[:, 0]."""
    if conversational:
        texts = [tokenizer.apply_chat_template(p + c, tools=None, tokenize=False) for p, c in zip(prompts, completions)]
    else:
        texts = [p + c for p, c in zip(prompts, completions)]
    enc = tokenizer(texts, return_tensors="pt", padding=True, padding_side="right", add_special_tokens=False)
    dev = next(model.parameters()).device
    enc = {k: v.to(dev) for k, v in enc.items()}
    with torch.inference_mode():
        return model(**enc).logits[:, 0].float().cpu().numpy().astype(np.float32)


class SCGRPOTrainer:
    def __init__(
        self,
        model: Union[str, tuple],
        reward_funcs: Union[Callable, list],
        args: Optional[GRPOConfig] = None,
        train_dataset=None,
        eval_dataset=None,
        processing_class=None,
        reward_processing_classes=None,
        callbacks=None,
        optimizers=(None, None),
        peft_config=None,
        max_pixels: Optional[int] = 12845056,
        min_pixels: Optional[int] = 3136,
        attn_implementation: str = "flash_attention_2",
        use_vllm_for_gen: bool = True,
    ):
        if args is None:
            name = model if isinstance(model, str) else "model"
            args = GRPOConfig(output_dir=f"{os.path.basename(name.rstrip('/'))}-GRPO")
        self.args = args
        mik = args.model_init_kwargs or {}
        if isinstance(model, str):
            td = mik.get("torch_dtype")
            if not (td is None or td == "auto" or isinstance(td, torch.dtype)):
                if isinstance(td, str) and hasattr(torch, td):
                    td = getattr(torch, td)
                else:
                    raise ValueError("Invalid `torch_dtype` passed to `GRPOConfig`. Expected either 'auto' or a string representing "
                                     f"a `torch.dtype` (e.g., 'float32'), but got {td}.")  # REF:108-111
            mid = model.lower()
            if not any(t in mid for t in ("qwen2.5-vl", "qwen2.5_vl", "qwen2.5vl", "qwen2-vl", "qwen2_vl", "qwen2vl", "llava-ov", "llava_ov", "llava_si",
                                          "llava-next", "llava_next", "llava_1_6", "llava-1_5", "llava_1_5")):
                raise ValueError(f"{model}: the reference's model switch (REF:116-137) knows Qwen2-VL, Qwen2.5-VL, LLaVA-OneVision, LLaVA-NeXT and LLaVA-1.5 ids; "
                                 "every other id falls to its AutoModelForCausalLM branch (text-only models), which is not part of this path.  Which family a "
                                 "directory holds is read from its config.json, not from the path")
        elif mik:
            raise ValueError("You passed `model_init_kwargs` to the `GRPOConfig`, but your model is already instantiated. "
                             "This argument can only be used when the `model` argument is a string.")  # REF:141-145
        if peft_config is not None:
            raise ValueError("peft_config: LoRA is not part of this path.  The reference wraps the policy with get_peft_model (REF:149-150) but pushes the wrapped "
                             "state_dict -- base_model.model.* / lora_A / lora_B names, adapters not merged -- into vLLM's load_weights (REF:569-579), and generation "
                             "without vLLM raises (REF:720): its LoRA branch cannot complete a step, and none of its launch scripts asks for it")
        self.device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        self.hf_config = None
        if isinstance(model, str):
            self.model_id = model
            self.cfg, self.policy = load_checkpoint(model, self.device, trainable=True)
            with open(os.path.join(model, "config.json")) as f:
                self.hf_config = json.load(f)
        else:
            self.model_id = "in-memory-qwen2.5-vl"
            self.cfg, weights = model
            if isinstance(weights, ParamStore):      # a resident trainable store (bench.py: random-init weights of a full-size model, no host copy)
                self.policy = weights
            else:
                self.policy = ParamStore(self.cfg, self.device, trainable=True)
                self.policy.load_named(weights)
        # frozen reference = the starting checkpoint (REF:152-182)
        self.ref = ParamStore(self.cfg, self.device, trainable=False)
        self.ref.copy_from(self.policy)
        if processing_class is None:
            if not isinstance(model, str):
                raise ValueError("processing_class is required when the model is passed in memory")
            from transformers import AutoProcessor
            processing_class = AutoProcessor.from_pretrained(model)
            if hasattr(processing_class, "image_processor"):
                processing_class.image_processor.max_pixels = max_pixels  # REF:192-193
                processing_class.image_processor.min_pixels = min_pixels
        if self.cfg.is_llava and hasattr(processing_class, "tokenizer"):
            processing_class.tokenizer.padding_side = "left"              # REF:218-219
        self.processing_class = processing_class
        self.reward_funcs, self.reward_processing_classes = resolve_reward_funcs(reward_funcs, reward_processing_classes, mik if isinstance(model, str) else None, self.device)
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.use_vllm = use_vllm_for_gen
        group = None
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            group = dist.group.WORLD
        self.engine = SCGRPOEngine(self.cfg, self.policy, self.ref, GRPOArgs(
            num_generations=args.num_generations, max_prompt_length=args.max_prompt_length, max_completion_length=args.max_completion_length,
            beta=args.beta, temperature=args.temperature, learning_rate=args.learning_rate, weight_decay=args.weight_decay,
            adam_beta1=args.adam_beta1, adam_beta2=args.adam_beta2, adam_epsilon=args.adam_epsilon, max_grad_norm=args.max_grad_norm,
            gradient_accumulation_steps=args.gradient_accumulation_steps, micro_batch_seqs=args.micro_batch_seqs, seed=args.seed,
            # under data parallelism the policy is "auto" even without the flag: it recomputes only where the static budget (which then reserves RCCL's buffers) is tight
            recompute="auto" if (args.gradient_checkpointing or group is not None) else "off"), group=group)
        self.state = type("State", (), {"global_step": 0})()
        self._metrics = defaultdict(list)
        self.log_history = []
        self._prefetcher = None
        self.prefetch_used = False          # a prefetch worker was started at some point (train() stops it again on the way out)

    # ---- batch construction (host) -------------------------------------------------------------------------------
    def _prepare(self, inputs: list[dict]):
        return prepare_batch(self.processing_class, inputs)

    def prefetch(self, micro_batches: list[list[dict]]) -> list:
        """Start the host preparation (REF:600-625) of the given micro-batches on the worker thread; hand the result to training_step(prepared=...)."""
        if self._prefetcher is None:
            import copy
            from .prefetch import BatchPrefetcher
            # the worker thread gets its OWN copy of the processor: HF fast tokenizers are not thread-safe (a padding=True call on one thread while the main
            # thread runs batch_decode can raise "Already borrowed"); a processor that cannot be copied is shared behind a lock the decode side takes too
            try:
                pc = copy.deepcopy(self.processing_class)
                prep = lambda inputs: prepare_batch(pc, inputs)
            except Exception:
                prep = lambda inputs: self._locked(self._prepare, inputs)
                self._pc_lock = threading.Lock()
            self._prefetcher = BatchPrefetcher(prep, self.device)
            self.prefetch_used = True
        return [self._prefetcher.submit(inputs) for inputs in micro_batches]

    def _locked(self, fn, *a):
        lock = self.__dict__.get("_pc_lock")
        if lock is None:
            return fn(*a)
        with lock:
            return fn(*a)

    def close(self):
        """Stop the prefetch worker (queued preparations are dropped)."""
        if self._prefetcher is not None:
            self._prefetcher.shutdown()
            self._prefetcher = None

    def _rewards(self, inputs, completion_ids: np.ndarray):
        G = self.args.num_generations
        texts = self._locked(lambda: self.processing_class.batch_decode(completion_ids, skip_special_tokens=True))       # (the lock exists only when the prefetch worker shares this processor)
        conversational = isinstance(inputs[0]["prompt"], list)
        completions = [[{"role": "assistant", "content": t}] for t in texts] if conversational else texts
        prompts = [ex["prompt"] for ex in inputs for _ in range(G)]
        kw = {k: [ex[k] for ex in inputs for _ in range(G)] for k in inputs[0] if k not in ("prompt", "completion")}
        cols = []
        for f, c in zip(self.reward_funcs, self.reward_processing_classes):
            if isinstance(f, torch.nn.Module):
                cols.append(reward_model_scores(f, c, prompts, completions, conversational))
            else:
                cols.append(np.asarray(f(prompts=prompts, completions=completions, current_step=self.state.global_step, **kw), dtype=np.float32))
        return np.stack(cols, 1)

    # ---- reference API ---------------------------------------------------------------------------------------------
    def compute_loss(self, model, inputs, return_outputs=False, num_items_in_batch=None, last_micro_step=True, _defer=False, _prepared=None, _batch=None, _completions=None):
        """One SC-GRPO micro-step on `inputs` (list of dataset rows).  Returns the loss value; gradients are
        accumulated inside the engine (there is no autograd graph to hand back).
        _defer (training_step, last micro-batch): returns a callable that yields the loss once called -- the two device-side means (loss, KL) are then read
        AFTER the optimizer has been enqueued, so the GPU does not idle while the host returns from the read.
        _prepared: a Future from `prefetch` for these same rows (the batch was built while the previous micro-batch ran); None: built here, as the reference does."""
        if return_outputs:
            raise ValueError("The GRPOTrainer does not support returning outputs")  # REF:587-588
        if _batch is not None:          # training_step already built it (and rolled the completions out with the other micro-batches of the step: _completions)
            batch = _batch
        elif _prepared is not None:
            from .prefetch import BatchPrefetcher
            batch = BatchPrefetcher.ready(_prepared.result())
        else:
            batch = self._prepare(inputs)
        # the engine's whole micro-step (SCGRPOEngine.step): vision tower once per image, rollout whose prefill / decode steps double as the policy's
        # training forward when the micro-batch holds whole groups, rewards evaluated on the host while the reference pass is in the GPU queue
        out = self.engine.step(batch, lambda comp: self._rewards(inputs, comp), do_optimizer_step=False, last_micro_step=last_micro_step, return_outputs=True, defer_metrics=_defer,
                               completions=_completions)
        m = out["metrics"]
        self._metrics["completion_length"].append(m["completion_length"])
        rp = out["rewards_per_func"].mean(0)
        for i, f in enumerate(self.reward_funcs):
            self._metrics[f"rewards/{reward_func_name(f)}"].append(float(rp[i]))
        self._metrics["reward"].append(m["reward"])
        self._metrics["reward_std"].append(m["reward_std"])

        def finish():
            if "finalize" in m:
                m.pop("finalize")()
            self._metrics["kl"].append(m["kl"])
            return m["loss"]
        return finish if _defer else finish()

    def log(self, logs: dict, start_time=None):
        metrics = {k: sum(v) / len(v) for k, v in self._metrics.items()}
        metrics = average_over_ranks(metrics)      # the reference logs rank-averaged metrics (REF:821-827 over accelerator.gather_for_metrics values)
        if "loss" in logs:
            logs = {**logs, **average_over_ranks({"loss": logs["loss"]})}
        logs = {**logs, **metrics}
        self.log_history.append(logs)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(logs), flush=True)
        self._metrics.clear()

    def _lr(self, step, total):
        a = self.args
        return schedule.lr_at(step, total, a.learning_rate, a.warmup_steps, a.lr_scheduler_type)

    def training_step(self, micro_batches: list[list[dict]], prepared: Optional[list] = None) -> list[float]:
        """One optimizer step = gradient_accumulation_steps micro-batches through compute_loss (the data-parallel gradient buckets leave during the
        last one's backward) + clip / AdamW / weight-copy refresh.  What transformers.Trainer.training_step + the optimizer block of its inner loop do
        around the reference's compute_loss (TF:trainer.py:1892-1961, :1785); train() and bench.py both go through here."""
        last = len(micro_batches) - 1
        ready = comps = None
        rows_total = sum(len(mb) for mb in micro_batches) * self.args.num_generations
        if self.args.batch_rollouts and len(micro_batches) > 1 and rows_total <= 64:
            # (<= 64 sequences: one 64-row tile of the decode GEMMs -- beyond that a decode step streams the weights once per tile and batching buys nothing, while
            # a micro-batch of whole groups keeps the rollout -> training hand-over)
            # ONE rollout for the whole optimizer step (GRPOConfig.batch_rollouts): all prompts left-padded into one batch, completions handed back per micro-batch
            from .prefetch import BatchPrefetcher
            ready = [BatchPrefetcher.ready(prepared[k].result()) if prepared is not None else self._prepare(inputs) for k, inputs in enumerate(micro_batches)]
            both = combine_batches(ready, self.cfg.pad_token_id)
            if both is not None:
                G = self.args.num_generations
                toks = self.engine.rollout(both)
                comps, r = [], 0
                for b in ready:
                    n = len(b["input_ids"]) * G
                    comps.append(trim_completions(toks[r: r + n], self.cfg.eos_token_id))
                    r += n
        losses = [self.compute_loss(None, inputs, last_micro_step=(k == last), _defer=(k == last), _prepared=None if (prepared is None or ready is not None) else prepared[k],
                                    _batch=None if ready is None else ready[k], _completions=None if comps is None else comps[k])
                  for k, inputs in enumerate(micro_batches)]
        self.engine.optimizer_step()
        losses[last] = losses[last]()       # the last micro-batch's loss / KL means are read after the optimizer launches are in the queue
        self.state.global_step += 1
        return losses

    def train(self, resume_from_checkpoint: Optional[str] = None):
        """resume_from_checkpoint: a checkpoint-N directory written by this trainer (weights + optimizer.safetensors + trainer_state.json), or True for
        the last one under output_dir (transformers.Trainer.train semantics)."""
        a = self.args
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        start = 0
        if resume_from_checkpoint:
            ck = last_checkpoint(a.output_dir) if resume_from_checkpoint is True else resume_from_checkpoint
            if ck is None:
                raise ValueError(f"No valid checkpoint found in output directory ({a.output_dir})")
            st = load_training_state(self.policy, ck)
            self.engine.opt_step, self.state.global_step, start = st["opt_step"], st["global_step"], st["global_step"]
        rows = list(self.train_dataset)
        bs, ga = a.per_device_train_batch_size, a.gradient_accumulation_steps
        # prompts are sharded over ranks, each rank keeps whole groups.  Every rank derives the step count from the GLOBAL
        # row count and holds ceil(n / world) rows per epoch (wrap-around padding, as DistributedSampler): equal collective sequences on all ranks
        total = schedule.total_steps(len(rows), world, bs, ga, a.num_train_epochs, a.max_steps)
        sampler = schedule.RankSampler(len(rows), rank, world, seed=a.seed, shuffle=a.shuffle)
        t0 = time.time()
        steps_per_epoch = max(1, schedule.total_steps(len(rows), world, bs, ga, 1.0, -1))

        def micro_rows(step):      # the rows of optimizer step `step`: a function of the step number alone (the sampler order is fixed up front)
            base = step * bs * ga
            return [[rows[sampler.index(base + k * bs + j)] for j in range(bs)] for k in range(ga)]

        try:
            return self._train_loop(start, total, micro_rows, steps_per_epoch, rank, t0)
        finally:
            self.close()        # also after an exception: no queued preparation keeps running behind a dead training loop

    def _train_loop(self, start, total, micro_rows, steps_per_epoch, rank, t0):
        a = self.args
        window_loss, window_from = 0.0, start
        nxt = self.prefetch(micro_rows(start)) if (a.prefetch_batches and start < total) else None
        for step in range(start, total):
            self.engine.args.learning_rate = self._lr(step, total)
            micro, prepared = micro_rows(step), nxt
            # step + 1's batches are built on the worker thread while this step runs on the GPU
            nxt = self.prefetch(micro_rows(step + 1)) if (a.prefetch_batches and step + 1 < total) else None
            losses = self.training_step(micro, prepared)
            window_loss += float(np.mean(losses))
            if self.state.global_step % a.logging_steps == 0:
                # transformers.Trainer._maybe_log_save_evaluate: the mean step loss since the last log line, rounded to 4 places; the scheduler has already stepped,
                # so `learning_rate` is the NEXT step's; `epoch` = fraction of the data seen
                self.log({"loss": round(window_loss / (self.state.global_step - window_from), 4), "grad_norm": self.engine.grad_norm(),
                          "learning_rate": self._lr(step + 1, total), "epoch": round(self.state.global_step / steps_per_epoch, 2), "step": self.state.global_step,
                          "elapsed_s": round(time.time() - t0, 2)})
                window_loss, window_from = 0.0, self.state.global_step
            if a.save_steps and self.state.global_step % a.save_steps == 0 and rank == 0:
                ck = os.path.join(a.output_dir, f"checkpoint-{self.state.global_step}")
                self.save_model(ck)
                save_training_state(self.policy, ck, self.engine.opt_step, self.state.global_step)
        return self.log_history

    def save_model(self, output_dir: Optional[str] = None):
        save_checkpoint(self.policy, output_dir or self.args.output_dir, self.hf_config)
        pc = self.processing_class
        if hasattr(pc, "save_pretrained"):
            pc.save_pretrained(output_dir or self.args.output_dir)

    def push_to_hub(self, **kw):
        raise RuntimeError("push_to_hub: no network egress in this environment")
