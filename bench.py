#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json on MI355X: GRPO samples/sec for the full SC-GRPO step
(hipGraph group rollout + CPU rewards + frozen-ref forward + policy forward/backward + DDP all-reduce + AdamW)
on Qwen2.5-VL-3B shapes, 8 prompts x group 8 per GPU, one 448x448 image + 512 prompt positions, 256 new tokens.
The timed step is the drop-in API itself: `SCGRPOTrainer.training_step` -> `compute_loss` (the reference's entry, REF
train/stage_rl/trainer/sc_grpo_trainer.py:586) -> `SCGRPOEngine.step` -> AdamW, fed by a synthetic processor.

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a torchrun environment: re-executes itself under `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU,
      RCCL); under torchrun it checks WORLD_SIZE == N.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline        : the launch that owns the step, named in `dominant` -- the rollout's decode replay (one hipGraph launch per generated token, HBM-bound: 56 % of the
                    step) or the GEMM family, whichever takes more of the timed region; a copy of one of the two objects below.
  roofline_decode : HBM roofline of the decode replay: algorithmic bytes per step (every decode-packed weight once + the K/V of every live context) / the HIP-event
                    time of the replay loop; `achieved_with_fetched_kv_bytes` = the same with the K/V bytes the attention launches really fetch (PMC).
  roofline_gemm   : gemm_nt_256 / gemm_nt_128 (bf16 MFMA).  achieved = sum of algorithmic 2*M*N*K over its launches in the timed region / the length of the UNION
                    of their HIP-event intervals on the launch streams (weight-gradient GEMMs overlap the dgrad chain on a side stream; the per-launch-sum
                    figure is printed beside it).
  cpu_baseline    : the CPU oracle (oracle/qwen25vl.py, kind "port") running the components of one B=1 x G=8 step -- vision tower, prefill,
                    KV-cached greedy decode steps, reference forward, policy forward+backward, head -- at full 3B width on this box's host
                    cores, each timed on a bounded sample and multiplied by its count in the step (layers x 36, blocks x 32, decode steps x 255).
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16


def _newest_profile(suffix: str):
    """profiles/rNN_<suffix> with the highest round NN (the counters / records of the newest build that re-took them), or None."""
    import re
    best = None
    try:
        for f in os.listdir(os.path.join(ROOT, "profiles")):
            m = re.fullmatch(r"r(\d+)_" + re.escape(suffix), f)
            if m and (best is None or int(m.group(1)) > best[0]):
                best = (int(m.group(1)), f)
    except OSError:
        pass
    return best[1] if best else None


def _skinny_pmc():
    """HBM-side bytes of the decode step from the PMC passes recorded under profiles/ (3B shapes): every kernel of the step (r02_decode_pmc.json: FETCH / WRITE per
    kernel, summed over the launches of one step), else the heaviest kernel alone (r01_skinny_pmc.json)."""
    try:
        rec = _newest_profile("decode_pmc.json")
        d = json.load(open(os.path.join(ROOT, "profiles", rec)))
        attn = next(k for k in d["kernels"] if "attn_decode" in k["kernel"])
        steps = attn["launches"] / 36.0
        total = sum((k["read_bytes_corrected"] + k["write_bytes"]) * k["launches"] for k in d["kernels"]) / steps
        kv_fetched = attn["read_bytes_corrected"] * attn["launches"] / steps
        return {"kernel": "all kernels of one decode step (Qwen2.5-VL-3B shapes, 64 sequences, context ~520)", "bytes_per_launch": total, "algorithmic_bytes_per_launch": None,
                "kv_bytes_fetched_per_step": kv_fetched,
                "kv_note": ("bytes the paged-attention launches of one step pull through the L2's fabric side: the G sequences of a prompt group share their prompt's pages and run on one "
                            "XCD, so its L2 serves all but the first reader -- a fraction of the ALGORITHMIC K/V bytes (every sequence reading its whole context) counted in `achieved`"),
                "source": f"profiles/{rec} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 read correction)", "round": int(rec[1:3]),
                "decode_cus_profiled": d.get("decode_cus")}
    except Exception:
        pass
    try:
        rec = _newest_profile("skinny_pmc.json")
        k = json.load(open(os.path.join(ROOT, "profiles", rec)))
        return {"kernel": k["kernel"], "bytes_per_launch": k["traffic_bytes"], "algorithmic_bytes_per_launch": k["algorithmic_bytes"], "MNK": k["MNK"], "source": "profiles/" + rec, "round": int(rec[1:3])}
    except Exception:
        return None


def power_limit():
    """Shader clock and socket power under a back-to-back gemm_nt stream, from the newest record tools/gemm_power.py left under profiles/ (a separate run: rocm-smi
    sampling next to the bench would perturb it)."""
    for f in (_newest_profile("gemm_power.json"),):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", f)))
            d["dense_bf16_peak_at_that_clock_tflops"] = MFMA_BF16_DENSE_PEAK_TFLOPS * d["sclk_mhz_under_gemm_stream"] / 2400.0
            return d
        except Exception:
            pass
    return {"source": "profiles/r02_gemm_power.txt", "sclk_mhz_under_gemm_stream": 1900, "sclk_mhz_idle": 2400, "socket_power_w": 1388,
            "dense_bf16_peak_at_that_clock_tflops": MFMA_BF16_DENSE_PEAK_TFLOPS * 1900 / 2400}


def decode_roofline(cfg, pol, events, n_seq, gen_len):
    """HBM roofline of the rollout's decode step (one hipGraph replay = ~250 kernels, 64 live sequences): algorithmic bytes per step
    = every decode-packed weight once + the K/V of every live context, / the HIP-event time of the replay loop (events on the
    stream the graph is replayed on)."""
    if not events:
        return None
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in events)
    steps = sum(n for _, _, n, _ in events)
    w_bytes = pol.flat_pk.numel() * 2 + (pol.flat_pk8.numel() if getattr(pol, "decode_fp8", False) else 0)
    kv_tok = 2 * cfg.num_hidden_layers * cfg.num_key_value_heads * cfg.head_dim * 2
    # context read by decode step j (1-based) = prompt length + j tokens (the new token's own K/V included)
    kv_bytes = sum(kv_tok * (plen * n + n_seq * n * (n + 1) // 2) for _, _, n, plen in events) / max(steps, 1)
    ach = (w_bytes + kv_bytes) / (ms / steps * 1e-3) / 1e9
    pmc = _skinny_pmc() if (cfg.hidden_size, cfg.num_hidden_layers, cfg.v_arch) == (2048, 36, "qwen2_5_vl") else None      # the PMC record is of the 3B shapes
    ach_fetched = None
    if pmc and pmc.get("kv_bytes_fetched_per_step"):      # the same rate with the K/V term replaced by what the attention launches really fetch (PMC): the byte rate the HBM sees
        ach_fetched = (w_bytes + pmc["kv_bytes_fetched_per_step"]) / (ms / steps * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "decode step (one hipGraph replay: skinny GEMMs on decode-packed weights + paged attention + RMSNorm + sampling)", "achieved": ach,
            "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "ms_per_decode_step": ms / steps, "decode_steps": steps, "total_ms": ms,
            "algorithmic_bytes_per_step": {"weights": w_bytes, "kv": kv_bytes}, "achieved_with_fetched_kv_bytes": ach_fetched,
            "traffic": (pmc or {}).get("bytes_per_launch"), "traffic_detail": pmc}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)     # (the caching allocator settles in the SECOND step: 7 device allocations there, none later -- tools/alloc_steps.py)
    ap.add_argument("--model", default="3b", choices=["3b", "7b", "qwen2vl_2b", "llava_ov_7b", "llava15_7b", "llava_next_7b", "tiny"])
    ap.add_argument("--decode-weights", default="bf16", choices=["bf16", "fp8"],
                    help="fp8: the rollout streams gate|up / down / lm_head as e4m3 + per-row scales (opt-in, BASELINE config 5; NOT the headline precision)")
    ap.add_argument("--prompts", type=int, default=8)
    ap.add_argument("--group", type=int, default=8)
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--gen-len", type=int, default=256)
    ap.add_argument("--micro-batch", type=int, default=0, help="sequences per ref/policy pass; 0 = 64 (3B) / 32 (7B: 20480-token activations of 28 wide layers do not fit next to the fp32 optimizer state)")
    ap.add_argument("--no-repeated-rows-leg", action="store_true", help="skip the extra (untimed) step in the reference's repeated-prompt-rows layout")
    ap.add_argument("--workload", default="sc_grpo", choices=["sc_grpo", "pa_sft"], help="sc_grpo = the north-star SC-GRPO step (default); pa_sft = BASELINE config 2 (PA-SFT, bs 16, 448^2 image, 512 prompt + 256 supervised tokens)")
    ap.add_argument("--sft-batch", type=int, default=16)
    ap.add_argument("--gradient-checkpointing", default=None, choices=["off", "auto", "on"], help="decoder activation recompute policy of the step (the reference scripts' "
                    "--gradient_checkpointing true = auto): lets e.g. --model 7b run one 64-sequence micro-batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-real-processor-legs", action="store_true", help="skip the two extra (untimed for `value`) loops that feed uint8 images through the HF image processor inside the step")
    ap.add_argument("--no-real-shapes-leg", action="store_true", help="skip the (untimed for `value`) leg at the reference scripts' shapes: B=1 x G=4, accum 2, ragged prompts, EOS live, 512 tokens")
    ap.add_argument("--real-shapes-steps", type=int, default=4)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-full-step", action="store_true", help="run the oracle's REAL B = 1 x G SC-GRPO step once at full 3B size on the host cores (about 6 minutes, "
                    "~130 GB of host memory; no GPU work) and print its JSON record -- the committed record is the newest profiles/rNN_cpu_full_step.json (round 6: r06); the default run's cpu_baseline "
                    "stays the bounded component sample")
    ap.add_argument("--launch-check", action="store_true", help="only the N-rank launch contract, no GPU work: respawn under torch.distributed.run when needed, assert WORLD_SIZE == --gpus, "
                    "rendezvous over gloo on 127.0.0.1, max-reduce a per-rank value, ONE JSON line from rank 0 (what tests/test_ddp_gloo.py runs on CPU)")
    ap.add_argument("--check", action="store_true", help="with --cpu-full-step: the full-size PARITY record instead of the timing record -- the same weights, prompt and completion "
                    "tokens through SCGRPOEngine.loss_and_grads on cuda:0 and through the fp32 oracle on the host (plus the oracle in bf16 as the yardstick), at P / C / G of the "
                    "headline, full depth; prints one JSON object (committed: profiles/r04_full_size_parity.json)")
    ap.add_argument("--check-forward-only", action="store_true", help="--check: the oracle runs without autograd (log-probs, KL, loss; no gradient cosines) -- the default for "
                    "--model llava_ov_7b, whose eager-attention autograd graph at 4019 positions x 28 layers does not fit the GPU box's 300 GiB memory cgroup")
    ap.add_argument("--check-noise", type=float, default=0.02, help="--check: element-wise relative noise policy = reference x (1 + noise)")
    ap.add_argument("--check-greedy-tokens", type=int, default=24, help="--check: greedy tokens compared between the hipGraph rollout and the oracle's KV-cached decode")
    return ap.parse_args()


CANNED = [
    "<think>surface looks uniform</think><answer>no</answer>",
    "<think>a line on the left</think><location>upper left</location><type>scratch</type><answer>yes</answer>",
    "<think>dark blob</think><location>center</location><type>stain</type><answer>yes</answer>",
    "<think>hmm</think><location>bottom right</location><type>hole</type><answer>no</answer>",
    "no tags at all",
    "<think>x</think><location>top left corner</location><type>surface scratch</type><answer>yes</answer>",
    "<think>y</think><location>left</location><type>structural anomaly</type><answer>yes</answer>",
    "<think>z</think><location>top</location><type>scrach</type><answer>yes</answer> extra",
]
SOLUTION = "<think>gt</think><location>top left</location><type>scratch</type><answer>yes</answer>"


def synth_batch_llava(cfg, n_prompts, n_text, seed, image_hw=(448, 448)):
    """BASELINE config 5 input: one 448 x 448 image per prompt through the any-resolution path (fitted to 768 x 768: 2 x 2 crops + the base image =
    5 crops of 384 x 384 -> 729 + 54 x 55 = 3699 packed image tokens) + 3 prefix ids + n_text text ids."""
    from iadr1_amd import llava_ov as lo
    rs = np.random.RandomState(seed)
    if cfg.llava_family == "llava":       # LLaVA-1.5: the processor resizes / centre-crops to one 336 x 336 crop: 576 image tokens
        n_img, nc = cfg.v_tokens, 1
    else:
        n_img = lo.num_image_tokens(image_hw, cfg.image_grid_pinpoints, cfg.v_image_size, cfg.v_side, cfg.anyres_max)
        nc = lo.num_crops(image_hw, cfg.image_grid_pinpoints, cfg.v_image_size)
    rows = [rs.randint(1000, 150000, 3).tolist() + [cfg.image_token_id] * n_img + rs.randint(1000, 150000, n_text).tolist() for _ in range(n_prompts)]
    ids = np.array(rows, dtype=np.int64)
    px = rs.standard_normal((n_prompts * nc, 3, cfg.v_image_size, cfg.v_image_size)).astype(np.float32)
    out = {"input_ids": ids, "attention_mask": np.ones_like(ids), "pixel_values": torch.from_numpy(px)}
    if cfg.llava_family != "llava":
        out["image_sizes"] = [image_hw] * n_prompts
    return out


def synth_batch(cfg, n_prompts, prompt_len, seed):
    """3 prefix ids + <|vision_start|> + 256 x <|image_pad|> + <|vision_end|> + text ids,
    one 448x448 image (grid 1x32x32 -> 1024 patches) per prompt; pixel rows ~ N(0,1) as after normalisation."""
    rs = np.random.RandomState(seed)
    grid = (1, 32, 32)
    n_img = 256
    n_text = prompt_len - (3 + 1 + n_img + 1)
    assert n_text >= 0
    hi = min(150000, cfg.vocab_size - 8, cfg.vision_start_token_id)
    lo = min(1000, hi - 1)
    rows = []
    for _ in range(n_prompts):
        rows.append(rs.randint(lo, hi, 3).tolist() + [cfg.vision_start_token_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_token_id] + rs.randint(lo, hi, n_text).tolist())
    ids = np.array(rows, dtype=np.int64)
    px = rs.standard_normal((n_prompts * 1024, cfg.patch_dim)).astype(np.float32)
    return {"input_ids": ids, "attention_mask": np.ones_like(ids), "pixel_values": torch.from_numpy(px), "image_grid_thw": [grid] * n_prompts}


class GemmTimer:
    """HIP-event timing of every gemm_nt launch on the launch stream (torch's current stream).  A launch on a CU-masked stream (the co-scheduled reference pass:
    iadr1_amd/overlap.py, 64 of 256 CUs) is priced against THAT share of the device: its interval counts share x length (a GEMM confined to a quarter of the
    CUs for 4 ms has used what a whole-device GEMM uses in 1 ms), so `achieved` stays FLOPs per whole-device second."""

    def __init__(self):
        self.records = []
        self.enabled = False

    def install(self):
        from iadr1_amd import ops
        orig = ops.gemm_nt
        timer = self

        def timed(a, b, bias=None, out=None, out_dtype=torch.bfloat16, accumulate=False, act=0):
            if not timer.enabled:
                return orig(a, b, bias=bias, out=out, out_dtype=out_dtype, accumulate=accumulate, act=act)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(a, b, bias=bias, out=out, out_dtype=out_dtype, accumulate=accumulate, act=act)
            e1.record()
            timer.records.append((e0, e1, 2.0 * a.shape[0] * b.shape[0] * a.shape[1], (a.shape[0], b.shape[0], a.shape[1], ('acc' if accumulate else str(out_dtype if out is None else out.dtype)[6:]) + timer.tag()), timer.share()))
            return r

        ops.gemm_nt = timed
        orig_f = ops.gemm_swiglu_fused

        def timed_f(x, w_gu, gu, a_out):      # gate|up GEMM with the SwiGLU epilogue: the same gemm_nt_256 main loop, 2*T*2I*K FLOPs
            if not timer.enabled:
                return orig_f(x, w_gu, gu, a_out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_f(x, w_gu, gu, a_out)
            e1.record()
            timer.records.append((e0, e1, 2.0 * x.shape[0] * w_gu.shape[0] * x.shape[1], (x.shape[0], w_gu.shape[0], x.shape[1], ('bfloat16+swiglu' if gu is not None else 'swiglu only') + timer.tag()), timer.share()))

        ops.gemm_swiglu_fused = timed_f

        # lm_head through the linear_logprob kernels: gemm_nt_256's main loop with the log-sum-exp / dlogits epilogues (the forward pair's second launch,
        # a per-row merge of ~0.1 ms, sits inside its interval)
        def wrap_head(name, tag):
            orig_h = getattr(ops, name)

            def timed_h(h, w, *a, **kw):
                if not timer.enabled:
                    return orig_h(h, w, *a, **kw)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig_h(h, w, *a, **kw)
                e1.record()
                timer.records.append((e0, e1, 2.0 * h.shape[0] * w.shape[0] * h.shape[1], (h.shape[0], w.shape[0], h.shape[1], tag + timer.tag()), timer.share()))
                return r
            setattr(ops, name, timed_h)
        wrap_head("linear_logprob", "lse epilogue")
        wrap_head("linear_logprob_dlogits", "dlogits epilogue")

    @staticmethod
    def share():
        from iadr1_amd import hip
        return hip.cu_share()

    def tag(self):
        sh = self.share()
        return "" if sh >= 1.0 else f" @{sh:.2f} of the CUs"

    def summary(self):
        """(launches, CU-share-weighted sum of launch durations in s, FLOPs)"""
        t = sum(r[0].elapsed_time(r[1]) * r[4] for r in self.records) * 1e-3
        fl = sum(r[2] for r in self.records)
        return len(self.records), t, fl

    def masked_launches(self):
        return sum(1 for r in self.records if r[4] < 1.0)

    def busy_seconds(self):
        """Length of the UNION of the launch intervals, per CU share, weighted by the share.  The weight-gradient GEMMs run on a side stream concurrently with the
        dgrad GEMMs of the main stream; the sum of per-launch durations then counts shared time twice, the union counts it once (= the sum when nothing
        overlaps).  Launches on a CU-masked stream form their own group (they run next to the decode replays, never next to other GEMMs)."""
        if not self.records:
            return 0.0
        ref = self.records[0][0]
        total = 0.0
        for sh in sorted({r[4] for r in self.records}):
            iv = sorted((ref.elapsed_time(e0), ref.elapsed_time(e1)) for e0, e1, _, _, s_ in self.records if s_ == sh)
            busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
            for s_, e_ in iv[1:]:
                if s_ > cur_e:
                    busy += cur_e - cur_s
                    cur_s, cur_e = s_, e_
                else:
                    cur_e = max(cur_e, e_)
            total += (busy + cur_e - cur_s) * sh
        return total * 1e-3

    def by_shape(self, top=14):
        agg = {}
        for e0, e1, f, key, sh in self.records:
            a = agg.setdefault(key, [0, 0.0, 0.0, sh])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += f
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1] * kv[1][3])[:top]
        return [{"MNK_out": list(k), "calls": v[0], "ms": round(v[1] * 1e3, 2), "TF": round(v[2] / max(v[1], 1e-9) / 1e12, 1),
                 **({"TF_per_whole_device": round(v[2] / max(v[1] * v[3], 1e-9) / 1e12, 1)} if v[3] < 1.0 else {})} for k, v in rows]


def cpu_baseline(cfg_dict_3b, seconds_budget, P=512, C=256, G=8):
    """the CPU restatement (oracle, kind "port") running ONE SC-GRPO step for B = 1 prompt x G = 8 completions on the
    host cores -- rollout = prefill once + greedy decode loop with a key/value cache, reference forward, policy forward + backward, fp32.  The
    full-size step would take ~10 minutes of CPU, so each COMPONENT runs at full 3B width on a bounded sample and is multiplied by its count
    in the step: decoder layer x 36, ViT block x 32 (+ patch embed / merger once per image), decode steps x (C - 1), lm_head per position."""
    from oracle import qwen25vl as oq
    cores = int(os.environ.get("IADR1_CPU_THREADS", min(os.cpu_count() or 1, 32)))  # 32 threads measured fastest on the 2x64-core host (16: same, 64: 0.6x, 128: 0.35x)
    torch.set_num_threads(cores)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixture_util as fx
    L_full, V_full = cfg_dict_3b["text"]["num_hidden_layers"], cfg_dict_3b["vision"]["depth"]
    gen = torch.Generator().manual_seed(0)

    def model(n_layers, v_depth, vocab, grad):
        d = json.loads(json.dumps(cfg_dict_3b))
        d["text"]["num_hidden_layers"], d["text"]["vocab_size"] = n_layers, vocab
        d["vision"]["depth"], d["vision"]["fullatt_block_indexes"] = v_depth, [v_depth - 1] if v_depth else []
        d.update(image_token_id=vocab - 8, vision_start_token_id=vocab - 7, vision_end_token_id=vocab - 6, eos_token_id=1, pad_token_id=2)
        w = {k: (torch.randn(sh, generator=gen) * 0.02) for k, sh in fx.param_shapes(d).items()}
        return d, oq.Qwen25VLOracle(d, w, requires_grad=grad)

    def timed(fn, reps=1, samples=3):
        """MEDIAN of `samples` timings (each the mean of `reps` calls) after a first-touch run (page faults, thread pool): a single-shot timing of a
        0.1-1 s component on a 256-thread host moved by 2x from run to run, and the differences below then went negative (BENCH_r02: -0.335 s)."""
        fn()
        ts = []
        for _ in range(samples):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            ts.append((time.perf_counter() - t0) / reps)
        return sorted(ts)[len(ts) // 2]

    pos0 = lambda x: max(0.0, x)               # a component's time is a DIFFERENCE of two medians: never below zero

    S = P + C
    grid = (1, 32, 32)
    # ---- vision tower per image: (depth 2) - (depth 1) = one block; depth 1 - block = patch embed + merger ---------------------------
    pv = torch.randn(1024, 1176, generator=gen)
    vt = {}
    for depth in (1, 2):
        _, m = model(0, depth, 4096, True)
        vt[depth, "f"] = timed(lambda: m.visual(pv, [grid]).detach())
        def fb():
            for _, t in m.parameters():
                t.grad = None
            m.visual(pv, [grid]).sum().backward()
        vt[depth, "fb"] = timed(fb)
    vit = {k: pos0(vt[2, k] - vt[1, k]) * V_full + pos0(2 * vt[1, k] - vt[2, k]) for k in ("f", "fb")}      # seconds per image, full depth
    # ---- decoder layer on the training rows [G, S] and the prefill row [1, P]; one layer = (1 layer) - (0 layers) -------------------
    vocab_small = 4096
    ids = torch.randint(3, vocab_small - 16, (G, S), generator=gen)
    mask = torch.ones(G, S, dtype=torch.long)
    lay = {}
    for nl in (0, 1):
        d, m = model(nl, 0, vocab_small, True)
        x = m.embed(ids).detach()
        pos = torch.arange(S).view(1, 1, S).expand(3, G, S)
        lay[nl, "f"] = timed(lambda: m.text_model(x, mask, pos).detach())
        def fb():
            for _, t in m.parameters():
                t.grad = None
            m.text_model(x, mask, pos).sum().backward()
        lay[nl, "fb"] = timed(fb)
        lay[nl, "prefill"] = timed(lambda: m.text_model(x[:1, :P], mask[:1, :P], pos[:, :1, :P]).detach())
    # KV-cached decode step of the G sequences at mid context (P + C/2 keys) over NDEC distinct layers: a decode step is a WEIGHT STREAM (G rows against every matrix),
    # and one layer's 344 MB of fp32 weights would be served from the host's last-level cache -- the full-size run (profiles/r03_cpu_full_step.json: 1.29 s per decode
    # step, 36 layers = 12.4 GB) showed the single-layer timing of rounds 1-2 to be 4.7x too fast
    NDEC = 6
    d, m = model(NDEC, 0, vocab_small, False)
    st = {"cache": [[torch.randn(G, d["text"]["num_key_value_heads"], P + C // 2, 128, generator=gen)] * 2 for _ in range(NDEC)], "mask": torch.ones(G, P + C // 2, dtype=torch.long),
          "deltas": torch.zeros(G, dtype=torch.long)}
    tok = torch.randint(3, vocab_small - 16, (G,), generator=gen)
    def dec():
        st["cache"] = [[kv[0][:, :, : P + C // 2], kv[1][:, :, : P + C // 2]] for kv in st["cache"]]
        st["mask"] = st["mask"][:, : P + C // 2]
        m.decode_step_cached(tok, st)
    lay["dec_layers_plus_small_head"] = timed(dec, reps=2)
    del m, st
    t_layer = {k: pos0(lay[1, k] - lay[0, k]) for k in ("f", "fb", "prefill")}
    # ---- lm_head + log-softmax + gather at the full vocabulary: the reference projects ALL S positions of every row (REF:505); timed on R rows --
    V = cfg_dict_3b["text"]["vocab_size"]
    H = cfg_dict_3b["text"]["hidden_size"]
    Wh = (torch.randn(V, H, generator=gen) * 0.02).requires_grad_(True)
    R = 1024
    hrows = torch.randn(R, H, generator=gen)
    tg = torch.randint(0, V, (R,), generator=gen)
    head_f = timed(lambda: torch.log_softmax(hrows @ Wh.t(), -1).gather(-1, tg.view(-1, 1)).detach())
    def head_fb():
        Wh.grad = None
        torch.log_softmax(hrows @ Wh.t(), -1).gather(-1, tg.view(-1, 1)).sum().backward()
    head_fb_t = timed(head_fb)
    with torch.no_grad():
        hd = torch.randn(G, H, generator=gen)
        head_dec = timed(lambda: (hd @ Wh.t()).argmax(-1), reps=4)           # decode: G rows against the whole matrix (weight-read bound)
        small = (torch.randn(vocab_small, H, generator=gen) * 0.02)
        head_dec_small = timed(lambda: (hd @ small.t()).argmax(-1), reps=4)
    t_dec_layer = pos0(lay["dec_layers_plus_small_head"] - head_dec_small) / NDEC
    rows_all = G * S
    parts = {
        "rollout_vision_1_image_fwd": vit["f"],
        "rollout_prefill_1_prompt": L_full * t_layer["prefill"] + head_f / R,
        "rollout_decode_255_steps_kv_cached": (C - 1) * (L_full * t_dec_layer + head_dec),
        "reference_forward_G_rows": G * vit["f"] + L_full * t_layer["f"] + head_f * rows_all / R,
        "policy_forward_backward_G_rows": G * vit["fb"] + L_full * t_layer["fb"] + head_fb_t * rows_all / R,
    }
    assert all(v >= 0.0 for v in parts.values()), parts
    step_s = sum(parts.values())
    return {"value": G / step_s, "unit": "samples/s", "cores": cores, "kind": "port", "seconds_per_step_B1_G8": step_s,
            "parts_seconds": {k: round(v, 3) for k, v in parts.items()},
            "sample": (f"oracle (fp32 torch CPU restatement) components of ONE B=1 x G={G} SC-GRPO step at full Qwen2.5-VL-3B width, P={P}, C={C}: ViT block fwd / fwd+bwd "
                       f"(1 image), decoder layer fwd / fwd+bwd on the [{G}, {S}] training rows, prefill layer on [1, {P}], KV-cached greedy decode step of {G} sequences over {NDEC} distinct layers (a weight stream: one layer alone runs from the host's cache) "
                       f"at {P + C // 2} cached keys, lm_head + log-softmax + gather on {R} rows of the {V}-token vocabulary (all S positions per row as REF:505), "
                       f"decode head on {G} rows; each the median of 3 timings after a first-touch run (differences of medians clamped at 0), multiplied by its count in the step (layers x {L_full}, ViT blocks x {V_full}, "
                       f"decode steps x {C - 1}); ViT recomputed per sequence in the training passes as the reference does")}


def cpu_full_step(cfg_dict_3b, P=512, C=256, G=8):
    """`--cpu-full-step`: ONE whole SC-GRPO step of the oracle (kind "port") at the full Qwen2.5-VL-3B size on the host cores, nothing extrapolated --
    the workload SURVEY section 8(d) names for the CPU baseline: B = 1 prompt x G completions, one 448 x 448 image, P prompt positions, C greedy tokens
    (EOS suppressed), fp32, eager attention: vision tower + prefill once (what vLLM's prefix cache gives the reference), KV-cached decode of the G sequences,
    canned-string rewards, frozen-reference forward and policy forward + backward on the [G, P + C] rows with all-position logits and the ViT recomputed
    per row (REF:505,625-628), AdamW step."""
    from oracle import qwen25vl as oq
    from oracle import sc_grpo as og
    from iadr1_amd import rewards as rw
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixture_util as fx
    cores = int(os.environ.get("IADR1_CPU_THREADS", min(os.cpu_count() or 1, 32)))
    torch.set_num_threads(cores)
    d = json.loads(json.dumps(cfg_dict_3b))
    d.update(image_token_id=151655, video_token_id=151656, vision_start_token_id=151652, vision_end_token_id=151653, eos_token_id=151645, pad_token_id=151643)
    gen = torch.Generator().manual_seed(0)
    t_0 = time.perf_counter()
    w = {}
    for k, sh in fx.param_shapes(d).items():
        if k.endswith("norm.weight") or "norm1.weight" in k or "norm2.weight" in k or k.endswith("ln_q.weight") or "layernorm.weight" in k:
            w[k] = torch.ones(sh)
        elif k.endswith(".bias"):
            w[k] = torch.zeros(sh)
        else:
            w[k] = torch.randn(sh, generator=gen) * 0.02
    pol = oq.Qwen25VLOracle(d, w, requires_grad=True, copy=False)
    ref = oq.Qwen25VLOracle(d, {k: t.detach() for k, t in w.items()}, copy=False)        # the frozen reference shares the (identical) initial weights
    opt = torch.optim.AdamW([t for _, t in pol.parameters()], lr=1e-6, weight_decay=0.0)
    t_init = time.perf_counter() - t_0
    grid = (1, 32, 32)
    rs = np.random.RandomState(1234)
    n_text = P - (3 + 1 + 256 + 1)
    row = rs.randint(1000, 150000, 3).tolist() + [d["vision_start_token_id"]] + [d["image_token_id"]] * 256 + [d["vision_end_token_id"]] + rs.randint(1000, 150000, n_text).tolist()
    ids = torch.tensor([row], dtype=torch.long)
    mask = torch.ones_like(ids)
    px = torch.from_numpy(rs.standard_normal((1024, 1176)).astype(np.float32))
    parts = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        lg, st = pol.prefill_cached(ids, mask, px, [grid])                  # vision tower + prefill, once per prompt
        parts["rollout_vision_and_prefill_1_prompt"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        st["cache"] = [[k.expand(G, -1, -1, -1).contiguous(), v.expand(G, -1, -1, -1).contiguous()] for k, v in st["cache"]]
        st["mask"], st["deltas"] = st["mask"].expand(G, -1).contiguous(), st["deltas"].expand(G).contiguous()
        lg = lg.expand(G, -1)
        comp = []
        for it in range(C):
            lg = lg.clone()
            lg[:, d["eos_token_id"]] = -float("inf")                         # EOS suppressed: fixed-length completions, like the GPU line
            nxt = lg.argmax(-1)
            comp.append(nxt)
            if it + 1 < C:
                lg = pol.decode_step_cached(nxt, st)
        comp = torch.stack(comp, 1)
        parts[f"rollout_decode_{C - 1}_steps_kv_cached"] = time.perf_counter() - t1
    del st
    t2 = time.perf_counter()
    wrapped = [[{"role": "assistant", "content": CANNED[i % len(CANNED)]}] for i in range(G)]
    rew = torch.tensor(np.stack([rw.accuracy_reward(wrapped, [SOLUTION] * G), rw.consistency_reward(wrapped, [SOLUTION] * G)], 1).astype(np.float32))
    parts["rewards"] = time.perf_counter() - t2
    t3 = time.perf_counter()
    out = og.sc_grpo_step(pol, ref, ids, mask, px, [grid], [r.tolist() for r in comp], rew, G, 0.04, d["eos_token_id"], d["pad_token_id"])
    parts["policy_and_reference_forward_G_rows"] = time.perf_counter() - t3
    t4 = time.perf_counter()
    out["loss"].backward()
    parts["policy_backward"] = time.perf_counter() - t4
    t5 = time.perf_counter()
    opt.step()
    parts["adamw"] = time.perf_counter() - t5
    step_s = time.perf_counter() - t0
    import resource
    return {"value": G / step_s, "unit": "samples/s", "cores": cores, "kind": "port", "seconds_per_step_B1_G8": step_s, "parts_seconds": {k: round(v, 3) for k, v in parts.items()},
            "loss": float(out["loss"].detach()), "kl": out["metrics"]["kl"], "weights_init_seconds": round(t_init, 1), "host_peak_rss_GB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20,
            "sample": (f"ONE complete B=1 x G={G} SC-GRPO step of the oracle (fp32 torch CPU restatement) at full Qwen2.5-VL-3B size, nothing extrapolated: P={P} (256 image + "
                       f"{P - 256} text / special positions), C={C} greedy tokens with EOS suppressed, vision + prefill once per prompt, KV-cached decode of {G} sequences, "
                       f"reference + policy forward on [{G}, {P + C}] rows (all-position logits, ViT per row as REF:505,625-628), backward, AdamW; {cores} threads")}


def start_memory_watchdog():
    """The GPU box's memory cgroup is 300 GiB and a process that exceeds it takes the box down with it (this round's one gpurun strike): a daemon thread that
    ends the process once the cgroup's usage passes 80 % of its limit."""
    import threading

    def _watch():
        try:
            lim = int(open("/sys/fs/cgroup/memory.max").read())
        except Exception:
            return
        while True:
            time.sleep(1.0)
            try:
                if int(open("/sys/fs/cgroup/memory.current").read()) > 0.80 * lim:
                    sys.stderr.write("bench.py: host memory above 80 % of the cgroup limit -- aborting\n")
                    sys.stderr.flush()
                    os._exit(3)
            except Exception:
                return
    threading.Thread(target=_watch, daemon=True).start()


def full_size_parity(a):
    """`--cpu-full-step --check`: parity AT THE HEADLINE'S SHAPE (VERDICT r3 "missing" #2).  One prompt (448 x 448 image = 256 image tokens, P positions) x G
    completions of C tokens through the UNREDUCED Qwen2.5-VL-3B (36 + 32 layers, 151 936-token head): `SCGRPOEngine.loss_and_grads` on cuda:0 (shared-prefix layout,
    the kernels of the measured step) against `oracle.sc_grpo.sc_grpo_step` in fp32 on the host on the SAME weights / prompt / completion tokens / rewards
    (REF sc_grpo_trainer.py:586-819), and against the same oracle in bf16 -- the precision `--bf16` gives the reference -- as the yardstick.
    Weights: ParamStore.init_random(seed 0) with the embedding x 2 (logit std ~ 1.8), policy = reference x (1 + noise) element-wise.  Completions: the engine's own
    SAMPLED rollout (temperature 0.9 / top-k 50 / top-p 0.9: eight different rows; a greedy rollout of one prompt gives G identical rows), one row cut by an EOS.
    Also recorded: greedy token ids of the hipGraph rollout against the oracle's KV-cached greedy decode for the first tokens, with the oracle's top-2 logit gap at
    the first disagreement (random-init logits are nearly flat; the bit-exact greedy check against HF lives in tests/ on fixtures with real structure)."""
    import resource
    from oracle import qwen25vl as oq
    from oracle import sc_grpo as og
    from iadr1_amd.params import ParamStore, VLMConfig
    from iadr1_amd.sc_grpo import GRPOArgs, SCGRPOEngine
    P, C, G = a.prompt_len, a.gen_len, a.group
    dev = "cuda:0"
    cores = int(os.environ.get("IADR1_CPU_THREADS", min(os.cpu_count() or 1, 32)))
    torch.set_num_threads(cores)
    llava = a.model == "llava_ov_7b"
    cfg = VLMConfig.llava_ov_7b() if llava else (VLMConfig.qwen25vl_7b() if a.model == "7b" else VLMConfig.qwen25vl_3b())
    d3 = json.loads(json.dumps(D3))
    if llava:                  # BASELINE config 5: SigLIP-so400m tower + any-resolution packing + Qwen2-7B decoder (params._llava_ov_7b)
        from oracle import llava_ov as oo
        d3 = {"text": {"vocab_size": 152064, "hidden_size": 3584, "intermediate_size": 18944, "num_hidden_layers": 28, "num_attention_heads": 28, "num_key_value_heads": 4,
                       "rms_norm_eps": 1e-6, "rope_theta": 1e6},
              "vision": {"arch": "siglip", "depth": 26, "hidden_size": 1152, "intermediate_size": 4304, "num_heads": 16, "in_channels": 3, "patch_size": 14, "image_size": 384,
                         "layer_norm_eps": 1e-6},
              "image_grid_pinpoints": [[384 * i, 384 * j] for i in range(1, 7) for j in range(1, 7)], "anyres_max": 9, "video_token_id": 151647}
    if a.model == "7b":       # BASELINE config 4's structure: untied 152 064-token head, 28 layers of 3584, 28:4 heads, MLP 18944; the ViT's merger projects to 3584
        d3["text"].update(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4)
        d3["vision"]["out_hidden_size"] = 3584
    d3.update(image_token_id=cfg.image_token_id, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id, tie_word_embeddings=cfg.tie_word_embeddings)
    if not llava:
        d3.update(video_token_id=151656, vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id)
    make_oracle = (lambda w, **kw: oo.LlavaOVOracle(d3, w, **kw)) if llava else (lambda w, **kw: oq.Qwen25VLOracle(d3, w, **kw))
    fwd_only = bool(a.check_forward_only or llava)
    start_memory_watchdog()
    T = {"t0": time.time()}
    pol, ref = ParamStore(cfg, dev, trainable=True), ParamStore(cfg, dev, trainable=False)
    ref.init_random(seed=0)
    ref.w("embed").mul_(2.0)
    if not cfg.tie_word_embeddings:
        ref.w(ref.lm_head_name()).mul_(2.0)          # untied head: the logits' spread comes from lm_head
    ref.finalize()
    pol.flat.copy_(ref.flat)
    gen = torch.Generator(device=dev).manual_seed(7)
    for lo in range(0, pol.flat.numel(), 1 << 28):
        v = pol.flat[lo: lo + (1 << 28)]
        v.copy_((v.float() * (1.0 + a.check_noise * torch.randn(v.shape, generator=gen, device=dev))).to(torch.bfloat16))
    pol.finalize()
    grid = (1, 32, 32)
    rs = np.random.RandomState(1234)
    if llava:     # one 448 x 448 image through the any-resolution path: 5 crops of 384 x 384 -> 3699 packed image tokens, + 3 prefix ids + 253 text ids (the bench's prompt)
        sb = synth_batch_llava(cfg, 1, 253, seed=1234)
        ids, mask, px = sb["input_ids"], sb["attention_mask"], sb["pixel_values"].numpy()
        P = ids.shape[1]
        vis_arg = [tuple(z) for z in sb["image_sizes"]]
        batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": sb["pixel_values"], "image_sizes": sb["image_sizes"]}
    else:
        n_text = P - (3 + 1 + 256 + 1)
        row = rs.randint(1000, 150000, 3).tolist() + [cfg.vision_start_token_id] + [cfg.image_token_id] * 256 + [cfg.vision_end_token_id] + rs.randint(1000, 150000, n_text).tolist()
        ids, mask = np.array([row], dtype=np.int64), np.ones((1, P), dtype=np.int64)
        px = rs.standard_normal((1024, cfg.patch_dim)).astype(np.float32)
        vis_arg = [grid]
        batch = {"input_ids": ids, "attention_mask": mask, "pixel_values": torch.from_numpy(px), "image_grid_thw": [grid]}
    eng = SCGRPOEngine(cfg, pol, ref, GRPOArgs(num_generations=G, max_prompt_length=4096, max_completion_length=C, beta=0.04, micro_batch_seqs=G, seed=11))
    n_greedy = 0 if llava else max(0, min(a.check_greedy_tokens, C))        # (the KV-cached greedy oracle exists for the Qwen structure)
    toks_greedy = eng.rollout(batch, greedy=True)[0, :n_greedy].tolist() if n_greedy else []
    # the logits behind those greedy tokens, from BOTH device paths (which path loses what): the decode kernels' (the rollout stopped after k + 1
    # tokens leaves the logits of token k in its buffer; k = 0 is the prefill) and the training kernels' (teacher-forced forward over [prompt | greedy tokens])
    dec_logits = trn_logits = None
    if n_greedy:
        dec_logits = []
        for k in range(n_greedy):
            eng.args.max_completion_length = k + 1
            eng.rollout(batch, greedy=True)
            dec_logits.append(eng._rollout.logits[0].float().cpu().numpy().copy())
        eng.args.max_completion_length = C
        dec_logits = np.stack(dec_logits)
        vis_g = eng.vision_policy(batch, save=False)
        gpr_g, off_g = eng._per_row_images(batch, vis_g["grids"], vis_g["rows"])
        ids_g = np.concatenate([ids, np.asarray([toks_greedy], dtype=np.int64)], 1)
        plan_g = eng.pol.text_plan(ids_g, np.ones_like(ids_g), gpr_g, off_g)
        hf_g, _ = eng.pol.text_forward(plan_g, vis_g["img"], save=False)
        trn_logits = eng.pol.logits_rows(hf_g, torch.arange(P - 1, P - 1 + n_greedy, device=dev, dtype=torch.int64)).float().cpu().numpy()
        del hf_g, plan_g, vis_g
    comps = eng.rollout(batch, greedy=False)
    comps = [r.tolist() for r in comps]
    comps[min(3, G - 1)] = comps[min(3, G - 1)][: C // 3] + [cfg.eos_token_id]     # one completion ends early: the EOS mask and the ragged rows (llava: the rotation quirk) are part of the check
    wrapped = [[{"role": "assistant", "content": CANNED[i % len(CANNED)]}] for i in range(G)]
    from iadr1_amd import rewards as rw
    rew = np.stack([rw.accuracy_reward(wrapped, [SOLUTION] * G), rw.consistency_reward(wrapped, [SOLUTION] * G)], 1).astype(np.float32)
    out = eng.loss_and_grads(batch, comps, rew)
    torch.cuda.synchronize()
    T["hip"] = time.time()
    Lm = cfg.num_hidden_layers
    names = ["model.norm.weight", f"model.layers.{Lm - 1}.post_attention_layernorm.weight", f"model.layers.{Lm // 2 - 1}.self_attn.k_proj.bias", "model.layers.0.input_layernorm.weight",
             "visual.merger.ln_q.weight", "visual.blocks.0.norm1.weight"]
    if llava:
        names = ["language_model." + n for n in names[:4]] + ["image_newline", "multi_modal_projector.linear_2.bias", "vision_tower.vision_model.encoder.layers.0.layer_norm1.weight"]
    grads = {}
    if not fwd_only:
        grads = pol.export_named(source="grad")
        grads = {n: grads[n].numpy().reshape(-1).astype(np.float64) for n in names}
    hip_lp, hip_lr, hip_cm, mt = out["logps"].cpu().numpy(), out["ref_logps"].cpu().numpy(), np.asarray(out["completion_mask"]), dict(out["metrics"])
    w_pol, w_ref = pol.export_named(), ref.export_named()
    del eng, out, pol, ref
    torch.cuda.empty_cache()
    T["export"] = time.time()
    o_pol = make_oracle(w_pol, requires_grad=False if fwd_only else set(names), copy=False)
    o_ref = make_oracle(w_ref, copy=False)
    # greedy ids: the oracle's KV-cached decode on the policy weights
    greedy = {"tokens_compared": 0}
    if n_greedy:
        with torch.no_grad():
            lg, st = o_pol.prefill_cached(torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(px), [grid])
            want_tok, gaps, ora_logits = [], [], []
            for it in range(n_greedy):
                top2 = torch.topk(lg[0], 2)
                want_tok.append(int(top2.indices[0]))
                gaps.append(float(top2.values[0] - top2.values[1]))
                ora_logits.append(lg[0].float().numpy().copy())
                if it + 1 < n_greedy:
                    lg = o_pol.decode_step_cached(torch.tensor([toks_greedy[it]]), st)     # teacher-forced with the HIP token: every position is compared on the same prefix
            del st
            ora_logits = np.stack(ora_logits)
            # the reference's precision on the same prefix: the oracle's cached decode in bf16
            b16_logits = None
            try:
                o16 = make_oracle({k: t for k, t in w_pol.items() if not (k == "lm_head.weight" and cfg.tie_word_embeddings)}, dtype=torch.bfloat16)
                lg16, st16 = o16.prefill_cached(torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(px), [grid])
                b16_logits = []
                for it in range(n_greedy):
                    b16_logits.append(lg16[0].float().numpy().copy())
                    if it + 1 < n_greedy:
                        lg16 = o16.decode_step_cached(torch.tensor([toks_greedy[it]]), st16)
                b16_logits = np.stack(b16_logits)
                del o16, st16
            except Exception as ex:       # noqa: BLE001 -- optional yardstick
                b16_logits = None
        agree = [int(x == y) for x, y in zip(toks_greedy, want_tok)]
        first_bad = agree.index(0) if 0 in agree else -1
        def logit_error(x):
            """Against the fp32 oracle's logits at the same positions: argmax agreement, the error of the TOP-2 GAP (what decides an argmax), the largest logit error
            over the oracle's 50 most likely tokens (a constant shift of a row removed: softmax does not see it)."""
            if x is None:
                return None
            top = np.argsort(-ora_logits, 1)[:, :50]
            rows_ = np.arange(len(x))
            d = np.take_along_axis(x, top, 1) - np.take_along_axis(ora_logits, top, 1)
            d = d - d.mean(1, keepdims=True)
            gap_err = (x[rows_, top[:, 0]] - x[rows_, top[:, 1]]) - (ora_logits[rows_, top[:, 0]] - ora_logits[rows_, top[:, 1]])
            return {"argmax_agree": int((x.argmax(1) == ora_logits.argmax(1)).sum()), "top2_gap_error_std": float(gap_err.std()), "top2_gap_error_max": float(np.abs(gap_err).max()),
                    "top50_logit_error_max": float(np.abs(d).max()), "top50_logit_error_rms": float(np.sqrt((d ** 2).mean()))}
        greedy = {"tokens_compared": n_greedy, "agree": int(sum(agree)), "first_disagreement": first_bad,
                  "oracle_top2_logit_gap_at_disagreements": [round(g_, 5) for g_, ok in zip(gaps, agree) if not ok], "median_top2_gap": float(np.median(gaps)),
                  "logit_error_vs_fp32_oracle": {"hip_decode_kernels": logit_error(dec_logits), "hip_training_kernels": logit_error(trn_logits), "bf16_oracle": logit_error(b16_logits)},
                  "decode_vs_training_kernels": ({"top50_logit_diff_max": float(np.abs(np.take_along_axis(dec_logits - trn_logits, np.argsort(-ora_logits, 1)[:, :50], 1)).max()),
                                                  "argmax_agree": int((dec_logits.argmax(1) == trn_logits.argmax(1)).sum())} if dec_logits is not None else None),
                  "note": ("teacher-forced on the HIP tokens.  A greedy token flips when the error of the top-2 gap exceeds the gap: `logit_error_vs_fp32_oracle` gives that error's "
                           "spread for the decode kernels, the training kernels and the oracle run in bf16 (the reference's precision) on the same prefix")}
    T["greedy"] = time.time()
    with (torch.no_grad() if fwd_only else contextlib.nullcontext()):
        want = og.sc_grpo_step(o_pol, o_ref, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(px), vis_arg, comps, torch.from_numpy(rew), G, 0.04,
                               cfg.eos_token_id, cfg.pad_token_id, rotate_right_padded_rows=llava)
    T["oracle_fwd"] = time.time()
    if not fwd_only:
        want["loss"].backward()
    T["oracle_bwd"] = time.time()
    m = want["completion_mask"].bool().numpy()
    assert np.array_equal(hip_cm.astype(bool), m), "completion masks differ"
    w_lp, w_lr = want["logps"].detach().numpy(), want["ref_logps"].numpy()
    # the reference's precision: the same oracle in bf16 (log-softmax in fp32 as REF:509)
    bf = None
    try:
        ids_t = torch.cat([torch.from_numpy(ids).repeat(G, 1), og.right_pad(comps, cfg.pad_token_id)], 1)
        mask_t = torch.cat([torch.from_numpy(mask).repeat(G, 1), want["completion_mask"].long()], 1)
        if llava:
            ids_t, mask_t = oo.ensure_left_padding(ids_t, mask_t, cfg.pad_token_id)
        px_t = torch.from_numpy(px)
        with torch.no_grad():
            lps = []
            for wd in (w_pol, w_ref):
                o16 = make_oracle({k: t for k, t in wd.items() if not (k == "lm_head.weight" and cfg.tie_word_embeddings)}, dtype=torch.bfloat16)
                lps.append(o16.per_token_logps(ids_t, mask_t, px_t.repeat(G, *[1] * (px_t.dim() - 1)), vis_arg * G)[:, P - 1:].float().numpy())
                del o16
        kl16 = float(og.grpo_loss(torch.from_numpy(lps[0]), torch.from_numpy(lps[1]), want["advantages"], want["completion_mask"].float(), 0.04)[2])
        bf = {"dlogp_policy_max": float(np.abs(lps[0][m] - w_lp[m]).max()), "dlogp_policy_mean": float(np.abs(lps[0][m] - w_lp[m]).mean()),
              "dlogp_ref_max": float(np.abs(lps[1][m] - w_lr[m]).max()), "dlogp_ref_mean": float(np.abs(lps[1][m] - w_lr[m]).mean()), "kl": kl16}
    except Exception as ex:      # the yardstick is optional: a host without a usable bf16 GEMM must not lose the record
        bf = {"error": repr(ex)}
    T["bf16"] = time.time()
    wl, wk = float(want["loss"].detach()), float(want["metrics"]["kl"])
    cos = {} if not fwd_only else {"skipped": "forward-only check (--check-forward-only / llava_ov_7b): no autograd graph on the host"}
    for n in ([] if fwd_only else names):
        x, y = grads[n], dict(o_pol.parameters())[n].grad.numpy().reshape(-1).astype(np.float64)
        cos[n] = {"cosine": float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30)), "norm_ratio": float(np.linalg.norm(x) / (np.linalg.norm(y) + 1e-30))}
    e_p, e_r = hip_lp[m] - w_lp[m], hip_lr[m] - w_lr[m]
    rec = {"shape": {"model": "LLaVA-OneVision-SI-7B shapes, unreduced (SigLIP-so400m 26 layers, 5 crops -> 3699 packed image tokens, Qwen2-7B decoder 28 x 3584, vocab 152064, untied head)" if llava else (f"Qwen2.5-VL-{'7B' if a.model == '7b' else '3B'}, unreduced ({cfg.num_hidden_layers} decoder layers of {cfg.hidden_size}, 32 ViT blocks, vocab {cfg.vocab_size}, "
                               f"{'untied' if not cfg.tie_word_embeddings else 'tied'} head)"), "prompts": 1, "G": G, "P": P, "C": C, "scored_tokens": int(m.sum()),
                     "policy": f"reference x (1 + {a.check_noise} N(0,1)) element-wise", "completions": "sampled by the engine's hipGraph rollout (T 0.9, top-k 50, top-p 0.9), row 3 cut by EOS"},
           "hip_vs_fp32_oracle": {"dlogp_policy_max": float(np.abs(e_p).max()), "dlogp_policy_mean": float(np.abs(e_p).mean()), "dlogp_ref_max": float(np.abs(e_r).max()),
                                  "dlogp_ref_mean": float(np.abs(e_r).mean()), "err_of_ref_minus_policy_std": float(np.std(e_r - e_p)),
                                  "logp_range": [float(w_lp[m].min()), float(w_lp[m].max())],
                                  "kl_hip": mt["kl"], "kl_oracle": wk, "kl_rel_err": abs(mt["kl"] - wk) / max(wk, 1e-30), "loss_hip": mt["loss"], "loss_oracle": wl, "loss_abs_err": abs(mt["loss"] - wl),
                                  "metrics_hip": {k: float(v) for k, v in mt.items() if isinstance(v, (int, float))},
                                  "metrics_oracle": {k: float(v) for k, v in want["metrics"].items() if isinstance(v, (int, float))},
                                  "gradients": cos},
           "bf16_oracle_vs_fp32_oracle": bf, "greedy_ids": greedy,
           "seconds": {"hip_rollouts_and_step": round(T["hip"] - T["t0"], 1), "export": round(T["export"] - T["hip"], 1), "oracle_greedy": round(T["greedy"] - T["export"], 1),
                       "oracle_fp32_forward": round(T["oracle_fwd"] - T["greedy"], 1), "oracle_fp32_backward": round(T["oracle_bwd"] - T["oracle_fwd"], 1), "oracle_bf16": round(T["bf16"] - T["oracle_bwd"], 1)},
           "host": {"threads": cores, "peak_rss_GB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20}, "device": torch.cuda.get_device_name(0)}
    if bf and "kl" in bf:
        rec["bf16_oracle_vs_fp32_oracle"]["kl_rel_err"] = abs(bf["kl"] - wk) / max(wk, 1e-30)
    return rec


def pa_sft_full_size_parity(a):
    """`--cpu-full-step --check --workload pa_sft`: BASELINE config 2's step at full size -- the unreduced Qwen2.5-VL-3B, `--sft-batch` sequences of [448^2 image + 512 prompt
    positions | 256 supervised tokens] -- through `SFTEngine` on cuda:0 and through the fp32 oracle (`Qwen25VLOracle.sft_loss` + torch AdamW) on the host, from the same
    seed-0 weights: the loss of three consecutive AdamW steps (lr 1e-5, no weight decay, clip 1.0), the step-1 gradients of six named tensors, the step-1 gradient norm.
    REF llamafactory/train/sft/trainer.py:92-107 -> TF loss/loss_utils.py:32-71.  The north star's "loss curve within 1e-3" is read against this record."""
    import resource
    from oracle import qwen25vl as oq
    from iadr1_amd.params import ParamStore, VLMConfig
    from iadr1_amd.sft import SFTArgs, SFTEngine
    dev = "cuda:0"
    cores = int(os.environ.get("IADR1_CPU_THREADS", min(os.cpu_count() or 1, 32)))
    torch.set_num_threads(cores)
    q2 = a.model == "qwen2vl_2b"        # BASELINE config 1's model: Qwen2-VL-2B (LayerNorm / QuickGELU ViT without windows), vision tower + projector frozen as in the reference
    cfg = VLMConfig.qwen2vl_2b() if q2 else VLMConfig.qwen25vl_3b()
    d3 = json.loads(json.dumps(D3))
    if q2:
        d3["text"].update(hidden_size=1536, intermediate_size=8960, num_hidden_layers=28, num_attention_heads=12, num_key_value_heads=2)
        d3["vision"] = {"arch": "qwen2_vl", "depth": 32, "hidden_size": 1280, "intermediate_size": 5120, "num_heads": 16, "in_channels": 3, "patch_size": 14, "spatial_merge_size": 2,
                        "temporal_patch_size": 2, "window_size": 0, "out_hidden_size": 1536, "fullatt_block_indexes": list(range(32))}
    d3.update(image_token_id=cfg.image_token_id, video_token_id=151656, vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id,
              eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id, tie_word_embeddings=True)
    B, P, C, STEPS = a.sft_batch, a.prompt_len, a.gen_len, 3
    start_memory_watchdog()
    T = {"t0": time.time()}
    p = ParamStore(cfg, dev, trainable=True, with_decode_pack=False)
    p.init_random(seed=0)
    p.w("embed").mul_(2.0)
    p.finalize()
    w0 = p.export_named()
    from iadr1_amd.sft import frozen_parameter_rule
    frozen = frozen_parameter_rule("qwen2_vl") if q2 else None
    eng = SFTEngine(cfg, p, SFTArgs(learning_rate=1e-5, weight_decay=0.0, micro_batch_seqs=B, frozen=frozen))
    Lm = cfg.num_hidden_layers

    def make(seed):
        b = synth_batch(cfg, B, P, seed)
        rs = np.random.RandomState(seed + 1)
        resp = rs.randint(1000, min(150000, cfg.vision_start_token_id), (B, C)).astype(np.int64)
        ids = np.concatenate([b["input_ids"], resp], 1)
        labels = ids.copy()
        labels[:, :P] = -100
        return {"input_ids": ids, "attention_mask": np.ones_like(ids), "labels": labels, "pixel_values": b["pixel_values"], "image_grid_thw": b["image_grid_thw"]}
    batches = [make(4321 + i) for i in range(STEPS)]
    names = ["model.norm.weight", f"model.layers.{Lm - 1}.post_attention_layernorm.weight", f"model.layers.{Lm // 2 - 1}.self_attn.k_proj.bias", "model.layers.0.input_layernorm.weight",
             "model.layers.0.mlp.down_proj.weight"] + ([] if q2 else ["visual.merger.ln_q.weight", "visual.blocks.0.norm1.weight"])
    hip_loss, grads, hip_gn = [], None, None
    for k, bt in enumerate(batches):
        hip_loss.append(eng.loss_and_grads(dict(bt, pixel_values=bt["pixel_values"].to(dev))))
        if k == 0:
            g = p.export_named(source="grad")
            grads = {n: g[n].numpy().reshape(-1).astype(np.float64) for n in names}
            del g
        eng.optimizer_step()
        if k == 0:
            hip_gn = eng.grad_norm()
    torch.cuda.synchronize()
    T["hip"] = time.time()
    del eng, p
    torch.cuda.empty_cache()
    w_init = {k: v.detach().clone() for k, v in w0.items()}        # the first oracle trains w0's tensors in place (copy=False)
    train_names = {n for n in w0 if not (q2 and n.startswith("visual."))}        # config 1: the vision tower and the merger do not train (LLaMA-Factory's defaults)
    o = oq.Qwen25VLOracle(d3, w0, requires_grad=train_names, copy=False)
    params = [t for n, t in o.parameters() if n in train_names]
    opt = torch.optim.AdamW(params, lr=1e-5, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8)
    want_loss, cos, want_gn = [], {}, None
    for k, bt in enumerate(batches):
        opt.zero_grad(set_to_none=True)
        loss = o.sft_loss(torch.from_numpy(bt["input_ids"]), torch.from_numpy(bt["attention_mask"]), torch.from_numpy(bt["labels"]), bt["pixel_values"], bt["image_grid_thw"])
        loss.backward()
        want_loss.append(float(loss.detach()))
        if k == 0:
            for n in names:
                x, y = grads[n], o.w[n].grad.numpy().reshape(-1).astype(np.float64)
                cos[n] = {"cosine": float(x @ y / (np.linalg.norm(x) * np.linalg.norm(y) + 1e-30)), "norm_ratio": float(np.linalg.norm(x) / (np.linalg.norm(y) + 1e-30))}
        gn = float(torch.nn.utils.clip_grad_norm_(params, 1.0))
        if k == 0:
            want_gn = gn
        opt.step()
    T["oracle"] = time.time()
    # the yardstick for steps 2-3: the same fp32 arithmetic with the weights STORED as the reference's --bf16 run stores them -- bf16 in the forward / backward, an
    # fp32 master under AdamW (the HIP path's scheme): what separates the two curves above beyond step 1 is that storage, not the kernels
    del o, opt, params, w0
    o = oq.Qwen25VLOracle(d3, w_init, requires_grad=train_names, copy=False)
    params = [t for n, t in o.parameters() if n in train_names]
    master = [t.detach().clone() for t in params]
    opt = torch.optim.AdamW(params, lr=1e-5, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8)
    bf16w_loss = []
    for k, bt in enumerate(batches):
        opt.zero_grad(set_to_none=True)
        with torch.no_grad():
            for t_, m_ in zip(params, master):
                t_.copy_(m_.to(torch.bfloat16).float())
        loss = o.sft_loss(torch.from_numpy(bt["input_ids"]), torch.from_numpy(bt["attention_mask"]), torch.from_numpy(bt["labels"]), bt["pixel_values"], bt["image_grid_thw"])
        loss.backward()
        bf16w_loss.append(float(loss.detach()))
        with torch.no_grad():
            for t_, m_ in zip(params, master):
                t_.copy_(m_)
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        with torch.no_grad():
            for t_, m_ in zip(params, master):
                m_.copy_(t_)
    T["oracle_bf16w"] = time.time()
    return {"shape": {"model": "Qwen2-VL-2B, unreduced (28 decoder layers of 1536, 32 ViT blocks, vocab 151936); vision tower + merger frozen" if q2 else "Qwen2.5-VL-3B, unreduced (36 decoder layers, 32 ViT blocks, vocab 151936)", "sequences": B, "prompt_positions": P, "supervised_tokens": C,
                      "steps": STEPS, "optimizer": "AdamW lr 1e-5, betas (0.9, 0.999), eps 1e-8, weight decay 0, clip 1.0; " + ("language model trains" if q2 else "everything trains")},
            "loss_hip": hip_loss, "loss_oracle_fp32": want_loss, "loss_abs_diff": [abs(x - y) for x, y in zip(hip_loss, want_loss)],
            "loss_oracle_bf16_stored_weights": bf16w_loss, "loss_abs_diff_vs_bf16_stored_weights": [abs(x - y) for x, y in zip(hip_loss, bf16w_loss)],
            "grad_norm_step1": {"hip": hip_gn, "oracle": want_gn, "ratio": hip_gn / max(want_gn, 1e-30)}, "gradients_step1": cos,
            "note": ("step 1 compares the forward / backward alone (same weights); steps 2-3 also carry the optimizer: the HIP path keeps bf16 parameters under an fp32 master "
                     "(the reference's --bf16), the oracle is fp32 throughout"),
            "seconds": {"hip_3_steps": round(T["hip"] - T["t0"], 1), "oracle_3_steps": round(T["oracle"] - T["hip"], 1), "oracle_bf16_stored_weights_3_steps": round(T["oracle_bf16w"] - T["oracle"], 1)},
            "host": {"threads": cores, "peak_rss_GB": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20}, "device": torch.cuda.get_device_name(0)}


def run_pa_sft(a, cfg, dev, rank, world):
    """BASELINE.json config 2: one PA-SFT optimizer step = forward(labels) + backward + AdamW on `--sft-batch` sequences of
    [448^2 image + prompt (512 positions) | 256 supervised response tokens] (llamafactory supervised masking: prompt labels -100)."""
    from iadr1_amd.params import ParamStore
    from iadr1_amd.sft import SFTArgs, SFTEngine
    p = ParamStore(cfg, dev, trainable=True, with_decode_pack=False)
    p.init_random(seed=0)
    # the trainable set of the reference's PA-SFT (LLaMA-Factory defaults): Qwen2-VL (a registered composite family) trains with its vision tower and projector frozen,
    # Qwen2.5-VL (not registered in the vendored LLaMA-Factory) trains whole -- iadr1_amd.sft.frozen_parameter_rule
    from iadr1_amd.sft import frozen_parameter_rule
    frozen = frozen_parameter_rule("qwen2_vl" if cfg.v_arch == "qwen2_vl" else "qwen2_5_vl")
    eng = SFTEngine(cfg, p, SFTArgs(learning_rate=1e-5, weight_decay=0.1, micro_batch_seqs=a.sft_batch, frozen=frozen))
    timer = GemmTimer()
    timer.install()
    B, P, C = a.sft_batch, a.prompt_len, a.gen_len
    model_name = "Qwen2-VL-2B" if a.model == "qwen2vl_2b" else f"Qwen2.5-VL-{a.model.upper()}"

    def make(seed):
        b = synth_batch(cfg, B, P, seed)
        rs = np.random.RandomState(seed + 1)
        resp = rs.randint(1000, min(150000, cfg.vision_start_token_id), (B, C)).astype(np.int64)
        ids = np.concatenate([b["input_ids"], resp], 1)
        labels = ids.copy()
        labels[:, :P] = -100
        return {"input_ids": ids, "attention_mask": np.ones_like(ids), "labels": labels, "pixel_values": b["pixel_values"].to(dev), "image_grid_thw": b["image_grid_thw"]}

    batches = [make(4321 + 7919 * rank + i) for i in range(a.warmup + a.steps)]

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    k = 0
    for _ in range(a.warmup):
        eng.loss_and_grads(batches[k])
        eng.optimizer_step()
        k += 1
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    loss = None
    for _ in range(a.steps):
        loss = eng.loss_and_grads(batches[k])
        eng.optimizer_step()
        k += 1
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        # which step structure every rank ran (the decision is collective -- overlap.agree_across_ranks -- so these must all be equal; printed so that a run shows it)
        mine = {"rank": rank, "co_scheduled": bool(getattr(eng, "last_step_shadowed", False)), "decode_stream_cus": int(getattr(eng._rollout, "decode_cus", 0) or 0) or None,
                "recompute_forced": bool(eng.pol.__dict__.get("_recompute_forced"))}
        per_rank_structure = [None] * world
        dist.all_gather_object(per_rank_structure, mine)
        dist.destroy_process_group()
    elif os.environ.get("IADR1_FORCE_REDUCE"):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()
    if rank == 0:
        n_launch, t_gemm, fl_gemm = timer.summary()
        ach = fl_gemm / max(t_gemm, 1e-9) / 1e12
        emit({
            "metric": f"PA-SFT samples/sec (bs={B}, img448, {P}+{C} tok) {model_name}", "value": world * B * a.steps / dt, "unit": "samples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic", "config": {"workload": f"{model_name} PA-SFT step (BASELINE config {1 if a.model == 'qwen2vl_2b' else 2}): {B} sequences x (448x448 image + {P} prompt positions + {C} supervised tokens), forward(labels) + backward + AdamW, random-init weights", "parallelism": f"dp{world}",
                                            "trainable": ("language model (vision tower + projector frozen: LLaMA-Factory's defaults for the registered qwen2_vl family)" if frozen is not None else "all parameters (qwen2_5_vl is not a registered composite family of the reference's LLaMA-Factory: nothing is frozen)")},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_256 / gemm_nt_128 (v_mfma_f32_16x16x32_bf16)", "achieved": ach, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": None, "launches": n_launch, "kernel_time_frac_of_step": t_gemm / dt},
            "last_loss": loss, "tokens_per_s": world * B * (P + C) * a.steps / dt, "cpu_baseline": None,
            "hbm": {"peak_allocated_GB": torch.cuda.max_memory_allocated() / 2**30}})


class SynthProcessor:
    """Stands in for the HF AutoProcessor the trainer calls (no tokenizer / image files offline): the chat template renders a marker, the
    processor call returns the synthetic prompt tensors of the step that the dataset rows name (already resident in HBM -- inputs are made resident
    before the timed region), batch_decode returns the canned completion strings (so that the reward plugins do real work on real text)."""

    def __init__(self, batches, texts):
        self.batches, self.texts = batches, texts

    def apply_chat_template(self, conv, add_generation_prompt=True, tokenize=False):
        return "SYNTHETIC PROMPT"

    def __call__(self, text=None, images=None, **kw):
        step = images[0][1]
        assert all(im[1] == step for im in images) and len(images) == len(text)
        return self.batches[step]

    def batch_decode(self, ids, skip_special_tokens=True):
        return [self.texts[i % len(self.texts)] for i in range(len(ids))]


class RealImageProcessor(SynthProcessor):
    """The `real_processor` legs: the processor call does the reference's real host work on real pixels -- uint8 [448, 448, 3] images
    (numpy RandomState(1234 + i)) read back from PNG files (PIL decode, REF:610) through the HF `Qwen2VLImageProcessor` (smart_resize,
    rescale, normalise, patchify -> fp32 [1024, 1176] per image, REF:612-622) -- on the host, per step, as `compute_loss` does.  The prompt token ids stay the
    synthetic ones of the step the text names (no tokenizer exists offline)."""

    def __init__(self, batches, texts):
        super().__init__(batches, texts)
        from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
        self.ip = Qwen2VLImageProcessorPil()
        self.ip.max_pixels, self.ip.min_pixels = 480000, 3136          # the launch scripts' --max_pixels (REF:192-193); 448 x 448 = 200 704 pixels pass unresized

    def apply_chat_template(self, conv, add_generation_prompt=True, tokenize=False):
        return conv[0]["content"][-1]["text"]          # "SYNTHETIC PROMPT <step>": names the synthetic token ids of the step

    def __call__(self, text=None, images=None, **kw):
        step = int(text[0].rsplit(" ", 1)[1])
        enc = self.ip(images=images, return_tensors="pt")
        b = self.batches[step]
        assert enc["pixel_values"].shape == b["pixel_values"].shape, (enc["pixel_values"].shape, b["pixel_values"].shape)
        return {"input_ids": b["input_ids"], "attention_mask": b["attention_mask"], "pixel_values": enc["pixel_values"], "image_grid_thw": enc["image_grid_thw"]}


def real_processor_legs(tr, batches, n_prompts, steps, first_step, N):
    """Same step, host work included: (1) the reference's form -- `prepare_batch` inside compute_loss, synchronously; (2) with the trainer's prefetch
    (micro-batch k+1 prepared on a worker thread + pinned upload on a copy stream while the GPU runs k), as SCGRPOTrainer.train() drives it."""
    import tempfile
    from PIL import Image
    tmp = tempfile.mkdtemp(prefix="iadr1_bench_img_")
    n_steps = 2 * (steps + 1)
    assert first_step + n_steps <= len(batches)
    paths = {}
    for s in range(first_step, first_step + n_steps):
        for j in range(n_prompts):
            i = s * n_prompts + j
            paths[s, j] = os.path.join(tmp, f"img_{i}.png")
            Image.fromarray(np.random.RandomState(1234 + i).randint(0, 256, (448, 448, 3)).astype(np.uint8)).save(paths[s, j], compress_level=1)
    rows = lambda s: [{"prompt": [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": f"SYNTHETIC PROMPT {s}"}]}], "image": [paths[s, j]], "solution": SOLUTION}
                      for j in range(n_prompts)]
    saved = tr.processing_class
    tr.processing_class = RealImageProcessor(batches, CANNED)
    res = {}
    try:
        t_host = time.perf_counter()
        tr._prepare(rows(first_step))
        res["host_prepare_seconds_per_step"] = time.perf_counter() - t_host
        s = first_step
        for mode in ("inline", "prefetch"):
            tr.training_step([rows(s)])           # one untimed step of the leg
            s += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nxt = tr.prefetch([rows(s)]) if mode == "prefetch" else None
            for k in range(steps):
                prepared = nxt
                nxt = tr.prefetch([rows(s + 1)]) if (mode == "prefetch" and k + 1 < steps) else None
                tr.training_step([rows(s)], prepared)
                s += 1
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[mode] = {"samples_per_s": N * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps}
    finally:
        tr.processing_class = saved
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    res["note"] = ("the headline `value` runs on pixel patches already resident in HBM; these legs add the per-step host work of a real run -- PNG decode + HF Qwen2VLImageProcessor "
                   f"on {n_prompts} uint8 448x448 images -- inline (the reference's form, REF:600-625) and through the trainer's prefetch thread (the default of SCGRPOTrainer.train())")
    return res


def synth_prompt_batch(cfg, grid_hw, n_text, seed, dev):
    """One prompt: 3 prefix ids + <|vision_start|> + (h/2 * w/2) x <|image_pad|> + <|vision_end|> + n_text text ids; pixel patches of an h x w patch grid."""
    rs = np.random.RandomState(seed)
    h, w = grid_hw
    n_img = (h // 2) * (w // 2)
    hi = min(150000, cfg.vocab_size - 8, cfg.vision_start_token_id)
    lo = min(1000, hi - 1)
    row = rs.randint(lo, hi, 3).tolist() + [cfg.vision_start_token_id] + [cfg.image_token_id] * n_img + [cfg.vision_end_token_id] + rs.randint(lo, hi, n_text).tolist()
    ids = np.array([row], dtype=np.int64)
    px = torch.from_numpy(rs.standard_normal((h * w, cfg.patch_dim)).astype(np.float32)).to(dev)
    return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.ones_like(torch.from_numpy(ids)), "pixel_values": px, "image_grid_thw": torch.tensor([[1, h, w]])}


def real_shapes_leg(tr, cfg, dev, steps):
    """The shapes the reference's launch scripts actually run (REF scripts/train/SC_GRPO/SC_GRPO_Qwen_Instruct_2_5_VL_3B.sh:49-59): per_device_train_batch_size 1,
    num_generations 4, gradient_accumulation_steps 2, max_prompt_length 4096, max_completion_length 512, max_pixels 480000, sampling with EOS LIVE -- on RAGGED
    prompts: five image sizes <= 480 000 pixels (patch grids 32x32, 46x34, 50x48, 24x40, 34x46: 256 / 391 / 600 / 240 / 391 image tokens) and 100-800 text tokens, a
    different prompt length in every micro-batch.  Reported beside the headline, never as `value`: samples/s with one rollout per optimizer step (the default,
    GRPOConfig.batch_rollouts) and with one per micro-batch (the reference's order), and the counts of everything a changing shape can cost -- KV-pool rebuilds,
    hipGraph captures and their seconds, training-arena re-keys, EOS polls -- plus the completion lengths.  Random-init weights never emit the real EOS id
    (1 token of 151 936), so for this leg the EOS id is set to a token a probe rollout sampled ~0.7 % of the time: completions then end at ragged lengths
    (geometric, mean ~140).  Everything is restored afterwards."""
    from iadr1_amd import rollout as ro
    eng, pol = tr.engine, tr.engine.pol.p
    grids = [(32, 32), (46, 34), (50, 48), (24, 40), (34, 46)]
    rs = np.random.RandomState(99)
    n_micro = 2 + 2 * (2 + 2 * steps)
    batches = {("rs", k): synth_prompt_batch(cfg, grids[k % len(grids)], int(rs.randint(100, 801)), 4321 + k, dev) for k in range(n_micro)}
    saved_proc, saved_args, saved_eos = tr.processing_class, dict(vars(eng.args)), cfg.eos_token_id
    saved_tr = (tr.args.per_device_train_batch_size, tr.args.num_generations, tr.args.gradient_accumulation_steps, tr.args.max_completion_length, tr.args.batch_rollouts)

    class P_(SynthProcessor):
        def __call__(self, text=None, images=None, **kw):
            return self.batches[images[0][1]]
    chat = [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "Is there any defect in the image?"}]}]
    rows = lambda k: [{"prompt": chat, "image": [("synthetic", ("rs", k), 0)], "solution": SOLUTION}]
    res = {}
    try:
        tr.processing_class = P_(batches, CANNED)
        eng.args.num_generations, eng.args.max_completion_length, eng.args.max_prompt_length = 4, 512, 4096
        eng.args.gradient_accumulation_steps, eng.args.suppress_eos, eng.args.micro_batch_seqs = 2, False, 4
        tr.args.per_device_train_batch_size, tr.args.num_generations, tr.args.gradient_accumulation_steps, tr.args.max_completion_length = 1, 4, 2, 512
        # probe: which token does this random model sample ~0.7 % of the time?  That token plays EOS for the leg.
        from iadr1_amd.trainer import combine_batches
        probe = eng.rollout(combine_batches([batches[("rs", 0)], batches[("rs", 1)]], cfg.pad_token_id))
        ids_, cnt = np.unique(probe, return_counts=True)
        freq = cnt / probe.size
        pick = int(np.argmin(np.abs(freq - 0.007)))
        cfg.eos_token_id = int(ids_[pick])
        res["synthetic_eos"] = {"token": cfg.eos_token_id, "probe_frequency": float(freq[pick]), "distinct_tokens_in_probe": int(len(ids_))}
        k0 = 2
        for mode, on in (("one_rollout_per_optimizer_step", True), ("one_rollout_per_micro_batch", False)):
            tr.args.batch_rollouts = on
            tr.training_step([rows(k0), rows(k0 + 1)])        # one untimed optimizer step of the mode: pools, arenas and graph of its geometry
            k0 += 2
            torch.cuda.synchronize()
            st0 = dict(ro.STATS)
            tr._metrics.clear()
            t0 = time.perf_counter()
            first = k0
            for s_ in range(steps):
                tr.training_step([rows(k0), rows(k0 + 1)])
                k0 += 2
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            d = {k: ro.STATS[k] - st0[k] for k in ro.STATS}
            res[mode] = {"samples_per_s": 8 * steps / dt, "ms_per_optimizer_step": dt / steps * 1e3, "optimizer_steps": steps, "micro_batches": 2 * steps,
                         "prompt_lengths": [int(batches[("rs", k)]["input_ids"].shape[1]) for k in range(first, k0)],
                         "mean_completion_length_per_micro_batch": [round(float(x), 1) for x in tr._metrics.get("completion_length", [])],
                         "rollouts": d["rollouts"], "decode_steps_run": d["decode_steps"], "decode_steps_if_no_eos": d["rollouts"] * 511,
                         "kv_pool_rebuilds": d["pool_builds"], "graph_captures": d["graph_captures"], "graph_capture_seconds": round(d["capture_seconds"], 4),
                         "arena_rekeys": d["trace_rekeys"], "eos_polls_host": d["eos_polls"], "host_drains_for_eos": 0}
        res["note"] = ("reference launch-script shapes (B=1 x G=4, accum 2, max_completion_length 512, ragged prompts, EOS live); each mode after ONE untimed optimizer step; "
                       "the EOS poll waits for a 12-step-old flag copy in pinned memory, never for the newest work (no queue drain); not part of `value`")
    finally:
        cfg.eos_token_id = saved_eos
        tr.processing_class = saved_proc
        for k, v in saved_args.items():
            setattr(eng.args, k, v)
        tr.args.per_device_train_batch_size, tr.args.num_generations, tr.args.gradient_accumulation_steps, tr.args.max_completion_length, tr.args.batch_rollouts = saved_tr
    return res


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a torchrun environment: run the same command line as N ranks of one node (one rank per GPU over RCCL)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__), *sys.argv[1:]]
    raise SystemExit(subprocess.call(cmd, env=env))


D3 = {"text": {"vocab_size": 151936, "hidden_size": 2048, "intermediate_size": 11008, "num_hidden_layers": 36, "num_attention_heads": 16,
               "num_key_value_heads": 2, "rms_norm_eps": 1e-6, "rope_theta": 1e6, "mrope_section": [16, 24, 24]},
      "vision": {"depth": 32, "hidden_size": 1280, "intermediate_size": 3420, "num_heads": 16, "in_channels": 3, "patch_size": 14,
                 "spatial_merge_size": 2, "temporal_patch_size": 2, "window_size": 112, "out_hidden_size": 2048, "fullatt_block_indexes": [7, 15, 23, 31]},
      "tie_word_embeddings": True}


_REAL_STDOUT = None


def emit(obj):
    """The ONE JSON line of the run, on the process's real stdout."""
    out = _REAL_STDOUT if _REAL_STDOUT is not None else sys.stdout
    out.write(json.dumps(obj) + "\n")
    out.flush()


def claim_stdout():
    """The bench contract is ONE JSON line on rank 0's stdout.  Libraries print there too (RCCL writes a five-line version banner to stdout when a communicator is
    created), so file descriptor 1 is pointed at stderr for the life of the process and the JSON line goes to a saved duplicate of the original stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def main():
    a = parse()
    if a.cpu_full_step and a.check:      # the full-size parity record: HIP engine on cuda:0 against the oracle on the host cores
        import iadr1_amd  # noqa: F401
        if a.workload == "pa_sft":
            emit({"pa_sft_full_size_parity": pa_sft_full_size_parity(a)})
            return
        rec = full_size_parity(a)
        emit({"full_size_parity": rec})
        return
    if a.cpu_full_step:      # host cores only: the oracle's real step, once
        import iadr1_amd  # noqa: F401
        emit({"cpu_full_step": cpu_full_step(D3, P=a.prompt_len, C=a.gen_len, G=a.group)})
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(a.gpus)          # (the ranks inherit this process's stdout)
    claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} launched with WORLD_SIZE={world}: the two must agree (one rank per GPU)")
    if a.launch_check:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        assert dist.get_world_size() == world == a.gpus and dist.get_rank() == rank
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)           # the max-over-ranks step of the timing contract
        seen = [None] * world
        dist.all_gather_object(seen, (rank, local))
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            emit({"launch_check": True, "n_gpus": a.gpus, "world_size": world, "max_over_ranks": float(t), "ranks_seen": sorted(r for r, _ in seen),
                              "local_ranks": sorted(l for _, l in seen), "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"})
        return
    # IADR1_BENCH_SHARE_GPU=1 + IADR1_BENCH_BACKEND=gloo: every rank on cuda:0 with the exchange over gloo -- how the N > 1 path of this file runs end to end on a
    # ONE-GPU box (tests/test_hip_model.py); RCCL refuses two ranks on one device.  Never the measured configuration.
    share_gpu = os.environ.get("IADR1_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("IADR1_BENCH_BACKEND", "nccl")
    if share_gpu:
        local = 0
        os.environ["LOCAL_RANK"] = "0"          # the trainer places its parameter stores on cuda:LOCAL_RANK
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rccl_ranks = 1
    if world > 1 or os.environ.get("IADR1_FORCE_REDUCE"):   # IADR1_FORCE_REDUCE=1 under torchrun --nproc-per-node 1: the RCCL exchange path on one GPU
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        rccl_ranks = dist.get_world_size()
        assert rccl_ranks == world
    if a.decode_weights == "fp8":
        os.environ["IADR1_DECODE_WEIGHTS"] = "fp8"       # read by every ParamStore of this process
    import iadr1_amd  # noqa: F401
    from iadr1_amd import rewards
    from iadr1_amd.params import ParamStore, VLMConfig
    from iadr1_amd.trainer import GRPOConfig, SCGRPOTrainer

    if a.model == "3b":
        cfg = VLMConfig.qwen25vl_3b()
    elif a.model == "7b":
        cfg = VLMConfig.qwen25vl_7b()
    elif a.model == "qwen2vl_2b":
        cfg = VLMConfig.qwen2vl_2b()      # BASELINE config 1 (Qwen2-VL-2B PA-SFT): LayerNorm / QuickGELU ViT, no window attention
    elif a.model == "llava_ov_7b":
        cfg = VLMConfig.llava_ov_7b()     # BASELINE config 5 (LLaVA-OneVision-SI-7B SC-GRPO): SigLIP tower + any-resolution packing + Qwen2-7B decoder
    elif a.model == "llava15_7b":
        cfg = VLMConfig.llava15_7b()      # SC_GRPO_LLaVA_1_5.sh: CLIP tower, one 336-pixel crop (576 image tokens), vicuna-7b decoder
    elif a.model == "llava_next_7b":
        cfg = VLMConfig.llava_next_7b()   # SC_GRPO_LLaVA_1_6.sh: CLIP tower, any-resolution crops, Mistral-7B decoder
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import fixture_util as fx
        cfg = VLMConfig.from_dict(fx.TINY)
    if a.workload == "pa_sft":
        return run_pa_sft(a, cfg, dev, rank, world)
    llava = a.model in ("llava_ov_7b", "llava15_7b", "llava_next_7b")
    if a.micro_batch <= 0:
        # llava_ov_7b: 3699 image tokens per prompt -> one group (prompt + its 8 completions) per pass keeps the saved activations at ~26 GB
        a.micro_batch = 32 if a.model in ("7b", "llava15_7b") else (a.group if llava else 64)
    pol = ParamStore(cfg, dev, trainable=True)
    pol.init_random(seed=0)
    N = a.prompts * a.group
    # inputs are generated and made resident in HBM BEFORE the timed region (the processor's fp32 patches stay fp32: the bf16 cast is part of the step)
    batches = []
    real_legs = world == 1 and not llava and not a.no_real_processor_legs and a.prompt_len >= 261
    leg_steps = min(a.steps, 4)       # the two real-processor legs are bounded: they are a comparison beside `value`, not the timed region
    for step_id in range(a.warmup + a.steps + 1 + (2 * (leg_steps + 1) + 1 if real_legs else 0)):
        if llava:
            b = synth_batch_llava(cfg, a.prompts, 253, seed=1234 + 7919 * rank + step_id)
            bb = {"input_ids": torch.from_numpy(b["input_ids"]), "attention_mask": torch.from_numpy(b["attention_mask"]), "pixel_values": b["pixel_values"].to(dev)}
            if "image_sizes" in b:
                bb["image_sizes"] = torch.tensor(b["image_sizes"])
            batches.append(bb)
            continue
        b = synth_batch(cfg, a.prompts, a.prompt_len, seed=1234 + 7919 * rank + step_id)
        batches.append({"input_ids": torch.from_numpy(b["input_ids"]), "attention_mask": torch.from_numpy(b["attention_mask"]), "pixel_values": b["pixel_values"].to(dev),
                        "image_grid_thw": torch.tensor(b["image_grid_thw"])})
    chat = [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "Is there any defect in the image?"}]}]
    rows = lambda step_id: [{"prompt": chat, "image": [("synthetic", step_id, j)], "solution": SOLUTION} for j in range(a.prompts)]
    # the reference's API: trainer object + reward plugins (train/stage_rl/grpo_ad.py:188-199); the frozen reference model is the trainer's own copy
    tr = SCGRPOTrainer((cfg, pol), [rewards.accuracy_reward, rewards.consistency_reward],
                       args=GRPOConfig(output_dir="/tmp/iadr1_bench", per_device_train_batch_size=a.prompts, num_generations=a.group,
                                       max_completion_length=a.gen_len, micro_batch_seqs=a.micro_batch, seed=1234 + rank, save_steps=0,
                                       max_prompt_length=None if llava else a.prompt_len),
                       train_dataset=None, processing_class=SynthProcessor(batches, CANNED))
    eng = tr.engine
    eng.args.suppress_eos = True            # fixed-length completions, every sequence generates gen_len tokens
    if a.gradient_checkpointing is None:
        # one GPU, no process group: "off" (the headline keeps its activations).  Under a process group: "auto" -- the static budget of
        # Engine.recompute_wanted then reserves RCCL's buffers, so the 7B-class configurations recompute instead of sitting at 255 of 288 GB
        a.gradient_checkpointing = "auto" if rccl_ranks > 1 or os.environ.get("IADR1_FORCE_REDUCE") else "off"
    eng.args.recompute = a.gradient_checkpointing
    eng.args.use_hip_graph = not a.no_graph
    timer = GemmTimer()
    timer.install()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    step_id = 0
    for _ in range(a.warmup):
        tr.training_step([rows(step_id)])
        step_id += 1
    barrier()
    ms0 = torch.cuda.memory_stats()
    alloc_trace = os.environ.get("IADR1_ALLOC_TRACE") == "1"     # who asks the device for memory inside the timed region (stderr; a diagnosis switch)
    if alloc_trace:
        torch.cuda.memory._record_memory_history(enabled="all", context="alloc", stacks="python", max_entries=200000)
    timer.enabled = True
    if eng._rollout is not None:
        eng._rollout.decode_events = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.training_step([rows(step_id)])
        step_id += 1
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    ms1 = torch.cuda.memory_stats()      # read HERE: the extra legs below (other layout, real processor) are not the timed region
    hbm = {"peak_allocated_GB": torch.cuda.max_memory_allocated() / 2**30, "peak_reserved_GB": torch.cuda.max_memory_reserved() / 2**30,
           "reserved_after_timed_region_GB": torch.cuda.memory_reserved() / 2**30,
           "alloc_retries": ms1.get("num_alloc_retries", 0),
           "device_allocs_in_timed_region": ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0),
           "device_frees_in_timed_region": ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)}
    if alloc_trace:
        snap = torch.cuda.memory._snapshot()
        torch.cuda.memory._record_memory_history(enabled=None)
        import collections
        print("[alloc-trace] events by action:", dict(collections.Counter(ev["action"] for tr_ in snap["device_traces"] for ev in tr_)), file=sys.stderr)
        for tr_ in snap["device_traces"]:
            for ev in tr_:
                if ev["action"] in ("segment_alloc", "segment_free"):
                    fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", []) if "iad-r1_amd" in f["filename"] or "bench.py" in f["filename"]][:4]
                    print(f"[alloc-trace] {ev['action']:13s} {ev['size'] / 2**20:9.1f} MiB  {' <- '.join(fr)}", file=sys.stderr)
    metrics = {k: (sum(v) / len(v) if v else None) for k, v in tr._metrics.items()}
    traced = bool(getattr(eng, "last_step_traced", False))      # read now: the extra (untimed) leg below runs the other layout
    co_sched = None
    if getattr(eng, "last_step_shadowed", False):
        ro, sh = eng._rollout, eng._shadow
        ncu_dev = torch.cuda.get_device_properties(dev).multi_processor_count
        mlp = bool(ro.trace and ro.trace.get("mlp_on_shadow"))
        co_sched = {"what": "the frozen reference's teacher-forced pass (vision tower, decoder, lm_head log-probs) runs UNDER the rollout on a second HIP stream, one chunk of decode steps' "
                            "rows at a time (iadr1_amd/overlap.py; bit-equal to the one-shot pass: tests/test_hip_model.py::test_chunked_reference_pass_is_bit_equal_to_the_one_shot_pass)",
                    "side_stream_cus": (ncu_dev - ro.decode_cus) if ro.decode_cus else None, "decode_stream_cus": ro.decode_cus or ncu_dev, "chunk_decode_steps": sh.steps,
                    "policy_mlp_rows_rebuilt_on_side_stream": mlp,
                    "rebuilt_gemm_tflop_per_step": (2.0 * N * a.gen_len * 2 * cfg.intermediate_size * cfg.hidden_size * cfg.num_hidden_layers / 1e12) if mlp else 0.0,
                    "rebuilt_note": "the policy's gate|up + SwiGLU rows of the completion tokens are recomputed from the decode steps' stored h2 rows instead of being stored by the decode "
                                    "kernel (iadr1_gemm_swiglu_rows_bf16); REDUNDANT FLOPs: not counted in roofline_gemm's achieved" if mlp else None,
                    "switch": "IADR1_OVERLAP_CUS=" + os.environ.get("IADR1_OVERLAP_CUS", "auto") + " (0: the reference pass after the rollout on the whole device)",
                    "roofline_note": "roofline_decode is measured WHILE the side stream runs: the decode replays own decode_stream_cus CUs and share HBM with the reference pass, so their "
                                     "ms_per_decode_step is higher (3B: 3.06 vs 2.80 ms alone on 256 CUs) although the step is shorter; roofline_gemm prices side-stream launches at their CU share"}
    dec_ev, eng._rollout.decode_events = eng._rollout.decode_events, None
    per_rank_ms = [dt / a.steps * 1e3]
    per_rank_structure = None
    exposed_ms = eng.reducer.exposed_ms() if eng.reducer.active else None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)                       # each rank's own wall time between the two barriers (they differ by the skew of the closing barrier only)
        per_rank_ms = [float(x) / a.steps * 1e3 for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        # which step structure every rank ran (the decision is collective -- overlap.agree_across_ranks -- so these must all be equal; printed so that a run shows it)
        mine = {"rank": rank, "co_scheduled": bool(getattr(eng, "last_step_shadowed", False)), "decode_stream_cus": int(getattr(eng._rollout, "decode_cus", 0) or 0) or None,
                "recompute_forced": bool(eng.pol.__dict__.get("_recompute_forced"))}
        per_rank_structure = [None] * world
        dist.all_gather_object(per_rank_structure, mine)
    # one extra step OUTSIDE the timed region in the reference's own layout (every prompt repeated in all G rows of its group,
    # IADR1_SHARE_PREFIX=0) so the line also says what the dedup of the prompt tokens is worth; N=1 only
    repeated = None
    if world == 1 and not a.no_repeated_rows_leg and eng.args.share_prefix and a.group > 1:
        try:
            eng.args.share_prefix, eng.args.micro_batch_seqs = False, min(a.micro_batch, 32)
            tr.training_step([rows(step_id)])          # buffers of this layout
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tr.training_step([rows(step_id)])
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            repeated = {"samples_per_s": N / d1, "ms_per_step": d1 * 1e3, "micro_batch_seqs": eng.args.micro_batch_seqs, "steps": 1,
                        "note": "same step with the prompt tokens recomputed in all G rows (the reference's [B*G, P+C] layout); not part of `value`"}
        except Exception as exc:  # the extra leg must never cost the headline line
            repeated = {"error": repr(exc)[:200]}
        finally:
            eng.args.share_prefix, eng.args.micro_batch_seqs = True, a.micro_batch
    real = None
    if real_legs:
        try:
            real = real_processor_legs(tr, batches, a.prompts, leg_steps, step_id + 1, N)
            real["prefetch_vs_headline"] = real["prefetch"]["samples_per_s"] / (world * N * a.steps / dt)
        except Exception as exc:
            real = {"error": repr(exc)[:300]}
    shapes = None
    if world == 1 and a.model in ("3b", "7b", "qwen2vl_2b", "tiny") and not a.no_real_shapes_leg and not llava:
        timer.enabled = False
        try:
            shapes = real_shapes_leg(tr, cfg, dev, a.real_shapes_steps)
        except Exception as exc:
            import traceback
            shapes = {"error": repr(exc)[:300], "where": traceback.format_exc()[-600:]}
    if rank == 0:
        n_launch, t_sum, fl_gemm = timer.summary()
        t_gemm = timer.busy_seconds()            # union of the launch intervals (wgrad GEMMs overlap dgrad GEMMs on a second stream)
        ach = fl_gemm / max(t_gemm, 1e-9) / 1e12
        # HBM-side traffic of the heaviest GEMM shape and the MFMA-pipe busy fraction from the PMC passes recorded under profiles/ (separate rocprofv3
        # --pmc runs, gfx950 FETCH_SIZE correction applied there); null when the record is absent
        traffic = None
        pmc_file = _newest_profile("gemm_pmc.json")
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            k0 = pmc["kernels"][0]
            traffic = {"bytes_per_launch": k0["traffic_bytes"], "algorithmic_bytes_per_launch": k0["algorithmic_bytes"], "MNK": k0["MNK"], "source": "profiles/" + pmc_file, "round": int(pmc_file[1:3]),
                       "other_forms": [{"kernel": k.get("kernel"), "MNK": k.get("MNK"), "traffic_bytes": k.get("traffic_bytes"), "algorithmic_bytes": k.get("algorithmic_bytes")} for k in pmc["kernels"][1:]]}
        except Exception:
            pass
        mfma_busy = None
        try:
            mb_file = _newest_profile("mfma_busy.json")
            mb = json.load(open(os.path.join(ROOT, "profiles", mb_file)))
            import re as _re
            g0 = next(k for k in mb["kernels"] if _re.search(r"gemm_nt_256<0(, false)*>", k.get("kernel", "")) and k.get("grid") == 3522560)
            mfma_busy = {"kernel": "gemm_nt_256 [20480 x 22016 x 2048]", "mfma_busy_frac": g0["mfma_busy_frac"], "lds_conflict_frac": g0.get("lds_conflict_frac"),
                         "source": f"profiles/{mb_file} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, separate profiled pass)"}
        except Exception:
            pass
        model_name = ({"llava_ov_7b": "LLaVA-OneVision-SI-7B", "llava15_7b": "LLaVA-1.5-7B", "llava_next_7b": "LLaVA-NeXT-Mistral-7B"}[a.model] if llava
                      else ("Qwen2-VL-2B" if a.model == "qwen2vl_2b" else "Qwen2.5-VL-" + a.model.upper()))
        image_note = ("448x448 image through the family's processor geometry (model_note) + 256 text positions" if llava
                      else f"448x448 image (1024 patches -> 256 tokens) + {a.prompt_len} prompt positions")
        out = {
            "metric": "GRPO samples/sec (img448+512tok, group=8) Qwen2.5-VL-3B" if a.model == "3b" else f"GRPO samples/sec {a.model}",
            "model_note": ({"llava_ov_7b": "LLaVA-OneVision-SI-7B shapes (BASELINE config 5): 448x448 image -> 5 crops of 384x384 through the SigLIP tower -> 3699 packed image tokens + 256 text positions per prompt",
                            "llava15_7b": "llava-1.5-7b shapes: one 336x336 crop through the CLIP tower -> 576 image tokens + 256 text positions per prompt",
                            "llava_next_7b": "llava-v1.6-mistral-7b shapes: 448x448 image -> 5 crops of 336x336 through the CLIP tower -> 2928 packed image tokens + 256 text positions per prompt"}.get(a.model)
                           if llava else None),
            "value": world * N * a.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{model_name} SC-GRPO step: {a.prompts} prompts x group {a.group} per GPU, {image_note}, {a.gen_len} generated tokens (EOS suppressed), random-init weights, rollout + ref fwd + policy fwd/bwd + AdamW",
                       "api_entry": "SCGRPOTrainer.training_step -> SCGRPOTrainer.compute_loss (REF sc_grpo_trainer.py:586) -> SCGRPOEngine.step; reward plugins accuracy_reward + consistency_reward on canned completion strings",
                       "per_gpu_sequences": N, "micro_batch_seqs": a.micro_batch, "gradient_checkpointing": a.gradient_checkpointing, "reference_forward": "bf16", "decode_weights": a.decode_weights, "hip_graph_rollout": not a.no_graph, "parallelism": f"dp{world}", "rccl_ranks": rccl_ranks, "collective_backend": backend if (world > 1 or os.environ.get("IADR1_FORCE_REDUCE")) else None,
                       "grad_exchange": ({"wire": eng.reducer.wire, "algo": eng.reducer.algo, "bytes_on_wire": getattr(eng.reducer, "last_bytes_on_wire", 0), "n_buckets": getattr(eng.reducer, "last_n_buckets", 0),
                                          "exposed_ms": exposed_ms, "exposed_note": "GPU time the compute stream waited in GradReducer.finish() for the exchange after backward ended (last timed step, this rank): what did not hide under backward",
                                          "staging_bytes": eng.reducer.staging_bytes()}
                                         if eng.reducer.active else None),
                       "dedup": ("ViT once per image; prompt tokens once per group in the ref / policy passes (shared-prefix attention: identical math to the "
                                 "reference's G repeated rows, parity-tested); the rollout's prefill is the prompt part of the policy's training forward"
                                 + ("; the rollout's decode steps write the completion rows of the policy's activation arena (side outputs of the decode kernels), so the "
                                    "policy's forward over the completions is the decode itself and is not run a second time before backward (pinned to the oracle by "
                                    "tests/test_hip_model.py::test_api_step_with_rollout_handover_matches_the_oracle)" if traced else ""))
                       if eng.args.share_prefix else "ViT once per image"},
            "per_rank_ms_per_step": [round(x, 2) for x in per_rank_ms],
            "per_rank_step_structure": per_rank_structure,
            "repeated_rows_layout": repeated,
            "real_processor": real,
            "real_shapes": shapes,
            "samples_per_sec_per_gpu": N * a.steps / dt,
            "roofline_gemm": {"bound": "mfma", "kernel": "gemm_nt_256 / gemm_nt_128 (v_mfma_f32_16x16x32_bf16)", "achieved": ach, "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / MFMA_BF16_DENSE_PEAK_TFLOPS, "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_detail": traffic,
                         "traffic_note": ("FETCH_SIZE / WRITE_SIZE are counted at the L2's fabric side: the 3.4x over the algorithmic bytes are operand-panel re-reads that miss the 4 MB L2 of an XCD "
                                          "(64 resident 256x256 tiles arranged 4 x 16 reuse a panel 6.4x) and are served by the 256 MB memory-side cache -- A + B of this launch are 174 MB; "
                                          "the kernel is MFMA / power bound (mfma_busy, power_limit), not traffic bound"),
                         "mfma_busy": mfma_busy,
                         "power_limit": power_limit(),
                         "launches": n_launch, "launches_on_cu_masked_streams": timer.masked_launches(), "kernel_time_frac_of_step": t_gemm / dt,
                         "timing": "sum of algorithmic FLOPs of the launches / length of the union of their HIP-event intervals (weight-gradient GEMMs run on a side stream "
                                   "concurrently with the dgrad GEMMs; equals FLOPs / sum of launch durations when nothing overlaps: IADR1_WGRAD_STREAM=0); intervals of launches on a "
                                   "CU-masked stream (co_scheduling) count share-of-CUs x length, i.e. whole-device seconds",
                         "achieved_by_sum_of_launch_durations": fl_gemm / max(t_sum, 1e-9) / 1e12,
                         "whole_step_executed_gemm_tflops": fl_gemm / dt / 1e12, "whole_step_frac_of_mfma_peak": fl_gemm / dt / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS},
            "roofline_decode": decode_roofline(cfg, pol, dec_ev, N, a.gen_len),
            "co_scheduling": co_sched,
            "last_step_metrics": metrics,
            "gemm_by_shape": timer.by_shape(),
            "hbm": hbm,
        }
        # `roofline` = the launch that OWNS the step: the decode replay (HBM-bound, one hipGraph launch per generated token) when the replays take more
        # of the step than the GEMM family does, else the GEMM family; both objects stay in the line under their own names.
        dec = out["roofline_decode"]
        dec_frac = (dec["total_ms"] / a.steps) / (dt / a.steps * 1e3) if dec else 0.0
        gemm_frac = out["roofline_gemm"]["kernel_time_frac_of_step"]
        if dec:
            dec["kernel_time_frac_of_step"] = dec_frac
        dom = "decode" if (dec and dec_frac >= gemm_frac) else "gemm"
        out["roofline"] = dict(out["roofline_decode"] if dom == "decode" else out["roofline_gemm"])
        out["roofline"]["dominant"] = dom
        out["roofline"]["share_of_step"] = {"decode_replays": dec_frac, "gemm_family": gemm_frac}
        if not a.no_cpu_baseline and a.model == "3b" and world == 1:     # rank 0 at N = 1 only: the other ranks of a multi-GPU run would sit in the closing barrier
            live = cpu_baseline(D3, a.cpu_seconds, P=a.prompt_len, C=a.gen_len, G=a.group)
            out["cpu_baseline"] = live
            try:
                # The QUOTED baseline is the one complete full-size step of the same oracle (bench.py --cpu-full-step: ~12 minutes of host time and 127 GB, run once on the
                # GPU box's host, committed under profiles/): the bounded component sample of this run -- each component timed in isolation, multiplied by its count -- is
                # 2.1-2.5x optimistic and is carried beside it as `live_component_sample`, so that a change of host shows up.
                full_file = _newest_profile("cpu_full_step.json")
                full = json.load(open(os.path.join(ROOT, "profiles", full_file)))["cpu_full_step"]
                out["cpu_baseline"] = {"value": full["value"], "unit": "samples/s", "cores": full["cores"], "kind": "port", "seconds_per_step_B1_G8": full["seconds_per_step_B1_G8"],
                                       "sample": ("ONE complete B=1 x G=8 SC-GRPO step of the oracle at full Qwen2.5-VL-3B size (P=512, C=256), nothing extrapolated, measured once on this pool's "
                                                  "GPU-box host with bench.py --cpu-full-step (profiles/" + full_file + "); this run's own bounded sample: live_component_sample"),
                                       "parts_seconds": full.get("parts_seconds"), "source": "profiles/" + full_file, "round": int(full_file[1:3]),
                                       "live_component_sample": live, "live_over_record": live["value"] / full["value"]}
            except Exception:
                pass
        else:
            out["cpu_baseline"] = None
        emit(out)
    if world > 1 or os.environ.get("IADR1_FORCE_REDUCE"):
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
