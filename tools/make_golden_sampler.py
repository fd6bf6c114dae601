#!/usr/bin/env python3
"""Third-party pin of the rollout sampler's FILTER semantics (tests/golden/sampler_hf.npz).

The reference samples through vLLM `SamplingParams(temperature, top_p=0.9, top_k=50)`
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:353-358).  vllm==0.7.3 is not importable here and its RNG
stream is not reproducible; what CAN be pinned to code we did not write is the candidate set and the renormalised
distribution the draw is taken from.  transformers' logits warpers -- the chain `generate(do_sample=True, temperature, top_k,
top_p)` applies, and the same rule as vLLM's `_apply_top_k_top_p` (ascending sort, drop while cumulative probability
<= 1 - top_p, on the top-k-masked distribution) -- are run on fixed logits:

    TemperatureLogitsWarper(t) -> TopKLogitsWarper(k) -> TopPLogitsWarper(p) -> softmax

and the kept index set + probabilities of every row are stored.  Consumers: tests/test_oracle_model.py (oracle/sampler.py
`candidates` against the sets / probabilities, CPU) and tests/test_hip_kernels.py (sample.hip: membership of every draw and
the empirical frequencies of 10^4 draws within 3 sigma, GPU).

Rows cover: gaussian logits at three scales (flat -> top-p cuts nothing below k, peaked -> top-p cuts to a handful), a
one-hot-like row (a single candidate), exact ties INSIDE the kept set, a vocabulary smaller than top_k, bf16-rounded logits
(many near-equal values), and the three (temperature, top_k, top_p) settings the path uses (0.9/50/0.9 = the scripts' values,
1.0/50/0.9 = GRPOConfig's default temperature, 0.7/20/0.8 as an off-default check).  A row is rejected at generation time when
its top-p boundary is within 1e-4 of a cumulative probability (fp32 summation order may then legitimately move the boundary token).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import transformers
from transformers import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "sampler_hf.npz")
KMAX = 64


def hf_filter(logits: np.ndarray, t: float, k: int, p: float):
    x = torch.from_numpy(logits.astype(np.float32))[None]
    ids = torch.zeros(1, 1, dtype=torch.long)
    x = TemperatureLogitsWarper(t)(ids, x)
    x = TopKLogitsWarper(min(k, x.shape[-1]))(ids, x)
    x = TopPLogitsWarper(p)(ids, x)
    pr = torch.softmax(x, -1)[0].numpy()
    keep = np.nonzero(np.isfinite(x[0].numpy()))[0]
    order = keep[np.lexsort((keep, -pr[keep]))]       # probability descending, index ascending
    return order.astype(np.int64), pr[order].astype(np.float32)


def boundary_margin(logits, t, k, p):
    z = np.sort(logits.astype(np.float64))[::-1][: min(k, logits.size)] / t
    w = np.exp(z - z[0])
    c = np.cumsum(w) / w.sum()
    before = np.concatenate([[0.0], c[:-1]])
    return float(np.min(np.abs(before[1:] - p))) if len(before) > 1 else 1.0


def main():
    rng = np.random.RandomState(20260929)
    V = 4096
    rows, settings = [], []

    def add(x, t=0.9, k=50, p=0.9):
        assert boundary_margin(x, t, k, p) > 1e-4, "row too close to the top-p boundary: change the seed"
        rows.append(np.asarray(x, dtype=np.float32))
        settings.append((t, k, p))

    for scale in (0.5, 2.5, 8.0):
        for _ in range(3):
            add(rng.randn(V) * scale)
    x = rng.randn(V) * 2.0
    x[1234] = 40.0
    add(x)                                               # one candidate
    x = rng.randn(V) * 1.0
    x[[7, 300, 2999]] = 6.0                              # exact ties inside the kept set
    x[[11, 12]] = 5.5
    add(x)
    add(rng.randn(37) * 1.5)                             # vocabulary < top_k
    add(torch.from_numpy((rng.randn(V) * 3.0).astype(np.float32)).to(torch.bfloat16).float().numpy())
    for _ in range(3):
        add(rng.randn(V) * 2.5, t=1.0)
    for _ in range(3):
        add(rng.randn(V) * 2.5, t=0.7, k=20, p=0.8)
    n = len(rows)
    vmax = max(r.size for r in rows)
    logits = np.full((n, vmax), -np.inf, dtype=np.float32)
    vocab = np.zeros(n, dtype=np.int64)
    ids = np.full((n, KMAX), -1, dtype=np.int64)
    probs = np.zeros((n, KMAX), dtype=np.float32)
    count = np.zeros(n, dtype=np.int64)
    for i, (r, (t, k, p)) in enumerate(zip(rows, settings)):
        logits[i, : r.size] = r
        vocab[i] = r.size
        o, pr = hf_filter(r, t, k, p)
        assert 1 <= len(o) <= KMAX and abs(float(pr.sum()) - 1.0) < 1e-5
        ids[i, : len(o)], probs[i, : len(o)], count[i] = o, pr, len(o)
    np.savez_compressed(OUT, logits=logits, vocab=vocab, settings=np.asarray(settings, dtype=np.float64), ids=ids, probs=probs, count=count,
                        meta=np.asarray(f"transformers {transformers.__version__} TemperatureLogitsWarper->TopKLogitsWarper->TopPLogitsWarper->softmax; torch {torch.__version__}; {os.path.basename(__file__)}"))
    print("wrote", OUT, "rows", n, "kept per row", count.tolist())


if __name__ == "__main__":
    sys.exit(main())
