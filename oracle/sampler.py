"""ORACLE (test infrastructure): CPU restatement of the rollout sampler's filter semantics
(temperature -> top-k -> top-p on the top-k-renormalised distribution -> multinomial), i.e. what
vLLM's SamplingParams(temperature, top_p=0.9, top_k=50) asks for at
/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:353-358 (vllm==0.7.3 is a third-party dependency
that is not under /root/reference; its sampler's RNG stream cannot be reproduced, so parity here is on the
candidate set + on an inverse-CDF draw with a documented Philox4x32-10 counter stream).  "parity unpinned"
by reference tests: the reference holds no test for its sampling."""
from __future__ import annotations

import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox_uniform(seed: int, row: int, step: int) -> float:
    k0, k1 = seed & MASK, (seed >> 32) & MASK
    c = [row & MASK, step & MASK, 0, 0]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c[3] ^ k1) & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return float(np.float32(c[0] >> 8) * np.float32(1.0 / 16777216.0))


def candidates(logits: np.ndarray, temperature: float, top_k: int, top_p: float):
    """(token ids in descending-probability order, their unnormalised fp32 weights) after top-k/top-p."""
    x = logits.astype(np.float32)
    order = np.lexsort((np.arange(x.size), -x))[: min(top_k, x.size)]  # value desc, index asc
    z = x[order] / np.float32(temperature)
    w = np.exp(z - z[0]).astype(np.float32)
    tot = np.float32(w.sum(dtype=np.float32))
    keep, before = [], np.float32(0.0)
    for j in range(len(w)):
        if j > 0 and before >= np.float32(top_p) * tot:
            break
        keep.append(j)
        before = np.float32(before + w[j])
    return order[keep], w[keep]


def sample_row(logits: np.ndarray, temperature: float, top_k: int, top_p: float, seed: int, row: int, step: int):
    """Returns (token, margin) where margin is the distance of u from the nearest CDF edge (tests skip
    near-edge draws, where fp32 summation order could legitimately flip the pick)."""
    if temperature <= 0:
        return int(np.argmax(logits)), 1.0
    ids, w = candidates(logits, temperature, top_k, top_p)
    cdf = np.cumsum(w.astype(np.float64))
    u = philox_uniform(seed, row, step) * cdf[-1]
    pick = int(np.searchsorted(cdf, u, side="right"))
    pick = min(pick, len(ids) - 1)
    margin = float(np.min(np.abs(cdf - u)) / cdf[-1])
    return int(ids[pick]), margin
