"""Host-side training-loop arithmetic both entry points share: learning-rate schedules, the per-epoch sample order and its
split over data-parallel ranks.  The reference gets all three from transformers.Trainer (TF:trainer.py `_inner_training_loop`,
TF:optimization.py:101-140 schedule lambdas, torch DistributedSampler through accelerate); restated here so that every rank derives the
SAME number of optimizer steps and the same collective sequence whatever the dataset size.
"""
from __future__ import annotations

import math

import numpy as np

SCHEDULES = ("linear", "cosine", "constant", "constant_with_warmup")


def lr_at(step: int, total: int, base_lr: float, warmup_steps: int = 0, kind: str = "linear") -> float:
    """Learning rate of optimizer step number `step` (0-based: the number of optimizer steps already taken, which is what HF's LambdaLR
    holds when that step runs -- so with warmup the very first step runs at lr 0, TF:optimization.py:101-104,134-140)."""
    if kind not in SCHEDULES:
        raise ValueError(f"lr_scheduler_type {kind!r} is not one of {SCHEDULES}")
    if kind == "constant":
        return base_lr
    if warmup_steps and step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    if kind == "constant_with_warmup":
        return base_lr
    progress = (step - warmup_steps) / max(1, total - warmup_steps)
    if kind == "cosine":
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
    return base_lr * max(0.0, (total - step) / max(1, total - warmup_steps))


def rows_per_rank(n_rows: int, world: int) -> int:
    """DistributedSampler(drop_last=False): every rank sees ceil(n / world) rows (the tail wraps around to the head)."""
    return (n_rows + world - 1) // world


def total_steps(n_rows: int, world: int, batch_size: int, grad_accum: int, num_train_epochs: float = 1.0, max_steps: int = -1) -> int:
    """Optimizer steps of a run -- a function of the GLOBAL dataset length only, identical on every rank
    (TF:trainer.py: len(dataloader) = ceil(rows per rank / bs); updates per epoch = max(len // grad_accum, 1); ceil(epochs * updates))."""
    if max_steps and max_steps > 0:
        return int(max_steps)
    per_rank = rows_per_rank(n_rows, world)
    n_batches = (per_rank + batch_size - 1) // batch_size
    return int(math.ceil(max(n_batches // grad_accum, 1) * num_train_epochs))


def epoch_order(n_rows: int, seed: int, epoch: int, shuffle: bool = True) -> np.ndarray:
    """Sample order of one epoch: a seeded permutation that is the same on every rank (seed + epoch, as DistributedSampler.set_epoch)."""
    if not shuffle:
        return np.arange(n_rows, dtype=np.int64)
    return np.random.RandomState((int(seed) + int(epoch)) % (2**32)).permutation(n_rows).astype(np.int64)


def shard(order: np.ndarray, rank: int, world: int) -> np.ndarray:
    """This rank's slice of an epoch order, padded by wrap-around so that all ranks hold exactly rows_per_rank rows (DistributedSampler)."""
    n = len(order)
    if n == 0:
        return order
    per = rows_per_rank(n, world)
    padded = np.concatenate([order, order[: per * world - n]]) if per * world > n else order
    while len(padded) < per * world:        # datasets shorter than the world size
        padded = np.concatenate([padded, order[: per * world - len(padded)]])
    return padded[rank::world]


class RankSampler:
    """Row indices for (rank, world), epoch after epoch; position `k` (0-based count of rows this rank has consumed since step 0) maps to a
    dataset index deterministically, so a resumed run continues exactly where the stopped one would have gone.
    Deviation from transformers.Trainer, stated: rows are consumed as ONE continuous stream with always-full batches (epoch = k // rows_per_rank), while
    `total_steps` uses HF's per-epoch formula.  When the batch size divides rows_per_rank and the accumulation count divides the batches of an epoch -- true
    for every reference launch script (batch 1; 2 accumulation steps over an even shard) -- the two coincide.  Otherwise HF ends an epoch with a short batch
    (and, in 4.51, a remainder update) where this sampler lets a batch straddle the epoch boundary: the run sees a few rows more or fewer per "epoch" and
    the logged epoch fraction drifts accordingly.  Step counts and the collective sequence are identical across ranks either way."""

    def __init__(self, n_rows: int, rank: int, world: int, seed: int = 42, shuffle: bool = True):
        if n_rows <= 0:
            raise ValueError("empty training set")
        self.n, self.rank, self.world, self.seed, self.shuffle = n_rows, rank, world, seed, shuffle
        self.per = rows_per_rank(n_rows, world)
        self._epoch, self._idx = -1, None

    def index(self, k: int) -> int:
        ep, off = divmod(int(k), self.per)
        if ep != self._epoch:
            self._epoch, self._idx = ep, shard(epoch_order(self.n, self.seed, ep, self.shuffle), self.rank, self.world)
        return int(self._idx[off])
