"""Weight prefetcher of the group rollout (the decode loop of /root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:637-683, vLLM's side of it):
one persistent launch per rollout on a small CU-masked stream pulls the decode step's weights into the 256 MB memory-side cache just ahead of the launches
that stream them (csrc/prefetch.hip, include/iadr1_hip.h iadr1_weight_prefetch), paced by the progress word the first kernel of every decoder layer stores.

Opt-in (IADR1_WPREFETCH_CUS > 0, co-scheduled step only) and OFF by default: prefetching the next layer's q|k|v + o weights takes 0.06 - 0.08 ms off the decode step
(2 - 2.6 %), prefetching the mlp streams does not pay at all, and inside the co-scheduled step the third queue costs the step more than the decode step gains
(profiles/EXPERIMENTS.md round 6 has every number).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import hip, ops


def wanted_cus() -> int:
    return int(os.environ.get("IADR1_WPREFETCH_CUS", "0"))


class WeightPrefetcher:
    """engine: the policy's Engine (its decode-packed weight copies are what the decode step streams); stream: a CU-masked stream that owns `n_cus` CUs."""

    def __init__(self, engine, stream, n_cus: int, what=("gu", "down"), lead: int = 0, nt: bool = False, next_what=()):
        self.e, self.stream, self.n_cus, self.lead, self.nt = engine, stream, int(n_cus), int(lead), bool(nt)
        c, P = engine.cfg, engine.p
        L = c.num_hidden_layers
        rows = []
        self._keep = []
        for i in range(L):
            segs = []
            for name in what:                 # this layer's streams, in the order the step reads them
                segs.append(P.wpk(f"layers.{i}.{name}.w"))
            for name in next_what:            # the NEXT layer's narrow projections (read before its mlp streams)
                segs.append(P.wpk(f"layers.{(i + 1) % L}.{name}.w"))
            self._keep += segs
            rows.append([[t.data_ptr(), t.numel() * t.element_size() // 16 * 16] for t in segs])
        self.n_units, self.n_seg = L, len(rows[0])
        self.bytes_per_unit = sum(b for _, b in rows[0])
        self.segs = ops.h2d(np.asarray(rows, dtype=np.int64), engine.dev)
        self.status = torch.zeros(4, dtype=torch.int32, device=engine.dev)
        self.status_host = torch.zeros(4, dtype=torch.int32).pin_memory()
        self.launched = 0

    def start(self, mark: torch.Tensor, first_mark: int, last_mark: int, stop: torch.Tensor, epoch: int, timeout_ms: int = 20000):
        """Enqueue the rollout's prefetcher on its own stream, with NO dependency on any other stream (it starts polling at once; `mark` reads 0 until the decode replays
        that follow on their stream publish the marks; it returns when `stop` reaches `epoch`)."""
        with torch.cuda.stream(self.stream):
            hip.call("weight_prefetch", self.segs, self.n_units, self.n_seg, mark, int(first_mark), int(last_mark), stop, int(epoch), self.lead, int(self.nt), self.n_cus,
                     int(timeout_ms), self.status)
            self.status_host.copy_(self.status, non_blocking=True)
        self.launched += 1

    def report(self) -> dict:
        """(synchronises the prefetcher's stream) what the last launch did, from block 0's point of view."""
        self.stream.synchronize()
        s = self.status_host.tolist()
        return {"timed_out": bool(s[0] & 1), "units_read": s[1], "units_stale": s[2], "mib_read": s[3], "mib_per_unit": round(self.bytes_per_unit / 2 ** 20, 1)}
