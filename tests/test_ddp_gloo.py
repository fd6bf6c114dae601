"""world_size=2 gloo (CPU) checks of the data-parallel path: the gradient reducer covers every element of the flat
gradient buffer exactly once (layer buckets launched from the backward hook + the remainder in finish()), and sums
across ranks; prompts are sharded per rank with no data-path collective other than this one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd.params import ParamStore, VLMConfig
    from iadr1_amd.sc_grpo import GradReducer

    cfg = VLMConfig.from_dict(fx.TINY)
    st = ParamStore(cfg, "cpu", trainable=True, with_transposes=False, with_decode_pack=False)
    g = torch.Generator().manual_seed(100 + rank)
    st.grad.copy_(torch.randn(st.n_total, generator=g))
    mine = st.grad.clone()
    red = GradReducer(st)
    assert red.world == world
    # backward order: last layer first; only some layers fire the hook (the rest must be swept by finish())
    for i in reversed(range(cfg.num_hidden_layers)):
        if i % 2 == 1:
            red.layer_ready(i)
    red.finish()
    others = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(others, mine)
    want = torch.stack(others).sum(0)
    ok = torch.allclose(st.grad, want, rtol=0, atol=1e-6)
    # a second step reuses the reducer cleanly
    st.grad.copy_(mine)
    red.finish()
    ok2 = torch.allclose(st.grad, want, rtol=0, atol=1e-6)
    # the other collective of the path: rank-averaged metrics for the log line
    from iadr1_amd.trainer import average_over_ranks
    avg = average_over_ranks({"reward": 1.0 + rank, "kl": 0.5 * rank, "completion_length": 10.0})
    ok3 = avg == {"reward": 1.0 + (world - 1) / 2, "kl": 0.25 * (world - 1), "completion_length": 10.0}
    q.put((rank, bool(ok), bool(ok2) and bool(ok3), red.layer_range[0][0] > 0))
    dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ok2, nonzero_first in res:
        assert ok and ok2 and nonzero_first, (rank, ok, ok2)


def test_layer_buckets_are_disjoint_and_cover_decoder_weights():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd.params import ParamStore, VLMConfig
    from iadr1_amd.sc_grpo import GradReducer

    cfg = VLMConfig.from_dict(fx.TINY)
    st = ParamStore(cfg, "cpu", trainable=True, with_transposes=False, with_decode_pack=False)
    red = GradReducer(st)
    prev_hi = None
    for i, (lo, hi) in enumerate(red.layer_range):
        assert lo < hi <= st.n_decay
        if prev_hi is not None:
            assert lo == prev_hi
        prev_hi = hi
        for k in ("qkv.w", "o.w", "gu.w", "down.w"):
            s = st.slots[f"layers.{i}.{k}"]
            assert lo <= s.offset and s.offset + int(np.prod(s.shape)) <= hi
    # decay / no-decay split used by the two AdamW launches
    for name, s in st.slots.items():
        assert (s.offset < st.n_decay) == s.decay, name
