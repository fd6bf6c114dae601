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


def _worker(rank, world, port, q, wire="fp32"):
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
    red = GradReducer(st, wire=wire, bucket_bytes=1 << 16)      # small buckets: every range is split into several collectives
    assert red.world == world
    # the bf16 wire stages through a RING of bucket-sized buffers (reused many times here: ~50 buckets through 3 slots), never a second copy of the model
    assert red.staging_bytes() == (3 * (1 << 16) if wire == "bf16" else 0)
    # backward order: last layer first; only some layers fire the hook (the rest must be swept by finish())
    for i in reversed(range(cfg.num_hidden_layers)):
        if i % 2 == 1:
            red.layer_ready(i)
    red.finish()
    others = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(others, mine)
    want = torch.stack(others).sum(0)
    # fp32 wire: the exact fp32 sum.  bf16 wire (the default, what DeepSpeed ZeRO-3 moves for a bf16 model): each addend and the sum rounded to bf16
    tol = dict(rtol=0, atol=1e-6) if wire == "fp32" else dict(rtol=2**-6, atol=2**-6)
    ok = torch.allclose(st.grad, want, **tol) and red.last_n_buckets > len(red.layer_range) + 1
    # a second step reuses the reducer cleanly
    st.grad.copy_(mine)
    red.finish()
    ok2 = torch.allclose(st.grad, want, **tol)
    # the other collective of the path: rank-averaged metrics for the log line
    from iadr1_amd.trainer import average_over_ranks
    avg = average_over_ranks({"reward": 1.0 + rank, "kl": 0.5 * rank, "completion_length": 10.0})
    ok3 = avg == {"reward": 1.0 + (world - 1) / 2, "kl": 0.25 * (world - 1), "completion_length": 10.0}
    q.put((rank, bool(ok), bool(ok2) and bool(ok3), red.layer_range[0][0] > 0))
    dist.destroy_process_group()


@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_grad_reducer_world2_gloo(wire):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, wire)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, ok2, nonzero_first in res:
        assert ok and ok2 and nonzero_first, (rank, ok, ok2)


def test_reducer_staging_is_bucket_sized_not_model_sized():
    """a full-model bf16 staging buffer (16.6 GB at 7B) would not fit next to the 7B training state.  The ring holds RING buckets of
    at most 256 MB each, whatever the model size (checked on the class constants; the 2-rank run above checks the tiny model's actual allocation)."""
    sys.path.insert(0, ROOT)
    import iadr1_amd  # noqa: F401
    from iadr1_amd.sc_grpo import GradReducer
    import inspect
    default_bucket = inspect.signature(GradReducer.__init__).parameters["bucket_bytes"].default
    assert GradReducer.RING * default_bucket <= 1 << 30


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


def _odd_rows_worker(rank, world, port, q):
    """Sharding of a dataset whose size is not a multiple of the world size: all ranks derive the same step count and issue the same number of
    gradient exchanges (a rank that stopped early would leave the others waiting in the all-reduce)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import iadr1_amd  # noqa: F401
    from iadr1_amd import schedule
    n_rows, bs, ga = 7, 1, 2
    total = schedule.total_steps(n_rows, world, bs, ga, num_train_epochs=2.0)
    sampler = schedule.RankSampler(n_rows, rank, world, seed=11)
    seen, i = [], 0
    t = torch.zeros(1)
    for step in range(total):
        for _ in range(ga):
            seen.append(sampler.index(i))
            i += bs
        dist.all_reduce(t)            # stands in for the per-step gradient exchange: hangs (test timeout) if the ranks disagree on `total`
    q.put((rank, total, seen))
    dist.destroy_process_group()


def test_odd_row_count_gives_every_rank_the_same_steps():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_odd_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, t0, s0), (_, t1, s1) = res
    assert t0 == t1 == 4                       # ceil(7/2) = 4 rows per rank and epoch -> 2 updates per epoch -> 4 steps over 2 epochs
    assert len(s0) == len(s1) == 8
    # within an epoch the two ranks cover every row (one row twice: wrap-around padding), and the epochs are ordered differently
    for ep in range(2):
        rows = s0[4 * ep: 4 * ep + 4] + s1[4 * ep: 4 * ep + 4]
        assert set(rows) == set(range(7)) and len(rows) == 8
    assert s0[:4] != s0[4:]


def test_lr_schedules_follow_the_hf_lambdas():
    sys.path.insert(0, ROOT)
    import math
    import iadr1_amd  # noqa: F401
    from iadr1_amd import schedule
    # transformers/optimization.py:101-104 (linear) and :134-140 (cosine): the step counter is the number of optimizer steps already taken
    assert schedule.lr_at(0, 100, 1e-5, 10, "cosine") == 0.0 and schedule.lr_at(5, 100, 1e-5, 10, "cosine") == 1e-5 * 0.5
    assert schedule.lr_at(10, 100, 1e-5, 10, "cosine") == 1e-5
    assert abs(schedule.lr_at(55, 100, 1e-5, 10, "cosine") - 1e-5 * 0.5 * (1 + math.cos(math.pi * 0.5))) < 1e-20
    assert schedule.lr_at(0, 10, 1e-6, 0, "linear") == 1e-6 and abs(schedule.lr_at(9, 10, 1e-6, 0, "linear") - 1e-7) < 1e-18
    assert schedule.lr_at(3, 10, 2.0, 4, "constant_with_warmup") == 1.5 and schedule.lr_at(7, 10, 2.0, 4, "constant") == 2.0
    with pytest.raises(ValueError):
        schedule.lr_at(0, 10, 1.0, 0, "polynomial")


def test_bench_multi_rank_launch_contract():
    """`python bench.py --gpus 2` without a torchrun environment re-executes itself as two ranks (127.0.0.1 rendezvous), every rank asserts WORLD_SIZE == --gpus,
    the per-rank values are max-reduced and ONLY rank 0 prints ONE JSON line -- the launch half of the bench contract, on CPU over gloo (`--launch-check`: no GPU
    work; the full N > 1 step runs in tests/test_hip_model.py::test_bench_two_ranks_on_one_gpu).  A WORLD_SIZE that disagrees with --gpus is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == rec["world_size"] == 2 and rec["ranks_seen"] == [0, 1] and rec["local_ranks"] == [0, 1] and rec["max_over_ranks"] == 2.0
    assert rec["master"].startswith("127.0.0.1:")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert bad.returncode != 0 and "must agree" in (bad.stderr + bad.stdout)


def _decisions_worker(rank, world, port, q):
    """The two per-rank decisions that change the STRUCTURE of a step are taken collectively: gradient recomputation
    (Engine.check_ddp_headroom: a rank whose own token-row bucket already forced it must still reach the all-reduce -- it used to return early and leave the
    others waiting in it) and co-scheduling (overlap.agree_across_ranks: one rank without a clean stream pair switches it off everywhere)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), IADR1_QUIET="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import fixture_util as fx
    import iadr1_amd  # noqa: F401
    from iadr1_amd import overlap
    from iadr1_amd.params import ParamStore, VLMConfig
    from iadr1_amd.vlm import Engine

    cfg = VLMConfig.from_dict(fx.TINY)
    eng = Engine(ParamStore(cfg, "cpu", trainable=True, with_transposes=False, with_decode_pack=False))
    if rank == 1:
        eng._recompute_forced = True           # what recompute_wanted() sets when THIS rank's bucket crossed the budget (ragged prompts: the other rank's did not)
    switched = eng.check_ddp_headroom("auto")  # every rank reaches the collective; a mismatch would pair it with the all_reduce below or hang (test timeout)
    t = torch.ones(1)
    dist.all_reduce(t)
    forced = bool(eng.__dict__.get("_recompute_forced"))
    # co-scheduling: rank 1's start-up probe found no clean stream pair -> nobody co-schedules; both fine -> both do
    agreed_mixed = overlap.agree_across_ranks(rank == 0)
    agreed_all = overlap.agree_across_ranks(True)
    q.put((rank, switched, forced, float(t.item()), agreed_mixed, agreed_all))
    dist.destroy_process_group()


def test_step_structure_decisions_are_collective():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_decisions_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, sw0, f0, t0, m0, a0), (_, sw1, f1, t1, m1, a1) = res
    assert sw0 is True and sw1 is False          # rank 0 switched because rank 1 had; rank 1 was already there
    assert f0 and f1                             # from the next step on BOTH recompute
    assert t0 == t1 == 2.0                       # the collectives stayed paired
    assert m0 is False and m1 is False and a0 is True and a1 is True
