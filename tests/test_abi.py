"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every symbol that
include/iadr1_hip.h declares; argument validation errors come back through iadr1_last_error()."""
import os

import iadr1_amd  # noqa: F401
from iadr1_amd import build, hip


def test_library_exports_every_declared_symbol():
    path = build.build(verbose=False)
    assert os.path.exists(path)
    L = hip.lib()
    assert len(hip.PROTOS) >= 28
    for name in hip.PROTOS:
        assert hasattr(L, name), name
    assert hip.version() >= 100


def test_header_cites_reference_for_each_family():
    txt = open(hip.HEADER).read()
    for needle in ("TF:", "REF:", "sc_grpo_trainer.py", "iadr1_gemm_nt_bf16", "iadr1_attn_fwd", "iadr1_sample_topk_topp"):
        assert needle in txt


def test_argument_errors_are_reported_without_a_gpu():
    L = hip.lib()
    rc = L.iadr1_gemm_nt_bf16(None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, None)
    assert rc < 0 and b"gemm_nt" in L.iadr1_last_error()
    rc = L.iadr1_attn_fwd(None, None, None, None, None, None, None, None, 1, 1, 0, 0, 4, 2, 1, 64, 8, 8, 8, 8, 1, 1.0, None)
    assert rc < 0 and b"head dim" in L.iadr1_last_error()


def test_committed_bench_line_has_the_contract_keys():
    """The newest committed driver-format line (profiles/r05_c_bench_3b.json, produced by `python bench.py` on the GPU box) carries every key of the bench
    contract: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, the
    `roofline` object of the launch that owns the step (named in `dominant`; the decode replay and the GEMM family both stay in the line under their own names)
    and the `cpu_baseline` object (kind "port", cores, sample)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r05_c_bench_3b.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["dominant"] in ("decode", "gemm") and d["roofline_decode"]["bound"] == "hbm" and d["roofline_gemm"]["bound"] == "mfma"
    share = r["share_of_step"]
    assert (share["decode_replays"] >= share["gemm_family"]) == (r["dominant"] == "decode") and r["bound"] == ("hbm" if r["dominant"] == "decode" else "mfma")
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert abs(d["value"] - d["n_gpus"] * 64 * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    co = d["co_scheduling"]       # the default step co-schedules the reference pass with the rollout: the line says with which CU split and what it recomputes
    assert co["side_stream_cus"] + co["decode_stream_cus"] == 256 and co["chunk_decode_steps"] % 16 == 0 and co["rebuilt_gemm_tflop_per_step"] > 0
    assert d["roofline_gemm"]["launches_on_cu_masked_streams"] > 0


def test_co_scheduling_host_logic(monkeypatch):
    """iadr1_amd/overlap.py without a GPU: the automatic switch (the shapes it was measured to pay on), the IADR1_OVERLAP_CUS forms, the chunk boundaries (with
    the policy's mlp rows on the side stream the last `steps` decode steps' rows are left to the whole device; without, one 16-step block), the time-blocked row
    layout of the reference's completion rows (a bijection onto [T0, T0 + N*C) in which the rows of a chunk are one contiguous range), XCD-wise CU lists."""
    import numpy as np
    from iadr1_amd import overlap
    from iadr1_amd.params import VLMConfig
    c3, c7, c2 = VLMConfig.qwen25vl_3b(), VLMConfig.qwen25vl_7b(), VLMConfig.qwen2vl_2b()
    monkeypatch.delenv("IADR1_OVERLAP_CUS", raising=False)
    assert overlap.shadow_cus(c3, 64, 256) == overlap.AUTO_CUS == 64 and overlap.shadow_cus(c2, 64, 256) == 64
    assert overlap.shadow_cus(c7, 64, 256) == 0 and overlap.shadow_cus(c3, 32, 256) == 0 and overlap.shadow_cus(c3, 8, 512) == 0 and overlap.shadow_cus(c3, 64, 250) == 0
    assert overlap.shadow_cus() == 0                                  # (no shape given: nothing to switch on)
    assert overlap.ChunkedRefPass.applicable(c3, 64, 256) and not overlap.ChunkedRefPass.applicable(c3, 8, 256)
    monkeypatch.setenv("IADR1_OVERLAP_CUS", "0")
    assert overlap.shadow_cus(c3, 64, 256) == 0 and not overlap.ChunkedRefPass.applicable(c3, 64, 256)
    monkeypatch.setenv("IADR1_OVERLAP_CUS", "96")
    assert overlap.shadow_cus(c7, 8, 48) == 96 and overlap.ChunkedRefPass.applicable(c7, 8, 48) and not overlap.ChunkedRefPass.applicable(c7, 8, 40)
    monkeypatch.setenv("IADR1_OVERLAP_CUS", "-1")
    assert overlap.ChunkedRefPass.unmasked() and overlap.ChunkedRefPass.applicable(c3, 8, 48)
    assert overlap.ChunkedRefPass.rebuilds_policy_mlp(c3, 16) and not overlap.ChunkedRefPass.rebuilds_policy_mlp(c3, 8)
    monkeypatch.delenv("IADR1_OVERLAP_TAIL", raising=False)
    p = overlap.ChunkedRefPass.__new__(overlap.ChunkedRefPass)
    p.steps, p.policy = 32, None
    assert p.boundaries(256) == set(range(32, 256, 32)) | {240} and p.boundaries(48) == {32}
    p.policy = (None, None)
    assert p.boundaries(256) == set(range(32, 256, 32))
    p.steps = 16
    assert p.boundaries(48) == {16, 32}
    # time-blocked layout: row(s, j) = T0 + (j // B) * N * B + s * B + j % B
    B, N, C, T0 = overlap.BLOCK, 5, 48, 7
    s_, j_ = np.meshgrid(np.arange(N), np.arange(C), indexing="ij")
    rows = T0 + (j_ // B) * N * B + s_ * B + j_ % B
    assert sorted(rows.reshape(-1).tolist()) == list(range(T0, T0 + N * C))
    for c0, c1 in ((0, 16), (16, 48)):
        r = np.sort(rows[:, c0:c1].reshape(-1))
        assert r[0] == T0 + (c0 // B) * N * B and r[-1] - r[0] + 1 == r.size == N * (c1 - c0)
    assert hip.xcd_cus([6, 7], total_cus=256) == [i for i in range(256) if i % 8 >= 6] and len(hip.xcd_cus(range(6), total_cus=256)) == 192
