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
    """The newest committed driver-format line (profiles/r05_a_bench_3b.json, produced by `python bench.py` on the GPU box) carries every key of the bench
    contract: metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, the
    `roofline` object of the launch that owns the step (named in `dominant`; the decode replay and the GEMM family both stay in the line under their own names)
    and the `cpu_baseline` object (kind "port", cores, sample)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r05_a_bench_3b.json")))
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
