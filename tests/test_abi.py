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
