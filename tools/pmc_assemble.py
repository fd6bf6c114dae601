#!/usr/bin/env python3
"""gpurun_out/pmc_{gemm,skinny}_{FETCH_SIZE,WRITE_SIZE}.json (tools/pmc_collect.sh) -> profiles/<TAG>_gemm_pmc.json, profiles/<TAG>_skinny_pmc.json (TAG = argv[1], default r03)."""
import json, os, sys
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"          # round tag of the output files
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ld = lambda n: json.load(open(os.path.join(ROOT, "gpurun_out", n)))
KiB = 1024
import re
def pick(rows, frag, grid=None):
    """frag "gemm_nt_256<0>" = template arguments (0) or (0, false, false); "gemm_nt_256<2,t>" = (2, true[, false]): the TN form"""
    m = re.fullmatch(r"(\w+)<(\d+)(,t)?>", frag)
    def ok(name):
        if not m:
            return frag in name
        k = re.search(re.escape(m.group(1)) + r"<\(?(?:int\)?)?(\d+)(?:, \(?(?:bool\)?)?(true|false|1|0))?(?:, \(?(?:bool\)?)?(true|false|1|0))?>", name)
        return bool(k) and k.group(1) == m.group(2) and ((k.group(2) in ("true", "1")) == bool(m.group(3))) and k.group(3) not in ("true", "1")
    r = [x for x in rows if ok(x["kernel"]) and (grid is None or x["grid"] == grid)]
    assert len(r) == 1, (frag, grid, [x["kernel"] for x in rows])
    return r[0]["avg_value"]
gf, gw = ld("pmc_gemm_FETCH_SIZE.json"), ld("pmc_gemm_WRITE_SIZE.json")
kern = []
for name, frag, (M, N, K), alg in (
        ("gemm_nt_256<bf16 out>", "gemm_nt_256<0>", (20480, 22016, 2048), None),
        ("gemm_nt_256<SwiGLU epilogue, only the activation written> (iadr1_gemm_swiglu_bf16, gate|up of the reference pass)", "gemm_nt_256<3>", (20480, 22016, 2048), (20480 * 2048 + 22016 * 2048 + 20480 * 11008) * 2),
        ("gemm_nt_256<bf16 out>", "gemm_nt_256<0>", (20480, 2048, 11008), None),
        ("gemm_nt_256<bf16 out>", "gemm_nt_256<0>", (22016, 2048, 20480), None),
        ("gemm_nt_256<row-blocked SwiGLU epilogue: gate|up AND activation written> (iadr1_gemm_swiglu_rows_bf16: the policy's mlp rows of one chunk of the co-scheduled pass, 64 sequences x 32 steps)",
         "gemm_nt_256<6>", (2048, 22016, 2048), (2048 * 2048 + 22016 * 2048 + 2048 * 22016 + 2048 * 11008) * 2),
        ("gemm_nt_256<fp32 accumulate, TN> (iadr1_gemm_tn_acc_bf16: dW_gu += dY^T . X from row-major operands, no transposed copies)", "gemm_nt_256<2,t>", (22016, 2048, 20480),
         (20480 * 22016 + 20480 * 2048) * 2 + 2 * 22016 * 2048 * 4)):
    grid = ((M + 255) // 256) * ((N + 255) // 256) * 512
    fr, wr = pick(gf, frag, grid), pick(gw, frag, grid)
    alg = alg if alg is not None else (M * K + N * K + M * N) * 2
    rd = fr * KiB * 2                                   # gfx950: FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md)
    kern.append({"kernel": name, "MNK": [M, N, K], "FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB": wr, "read_bytes_corrected": rd, "write_bytes": wr * KiB,
                 "traffic_bytes": rd + wr * KiB, "algorithmic_bytes": alg, "traffic_over_algorithmic": (rd + wr * KiB) / alg})
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; tools/pmc_collect.sh -> tools/pmc_assemble.py) on tools/gemm_pmc.py; a 1 GiB memset precedes each launch. Units of the raw counters: KiB. gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads -> doubled. FETCH_SIZE counts L2 misses sent to the fabric, Infinity-Cache hits included, so it is an upper bound of HBM reads. Shapes: the shared-prefix micro-batch (20480 token rows) of the 3B SC-GRPO step. Build: " + TAG + ".",
           "kernels": kern}, open(os.path.join(ROOT, "profiles", TAG + "_gemm_pmc.json"), "w"), indent=1)
sf, sw = ld("pmc_skinny_FETCH_SIZE.json"), ld("pmc_skinny_WRITE_SIZE.json")
fr, wr = pick(sf, "gemm_skinny_pers_kernel<8, 8"), pick(sw, "gemm_skinny_pers_kernel<8, 8")
alg = 22016 * 2048 * 2 + 64 * 11008 * 2
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only; tools/pmc_collect.sh -> tools/pmc_assemble.py) on tools/skinny_pmc.py: the decode gate|up stream (persistent fused-SwiGLU skinny GEMM, M=64, N=22016, K=2048, decode-packed X) on 12 rotating weight buffers, 36 launches. Raw counter unit KiB. gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced streaming reads -> doubled. The X re-reads (256 KB per block) are L2 hits and do not reach the fabric. Build: " + TAG + ".",
           "kernel": "gemm_skinny_pers_kernel<8, 8> (out_mode 3)", "MNK": [64, 22016, 2048], "FETCH_SIZE_KiB_raw": fr, "WRITE_SIZE_KiB": wr, "read_bytes_corrected": fr * KiB * 2,
           "write_bytes": wr * KiB, "traffic_bytes": fr * KiB * 2 + wr * KiB, "algorithmic_bytes": alg, "algorithmic_bytes_detail": "weights 22016*2048*2 = 90177536 + output 64*11008*2 = 1409024",
           "traffic_over_algorithmic": (fr * KiB * 2 + wr * KiB) / alg}, open(os.path.join(ROOT, "profiles", TAG + "_skinny_pmc.json"), "w"), indent=1)
for k in kern: print(k["kernel"][:40], k["MNK"], round(k["traffic_bytes"] / 1e9, 3), "GB", round(k["traffic_over_algorithmic"], 2))
print("skinny", round((fr * KiB * 2 + wr * KiB) / 1e6, 2), "MB", round((fr * KiB * 2 + wr * KiB) / alg, 3))
