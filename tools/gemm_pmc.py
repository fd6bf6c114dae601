#!/usr/bin/env python3
"""Launch the dominant GEMM shapes of the SC-GRPO step a few times (for rocprofv3 --pmc passes)."""
import os, sys
if len(sys.argv) >= 4 and sys.argv[1] == "--parse":     # python tools/gemm_pmc.py --parse <rocpd db> <counter>: avg counter value per gemm_nt grid
    import sqlite3, json
    db = sqlite3.connect(sys.argv[2])
    rows = db.execute("select kernel_name, grid_size, avg(value), count(*) from counters_collection where counter_name = ? and kernel_name like '%gemm_nt%' group by kernel_name, grid_size order by grid_size desc", (sys.argv[3],)).fetchall()
    print(json.dumps([{"kernel": r[0][:70], "grid": r[1], "avg_value": r[2], "launches": r[3]} for r in rows], indent=1))
    sys.exit(0)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
for M, N, K in [(20480, 22016, 2048), (20480, 2048, 11008), (22016, 2048, 20480)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB: evicts the 256 MiB Infinity Cache between launches
    for _ in range(3):
        flush.zero_()
        ops.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    del a, b, out, flush
# the gate|up projection as the reference pass runs it: SwiGLU in the epilogue, only the activation [M, I] is written (gemm_nt_256<3>)
M, I, K = 20480, 11008, 2048
a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = torch.randn(2 * I, K, device=dev).to(torch.bfloat16)
act = torch.empty(M, I, dtype=torch.bfloat16, device=dev)
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for _ in range(3):
    flush.zero_()
    ops.gemm_swiglu_fused(a, w, None, act)
torch.cuda.synchronize()
# round 5's two new forms: the row-blocked fused gate|up of the co-scheduled pass (gemm_nt_256<6>: one chunk = 64 sequences x 32 steps, rows 256 apart in a sequence-major
# arena, gate|up AND activation written) and the TN weight-gradient kernel straight from row-major dY / X (gemm_nt_256<2, true>: dW_gu [22016 x 2048] over 20480 token rows)
Mr = 64 * 32
xa = torch.randn(64 * 256, K, device=dev).to(torch.bfloat16)
gu_r = torch.empty(64 * 256, 2 * I, dtype=torch.bfloat16, device=dev); a_r = torch.empty(64 * 256, I, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    flush.zero_()
    ops.gemm_swiglu_rows(xa, w, gu_r, a_r, 64, 32, 256)
torch.cuda.synchronize()
del xa, gu_r, a_r, act, a
dy = torch.randn(20480, 2 * I, device=dev).to(torch.bfloat16); x = torch.randn(20480, K, device=dev).to(torch.bfloat16)
dw = torch.zeros(2 * I, K, dtype=torch.float32, device=dev)
for _ in range(3):
    flush.zero_()
    ops.hip.call("gemm_tn_acc_bf16", dy, x, dw, None, 2 * I, K, 20480, 2 * I, K, K, 1)
torch.cuda.synchronize()
