#!/usr/bin/env python3
"""Launch the decode gate|up stream (fused SwiGLU skinny GEMM, 3B shape, decode-packed X) on 12 rotating weight buffers (so the 256 MB
memory-side cache cannot hold them) for rocprofv3 --pmc passes; and summarise a pass:  python tools/skinny_pmc.py --parse <db> <counter>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) >= 4 and sys.argv[1] == "--parse":
    import sqlite3, json
    db = sqlite3.connect(sys.argv[2])
    rows = db.execute("select kernel_name, grid_size, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name, grid_size order by 3 desc", (sys.argv[3],)).fetchall()
    print(json.dumps([{"kernel": r[0][:80], "grid": r[1], "avg_value": r[2], "launches": r[3]} for r in rows[:8]], indent=1))
    sys.exit(0)
import torch
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
I, K, NL = 11008, 2048, 12
Wp = [ops.pack_gateup(torch.randn(2 * I, K, device=dev).to(torch.bfloat16)) for _ in range(NL)]
x = ops.pack_act(torch.randn(64, K, device=dev).to(torch.bfloat16))
a = ops.PackedAct(64, I, dev)
for rep in range(3):
    for w in Wp:
        ops.gemm_skinny(x, w, 2 * I, swiglu=True, out=a)
torch.cuda.synchronize()
