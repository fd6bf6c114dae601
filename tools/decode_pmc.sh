#!/bin/bash
# HBM-side traffic of every kernel of the decode step ("record the decode kernels' FETCH_SIZE too"): rocprofv3 --kernel-trace --pmc, one
# counter per pass (FETCH_SIZE, WRITE_SIZE), on tools/decode_step_time.py (64 sequences, 3B shapes, hipGraph replay).  Output: gpurun_out/<TAG>_decode_pmc.json (TAG = $1, default r03)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
# DECODE_ARGS: extra arguments of tools/decode_step_time.py (round 6: "--trace 1"); the decode-step switches come from the environment (round 6: IADR1_OVERLAP_CUS=0
# IADR1_DECODE_CUS=192 IADR1_DECODE_KS=1,8 = the launch set of the co-scheduled step -- grids sized for 192 CUs -- on an ordinary stream, rocprofv3 crashes with masked ones)
DECODE_ARGS=${DECODE_ARGS:-}
mkdir -p $R/gpurun_out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_dec_$ctr
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_dec_$ctr -o d -- python $R/tools/decode_step_time.py --reps 0 --steps 24 $DECODE_ARGS > /dev/null 2>&1
done
python3 - <<PY
import sqlite3, glob, json
out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/decode_step_time.py --reps 0 --steps 24: the decode step of the rollout, 64 sequences, "
               "Qwen2.5-VL-3B shapes, context 512 + t.  Per kernel and grid: average over its dispatches.  Raw unit KiB; gfx950 correction of MI355X_MICROARCH.md applied to reads "
               "(FETCH_SIZE counts half of the bytes of wide coalesced streaming reads -> x2): read_bytes = 2 * FETCH_SIZE * 1024.  Profiled passes serialise the graph's kernels.",
       "decode_cus": int("${IADR1_DECODE_CUS:-0}") or None, "decode_args": "${DECODE_ARGS}", "env": {k: v for k, v in __import__("os").environ.items() if k.startswith("IADR1_")},
       "kernels": []}
agg = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pmc_dec_{ctr}/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        for name, grid, val, n in db.execute("select kernel_name, grid_size, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name, grid_size", (ctr,)):
            if any(k in name for k in ("skinny", "attn_decode", "rmsnorm_fwd_row", "sample_", "decode_advance", "rope_table", "embed")):
                agg.setdefault((name[:90], grid), {})[ctr] = (val, n)
for (name, grid), c in sorted(agg.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[0]):
    f, w = c.get("FETCH_SIZE", (0, 0)), c.get("WRITE_SIZE", (0, 0))
    out["kernels"].append({"kernel": name, "grid": grid, "launches": f[1], "FETCH_SIZE_KiB_raw": f[0], "WRITE_SIZE_KiB": w[0], "read_bytes_corrected": 2 * f[0] * 1024, "write_bytes": w[0] * 1024})
json.dump(out, open("$R/gpurun_out/${TAG}_decode_pmc.json", "w"), indent=1)
for k in out["kernels"][:12]:
    print(f'{k["kernel"][:70]:70s} grid {k["grid"]:8d} n {k["launches"]:5d} read {k["read_bytes_corrected"]/1e6:8.2f} MB  write {k["write_bytes"]/1e6:7.2f} MB')
PY
