# MFMA-pipe busy fraction and LDS bank-conflict fraction of the training GEMMs and attention kernels (north star: "choices evidenced by
# rocprof HBM GB/s and MFMA-busy counters").  Counters only: rocprofv3 --pmc with --kernel-trace, one pass per probe.
# usage (GPU box): bash tools/mfma_pmc.sh [TAG]  ->  gpurun_out/<TAG>_mfma_busy.json   (copy to profiles/; TAG defaults to r03)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
SET="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
for probe in gemm_pmc attn_probe; do
  rm -rf /tmp/pmc_$probe
  rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_$probe -o p -- python $R/tools/$probe.py > /dev/null 2>&1
done
python - <<PY
import sqlite3, glob, json, os
out = {"note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE "
               "(one pass per probe: tools/gemm_pmc.py = the dominant GEMM shapes of the 3B SC-GRPO step, tools/attn_probe.py = attention on the shared-prefix training shape; "
               "tools/mfma_pmc.sh).  Per kernel and grid: averages over its dispatches.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs) "
               "(the MfmaUtil formula of the gfx94x derived-counter file, which ROCm 7.2 falls back to on gfx950); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE. "
               "Profiled passes run at a lower clock than un-profiled ones (MI355X_MICROARCH.md, DVFS): fractions, not rates, are the result.", "kernels": []}
for probe in ("gemm_pmc", "attn_probe"):
    for f in glob.glob(f"/tmp/pmc_{probe}/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        try:
            rows = db.execute("select kernel_name, grid_size, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%gemm_nt%' or kernel_name like '%attn_%' "
                              "group by kernel_name, grid_size, counter_name").fetchall()
        except Exception as e:
            out["kernels"].append({"error": repr(e), "probe": probe}); continue
        agg = {}
        for name, grid, ctr, val, n in rows:
            agg.setdefault((name, grid), {"launches": n})[ctr] = val
        for (name, grid), c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
            gui = c.get("GRBM_GUI_ACTIVE") or 0
            rec = {"probe": probe, "kernel": name[:90], "grid": grid, **{k: v for k, v in c.items()}}
            if gui:
                rec["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 256 * 4)      # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            if c.get("SQ_LDS_IDX_ACTIVE"):
                rec["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
            out["kernels"].append(rec)
json.dump(out, open(os.path.join("$R", "gpurun_out", "${TAG}_mfma_busy.json"), "w"), indent=1)
for k in out["kernels"]:
    print(k.get("kernel", k)[:70], k.get("grid"), "mfma_busy", round(k.get("mfma_busy_frac", -1), 3), "lds_conflict", round(k.get("lds_conflict_frac", -1), 4))
PY
