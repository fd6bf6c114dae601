cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SET="SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"
rm -rf /tmp/pmc_tn
IADR1_GEMM_TN=2 timeout 600 rocprofv3 --kernel-trace --pmc $SET -d /tmp/pmc_tn -o p -- python $R/tools/gemm_tn_probe.py 20480 > /dev/null 2>&1
python - <<'PY'
import sqlite3, glob
for f in glob.glob("/tmp/pmc_tn/**/*.db", recursive=True):
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, grid_size, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%gemm_nt_256%' group by kernel_name, grid_size, counter_name").fetchall()
    agg = {}
    for name, grid, ctr, val, n in rows:
        agg.setdefault((name[:70], grid), {"n": n})[ctr] = val
    for (name, grid), c in sorted(agg.items()):
        gui = c.get("GRBM_GUI_ACTIVE") or 1
        print(name, grid, "launches", c["n"], "mfma_busy %.3f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 256 * 4)),
              "lds_conflict %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1)),
              "lds_idx_active/gui %.3f" % (c.get("SQ_LDS_IDX_ACTIVE", 0) / (gui / 8 * 256)), "insts_lds %.3g" % c.get("SQ_INSTS_LDS", 0), "wait_lds/gui %.3f" % (c.get("SQ_WAIT_INST_LDS", 0) / (gui / 8 * 256 * 4)), "gui %.3g" % gui)
PY
