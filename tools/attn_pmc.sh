# PMC passes over the attention kernels on the 3B shared-prefix training shape (tools/attn_probe.py).  Counters only (--pmc with --kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_out -o p -- python $R/tools/attn_probe.py > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
for f in glob.glob('/tmp/pmc_out/**/*.db', recursive=True):
    db = sqlite3.connect(f)
    try:
        rows = db.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%attn%' group by kernel_name, counter_name order by kernel_name").fetchall()
    except Exception as e:
        print('ERR', e); continue
    for r in rows: print(r[0][:60].ljust(60), r[1].ljust(28), '%.4g' % r[2], r[3])
PY
done
