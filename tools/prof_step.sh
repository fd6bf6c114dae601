cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_h
IADR1_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o h -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-repeated-rows-leg > $R/gpurun_out/prof_h_bench.txt 2>&1
DB=$(find /tmp/prof_h -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/prof_h_stats.txt 2>&1
head -40 $R/gpurun_out/prof_h_stats.txt
