cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 1 0; do
rm -rf /tmp/prof_t
IADR1_REUSE_DECODE=$v rocprofv3 --kernel-trace --stats -d /tmp/prof_t -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-repeated-rows-leg > /dev/null 2>&1
echo "REUSE_DECODE=$v"
python $R/tools/rocpd_summary.py $(find /tmp/prof_t -name "*.db" | head -1) | grep -E "skinny|attn_decode|rmsnorm_fwd_row" | cut -c1-60,96-160
done
