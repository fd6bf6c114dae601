cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_g
rocprofv3 --kernel-trace -d /tmp/prof_g -o g -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-repeated-rows-leg > /dev/null 2>&1
python $R/tools/gpu_gaps.py $(find /tmp/prof_g -name "*.db" | head -1) 30 2.7
