# rocprofv3 kernel-trace summaries of the default bench step (two streams) and of the single-stream variant, plus the bench JSON lines.
# usage (GPU box): bash tools/prof_step.sh <tag>   -> gpurun_out/<tag>_*.{json,txt}
TAG=${1:-prof}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/${TAG}_bench_3b.json 2> $R/gpurun_out/${TAG}_bench_3b.err
rm -rf /tmp/prof_a /tmp/prof_b
rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o a -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-repeated-rows-leg --no-real-processor-legs --no-real-shapes-leg > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_a -name "*.db" | head -1) > $R/gpurun_out/${TAG}_bench_3b_kernel_stats.txt 2>&1
IADR1_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-repeated-rows-leg --no-real-processor-legs --no-real-shapes-leg > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_b -name "*.db" | head -1) > $R/gpurun_out/${TAG}_bench_3b_kernel_stats_single_stream.txt 2>&1
python $R/bench.py --workload pa_sft --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_pa_sft_3b.json 2>/dev/null
python $R/bench.py --model 7b --no-cpu-baseline --no-repeated-rows-leg > $R/gpurun_out/${TAG}_bench_7b.json 2>/dev/null
python $R/bench.py --model qwen2vl_2b --workload pa_sft --sft-batch 4 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_pa_sft_qwen2vl_2b.json 2>/dev/null
tail -c 600 $R/gpurun_out/${TAG}_bench_3b.json; echo; head -c 300 $R/gpurun_out/${TAG}_bench_pa_sft_3b.json; echo; head -c 300 $R/gpurun_out/${TAG}_bench_7b.json; echo; head -c 300 $R/gpurun_out/${TAG}_bench_pa_sft_qwen2vl_2b.json
