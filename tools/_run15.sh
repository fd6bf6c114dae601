cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
timeout 1200 python bench.py > gpurun_out/r05_d_bench_3b.json 2> gpurun_out/r05_d_bench_3b.err; echo "bench rc=$?"
B="--no-cpu-baseline --no-real-processor-legs --no-real-shapes-leg --no-repeated-rows-leg --steps 4 --warmup 2"
timeout 900 python bench.py $B --model 7b > gpurun_out/r05_d_bench_7b.json 2>/dev/null; echo "7b rc=$?"
timeout 900 python bench.py $B --model llava_ov_7b > gpurun_out/r05_d_bench_llava_ov_7b.json 2>/dev/null; echo "llava rc=$?"
timeout 900 python bench.py $B --model qwen2vl_2b > gpurun_out/r05_d_bench_2b.json 2>/dev/null; echo "2b rc=$?"
timeout 900 python bench.py --workload pa_sft --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r05_d_bench_pa_sft_3b.json 2>/dev/null; echo "sft rc=$?"
timeout 900 python bench.py --workload pa_sft --model qwen2vl_2b --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r05_d_bench_pa_sft_qwen2vl_2b.json 2>/dev/null; echo "sft2b rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05_d_bench_*.json')):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(r['value'],2), round(r['ms_per_step'],1), r.get('roofline',{}).get('frac'), (r.get('co_scheduling') or {}).get('side_stream_cus'), r.get('hbm',{}).get('device_allocs_in_timed_region'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
