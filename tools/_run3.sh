cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_model.py -q -x -m gpu -k "co_scheduling_is_on or policy_mlp_rows or chunked_reference" 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --no-real-processor-legs --no-real-shapes-leg --no-repeated-rows-leg --steps 3 --warmup 2"
for cfgs in "4:0" "4:64" "2:0" "2:64"; do
  IFS=: read pr cus <<< "$cfgs"
  IADR1_OVERLAP_CUS=$cus timeout 900 $B --prompts $pr > gpurun_out/ab3_$cfgs.log 2>&1
  echo "== prompts $pr cus $cus rc=$?"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"ms_per_decode_step": [0-9.]*' gpurun_out/ab3_$cfgs.log | head -3 | tr '\n' ' '; echo
done
timeout 1200 python bench.py > gpurun_out/bench_auto.json 2> gpurun_out/bench_auto.err; echo "bench rc=$?"; tail -c 600 gpurun_out/bench_auto.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/bench_auto.json').read().strip().splitlines()[-1])
print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['dominant'], round(r['roofline']['frac'],4), round(r['roofline_gemm']['frac'],4), r['roofline_gemm']['launches_on_cu_masked_streams'], r['co_scheduling'] and {k:r['co_scheduling'][k] for k in ('side_stream_cus','decode_stream_cus','chunk_decode_steps','policy_mlp_rows_rebuilt_on_side_stream','rebuilt_gemm_tflop_per_step')})
print(r['real_shapes'] and {k:(v.get('samples_per_s') if isinstance(v,dict) else v) for k,v in r['real_shapes'].items()})
print(r['roofline_decode']['ms_per_decode_step'], r['roofline']['share_of_step'])
PY
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
