cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-real-processor-legs --no-real-shapes-leg --no-repeated-rows-leg --steps 3 --warmup 2"
for cfgs in "7b:0" "7b:64" "7b:96" "qwen2vl_2b:0" "qwen2vl_2b:64" "7b:0"; do
  IFS=: read m cus <<< "$cfgs"
  IADR1_OVERLAP_CUS=$cus IADR1_OVERLAP_STATS=1 timeout 900 $B --model $m > gpurun_out/ab2_$cfgs.log 2>&1
  echo "== $cfgs rc=$?"; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"ms_per_decode_step": [0-9.]*' gpurun_out/ab2_$cfgs.log | head -3 | tr '\n' ' '; echo
  grep "side-stream phases" gpurun_out/ab2_$cfgs.log | tail -1 | grep -o "('rows\[[0-9]*,256)'[^)]*)\|('policy mlp\[[0-9]*,256)'[^)]*)\|('rows\[192[^)]*)\|('prompt'[^)]*)"
done
export IADR1_OVERLAP_CUS=64
timeout 2400 python -m pytest tests -q -x -m gpu --deselect tests/test_hip_model.py::test_full_size_3b_parity_at_the_headline_shape_forward 2>&1 | tail -15
