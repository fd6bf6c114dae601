cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-real-processor-legs --no-real-shapes-leg --no-repeated-rows-leg --steps 5 --warmup 2"
for cfgs in "auto:32" "64:48" "64:64" "96:32" "32:32" "auto:32"; do
  IFS=: read cus st <<< "$cfgs"
  IADR1_OVERLAP_CUS=$cus IADR1_OVERLAP_STEPS=$st timeout 900 $B > gpurun_out/sw_$cfgs.log 2>/dev/null
  echo "cus $cus steps $st: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"ms_per_decode_step": [0-9.]*' gpurun_out/sw_$cfgs.log | head -3 | tr '\n' ' ')"
done
