cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export IADR1_FORCE_REDUCE=1
for algo in all_reduce rs_ag; do
IADR1_REDUCE_ALGO=$algo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 4 --warmup 2 --no-cpu-baseline --no-real-processor-legs --no-real-shapes-leg --no-repeated-rows-leg > gpurun_out/ddp1_$algo.log 2> gpurun_out/ddp1_$algo.err
echo "rc=$? algo=$algo"; python - <<PY
import json
try:
    r=json.loads(open('gpurun_out/ddp1_$algo.log').read().strip().splitlines()[-1])
    print(round(r['value'],2), round(r['ms_per_step'],1), r['config']['grad_exchange'], (r.get('co_scheduling') or {}).get('side_stream_cus'))
except Exception as e:
    print('ERR', e); print(open('gpurun_out/ddp1_$algo.err').read()[-1500:])
PY
done
