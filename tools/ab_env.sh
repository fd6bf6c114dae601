# usage: bash tools/ab_env.sh VAR v1 v2 ... [-- bench args]: bench.py once per value of an environment switch, value / samples per s / ms per step
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for v in "${VALS[@]}"; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-repeated-rows-leg "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$VAR=$v', round(d['value'],3), round(d['ms_per_step'],2))"
done
