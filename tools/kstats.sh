#!/bin/bash
# rocprofv3 kernel-trace summary of a command: tools/kstats.sh <tag> <command...>  ->  gpurun_out/<tag>_kernel_stats.csv (+ top rows on stdout)
tag=$1; shift
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- "$@" > $out/${tag}_cmd.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
if [ -z "$f" ]; then echo "no kernel_stats.csv; log tail:"; tail -5 $out/${tag}_cmd.log; find /tmp/prof_$tag | head; exit 1; fi
cp $f $out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>7s} avg {float(r["AverageNs"])/1e3:9.2f} us  {r["Percentage"]:>6s}%')
PY
