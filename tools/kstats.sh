#!/bin/bash
# kernel statistics of a short config-3 bench run (top kernels), for A/B of one switch: bash tools/kstats.sh [VAR=VALUE]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}; out=$R/gpurun_out/kstats; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
[ -n "$1" ] && export "$1"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o k -- python3 $R/bench.py --config 3 --steps 8 --warmup 2 --cpu-seconds 0 --no-profile > $out/bench.json 2> $out/err.txt
cd $R
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-60s calls %4s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cut -c80-160 $out/bench.json
rm -rf $out/t
