#!/bin/bash
# Regenerates the measurement set kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/final_profiles.sh r03 "3 4 5"
# per config: 1. rocprofv3 kernel statistics of a short bench run;  2. (configs 3 and 4) PMC passes FETCH_SIZE / WRITE_SIZE
# (separate runs, kernel trace only) -> HBM bytes per launch (tools/pmc_traffic.py);  3. the bench line (with the CPU baseline leg)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r03}
cfgs=${2:-"3"}
out=$R/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
for c in $cfgs; do
  cd /tmp
  steps=15; [ "$c" != "3" ] && [ "$c" != "2" ] && steps=6
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats$c" -o s -- python3 "$R/bench.py" --config $c --steps $steps --warmup 3 --cpu-seconds 0 --no-verify \
      > "$out/stats_bench_config$c.json" 2> "$out/stats_bench_config$c.err"
  echo "config $c: stats pass done"
  find "$out/stats$c" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_config$c.csv" \;
  rm -rf "$out/stats$c"
  if [ "$c" = "3" ] || [ "$c" = "4" ]; then
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch$c" -o f -- python3 "$R/bench.py" --config $c --steps 3 --warmup 1 --cpu-seconds 0 --no-profile --no-verify \
        > /dev/null 2> "$out/fetch$c.err"
    echo "config $c: fetch pass done"
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write$c" -o w -- python3 "$R/bench.py" --config $c --steps 3 --warmup 1 --cpu-seconds 0 --no-profile --no-verify \
        > /dev/null 2> "$out/write$c.err"
    echo "config $c: write pass done"
    cd "$R"
    python3 tools/pmc_traffic.py "$out/fetch$c" "$out/write$c" > "$out/pmc_traffic_config$c.json"
    cp "$out/pmc_traffic_config$c.json" "profiles/${tag}_pmc_traffic_config$c.json"       # bench.py reads `traffic` from here
    rm -rf "$out/fetch$c" "$out/write$c"                                                  # raw traces stay on the box
  fi
  cd "$R"
  if [ "$c" = "3" ]; then python3 bench.py --config 3 > "$out/bench_config3.json" 2> "$out/bench_config3.err"
  elif [ "$c" = "2" ]; then python3 bench.py --config 2 --steps 1200 --cpu-seconds 15 > "$out/bench_config2.json" 2> "$out/bench_config2.err"
  elif [ "$c" = "4" ]; then python3 bench.py --config 4 --steps 60 --warmup 5 --cpu-seconds 25 --no-verify > "$out/bench_config4.json" 2> "$out/bench_config4.err"
  else python3 bench.py --config 5 --steps 40 --warmup 4 --cpu-seconds 20 > "$out/bench_config5.json" 2> "$out/bench_config5.err"; fi
  echo "config $c: bench done"
done
ls -la "$out"
