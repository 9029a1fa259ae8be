#!/bin/bash
# Regenerates the measurement set kept under profiles/ (run on the GPU box from the repo root):
#   bash tools/final_profiles.sh r01_final
# 1. rocprofv3 kernel statistics of the default bench command (config 3)
# 2. PMC passes FETCH_SIZE / WRITE_SIZE (separate runs, kernel trace only) -> HBM bytes per launch (tools/pmc_traffic.py)
# 3. the bench lines of config 3 (with the CPU baseline leg) and config 2
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r01_final}
out=$R/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- python3 "$R/bench.py" --config 3 --steps 15 --warmup 3 --cpu-seconds 0 \
    > "$out/stats_bench_config3.json" 2> "$out/stats_bench_config3.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o f -- python3 "$R/bench.py" --config 3 --steps 3 --warmup 1 --cpu-seconds 0 --no-profile \
    > /dev/null 2> "$out/fetch.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o w -- python3 "$R/bench.py" --config 3 --steps 3 --warmup 1 --cpu-seconds 0 --no-profile \
    > /dev/null 2> "$out/write.err"
echo "write pass done"
cd "$R"
python3 tools/pmc_traffic.py "$out/fetch" "$out/write" > "$out/pmc_traffic_config3.json"
cp "$out/pmc_traffic_config3.json" "profiles/${tag}_pmc_traffic_config3.json"       # bench.py reads `traffic` from here
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats_config3.csv" \;
python3 bench.py --config 3 > "$out/bench_config3.json" 2> "$out/bench_config3.err"
echo "bench config 3 done"
python3 bench.py --config 2 --steps 1200 --cpu-seconds 15 > "$out/bench_config2.json" 2> "$out/bench_config2.err"
echo "bench config 2 done"
rm -rf "$out/stats" "$out/fetch" "$out/write"                                      # raw traces stay on the box
ls -la "$out"
