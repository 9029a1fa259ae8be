mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_a.log 2>&1; echo "pytest rc $?"
tail -4 gpurun_out/gputests_a.log
python bench.py --cpu-seconds 0 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['stage_ms_per_step'])
PY
