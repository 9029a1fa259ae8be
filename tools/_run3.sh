mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_f.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -4 gpurun_out/gputests_f.log
for v in SURFH_ALPHA_RANGE=1 SURFH_ALPHA_RANGE=0 SURFH_ALPHA_RANGE=1; do
env $v python bench.py --cpu-seconds 0 --no-verify > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err; echo "bench rc $?"
python - $v <<'PY'
import json,sys
d=json.loads(open('gpurun_out/bench_f.json').read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round(d['ms_per_step'],3), {k[:22]:v for k,v in d['stage_ms_per_step'].items() if v>0.03})
PY
done
exit $rc
