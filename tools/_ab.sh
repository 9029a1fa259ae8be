mkdir -p gpurun_out
for v in "$@"; do
  for rep in 1 2; do
    env $v python bench.py --cpu-seconds 0 --no-verify --steps 100 > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - "$v" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1])
st=d['stage_ms_per_step']
print(sys.argv[1], round(d['value'],1), 'it/s', round(d['ms_per_step'],3), {k.split('_kernel')[0][:18]:v for k,v in st.items() if v>0.05})
PY
  done
done
