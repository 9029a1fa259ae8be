mkdir -p gpurun_out
export SURFH_REHEARSAL=1 MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=2
for n in 2 4; do
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --steps 10 --warmup 2 --cpu-seconds 0 --no-verify > gpurun_out/bench_mp$n.json 2> gpurun_out/bench_mp$n.err; echo "N=$n rc $?"
  python - $n <<'PY'
import json,sys
try:
    d=json.loads(open(f'gpurun_out/bench_mp{sys.argv[1]}.json').read().strip().splitlines()[-1])
    print(sys.argv[1], d['value'], d['ms_per_step'], d['config']['parallelism'][:160], d['config']['grad_norm_first_last'])
except Exception as e:
    print('no json', e)
PY
  grep -i "error\|traceback" gpurun_out/bench_mp$n.err | head -5
done
