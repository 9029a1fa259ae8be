python bench.py --cpu-seconds 0 --no-verify --steps 20 > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err; grep "plan built" gpurun_out/bench_f.err
