mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/gputests_i.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -3 gpurun_out/gputests_i.log
cd tools/exp/_bin && ./cc_bench | grep "forward\|adjoint\|ragged\|short"
exit $rc
