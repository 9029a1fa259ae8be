mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_distributed.py -m gpu -x -q -s > gpurun_out/gputests_m.log 2>&1; rc=$?; echo "pytest rc $rc"
grep "sharded vs\|passed\|failed\|Error\|error" gpurun_out/gputests_m.log | tail -12
exit $rc
