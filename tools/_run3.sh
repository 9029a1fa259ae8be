mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_c.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -4 gpurun_out/gputests_c.log
exit $rc
