mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_dft.py -m gpu -x -q > gpurun_out/gputests_j.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -3 gpurun_out/gputests_j.log
bash tools/_ab.sh SURFH_X=1 SURFH_X=1
exit $rc
