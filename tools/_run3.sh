mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dft.py -m gpu -x -q -s -k "fused_adjoint" > gpurun_out/t_fused.log 2>&1; rc=$?; echo "pytest rc $rc"
grep "fused adjoint\|passed\|failed\|Error\|fault" gpurun_out/t_fused.log | head -20
exit $rc
