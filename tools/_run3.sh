mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gputests_l.log 2>&1; rc=$?; echo "pytest rc $rc"
tail -3 gpurun_out/gputests_l.log
bash tools/_ab.sh SURFH_SPECTRAL_CG=0 SURFH_SPECTRAL_CG=1 SURFH_SPECTRAL_CG=0 SURFH_SPECTRAL_CG=1
exit $rc
