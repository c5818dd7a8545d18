mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r2_suite.log 2>&1; rc=$?
tail -25 gpurun_out/r2_suite.log; exit $rc
