set -e
mkdir -p gpurun_out
python -m pytest tests/test_predictor_gpu.py tests/test_soa_gpu.py tests/test_stack_gpu.py -m gpu -x -q > gpurun_out/r2_tests3.log 2>&1 || { tail -40 gpurun_out/r2_tests3.log; exit 1; }
tail -3 gpurun_out/r2_tests3.log
