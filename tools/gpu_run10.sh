mkdir -p gpurun_out
python -m pytest tests/test_ppo_gpu.py tests/test_soa_gpu.py tests/test_predictor_gpu.py -m gpu -x -q -k "fused or channels_last or epilogue or soa_update or heads" > gpurun_out/r2_tests10.log 2>&1 || { tail -40 gpurun_out/r2_tests10.log; exit 1; }
tail -3 gpurun_out/r2_tests10.log
timeout -k 10 500 python tools/ppo_bench.py --updates 1 --nhwc > gpurun_out/r2_ppo_conv1.json 2> gpurun_out/r2_ppo_conv1.err; tail -2 gpurun_out/r2_ppo_conv1.err; cat gpurun_out/r2_ppo_conv1.json
