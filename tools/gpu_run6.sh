mkdir -p gpurun_out
python -m pytest tests/test_minigrid_view_gpu.py tests/test_ppo_gpu.py tests/test_stack_gpu.py -m gpu -x -q -k "minigrid or view or occlu or mgstep or channels_last or epilogue or trainer or full_size or gen_obs" > gpurun_out/r2_tests6.log 2>&1 || { tail -40 gpurun_out/r2_tests6.log; exit 1; }
tail -3 gpurun_out/r2_tests6.log
python tools/view_bench.py > gpurun_out/r2_view_bench.json 2> gpurun_out/r2_view_bench.err; cat gpurun_out/r2_view_bench.json
timeout -k 10 500 python tools/ppo_bench.py --updates 1 --nhwc > gpurun_out/r2_ppo_nhwc_fused.json 2> gpurun_out/r2_ppo_nhwc_fused.err; tail -2 gpurun_out/r2_ppo_nhwc_fused.err; cat gpurun_out/r2_ppo_nhwc_fused.json
timeout -k 10 500 python tools/ppo_bench.py --updates 1 --nhwc --unfused --no_value_reuse > gpurun_out/r2_ppo_nhwc_unfused.json 2> gpurun_out/r2_ppo_nhwc_unfused.err; tail -1 gpurun_out/r2_ppo_nhwc_unfused.err; cat gpurun_out/r2_ppo_nhwc_unfused.json
