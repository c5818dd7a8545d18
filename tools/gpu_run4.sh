mkdir -p gpurun_out
timeout -k 10 500 python tools/ppo_bench.py --updates 1 --nhwc > gpurun_out/r2_ppo_nhwc.json 2> gpurun_out/r2_ppo_nhwc.err; tail -3 gpurun_out/r2_ppo_nhwc.err; cat gpurun_out/r2_ppo_nhwc.json
timeout -k 10 400 python tools/ppo_bench.py --updates 1 > gpurun_out/r2_ppo_nchw.json 2> gpurun_out/r2_ppo_nchw.err; tail -3 gpurun_out/r2_ppo_nchw.err; cat gpurun_out/r2_ppo_nchw.json
