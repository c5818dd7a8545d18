mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_bench_gpu.py -m gpu -x -q > gpurun_out/r2_tests14.log 2>&1 || { tail -30 gpurun_out/r2_tests14.log; exit 1; }
tail -2 gpurun_out/r2_tests14.log
# warm MIOpen's search cache for the configs[2] shapes, then trace the steady state
timeout -k 10 500 python3 tools/ppo_bench.py --updates 1 --nhwc > gpurun_out/r2_ppo_final_warm.json 2> gpurun_out/r2_ppo_final_warm.err; tail -1 gpurun_out/r2_ppo_final_warm.err
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ppo_final -- python3 tools/ppo_bench.py --updates 1 --nhwc > gpurun_out/r2_ppo_final_prof.json 2> gpurun_out/r2_ppo_final_prof.err
tail -1 gpurun_out/r2_ppo_final_prof.err; cat gpurun_out/r2_ppo_final_prof.json
find gpurun_out/prof_ppo_final -name '*kernel_trace.csv' -delete
