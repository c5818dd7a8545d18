mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. engine profiles: kernel stats + HBM traffic (separate pmc passes) of the default bench command
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/r2_bench_under_rocprof.json 2> gpurun_out/r2_prof_bench.err
f=$(find gpurun_out/prof_bench -name '*kernel_stats.csv' | head -1); head -6 $f | cut -c1-160
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch2 -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2> gpurun_out/r2_pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write2 -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2> gpurun_out/r2_pmc_write.err
python tools/traffic_from_pmc.py gpurun_out/pmc_fetch2 gpurun_out/pmc_write2 > gpurun_out/r2_traffic.json; cat gpurun_out/r2_traffic.json | head -30
# 2. PPO MFMA counters: warm MIOpen's find cache for these shapes first, then the counter pass
timeout -k 10 400 python3 tools/ppo_bench.py --envs 512 --T 16 --minibatch 8192 --updates 1 --nhwc > gpurun_out/r2_ppo_small.json 2> gpurun_out/r2_ppo_small.err; tail -1 gpurun_out/r2_ppo_small.err
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_ppo_fused2 -- python3 tools/ppo_bench.py --envs 512 --T 16 --minibatch 8192 --updates 1 --nhwc > gpurun_out/r2_ppo_fused_pmc2.json 2> gpurun_out/r2_ppo_fused_pmc2.err
python tools/mfma_util_from_pmc.py gpurun_out/pmc_ppo_fused2 30 > gpurun_out/r2_ppo_fused_mfma_util.csv; head -16 gpurun_out/r2_ppo_fused_mfma_util.csv | cut -c1-150
find gpurun_out -name '*counter_collection.csv' -size +20M -delete; find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.err; cat gpurun_out/r2_bench3.json
