mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ppo_fused -- python3 tools/ppo_bench.py --updates 1 --nhwc > gpurun_out/r2_ppo_fused_prof.json 2> gpurun_out/r2_ppo_fused_prof.err
tail -1 gpurun_out/r2_ppo_fused_prof.err; cat gpurun_out/r2_ppo_fused_prof.json
find gpurun_out/prof_ppo_fused -name '*kernel_trace.csv' -delete
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_ppo_fused -- python3 tools/ppo_bench.py --envs 512 --T 16 --minibatch 8192 --updates 1 --nhwc > gpurun_out/r2_ppo_fused_pmc.json 2> gpurun_out/r2_ppo_fused_pmc.err
tail -1 gpurun_out/r2_ppo_fused_pmc.err
python tools/mfma_util_from_pmc.py gpurun_out/pmc_ppo_fused 30 > gpurun_out/r2_ppo_fused_mfma_util.csv; head -12 gpurun_out/r2_ppo_fused_mfma_util.csv | cut -c1-150
find gpurun_out/pmc_ppo_fused -name '*.csv' -size +20M -delete
timeout -k 10 500 python bench.py --mode ppo --steps 1 --warmup 1 > gpurun_out/r2_bench_ppo_n1.json 2> gpurun_out/r2_bench_ppo_n1.err; tail -2 gpurun_out/r2_bench_ppo_n1.err; cat gpurun_out/r2_bench_ppo_n1.json
timeout -k 10 500 python bench.py --mode ppo --gpus 2 --envs 1024 --minibatch 8192 --steps 1 --warmup 1 > gpurun_out/r2_bench_ppo_n2.json 2> gpurun_out/r2_bench_ppo_n2.err; tail -3 gpurun_out/r2_bench_ppo_n2.err; cat gpurun_out/r2_bench_ppo_n2.json
timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_n2.json 2> gpurun_out/r2_bench_n2.err; tail -3 gpurun_out/r2_bench_n2.err; cat gpurun_out/r2_bench_n2.json
