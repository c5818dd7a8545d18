mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ppo_nhwc -- python3 tools/ppo_bench.py --updates 1 --nhwc > gpurun_out/r2_ppo_nhwc_prof.json 2> gpurun_out/r2_ppo_nhwc_prof.err
tail -2 gpurun_out/r2_ppo_nhwc_prof.err
f=$(find gpurun_out/prof_ppo_nhwc -name '*kernel_stats.csv' | head -1); echo $f; head -40 $f | cut -c1-200
find gpurun_out/prof_ppo_nhwc -name '*kernel_trace.csv' -size +60M -delete
