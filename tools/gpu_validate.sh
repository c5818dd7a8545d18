mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r2_suite2.log 2>&1; rc=$?
tail -14 gpurun_out/r2_suite2.log
[ $rc -eq 0 ] || exit $rc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench2 -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/r2_bench_under_rocprof2.json 2> gpurun_out/r2_prof_bench2.err
f=$(find gpurun_out/prof_bench2 -name '*kernel_stats.csv' | head -1); head -5 $f | cut -c1-170
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch3 -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2> gpurun_out/r2_pmc_fetch3.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write3 -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2> gpurun_out/r2_pmc_write3.err
python tools/traffic_from_pmc.py gpurun_out/pmc_fetch3 gpurun_out/pmc_write3 > gpurun_out/r2_traffic3.json; grep "traffic_over\|launches" -A1 gpurun_out/r2_traffic3.json | head
find gpurun_out -name '*counter_collection.csv' -size +20M -delete; find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err; cat gpurun_out/r2_bench_final.json | cut -c1-400
