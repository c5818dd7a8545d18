#!/bin/bash
# Round 3: evidence for the CURRENT build on one box: headline (driver command), kernel trace, HBM traffic counters,
# other workloads, MFMA counters of the PPO update.  Unbuffered + progress lines (a silent run is killed after 7 min).
set -o pipefail
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r3final; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err || exit 1
cut -c1-160 $o/bench_default.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > $o/bench_under_rocprof.json 2> $o/prof.err
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); cp $f $o/kernel_stats.csv; head -4 $o/kernel_stats.csv | cut -c1-170
find $o -name '*kernel_trace.csv' -size +5M -delete
bash tools/gpu_traffic.sh r03 > $o/traffic.log 2>&1; tail -5 $o/traffic.log; cp gpurun_out/r03_traffic.json $o/ 2>/dev/null
echo "== other workloads"
for extra in "--matrix-codes" "--view 7" "--variant v4" "--envs 16384" "--mode step"; do
    n=$(echo $extra | tr -d ' -'); python bench.py --steps 20 --warmup 5 --no-cpu-baseline --slab-check 1 $extra > $o/bench_$n.json 2>> $o/err.log || exit 1
done
for cfg in "v4 1024" "v6 1024" "v4 2048" "v6 2048" "v4 512" "v6 512"; do
    set -- $cfg
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --slab-check 1 --variant $1 --envs $2 > $o/bench_$1_$2.json 2>> $o/err.log || exit 1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3final/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print("%-28s %.3f G  kernel %.4f ms  frac %.3f  traffic %s" % (os.path.basename(f), d['value']/1e9, r['kernel_ms'], r['frac'], r.get('traffic')))
    except Exception as e: print(f, "ERR", e)
PY
echo "== MFMA counters of the PPO loop (un-profiled run first: MIOpen's search must not run under the counters)"
python tools/ppo_bench.py --envs 512 --T 16 --minibatch 8192 --updates 1 --nhwc > $o/ppo_small.json 2>> $o/err.log
timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $o/pmc_mfma -- python3 tools/ppo_bench.py --envs 512 --T 16 --minibatch 8192 --updates 1 --nhwc > $o/ppo_small_pmc.json 2> $o/pmc_mfma.err
python tools/mfma_util_from_pmc.py $o/pmc_mfma > $o/ppo_mfma_util.csv; head -12 $o/ppo_mfma_util.csv | cut -c1-150
find $o -name '*counter_collection.csv' -size +5M -delete; find $o -name '*kernel_trace.csv' -size +5M -delete
