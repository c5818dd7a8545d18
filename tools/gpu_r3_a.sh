#!/bin/bash
# Round 3, first GPU call: smoke, the slab-reuse reproducer (each mode once), the new / changed GPU tests, the headline
# line with median-of-R regions, full PPO at the reference's K_epochs = 10.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r3a; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
for mode in "1 1" "1 0" "0 1" "0 0"; do
    timeout -k 10 120 tools/vmm_reuse_repro $mode 3 >> $o/vmm_repro.log 2>&1 || { echo "repro mode $mode rc=$?" >> $o/vmm_repro.log; break; }
done
cat $o/vmm_repro.log
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py::test_slab_free_then_realloc_writes_every_row tests/test_dist_gpu.py tests/test_bench_gpu.py tests/test_her_gpu.py -x -q --durations=5 > $o/tests.log 2>&1; rc=$?
tail -15 $o/tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/bench_default.err || exit 1
cut -c1-200 $o/bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3a/bench_default.json")); r=d["roofline"]
print(d["value"]/1e9, d["ms_per_step"], r["frac"], r["regions"], r["gpu_ms_timed_total"], r["kernel_ms_samples"], r["traffic"], r["traffic_source"], d["config"]["slab_backing"])
PY
timeout -k 10 900 python bench.py --mode ppo --variant v6 --k-epochs 10 --minibatch 32768 --steps 1 --warmup 1 > $o/bench_ppo_v6_4096_k10.json 2> $o/bench_ppo_v6_4096_k10.err || exit 1
cut -c1-150 $o/bench_ppo_v6_4096_k10.json
timeout -k 10 600 python bench.py --mode ppo --variant v6 --k-epochs 1 --minibatch 32768 --steps 2 --warmup 1 > $o/bench_ppo_v6_4096_k1.json 2> $o/bench_ppo_v6_4096_k1.err || exit 1
cut -c1-150 $o/bench_ppo_v6_4096_k1.json
timeout -k 10 600 python bench.py --mode ppo --variant v4 --envs 1024 --k-epochs 10 --minibatch 8192 --steps 1 --warmup 1 > $o/bench_ppo_v4_1024_k10.json 2> $o/bench_ppo_v4_1024_k10.err || exit 1
cut -c1-150 $o/bench_ppo_v4_1024_k10.json
timeout -k 10 600 python bench.py --mode ppo --variant v4 --envs 1024 --k-epochs 1 --minibatch 8192 --steps 2 --warmup 1 > $o/bench_ppo_v4_1024_k1.json 2> $o/bench_ppo_v4_1024_k1.err || exit 1
cut -c1-150 $o/bench_ppo_v4_1024_k1.json
timeout -k 10 300 python bench.py --gpus 2 --mode ppo --variant v4 --envs 512 --k-epochs 1 --minibatch 8192 --steps 1 --warmup 1 > $o/bench_ppo_n2_gloo.json 2> $o/bench_ppo_n2_gloo.err || exit 1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3a/bench_ppo_n2_gloo.json")); print(d["n_gpus"], d["value"], d["config"]["grad_bucket"], d["config"]["slab_backing"])
PY
