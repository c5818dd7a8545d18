# Final validation of a build on one MI355X box: smoke, the whole GPU suite, the driver's bench line, the PPO loop,
# the two-rank rehearsal of the PPO + predictor loop (gloo: one GPU for two ranks), a kernel trace of the headline.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r2_suite_final.log 2>&1; rc=$?
tail -12 gpurun_out/r2_suite_final.log
[ $rc -eq 0 ] || exit $rc
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench_final.json 2> gpurun_out/r2_bench_final.err || exit 1
cut -c1-330 gpurun_out/r2_bench_final.json
timeout -k 10 400 python bench.py --mode ppo --k-epochs 1 --steps 2 --warmup 1 > gpurun_out/r2_bench_ppo_final.json 2> gpurun_out/r2_bench_ppo_final.err || exit 1
timeout -k 10 400 python bench.py --gpus 2 --mode ppo --k-epochs 1 --predictor --matrix-codes --envs 1024 --minibatch 8192 --steps 1 --warmup 1 > gpurun_out/r2_bench_ppo_predictor_n2_gloo.json 2> gpurun_out/r2_bench_ppo_predictor_n2_gloo.err || exit 1
python - <<'PY'
import json
for f in ("r2_bench_ppo_final", "r2_bench_ppo_predictor_n2_gloo"):
    d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1]); c = d["config"]
    print(f, d["n_gpus"], round(d["value"]), {k: c.get(k) for k in ("rollout_s", "update_s", "update_targets_s", "update_epoch_s")},
          c.get("collective"), {k: v for k, v in c.get("grad_bucket", {}).items() if k.startswith("ms") or k == "allreduces_timed"})
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -- python3 bench.py --no-cpu-baseline --steps 40 --warmup 10 > gpurun_out/r2_bench_under_rocprof_final.json 2> gpurun_out/r2_prof_final.err
f=$(find gpurun_out/prof_final -name '*kernel_stats.csv' | head -1); head -4 $f | cut -c1-170
find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
