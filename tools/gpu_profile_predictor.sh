# kernel trace of the configs[4] share (PPO + predictor head, 2048 envs, code frames): an un-profiled run first fills
# MIOpen's find cache, then the same command under rocprofv3; top 40 kernels by total time -> gpurun_out/
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --mode ppo --k-epochs 1 --predictor --matrix-codes --envs 2048 --steps 1 --warmup 1 > gpurun_out/r2_pred_warm.json 2> gpurun_out/r2_pred_warm.err || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pred -- python3 bench.py --mode ppo --k-epochs 1 --predictor --matrix-codes --envs 2048 --steps 1 --warmup 1 > gpurun_out/r2_pred_under_rocprof.json 2> gpurun_out/r2_prof_pred.err || exit 1
f=$(find gpurun_out/prof_pred -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/r2_ppo_predictor_kernel_stats_top40.csv", "w") as out:
    out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows[:40]:
        name = r["Name"][:160].replace('"', "'")
        out.write('"%s",%s,%s,%s,%.2f\n' % (name, r["Calls"], r["TotalDurationNs"], r["AverageNs"], 100 * float(r["TotalDurationNs"]) / tot))
print("total GPU ms", tot / 1e6)
for r in rows[:14]:
    print("%6.2f%%  %6s calls  %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
find gpurun_out/prof_pred -name '*kernel_trace.csv' -size +20M -delete
