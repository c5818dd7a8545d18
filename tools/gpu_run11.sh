mkdir -p gpurun_out/final
true > gpurun_out/r2_tests11.log 2>&1 || { tail -40 gpurun_out/r2_tests11.log; exit 1; }
tail -3 gpurun_out/r2_tests11.log
o=gpurun_out/final
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --view 7 > $o/bench_view7.json 2>> $o/err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --matrix-codes > $o/bench_codes.json 2>> $o/err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_default2.json 2>> $o/err.log
python - <<'PY'
import json,glob,os
for f in ("bench_view7","bench_codes","bench_default2"):
    d=json.load(open('gpurun_out/final/%s.json'%f)); r=d['roofline']
    print("%-20s %.3f G  kernel %.4f ms  frac %.3f  %s" % (f, d['value']/1e9, r['kernel_ms'], r['frac'], d['config']['output_buffers']))
PY
timeout -k 10 900 python bench.py --mode ppo --steps 2 --warmup 1 --her > $o/bench_ppo_v4_her.json 2>> $o/err.log; python -c "
import json;d=json.load(open('$o/bench_ppo_v4_her.json'));c=d['config'];print('ppo v4 her', d['value'], c['rollout_s'], c['update_s'], c['update_targets_s'], c['update_epoch_s'], c['her_records_per_iteration'])"
timeout -k 10 600 python bench.py --mode ppo --steps 2 --warmup 1 > $o/bench_ppo_v4.json 2>> $o/err.log; python -c "
import json;d=json.load(open('$o/bench_ppo_v4.json'));c=d['config'];print('ppo v4', d['value'], c['rollout_s'], c['update_s'], c['update_targets_s'], c['update_epoch_s'], d['roofline']['achieved'])"
