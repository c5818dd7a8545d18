mkdir -p gpurun_out/final
o=gpurun_out/final
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_default.json 2> $o/err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --torch-outputs > $o/bench_torch_two_streams.json 2>> $o/err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --variant v4 > $o/bench_v4.json 2>> $o/err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --view 7 > $o/bench_view7.json 2>> $o/err.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --matrix-codes > $o/bench_codes.json 2>> $o/err.log
for n in 16384 2048 1024 512; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs $n > $o/bench_envs$n.json 2>> $o/err.log; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs 1024 --variant v4 > $o/bench_envs1024_v4.json 2>> $o/err.log
python bench.py --mode step --no-cpu-baseline > $o/bench_step_mode.json 2>> $o/err.log
TW_SLAB_BACKING=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_slab_hipmalloc.json 2>> $o/err.log
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/final/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
        print("%-34s %.3f G  kernel %.4f ms  frac %.3f  fill %.0f" % (os.path.basename(f), d['value']/1e9, r['kernel_ms'], r['frac'], r.get('measured_fill_ceiling_GBs',0)))
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 600 python bench.py --mode ppo --k-epochs 1 --steps 1 --warmup 1 > $o/bench_ppo_v4.json 2>> $o/err.log; python -c "
import json;d=json.load(open('$o/bench_ppo_v4.json'));c=d['config'];print('ppo v4', d['value'], c['rollout_s'], c['update_s'], c['update_targets_s'], c['update_epoch_s'], d['roofline']['achieved'])"
timeout -k 10 600 python bench.py --mode ppo --k-epochs 1 --steps 1 --warmup 1 --her > $o/bench_ppo_v4_her.json 2>> $o/err.log; python -c "
import json;d=json.load(open('$o/bench_ppo_v4_her.json'));c=d['config'];print('ppo v4 her', d['value'], c['rollout_s'], c['update_s'], c['her_records_per_iteration'])"
python tools/ppo_kernels_bench.py > $o/ppo_kernels.json 2>> $o/err.log; tail -3 $o/ppo_kernels.json | cut -c1-300
