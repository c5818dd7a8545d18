mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/r2_tests9.log 2>&1 || { tail -40 gpurun_out/r2_tests9.log; exit 1; }
tail -3 gpurun_out/r2_tests9.log
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench9_$i.json 2> gpurun_out/r2_bench9.err; python -c "
import json;d=json.load(open('gpurun_out/r2_bench9_$i.json'));print(d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --variant v4 > gpurun_out/r2_bench9_v4.json 2>> gpurun_out/r2_bench9.err; python -c "
import json;d=json.load(open('gpurun_out/r2_bench9_v4.json'));print('v4', d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'])"
for n in 2048 1024 512; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs $n > gpurun_out/r2_bench9_$n.json 2>> gpurun_out/r2_bench9.err; python -c "
import json;d=json.load(open('gpurun_out/r2_bench9_$n.json'));print($n, d['value']/1e9, d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
