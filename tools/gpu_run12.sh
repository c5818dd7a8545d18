mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/r2_tests12.log 2>&1 || { tail -40 gpurun_out/r2_tests12.log; exit 1; }
tail -2 gpurun_out/r2_tests12.log
for cfg in "4096 v6" "4096 v4" "1024 v6" "1024 v4" "512 v4"; do set -- $cfg; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --envs $1 --variant $2 > gpurun_out/b.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/b.json'));print('$1 $2', round(d['value']/1e9,3), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],3))"; done
