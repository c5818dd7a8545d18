set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests1.log 2>&1 || { tail -30 gpurun_out/r2_tests1.log; exit 1; }
tail -3 gpurun_out/r2_tests1.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
cat gpurun_out/r2_bench1.json
python bench.py --steps 20 --warmup 5 --torch-outputs --no-cpu-baseline > gpurun_out/r2_bench1_torch.json 2>> gpurun_out/r2_bench1.err
cat gpurun_out/r2_bench1_torch.json
timeout -k 10 300 python tools/placement_probe2.py 3 > gpurun_out/r2_place.log 2>&1 || true
tail -25 gpurun_out/r2_place.log
