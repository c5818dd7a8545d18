set -e
mkdir -p gpurun_out
python -m pytest tests/test_engine_gpu.py tests/test_stack_gpu.py -m gpu -x -q > gpurun_out/r2_tests2.log 2>&1 || { tail -30 gpurun_out/r2_tests2.log; exit 1; }
tail -3 gpurun_out/r2_tests2.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench2.json 2> gpurun_out/r2_bench2.err
cat gpurun_out/r2_bench2.json
python bench.py --steps 20 --warmup 5 --torch-outputs --no-cpu-baseline > gpurun_out/r2_bench2_torch.json 2>> gpurun_out/r2_bench2.err
cat gpurun_out/r2_bench2_torch.json
timeout -k 10 300 python tools/placement_probe2.py 4 torch,hipMalloc,c2M > gpurun_out/r2_place3.log 2>&1 || true
tail -6 gpurun_out/r2_place3.log
