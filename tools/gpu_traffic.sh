#!/bin/bash
# HBM traffic of the headline kernel for the CURRENT build (run on the GPU box, from the repo root):
#   gpurun --timeout 900 -- 'bash tools/gpu_traffic.sh r03'
# Two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate passes, --kernel-trace only, the program itself after
# `--`), then tools/traffic_from_pmc.py -> gpurun_out/<tag>_traffic.json, which records the library's tw_build_id():
# copy it to profiles/ -- bench.py reports `roofline.traffic` only from a profile whose build id equals the running one.
set -e -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
    d=gpurun_out/pmc_$c
    rm -rf "$d"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$d" -- python3 bench.py --no-cpu-baseline \
        --steps 4 --warmup 2 --regions 4 --slab-check 1 > gpurun_out/${tag}_pmc_$c.json 2> gpurun_out/${tag}_pmc_$c.err
done
python3 tools/traffic_from_pmc.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/${tag}_traffic.json
cat gpurun_out/${tag}_traffic.json
find gpurun_out -name '*kernel_trace.csv' -size +20M -delete
find gpurun_out -name '*counter_collection.csv' -size +20M -delete
