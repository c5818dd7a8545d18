#!/bin/bash
set -o pipefail
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/quick; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tee $o/tests.log | tail -3
[ ${PIPESTATUS[0]} -eq 0 ] || exit 1
for cfg in "4 1024" "6 4096"; do python tools/stamp_pipe.py $cfg 2>&1 | grep -v "per-wave\|amdgpu.ids"; done
for cfg in "v4 1024" "v6 1024" "v4 2048" "v6 4096" "v4 4096"; do
    set -- $cfg
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --slab-check 1 --variant $1 --envs $2 > $o/bench_$1_$2.json 2>> $o/err.log || exit 1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/quick/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print("%-24s %.3f G  kernel %.4f ms  frac %.3f" % (os.path.basename(f), d['value']/1e9, r['kernel_ms'], r['frac']))
PY
