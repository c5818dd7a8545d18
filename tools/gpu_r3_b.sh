#!/bin/bash
# Round 3: logic-loop rewrite -- bit-exactness (engine + stack tests), then the small-batch shares and the headline.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r3b; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_stack_gpu.py -x -q --durations=5 > $o/tests.log 2>&1; rc=$?
tail -8 $o/tests.log
[ $rc -eq 0 ] || exit $rc
for cfg in "v4 1024" "v6 1024" "v4 2048" "v6 2048" "v4 4096" "v6 4096" "v4 512" "v6 512"; do
    set -- $cfg
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --slab-check 1 --variant $1 --envs $2 > $o/bench_$1_$2.json 2>> $o/err.log || exit 1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3b/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print("%-24s %.3f G  kernel %.4f ms  frac %.3f  samples %s" % (os.path.basename(f), d['value']/1e9, r['kernel_ms'], r['frac'], {k: round(v,4) for k,v in r['kernel_ms_samples'].items()}))
PY
