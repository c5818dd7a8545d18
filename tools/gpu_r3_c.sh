#!/bin/bash
# Round 3: logic-loop variants A/B in one process on one box, then correctness of the shipped build.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r3c; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
for v in 4 6; do
  AB_ENVS=1024 AB_VARIANT=$v python tools/ab_variants.py std noexp 2>&1 | grep round > $o/ab_v${v}_1024.log
  awk '{a[$3]=a[$3]" "$4} END{for(k in a) print k, a[k]}' $o/ab_v${v}_1024.log
done
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -x -q > $o/tests.log 2>&1; rc=$?
tail -3 $o/tests.log
[ $rc -eq 0 ] || exit $rc
for cfg in "v4 1024" "v6 1024" "v6 4096"; do
    set -- $cfg
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --slab-check 1 --variant $1 --envs $2 > $o/bench_$1_$2.json 2>> $o/err.log || exit 1
done
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob('gpurun_out/r3c/bench_*.json')):
    d=json.load(open(f)); r=d['roofline']
    print("%-24s %.3f G  kernel %.4f ms  frac %.3f" % (os.path.basename(f), d['value']/1e9, r['kernel_ms'], r['frac']))
PY
