#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r3d; mkdir -p $o
for v in 4 6; do
  for n in 1024 4096; do
    AB_ENVS=$n AB_VARIANT=$v python tools/ab_variants.py std noexp noemit 2>&1 | grep round > $o/ab_v${v}_$n.log
    echo "v$v $n"; awk '{a[$3]=a[$3]" "$4} END{for(k in a) print k, a[k]}' $o/ab_v${v}_$n.log
  done
  python tools/stamp_pipe.py $v 1024 2>&1 | tail -4
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --slab-check 1 --variant v4 --envs 1024 > $o/bench_prof.json 2> $o/prof.err
f=$(find $o/prof -name '*kernel_stats.csv' | head -1); head -5 $f | cut -c1-200
find $o -name '*kernel_trace.csv' -size +5M -delete
