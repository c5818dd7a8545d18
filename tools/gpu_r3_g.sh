#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r3g; mkdir -p $o
for cfg in "4 1024" "6 1024" "6 4096"; do python tools/stamp_pipe.py $cfg 2>&1 | grep -v "per-wave" | tail -3; done
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $o/suite.log 2>&1; rc=$?
tail -10 $o/suite.log
exit $rc
