#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r3e; mkdir -p $o
timeout -k 10 60 tools/issue_probe > $o/issue_probe.log 2>&1; cat $o/issue_probe.log
for v in 4 6; do python tools/stamp_pipe.py $v 1024 2>&1 | grep -v "per-wave" | tail -3; done
python tools/stamp_pipe.py 6 4096 2>&1 | grep -v "per-wave" | tail -3
