#!/bin/bash
# The whole GPU suite on one box (unbuffered: a silent run is taken for hung after 7 minutes).
export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -v --durations=8 2>&1 | tee gpurun_out/suite.log | grep -v "PASSED" | tail -40
exit ${PIPESTATUS[0]}
