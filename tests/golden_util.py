"""Helpers to read tests/golden/twoarmy_traces.npz (recorded from the reference by oracle/gen_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCAL = ["step_count", "step_move", "pone", "patrol", "up1", "right2", "Update_longitudinal",
        "Update_horizontal", "risk_count", "first_to_room2", "agent_dir"]
_cache = {}


def load_traces():
    if "traces" not in _cache:
        z = np.load(os.path.join(GOLDEN, "twoarmy_traces.npz"))
        n = int(z["n_traces"])
        traces = []
        for i in range(n):
            pre = "t%02d_" % i
            traces.append({k[len(pre):]: z[k] for k in z.files if k.startswith(pre)})
        _cache["traces"] = (traces, int(z["seed"]))
    return _cache["traces"]


def explicit_draws(trace):
    """For traces recorded with the reference's natural MT19937 stream: per-step uint32[8] words
    such that lo + word % n reproduces the recorded value.  Returns {t: words}."""
    out = {}
    for t, slot, lo, n, v in trace["draw_log"]:
        w = out.setdefault(int(t), np.zeros(8, np.uint32))
        if slot < 8:
            w[int(slot)] = v - lo
    return out
