#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from a rocprofv3 counter pass:

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d DIR -- python3 tools/ppo_bench.py ...
  python tools/mfma_util_from_pmc.py DIR > profiles/rNN_ppo_..._mfma_util.csv

GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over every SIMD of the chip, so
mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the share of SIMD-cycles, while the kernel ran, in which
the matrix pipe was busy."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: [0, 0.0, 0.0])
seen = set()
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            a = acc[r["Kernel_Name"]]
            key = (r["Dispatch_Id"], r["Kernel_Name"])
            if key not in seen:
                seen.add(key)
                a[0] += 1
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                a[1] += float(r["Counter_Value"])
            elif r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
                a[2] += float(r["Counter_Value"])
tot = sum(v[1] for v in acc.values()) or 1.0
w = csv.writer(sys.stdout)
w.writerow(["kernel", "calls", "share_of_gpu_active_pct", "mfma_busy_cycles", "mfma_util_pct"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    util = 100.0 * v[2] / (v[1] / 8.0 * 256 * 4) if v[1] else 0.0
    w.writerow([k[:100], v[0], "%.2f" % (100.0 * v[1] / tot), "%d" % v[2], "%.2f" % util])
