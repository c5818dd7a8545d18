#!/usr/bin/env python3
"""HBM traffic per launch of tw_pipe_kernel from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py \\
      --no-cpu-baseline --steps 8 --warmup 2
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ... (same)
  python tools/traffic_from_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r03_traffic.json
(tools/gpu_traffic.sh does all three on the GPU box)

FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 reports half of a wide coalesced read stream),
WRITE_SIZE is taken as is (exact for 16-byte-per-lane stores).  The median over the 128-step launches is used.
"""
import csv
import glob
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KERNEL = "tw_pipe_kernel<6, "        # <variant 6, envs per workgroup, layout>: the headline launches <6, 8, 1>
ALGORITHMIC = (4 + 17 * 17 * 3 + 289 * 4 + 8 + 4 + 1 + 1 + 2 * (289 + 289 + 48 * 4) / 128.0) * 4096 * 128


def counter_values(directory, counter):
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    vals.append(float(row["Counter_Value"]))
    return sorted(vals)


def build_id():
    """tw_build_id() of the library in the tree (the one the profiled bench.py run loaded)."""
    from twoarmy_amd import _lib
    return _lib.lib().tw_build_id().decode()


def git_head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True,
                                       stderr=subprocess.DEVNULL).strip()
    except Exception:
        return None                                     # the GPU box's snapshot carries no .git


def main():
    fetch = counter_values(sys.argv[1], "FETCH_SIZE")
    write = counter_values(sys.argv[2], "WRITE_SIZE")
    assert fetch and write, "no %s rows found" % KERNEL
    fb = 2.0 * statistics.median(fetch) * 1024.0
    wb = statistics.median(write) * 1024.0
    print(json.dumps({
        "round": 3, "kernel": KERNEL, "build_id": build_id(), "git_head": git_head(),
        "config": "bench.py default: v6, 4096 envs, T=128 steps per launch, view 17, record layout (tw_alloc_outputs)",
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE; median over the "
                  "128-step launches; counters are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a "
                  "wide coalesced read stream); WRITE_SIZE as is",
        "launches_seen": {"fetch": len(fetch), "write": len(write)},
        "fetch_size_kib_median": statistics.median(fetch), "write_size_kib_median": statistics.median(write),
        "fetch_size_kib_minmax": [fetch[0], fetch[-1]], "write_size_kib_minmax": [write[0], write[-1]],
        "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb,
        "traffic_bytes_per_launch": fb + wb, "algorithmic_bytes_per_launch": ALGORITHMIC,
        "traffic_over_algorithmic": (fb + wb) / ALGORITHMIC}, indent=1))


if __name__ == "__main__":
    main()
