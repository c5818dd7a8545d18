"""bench.py pieces that do not need a GPU: the roofline accounting and the CPU-baseline leg (oracle, short sample)."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_match_design():
    # 2041 B per env-step + 1540 B of plane/record traffic per launch amortised over T
    assert bench.algorithmic_bytes_per_env_step(17, 128) == 2041 + 1540 / 128.0
    assert bench.algorithmic_bytes_per_env_step(7, 128) == 4 + 147 + 1156 + 8 + 4 + 1 + 1 + 1540 / 128.0
    assert bench.algorithmic_bytes_per_env_step(17, 128, 289) == 2041 - 867 + 1540 / 128.0        # uint8 code frames


def test_traffic_profile_is_consistent_with_the_accounting():
    t, src = bench.traffic_from_profile("v6", 4096, 128, 17)
    alg = bench.algorithmic_bytes_per_env_step(17, 128) * 4096 * 128
    assert t is not None and 0.98 * alg < t < 1.10 * alg          # measured HBM bytes: no wasted re-reads / re-writes
    assert bench.traffic_from_profile("v4", 4096, 128, 17) == (None, None)   # only the profiled configuration carries a number
    with open(os.path.join(ROOT, src)) as f:
        assert abs(json.load(f)["algorithmic_bytes_per_launch"] - alg) < 1.0


def test_cpu_baseline_leg_runs_the_oracle():
    assert 1 <= bench.usable_cores() <= 64
    r = bench.cpu_baseline(6, 256, 17, seconds=0.5)
    assert r["kind"] == "port" and r["unit"] == "env-steps/s" and r["value"] > 1e4 and r["single_thread_value"] > 1e4
    assert r["cores"] == bench.usable_cores()
