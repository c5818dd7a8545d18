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


def test_traffic_profile_is_only_reported_for_the_build_it_was_taken_on(tmp_path):
    alg = bench.algorithmic_bytes_per_env_step(17, 128) * 4096 * 128
    for name, bid, t in (("r07_traffic.json", "aaaa", 1.01 * alg), ("r08_traffic.json", "bbbb", 1.02 * alg),
                         ("r06_traffic.json", None, 1.5 * alg)):
        d = {"traffic_bytes_per_launch": t, "algorithmic_bytes_per_launch": alg}
        if bid:
            d["build_id"] = bid
        (tmp_path / name).write_text(json.dumps(d))
    t, src = bench.traffic_from_profile("v6", 4096, 128, 17, build_id="aaaa", profiles_dir=str(tmp_path))
    assert t == 1.01 * alg and "r07_traffic.json" in src
    t, src = bench.traffic_from_profile("v6", 4096, 128, 17, build_id="bbbb", profiles_dir=str(tmp_path))
    assert t == 1.02 * alg and "r08_traffic.json" in src
    t, src = bench.traffic_from_profile("v6", 4096, 128, 17, build_id="cccc", profiles_dir=str(tmp_path))
    assert t is None and "stale" in src and "r06_traffic.json: build not recorded" in src    # never a stale number
    assert bench.traffic_from_profile("v4", 4096, 128, 17, build_id="aaaa", profiles_dir=str(tmp_path)) == (None, None)


def test_committed_traffic_profiles_are_consistent_with_the_accounting():
    alg = bench.algorithmic_bytes_per_env_step(17, 128) * 4096 * 128
    pdir = os.path.join(ROOT, "profiles")
    names = [f for f in os.listdir(pdir) if f.endswith("_traffic.json")]
    assert names
    for name in names:
        with open(os.path.join(pdir, name)) as f:
            prof = json.load(f)
        assert abs(prof["algorithmic_bytes_per_launch"] - alg) < 1.0
        assert 0.98 * alg < prof["traffic_bytes_per_launch"] < 1.10 * alg       # measured HBM bytes: no wasted re-reads / re-writes


def test_library_reports_the_build_id_of_its_sources():
    import hashlib
    import twoarmy_amd
    twoarmy_amd._lib.build()
    csrc = twoarmy_amd._lib.CSRC_DIR
    files = [os.path.join(csrc, f) for f in ("twoarmy_engine.hip", "ppo_kernels.hip", "minigrid_view.hip")] + \
        sorted(os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include")) if f.endswith(".h"))
    h = hashlib.sha256()
    for fn in files:
        with open(fn, "rb") as f:
            h.update(f.read())
    assert twoarmy_amd._lib.lib().tw_build_id().decode() == h.hexdigest()[:16]


def test_region_count_rule():
    assert bench.auto_regions(20, 0.17) == 295                  # the driver's command at 4096 envs: ~1 s of GPU time
    assert bench.auto_regions(20, 5.0) == 25                    # never fewer than 25 regions
    assert bench.auto_regions(4, 0.05) == 400                   # ... nor more than 400
    assert 0.9 < bench.auto_regions(40, 0.17) * 40 * 0.17 / 1000.0 < 1.1


def test_cpu_baseline_leg_runs_the_oracle():
    assert 1 <= bench.usable_cores() <= 64
    r = bench.cpu_baseline(6, 256, 17, seconds=0.5)
    assert r["kind"] == "port" and r["unit"] == "env-steps/s" and r["value"] > 1e4 and r["single_thread_value"] > 1e4
    assert r["cores"] == bench.usable_cores()
