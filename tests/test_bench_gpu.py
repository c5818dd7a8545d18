"""bench.py on the GPU, driven exactly like the driver drives it: one JSON line, `value` and `roofline.achieved` from the
same timed launches, engine slab outputs; and the --gpus 2 launcher on a one-GPU box (two ranks share the GPU, gloo)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_driver_command_line():
    r = _bench(["--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"])
    assert r["metric"] == "env-steps/sec" and r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 2
    assert r["config"]["env_steps_per_step"] == 4096 * 128 and "records" in r["config"]["output_buffers"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["launches_timed"] == 6 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # value (host wall clock) and achieved (events on the launch stream) describe the same launches
    implied = r["value"] * rf["algorithmic_bytes_per_env_step"] / 1e9
    assert 0.9 * rf["achieved"] < implied <= 1.001 * rf["achieved"]
    assert r["value"] > 1.0e9                                   # a launch-bound or fallback path would be far below


def test_two_ranks_on_one_gpu_through_the_launcher():
    r = _bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--envs", "1024"])
    assert r["n_gpus"] == 2
    c = r["config"]["collective"]
    assert c["world_size"] == 2 and c["sanity_allreduce_of_ones"] == 2.0


def test_two_ranks_ppo_mode_zero_copy_bucket():
    """--mode ppo on two ranks (they share the GPU: gloo): every optimiser step issues two all-reduces (actor's while the
    critic's backward runs, then the critic's), nothing is copied into the bucket, the line carries the bucket's timings."""
    r = _bench(["--gpus", "2", "--mode", "ppo", "--variant", "v4", "--envs", "256", "--minibatch", "8192", "--k-epochs", "2",
                "--steps", "1", "--warmup", "1", "--graph", "off"], timeout=900)
    assert r["n_gpus"] == 2 and r["dtype"] == "f32"
    gb = r["config"]["grad_bucket"]
    steps = gb["optimiser_steps_per_iteration"]
    assert steps == 2 * 4                                        # 256 x 128 samples / 8192 per minibatch, K = 2
    assert gb["allreduces_issued"] == 2 * steps and gb["gradients_copied_into_bucket"] == 0
    assert gb["floats"] == 2515206 and gb["ms_per_allreduce"] > 0 and gb["ms_exposed_per_optimiser_step_mean"] > 0
    assert r["config"]["slab_backing"]["agree"] is True
