"""bench.py --gpus N: the parent starts N ranks as child processes (torch.distributed.run), the ranks form a process
group, run the sanity all-reduce and rank 0 prints ONE JSON line with n_gpus == N.  On this CPU-only host the ranks
use gloo and --rehearse (no engine: the product path has no CPU fallback); on the GPU box the same launcher runs the
engine over RCCL."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    return p, lines


def test_gpus_2_spawns_two_ranks_and_reports_n_gpus_2():
    p, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1
    coll = res["config"]["collective"]
    assert coll["world_size"] == 2 and coll["backend"] == "gloo"
    assert coll["sanity_allreduce_of_ones"] == 2.0           # both ranks took part in the collective
    assert res["data"] == "rehearsal" and res["value"] == 0.0  # never mistaken for a measurement


def test_gpus_8_rehearsal_bucket_protocol_and_slab_backing_gather():
    """World size 8 over gloo: the launcher, the per-rank slab-backing gather and the two-group gradient-bucket protocol
    with ranks that ask for different step counts."""
    p, lines = _run(["--gpus", "8", "--steps", "3", "--warmup", "1", "--rehearse"], timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    cfg = res["config"]
    assert res["n_gpus"] == 8 and cfg["collective"]["sanity_allreduce_of_ones"] == 8.0
    assert cfg["slab_backing"] == {"per_rank": ["rehearsal"] * 8, "agree": True}
    gb = cfg["grad_bucket"]
    assert gb["optimiser_steps"] == 1 + 3 + 2                     # the max over ranks of (W + K + rank % 3)
    assert gb["allreduces_issued"] == 2 * gb["optimiser_steps"] and gb["gradients_copied_into_bucket"] == 0
    assert gb["replicas_identical"] is True


def test_ranks_that_disagree_on_the_slab_backing_abort_the_run():
    p, lines = _run(["--gpus", "4", "--steps", "1", "--warmup", "0", "--rehearse"], env_extra={"TW_REHEARSE_ODD_RANK": "2"},
                    timeout=600)
    assert p.returncode != 0 and not lines
    assert "disagree on the output slab backing" in p.stderr


def test_launched_by_torchrun_like_the_driver():
    """The driver's own form: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ..."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_world_size_mismatch_is_an_error():
    p, lines = _run(["--gpus", "2", "--rehearse"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and not lines


def test_failing_rank_fails_the_launcher():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the ranks would run the engine")
    # without --rehearse a CPU-only host has no engine: every rank exits non-zero and so must the parent
    p, lines = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert p.returncode != 0 and not lines
