#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched HIP Twoarmy step engine.

Workload (default, --mode rollout) = BASELINE.json configs[1]: MiniGrid-twoarmy-17x17-v6, 4096 envs on one
MI355X, batched HIP step() only (obs + state matrix + reward/done written every env-step), auto-reset on, view 17,
actions = Philox(seed 9981) policy indices resident in HBM.

A bench "step" = ONE ROLLOUT LAUNCH = ROLLOUT_T (128) environment steps of all envs of one GPU (tw_rollout,
include/twoarmy.h): `--steps K --warmup W` runs W untimed launches and then times exactly K launches, so
value = n_gpus * envs * 128 * K / wall.  The same K launches are bracketed by events on the launch stream and
`roofline.achieved` comes from that one region (value x bytes-per-env-step == achieved up to the host's
barrier/sync overhead, reported as roofline.wall_over_event).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--variant v6] [--view 17]
                  [--mode rollout|step|ppo] [--no-cpu-baseline]

--gpus N > 1 without a torch.distributed environment: this process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (before anything here touches the
GPU), relays rank 0's JSON line and exits with the child's code.  Every rank steps its own env range
(env ids rank*envs ..): the path shards with no data-path collective, scaling = weak.  The process group is
"nccl" (= RCCL over xGMI) with one GPU per rank; one sanity all-reduce is recorded in the output.

--mode ppo: BASELINE configs[2]/[3] loop (v4 by default): each step = one PPO iteration of the rank's envs
(128-step rollout with the actor in the loop + K-epoch update, K = 10 and minibatch = buffer / 16 like the reference unless
--k-epochs / --minibatch say otherwise); gradients go through dist.GradBucket (zero-copy views into one flat fp32 buffer, the
actor's all-reduce overlaps the critic's backward); its exposed and stand-alone times are reported per optimiser step.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROLLOUT_T = 128
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SEED = 9981                    # reference default, soa/train_ppo.py:25
FWD_FLOP_PER_SAMPLE_PER_NET = 47.5e6


def algorithmic_bytes_per_env_step(view, rollout_t, matrix_bytes=289 * 4):
    """Bytes that must cross HBM per env-step for the design built:
    action 4 + obs V*V*3 + state matrix 289*4 + pos 8 + reward 4 + terminated 1 + truncated 1,
    plus the per-launch load/store of the LDS-resident planes + record amortised over T."""
    per_step = 4 + view * view * 3 + matrix_bytes + 8 + 4 + 1 + 1
    per_launch = 2 * (289 + 289 + 48 * 4)
    return per_step + per_launch / float(rollout_t)


def traffic_from_profile(variant, n_envs, rollout_t, view, build_id=None, profiles_dir=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile of the SAME configuration AND the
    SAME build (a bench run cannot profile itself): the newest profiles/r*_traffic.json whose recorded `build_id`
    equals the running library's tw_build_id().  (None, reason) when the configuration differs from the profiled one
    or every committed profile was taken on another build -- a stale number is never reported.
    tools/gpu_traffic.sh re-takes the profile (two rocprofv3 --pmc passes) for the current build."""
    if not (variant == "v6" and n_envs == 4096 and rollout_t == 128 and view == 17):
        return None, None
    pdir = profiles_dir or os.path.join(ROOT, "profiles")
    if build_id is None:
        from twoarmy_amd import _lib
        build_id = _lib.lib().tw_build_id().decode()
    stale = []
    for name in sorted((f for f in os.listdir(pdir) if f.endswith("_traffic.json")), reverse=True):
        with open(os.path.join(pdir, name)) as f:
            prof = json.load(f)
        if prof.get("build_id") == build_id:
            return prof["traffic_bytes_per_launch"], "profiles/%s (build %s)" % (name, build_id)
        stale.append("%s: build %s" % (name, prof.get("build_id", "not recorded")))
    return None, "no counter profile of build %s (stale: %s); run tools/gpu_traffic.sh" % (build_id, "; ".join(stale))


def auto_regions(K, est_ms):
    """Number of timed regions of K launches each: at least 25, about one second of GPU time in total (a sampling
    monitor then sees the kernel), at most 400.  Every rank calls this with the SAME est_ms (the slowest rank's)."""
    return min(400, max(25, int(1000.0 / (K * est_ms)) + 1))


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut down by a cgroup CPU quota when there is one
    (the GPU boxes expose every core in the mask but schedule a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    per = int(f.read().split()[0])
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return min(n, 64)                      # 4096 envs: beyond 64 threads a shard is < 64 envs and threads only contend


def cpu_baseline(variant, n_envs, view, seconds=10.0):
    """CPU oracle ("port": oracle/twoarmy_oracle.c) on a bounded sample of the same workload: the same 4096 envs
    stepped with the same Philox action stream and auto-reset, sharded over every host core this process may use
    (~10 s), plus a short single-thread run for the per-core figure."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import twoarmy_oracle as orc
    orc.lib()
    cores = usable_cores()
    steps1, dt1 = orc.timed_rollout(variant, n_envs, 4.0, SEED, view=view, threads=1)
    steps, dt = orc.timed_rollout(variant, n_envs, seconds, SEED, view=view, threads=cores)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d envs x %d steps of the same workload (oracle/twoarmy_oracle.c, gcc -O2, %d threads, %.1f s)"
                      % (n_envs, steps // n_envs, cores, dt),
            "single_thread_value": steps1 / dt1, "host_cores_available": os.cpu_count()}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (rollout/ppo: launches / PPO iterations; "
                    "step mode: single env-batch steps); default 40 / 2 / 2560")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--variant", default=None, help="v6 (default for rollout/step) or v4 (default for ppo)")
    ap.add_argument("--view", type=int, default=17)
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step", "ppo"],
                    help="rollout: 128 env-steps per launch (headline); step: one tw_step launch per env-step; "
                         "ppo: rollout with the actor in the loop + PPO update, gradient all-reduce per optimiser step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--slab-check", type=int, default=3,
                    help="rollout mode: time the kernel into this many freshly allocated engine slabs at set-up and report "
                         "the values (evidence that placement no longer matters; nothing is selected); 1 = skip")
    ap.add_argument("--regions", type=int, default=0,
                    help="rollout/step mode: number of timed regions of --steps launches each (0 = auto: >= 25 regions and about "
                         "1 s of GPU time in total, at most 400); the median region is reported")
    ap.add_argument("--torch-outputs", action="store_true",
                    help="diagnostic: outputs from torch's caching allocator instead of the engine's slab")
    ap.add_argument("--matrix-codes", action="store_true",
                    help="BASELINE configs[4] variant: state matrix as uint8 codes (TW_F_MATRIX_CODE), not the headline")
    # ppo mode
    ap.add_argument("--minibatch", type=int, default=32768, help="ppo mode: samples per optimiser step (reference: buffer / 16, PPO.py:56,122: 32768 of 4096 x 128)")
    ap.add_argument("--k-epochs", type=int, default=10, help="ppo mode: epochs per update (reference: K_epochs = 10, PPO.py:65)")
    ap.add_argument("--amp", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--her", action="store_true")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="ppo mode: replay the rollout as one HIP graph (auto: on for <= 512 envs per GPU)")
    ap.add_argument("--nchw", action="store_true", help="ppo mode: literal NCHW nn.Sequential conv stacks (default: channels-last "
                    "+ fused epilogues)")
    ap.add_argument("--predictor", action="store_true", help="ppo mode: PPO + predictor head (configs[4])")
    ap.add_argument("--miopen-benchmark", action="store_true",
                    help="ppo mode: torch.backends.cudnn.benchmark = True (MIOpen benchmarks its solvers per conv shape once, "
                         "minutes, instead of taking its heuristic pick)")
    ap.add_argument("--rehearse", action="store_true",
                    help="launcher / process-group plumbing only, no engine (for CPU-only hosts: gloo); the line "
                         "carries value 0 and \"data\": \"rehearsal\" and is not a measurement")
    return ap


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Parent of an N-rank run: start the ranks as child processes (this process has not touched the GPU and never
    does), relay their output, return the child's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        sys.stdout.write(out)
        sys.stdout.flush()
        if out.startswith("{") and '"metric"' in out:
            line = out
    rc = proc.wait()
    if rc == 0 and line is not None and json.loads(line).get("n_gpus") != args.gpus:
        print("bench.py: the ranks reported n_gpus != %d" % args.gpus, file=sys.stderr)
        return 3
    if rc == 0 and line is None:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 4
    return rc


def init_ranks(args):
    """(rank, world, device or None, collective record).  One process per GPU; backend nccl (RCCL) when every rank has
    its own GPU, gloo otherwise (CPU-only rehearsal, or more ranks than GPUs on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ngpu = torch.cuda.device_count()
    if ngpu == 0 and not args.rehearse:
        raise SystemExit("bench.py: no GPU visible; the engine has no CPU path (use --rehearse for the launcher only)")
    dev = None
    if ngpu:
        torch.cuda.set_device(local_rank % ngpu)
        dev = torch.device("cuda", local_rank % ngpu)
    coll = {"backend": None, "world_size": world}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if ngpu >= world:                   # one rank per GPU: RCCL over xGMI
            dist.init_process_group(backend="nccl", device_id=dev)
        else:                               # rehearsal of the N>1 path with fewer GPUs than ranks
            dist.init_process_group(backend="gloo")
        coll["backend"] = dist.get_backend()
        one = torch.ones(1024, dtype=torch.float32, device=dev if coll["backend"] == "nccl" else "cpu")
        dist.all_reduce(one)
        coll["sanity_allreduce_of_ones"] = float(one[0].item())       # == world: every rank took part
        assert one.min().item() == world == one.max().item(), "sanity all-reduce saw %s ranks" % one[0].item()
        coll["gpus_visible_per_rank"] = ngpu
    return rank, world, dev, coll


def sync_all(dev, world):
    import torch
    import torch.distributed as dist
    if dev is not None:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if dev is not None:
        torch.cuda.synchronize()


def max_over_ranks(dt, dev, world):
    import torch
    import torch.distributed as dist
    if world == 1:
        return dt
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor([dt], dtype=torch.float64, device=dev if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def max_over_ranks_vec(dts, dev, world):
    """Element-wise MAX over ranks of a list of region times (one all-reduce)."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return list(dts)
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor(dts, dtype=torch.float64, device=dev if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t.tolist()]


def allreduce_alone_ms(bucket, dev, iters=20):
    """The bucket's collectives with nothing to overlap: `iters` x (one all-reduce per parameter group + the scale),
    events on the compute stream (wall clock on CPU tensors).  Every rank calls this the same number of times."""
    import torch
    if not bucket.active():
        return None
    was, bucket.timing = bucket.timing, False
    bucket()
    if dev is not None:
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
    t0 = time.perf_counter()
    for _ in range(iters):
        bucket()
    if dev is not None:
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / iters
    else:
        ms = (time.perf_counter() - t0) * 1e3 / iters
    bucket.timing = was
    bucket.n_reduces -= (iters + 1) * len(bucket.parts)
    return ms


def grad_bucket_record(nparam, n_reduces, exposed, alone_ms, opt_steps, update_s, n_copied):
    """config.grad_bucket of a --mode ppo line: what the gradient exchange cost per optimiser step."""
    step_ms = update_s * 1e3 / max(1, opt_steps) * 1.0
    mean_exposed = (sum(exposed) / len(exposed)) if exposed else None
    return {"floats": nparam, "bytes": 4 * nparam, "allreduces_issued": n_reduces, "optimiser_steps_per_iteration": opt_steps,
            "layout": "zero-copy: .grad views into one flat fp32 buffer, one all-reduce per network (actor's overlaps the "
                      "critic's backward), one scale kernel",
            "gradients_copied_into_bucket": n_copied,
            "ms_per_allreduce": alone_ms,
            "ms_per_allreduce_what": "both networks' all-reduces + scale, back to back with nothing to overlap (20 repeats)",
            "ms_exposed_per_optimiser_step_mean": mean_exposed,
            "ms_exposed_per_optimiser_step_max": max(exposed) if exposed else None,
            "ms_exposed_what": "end of the critic's backward -> gradients averaged, on the compute stream",
            "ms_per_optimiser_step": step_ms,
            "share_of_optimiser_step": (mean_exposed / step_ms) if (mean_exposed is not None and step_ms > 0) else None}


def slab_backing_of_ranks(mine, dev, world):
    """Every rank's output-slab backing, gathered on all ranks; ranks that disagree are a hard error (the headline
    depends on the placement: a silent hipMalloc fallback on some ranks must not hide in an aggregate)."""
    import torch.distributed as dist
    if world == 1:
        return {"per_rank": [mine], "agree": True}
    got = [None] * world
    dist.all_gather_object(got, mine)
    agree = all(g == got[0] for g in got)
    if not agree:
        if dist.get_rank() == 0:
            print("bench.py: ranks disagree on the output slab backing: %s" % got, file=sys.stderr, flush=True)
        dist.barrier()
        raise SystemExit(5)
    return {"per_rank": got, "agree": True}


def fill_ceiling_gbs(dev):
    """Write-only ceiling of this very box (a measured device ceiling beside the vendor peak)."""
    import torch
    buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    buf.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        buf.fill_(2)
    e1.record()
    torch.cuda.synchronize()
    return 5 * buf.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9


def run_engine_mode(args, rank, world, dev, coll):
    import torch
    from twoarmy_amd.engine import TwoarmyEngine
    vname = args.variant or "v6"
    variant = {"v4": 4, "v6": 6}[vname]
    N, V = args.envs, args.view
    rollout = args.mode == "rollout"
    T = ROLLOUT_T if rollout else 1
    K = args.steps if args.steps is not None else (40 if rollout else 2560)
    W = args.warmup if args.warmup is not None else (10 if rollout else 256)

    eng = TwoarmyEngine(variant, N, V, device=dev, seed=SEED, env_id0=rank * N)
    # the engine's own Philox stream, HBM-resident; launches cycle over a fixed window of it
    n_act = min(W + K, 64) if rollout else W + K
    actions = eng.fill_actions(n_act * T).view(n_act, T, N)
    slab_check_ms, slabs = None, None
    if args.torch_outputs or not rollout:
        out = eng.alloc_outputs(T if rollout else None, matrix_codes=args.matrix_codes, slab=False)
    else:
        out = eng.alloc_outputs(T, matrix_codes=args.matrix_codes)          # engine-owned slab (tw_alloc_outputs)
        if args.slab_check > 1:
            # untimed evidence, NOT a selection: the kernel time into a few more freshly allocated slabs; `out` (the
            # first allocation) is what the timed region uses whatever these say.  The env state is restored.
            state = eng.get_state()
            slabs = [out] + [eng.alloc_outputs(T, matrix_codes=args.matrix_codes) for _ in range(args.slab_check - 1)]
            for s_ in slabs:
                eng.time_rollout(T, s_, actions=actions[0], iters=2)
            slab_check_ms = [eng.time_rollout(T, s_, actions=actions[0], iters=4) for s_ in slabs]
            eng.set_state(*state)
            # the extra slabs stay allocated until the timed region is over: returning gigabytes of physical memory to
            # the driver right before it was followed (once) by a timed region 21 % slower than the same slab's check

    def run(i_begin, n):
        for i in range(i_begin, i_begin + n):
            if rollout:
                eng.rollout(T, out, actions=actions[i % n_act], autoreset=True, policy_idx=True)
            else:
                eng.step(actions[i % n_act, 0], out, autoreset=True, policy_idx=True)

    run(0, W)
    # clocks: at least ~50 ms of launches before the first timed region (a 4 ms region right after set-up ran 3-8 % off)
    ew0, ew1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ew0.record()
    run(0, 8)
    ew1.record()
    torch.cuda.synchronize()
    # every rank must run the same number of regions (each one has barriers): agree on the slowest rank's estimate
    est_ms = max_over_ranks(max(ew0.elapsed_time(ew1) / 8.0, 1e-3), dev, world)
    warm_launches = 8 + int(50.0 / est_ms) + 1
    run(0, warm_launches - 8)
    # R timed regions of exactly K launches each, every one bracketed by barrier + synchronize on both sides; the line
    # reports the MEDIAN region (value, ms_per_step, roofline.kernel_ms) and the spread of all of them
    R = args.regions if args.regions > 0 else auto_regions(K, est_ms)
    wall_s, ev_ms_all = [], []
    for r_ in range(R):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all(dev, world)
        t0 = time.perf_counter()
        e0.record()                 # torch's current stream == the stream handed to tw_rollout (engine._stream)
        run(W + r_ * K, K)
        e1.record()
        sync_all(dev, world)
        wall_s.append(time.perf_counter() - t0)
        ev_ms_all.append(e0.elapsed_time(e1))
    wall_s = max_over_ranks_vec(wall_s, dev, world)
    order = sorted(range(R), key=lambda i: wall_s[i])
    med = order[R // 2]
    dt, ev_ms = wall_s[med], ev_ms_all[med]
    del slabs

    layout = getattr(out["matrix"], "_tw_layout", None)
    mine = "torch caching allocator" if layout is None else ("hipMalloc" if "hipMalloc" in layout else "2 MiB mapped chunks") + \
        (" (FALLBACK: torch could not alias the mapped slab)" if getattr(out["matrix"], "_tw_fallback", False) else "")
    backing = slab_backing_of_ranks(mine, dev, world)
    bpe = algorithmic_bytes_per_env_step(V, T, 289 if args.matrix_codes else 289 * 4)
    k_ms = ev_ms / K                                    # device time per launch, same K launches as `value`
    bytes_per_launch = bpe * N * T
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
    traffic, traffic_src = (None, None) if args.matrix_codes or not rollout else traffic_from_profile(vname, N, T, V)
    fill_gbs = fill_ceiling_gbs(dev)
    if rank != 0:
        return None
    pipelined = rollout and os.environ.get("TW_PIPELINE", "1") != "0"
    res = {
        "metric": "env-steps/sec", "value": world * N * T * K / dt, "unit": "env-steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "MiniGrid-twoarmy-17x17-%s, %d envs/GPU, batched HIP step() only "
                               "(BASELINE configs[1])%s" % (vname, N, " + uint8 code frames (configs[4] storage)"
                                                               if args.matrix_codes else ""),
                   "step": "one tw_rollout launch = %d env-steps x %d envs" % (T, N) if rollout
                           else "one tw_step launch = 1 env-step x %d envs" % N,
                   "env_steps_per_step": N * T, "envs_per_gpu": N, "view": V, "steps_per_launch": T, "autoreset": True,
                   "actions": "Philox(seed=9981) policy indices 0..4 (4->done), resident in HBM",
                   "outputs_per_step": "obs u8[N,V,V,3] + state_matrix %s[N,289] + pos f32[N,2] + reward f32 + term u8 + trunc u8"
                                      % ("u8-code" if args.matrix_codes else "f32"),
                   "parallelism": "env-sharded x%d, no data-path collective" % world,
                   "collective": coll,
                   "output_buffers": "torch caching allocator, two streams" if (args.torch_outputs or not rollout)
                                     else "engine slab (tw_alloc_outputs): %s" % getattr(out["matrix"], "_tw_layout", "?"),
                   "slab_backing": backing,
                   "slab_check_ms": None if slab_check_ms is None else {
                       "what": "untimed set-up: kernel ms per launch into %d freshly allocated engine slabs; no selection, "
                               "the first one is used by the timed region" % len(slab_check_ms),
                       "ms": slab_check_ms,
                       "spread": (max(slab_check_ms) - min(slab_check_ms)) / min(slab_check_ms)}},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "tw_pipe_kernel" if pipelined else "tw_rollout_kernel",
                     "kernel_ms": k_ms, "launches_timed": K, "timed_region": "the same K launches as `value` "
                     "(events on the launch stream): the median of `regions` back-to-back regions of K launches",
                     "regions": R, "gpu_ms_timed_total": sum(ev_ms_all), "clock_warm_launches": warm_launches,
                     "kernel_ms_samples": {"min": min(ev_ms_all) / K, "median": sorted(ev_ms_all)[R // 2] / K,
                                           "max": max(ev_ms_all) / K, "max_over_min": max(ev_ms_all) / min(ev_ms_all)},
                     "region_wall_ms_samples": {"min": min(wall_s) * 1e3, "median": dt * 1e3, "max": max(wall_s) * 1e3},
                     "wall_over_event": dt * 1e3 / ev_ms,
                     "algorithmic_bytes_per_env_step": bpe, "bytes_per_launch": bytes_per_launch,
                     "survey_bytes_per_env_step": 2690, "us_per_env_batch_step": k_ms * 1e3 / T,
                     "launches_per_env_batch_step": 1.0 / T, "measured_fill_ceiling_GBs": fill_gbs,
                     "frac_of_measured_fill_ceiling": achieved / fill_gbs},
    }
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(variant, N, V)
    eng.close()
    return res


def run_ppo_mode(args, rank, world, dev, coll):
    """One step = one PPO iteration of this rank's envs: T-step rollout with the actor in the loop, then the update
    (K epochs x minibatches, both networks, one gradient-bucket all-reduce per optimiser step when world > 1)."""
    import torch
    import torch.distributed as dist
    from twoarmy_amd import dist as twdist
    from twoarmy_amd.engine import TwoarmyEngine
    from twoarmy_amd.soa.ppo_vec import VecPPOTrainer
    vname = args.variant or "v4"
    variant = {"v4": 4, "v6": 6}[vname]
    N, T = args.envs, ROLLOUT_T
    K = args.steps if args.steps is not None else 2
    W = args.warmup if args.warmup is not None else 1
    torch.manual_seed(SEED)
    if args.miopen_benchmark:
        torch.backends.cudnn.benchmark = True
    if args.predictor:
        from twoarmy_amd.soa.agent.PPO_Predictor import ppo_predictor as Agent
    else:
        from twoarmy_amd.soa.agent.PPO import PPO as Agent
    agent = Agent()
    agent.K_epochs = args.k_epochs
    agent.sample_seed = SEED + 7919 * rank
    agent.amp_dtype = torch.bfloat16 if args.amp == "bf16" else None
    agent.to(dev)
    if not args.nchw:
        agent.use_nhwc()
    twdist.broadcast_parameters([agent.actor, agent.critic])
    bucket = None
    if world > 1:
        # zero-copy bucket: gradients are views into one flat buffer, the actor's all-reduce overlaps the critic's backward
        bucket = twdist.GradBucket([list(agent.actor.parameters()), list(agent.critic.parameters())])
        bucket.timing = dev is not None
        agent.grad_sync = bucket
    eng = TwoarmyEngine(variant, N, 17, device=dev, seed=SEED, env_id0=rank * N)
    tr = VecPPOTrainer(agent, eng, rollout_steps=T, minibatch=args.minibatch, frame_codes=args.matrix_codes)
    tr.time_phases = True
    tr.use_graph = not args.predictor and (args.graph == "on" or (args.graph == "auto" and N <= 512))
    roll_s, upd_s, her_n, val_rows = [], [], [], []

    def iteration(timed):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.collect()
        if args.her:
            tr.relabel()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        h = 0 if tr.her is None else int(tr.her["t"].numel())
        tr.update()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tr.carry_over()
        if timed:
            roll_s.append(t1 - t0); upd_s.append(t2 - t1); her_n.append(h); val_rows.append(tr.value_rows)

    for _ in range(W):
        iteration(False)
    if bucket is not None:
        del bucket.events[:]
        bucket.n_reduces = 0
    sync_all(dev, world)
    t0 = time.perf_counter()
    for _ in range(K):
        iteration(True)
    sync_all(dev, world)
    dt = max_over_ranks(time.perf_counter() - t0, dev, world)
    ar_exposed = bucket.exposed_ms() if bucket is not None else []
    n_reduces = bucket.n_reduces if bucket is not None else 0
    ar_alone = allreduce_alone_ms(bucket, dev) if bucket is not None else None
    backing = slab_backing_of_ranks("n/a (ppo mode: torch tensors)", dev, world)
    if rank != 0:
        return None
    S = N * T
    samples = S + sum(her_n) / max(1, len(her_n))
    opt_steps = args.k_epochs * -(-int(samples) // args.minibatch)
    # critic forwards of the target pass as actually issued (V(s') = V(s of the next step) reuse: ~1 per sample instead of
    # the reference's 2, padding rows included), then K epochs x 2 nets x (fwd + 2x bwd) over every sample
    target_rows = sum(val_rows) / max(1, len(val_rows))
    flop_upd = FWD_FLOP_PER_SAMPLE_PER_NET * (target_rows + samples * args.k_epochs * 3 * 2)
    r, u = sum(roll_s) / K, sum(upd_s) / K
    nparam = sum(p.numel() for p in list(agent.actor.parameters()) + list(agent.critic.parameters()) if p.requires_grad)
    return {
        "metric": "env-steps/sec", "value": world * S * K / dt, "unit": "env-steps/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.amp == "fp32" else "bf16", "data": "synthetic",
        "config": {"workload": "MiniGrid-twoarmy-17x17-%s, %d envs/GPU, full PPO%s (BASELINE configs[%s]): %d-step rollouts "
                               "with the actor in the loop + update K=%d, minibatch %d"
                               % (vname, N, " + predictor head" if args.predictor else "",
                                  "4" if args.predictor else ("3" if world > 1 else "2"), T, args.k_epochs, args.minibatch),
                   "step": "one PPO iteration = %d env-steps + %d optimiser steps" % (S, opt_steps),
                   "envs_per_gpu": N, "rollout_s": r, "update_s": u, "rollout_env_steps_per_s_per_gpu": S / r,
                   "update_targets_s": tr.last_update_timing["targets_s"], "update_epoch_s": tr.last_update_timing["epoch_s"],
                   "conv_layout": "nchw (literal nn.Sequential)" if args.nchw else "nhwc + fused upsample/conv1 and conv epilogues",
                   "rollout_as_hip_graph": bool(tr.use_graph), "miopen_benchmark": bool(args.miopen_benchmark),
                   "her_records_per_iteration": sum(her_n) / max(1, len(her_n)),
                   "parallelism": "env-sharded x%d, one gradient-bucket all-reduce per optimiser step" % world,
                   "collective": coll,
                   "slab_backing": backing,
                   "grad_bucket": grad_bucket_record(nparam, n_reduces, ar_exposed, ar_alone, opt_steps * K, u,
                                                     None if bucket is None else bucket.n_copied)},
        "roofline": {"bound": "mfma", "achieved": flop_upd / u / 1e12, "peak": 157.3 if args.amp == "fp32" else 2500.0,
                     "unit": "TFLOP/s", "frac": flop_upd / u / 1e12 / (157.3 if args.amp == "fp32" else 2500.0),
                     "traffic": None, "what": "update phase: actor/critic conv/linear flops (47.5 MFLOP fwd per sample per "
                     "net; target pass = %.2f critic fwd per sample as issued, + K x (fwd + 2x bwd) x 2 nets) / update wall "
                     "time%s" % (target_rows / max(1.0, samples), "; the frozen world model's flops are NOT counted"
                                 if args.predictor else ""),
                     "target_forward_rows": target_rows,
                     "rollout_TFLOPs": S * FWD_FLOP_PER_SAMPLE_PER_NET / r / 1e12},
    }


def run_rehearsal(args, rank, world, dev, coll):
    """Launcher and process-group plumbing without the engine (CPU-only hosts, gloo): the per-rank slab-backing gather,
    and W + K "steps" that are optimiser steps of two small CPU networks through the same two-group gradient bucket
    protocol as the PPO loop (ranks take different numbers of local steps and agree on the maximum)."""
    import torch
    from twoarmy_amd import dist as twdist
    from twoarmy_amd.soa.ppo_vec import agree_on_steps
    K = args.steps if args.steps is not None else 2
    W = args.warmup if args.warmup is not None else 1
    mine = "rehearsal"
    if os.environ.get("TW_REHEARSE_ODD_RANK", "") == str(rank):
        mine = "hipMalloc"                          # test hook: one rank reports another backing -> the run must abort
    backing = slab_backing_of_ranks(mine, None, world)
    torch.manual_seed(SEED + rank)
    nets = [torch.nn.Linear(8, 4), torch.nn.Linear(8, 1)]
    twdist.broadcast_parameters(nets)
    bucket = twdist.GradBucket([list(n.parameters()) for n in nets])
    opts = [torch.optim.Adam(n.parameters(), 1e-3) for n in nets]
    n_steps = agree_on_steps(W + K + (rank % 3), "cpu") - W          # ranks ask for different counts; all take the max

    def step():
        x = torch.randn(16, 8)
        bucket.zero()
        nets[0](x).square().mean().backward()
        bucket.reduce_async(0)
        nets[1](x).square().mean().backward()
        bucket.reduce_async(1)
        bucket.finish()
        for o in opts:
            o.step()
    for _ in range(W):
        step()
    sync_all(None, world)
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    sync_all(None, world)
    dt = max_over_ranks(time.perf_counter() - t0, None, world)
    chk = torch.cat([p.detach().reshape(-1) for n in nets for p in n.parameters()]).double().sum().reshape(1)
    lo, hi = chk.clone(), chk.clone()
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank != 0:
        return None
    return {"metric": "env-steps/sec", "value": 0.0, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": dt / max(1, K) * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "rehearsal", "config": {"workload": "launcher rehearsal: no engine, nothing measured",
                                            "collective": coll, "slab_backing": backing,
                                            "grad_bucket": {"allreduces_issued": bucket.n_reduces,
                                                            "optimiser_steps": W + n_steps,
                                                            "gradients_copied_into_bucket": bucket.n_copied,
                                                            "replicas_identical": bool(lo.item() == hi.item())}}}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = build_parser().parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)             # parent: nothing below runs here
    import torch.distributed as dist
    rank, world, dev, coll = init_ranks(args)
    if args.rehearse:
        res = run_rehearsal(args, rank, world, dev, coll)
    elif args.mode == "ppo":
        res = run_ppo_mode(args, rank, world, dev, coll)
    else:
        res = run_engine_mode(args, rank, world, dev, coll)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
