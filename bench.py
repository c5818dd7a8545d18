#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of the batched HIP Twoarmy step engine.

Workload = BASELINE.json configs[1]: MiniGrid-twoarmy-17x17-v6, 4096 envs on one MI355X, batched
HIP step() only (obs + state matrix + reward/done written every step), auto-reset on, view 17,
actions = Philox(seed 9981) policy indices resident in HBM.

A "step" = one environment step of all envs of one GPU.  Steps are issued as rollouts of
ROLLOUT_T steps per launch (tw_rollout, include/twoarmy.h) plus one remainder launch, so exactly
K steps are timed.  With --gpus N (torch.distributed.run, one rank per GPU) every rank steps its
own 4096 envs (env ids rank*4096..): the path shards with no data-path collective, scaling = weak.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--envs 4096] [--variant v6] [--view 17]
                  [--no-cpu-baseline] [--mode rollout|step]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROLLOUT_T = 128
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
SEED = 9981                    # reference default, soa/train_ppo.py:25


def algorithmic_bytes_per_env_step(view, rollout_t, matrix_bytes=289 * 4):
    """Bytes that must cross HBM per env-step for the design built:
    action 4 + obs V*V*3 + state matrix 289*4 + pos 8 + reward 4 + terminated 1 + truncated 1,
    plus the per-launch load/store of the LDS-resident planes + record amortised over T."""
    per_step = 4 + view * view * 3 + matrix_bytes + 8 + 4 + 1 + 1
    per_launch = 2 * (289 + 289 + 48 * 4)
    return per_step + per_launch / float(rollout_t)


def traffic_from_profile(variant, n_envs, rollout_t, view):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile of the SAME configuration
    (a bench run cannot profile itself); None when the configuration differs from the profiled one."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if not (variant == "v6" and n_envs == 4096 and rollout_t == 128 and view == 17 and os.path.exists(path)):
        return None
    with open(path) as f:
        return json.load(f)["traffic_bytes_per_launch"]


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2560)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--variant", default="v6")
    ap.add_argument("--view", type=int, default=17)
    ap.add_argument("--mode", default="rollout", choices=["rollout", "step"],
                    help="rollout: ROLLOUT_T steps per launch (headline); step: one tw_step launch per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--placement-candidates", type=int, default=6,
                    help="output buffer sets probed at set-up (1 = take the first allocation)")
    ap.add_argument("--matrix-codes", action="store_true",
                    help="BASELINE configs[4] variant: state matrix as uint8 codes (TW_F_MATRIX_CODE), not the headline")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from twoarmy_amd.engine import TwoarmyEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ngpu = torch.cuda.device_count()
        torch.cuda.set_device(local_rank % ngpu)
        if ngpu >= world:                   # one rank per GPU: RCCL over xGMI
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:                               # rehearsal of the N>1 path on a box with fewer GPUs than ranks
            dist.init_process_group(backend="gloo")
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    variant = {"v4": 4, "v6": 6}[args.variant]
    N, K, W, V = args.envs, args.steps, args.warmup, args.view
    T = ROLLOUT_T if args.mode == "rollout" else 1

    eng = TwoarmyEngine(variant, N, V, device=dev, seed=SEED, env_id0=rank * N)
    actions = eng.fill_actions(W + K)                      # the engine's own Philox stream, HBM-resident
    placement_ms = None
    if args.mode == "rollout" and args.placement_candidates > 1:
        # untimed set-up: pick the HBM placement of the output streams (engine.alloc_outputs_tuned explains why)
        out, placement_ms = eng.alloc_outputs_tuned(T, candidates=args.placement_candidates, matrix_codes=args.matrix_codes)
    else:
        out = eng.alloc_outputs(T, matrix_codes=args.matrix_codes)

    def run(t_begin, n_steps):
        t = t_begin
        end = t_begin + n_steps
        while t < end:
            tt = min(T, end - t)
            sub = out if tt == T else {k: v[:tt] for k, v in out.items()}
            eng.rollout(tt, sub, actions=actions[t:t + tt], autoreset=True, policy_idx=True)
            t += tt

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, W)
    barrier()
    t0 = time.perf_counter()
    run(W, K)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # --- roofline leg: the same kernel timed with HIP events on its own launch stream
    bpe = algorithmic_bytes_per_env_step(V, T, 289 if args.matrix_codes else 289 * 4)
    iters = max(3, min(50, K // T))
    k_ms = eng.time_rollout(T, out, actions=actions[:T], autoreset=True, iters=iters)
    bytes_per_launch = bpe * N * T
    achieved = bytes_per_launch / (k_ms * 1e-3) / 1e9
    traffic = None if args.matrix_codes else traffic_from_profile(args.variant, N, T, V)
    torch.cuda.synchronize()
    # write-only ceiling of this very box (report a measured device ceiling beside the vendor peak)
    buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    buf.fill_(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        buf.fill_(2)
    e1.record()
    torch.cuda.synchronize()
    fill_gbs = 5 * buf.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del buf

    if rank == 0:
        res = {
            "metric": "env-steps/sec", "value": world * N * K / dt, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "MiniGrid-twoarmy-17x17-%s, %d envs/GPU, batched HIP step() only "
                                   "(BASELINE configs[1])%s" % (args.variant, N, " + uint8 code frames (configs[4] storage)"
                                                                   if args.matrix_codes else ""),
                       "envs_per_gpu": N, "view": V, "steps_per_launch": T, "autoreset": True,
                       "actions": "Philox(seed=9981) policy indices 0..4 (4->done), resident in HBM",
                       "outputs_per_step": "obs u8[N,V,V,3] + state_matrix %s[N,289] + pos f32[N,2] + reward f32 + term u8 + trunc u8"
                                          % ("u8-code" if args.matrix_codes else "f32"),
                       "parallelism": "env-sharded x%d, no collective" % world,
                       "output_placement": None if placement_ms is None else {
                           "what": "untimed set-up: %d output buffer sets allocated, the fastest kept (HBM placement of the "
                                   "two output streams moves the store bandwidth by up to 25 %%)" % len(placement_ms),
                           "probe_kernel_ms": placement_ms}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950-corrected)"
                                           if traffic is not None else None,
                         "kernel": "tw_pipe_kernel (+ flag-gated tw_rollout_kernel fallback launch)" if T >= 8 and os.environ.get("TW_PIPELINE", "1") != "0" else "tw_rollout_kernel", "kernel_ms": k_ms, "launches_timed": iters,
                         "algorithmic_bytes_per_env_step": bpe, "bytes_per_launch": bytes_per_launch,
                         "survey_bytes_per_env_step": 2690, "us_per_env_batch_step": k_ms * 1e3 / T,
                         "launches_per_env_batch_step": 2.0 / T, "measured_fill_ceiling_GBs": fill_gbs,
                         "frac_of_measured_fill_ceiling": achieved / fill_gbs},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(variant, N, V)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
