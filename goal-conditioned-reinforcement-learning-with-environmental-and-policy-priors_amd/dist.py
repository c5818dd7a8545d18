"""Multi-GPU glue: env batches shard across ranks (disjoint env-id ranges, no data-path exchange); the only
collective is ONE all-reduce of the flattened fp32 gradient bucket of actor + critic per optimiser step
(2 515 206 floats ~ 10 MB), averaged over ranks.  backend "nccl" is RCCL over xGMI on ROCm; the same code
runs on "gloo" for the CPU tests."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """One process per GPU, launched by torch.distributed.run; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # RCCL needs one GPU per rank; fewer GPUs than ranks (rehearsals, CPU tests) -> gloo
            backend = "nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= world else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local_rank


def shard_range(total_envs, rank, world):
    """Contiguous env-id range [lo, hi) of this rank."""
    per = total_envs // world
    extra = total_envs % world
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


class GradBucket:
    """Flattens the gradients of `params` into one persistent fp32 buffer and all-reduces it once."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.numel = n

    def __call__(self, _params=None):
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n


def broadcast_parameters(modules, src=0):
    """Identical replicas at start (ranks also use identical seeds; this makes it explicit)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            buf = t.data.contiguous()               # channels-last conv weights (use_nhwc) are not "contiguous" for c10d
            dist.broadcast(buf, src)
            if buf.data_ptr() != t.data.data_ptr():
                t.data.copy_(buf)


def global_adv_norm_(adv, eps=1e-8):
    """adv <- (adv - mean) / (std + eps) with mean / unbiased std over the samples of ALL ranks (the reference's
    commented normalisation, PPO.py:115, applied to the whole sharded batch): one all-reduce of (count, sum, sum of
    squares) in float64 -- SURVEY.md section 8e's optional second collective.  Works on any device / backend."""
    a = adv.double()
    st = torch.stack([torch.tensor(float(a.numel()), dtype=torch.float64, device=adv.device), a.sum(), (a * a).sum()])
    on_cpu = dist.get_backend() != "nccl"
    red = st.cpu() if on_cpu else st
    dist.all_reduce(red, op=dist.ReduceOp.SUM)
    n, s1, s2 = (float(x) for x in red)
    mean = s1 / n
    var = max(0.0, (s2 - n * mean * mean) / max(1.0, n - 1.0))
    adv.sub_(mean).div_(var ** 0.5 + eps)
    return adv
