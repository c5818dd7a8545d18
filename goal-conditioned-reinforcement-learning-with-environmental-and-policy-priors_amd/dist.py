"""Multi-GPU glue: env batches shard across ranks (disjoint env-id ranges, no data-path exchange); the only
collective is the all-reduce of the flattened fp32 gradient bucket of actor + critic per optimiser step
(2 515 206 floats ~ 10 MB: the actor's half goes out while the critic's backward runs), averaged over ranks.  backend "nccl" is RCCL over xGMI on ROCm; the same code
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
    """Zero-copy gradient bucket: ONE persistent flat fp32 buffer, every parameter's `.grad` is a VIEW into it
    (as_strided with the parameter's own strides, so channels-last conv weights get channels-last gradients), so an
    optimiser step is: `zero()` (one fill kernel), the backward passes accumulate straight into the bucket,
    `reduce_async(g)` per parameter group (one all-reduce each, on the backend's own stream: the actor's collective
    runs while the critic's backward is still computing), `finish()` (wait + one scale kernel).  No per-parameter
    copy kernel anywhere.

    groups: a list of parameter lists -- one all-reduce per group, laid out back to back in the flat buffer -- or a flat
    parameter list (one group).  PPO: [actor parameters, critic parameters] = 1 258 629 + 1 256 577 floats.

    Legacy use (`bucket()` after an optimiser's own `zero_grad()` detached the views): gradients that are not views of
    the bucket any more are copied in and re-attached, then everything is reduced -- correct, just not copy-free."""

    def __init__(self, groups):
        groups = list(groups)
        if groups and isinstance(groups[0], torch.nn.Parameter):
            groups = [groups]
        self.groups = [[p for p in g if p.requires_grad] for g in groups]
        self.params = [p for g in self.groups for p in g]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.views, self.parts = [], []
        off = 0
        for g in self.groups:
            lo = off
            for p in g:
                n = p.numel()
                dense = p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)
                v = self.flat[off:off + n].as_strided(p.size(), p.stride()) if dense and p.dim() > 0 \
                    else self.flat[off:off + n].view(p.shape)
                self.views.append(v)
                off += n
            self.parts.append(self.flat[lo:off])
        self._work = []
        self.n_reduces = 0                      # all-reduce calls issued (tests / bench bookkeeping)
        self.n_copied = 0                       # gradients that had to be copied in (0 on the zero-copy path)
        self.timing = False                     # record (last reduce_async -> finish) event pairs on the current stream
        self.events = []
        self._ev0 = None
        self.attach()

    # ---- zero-copy protocol
    def attach(self):
        """Point every parameter's .grad at its slice of the bucket (keeps the current contents of the bucket)."""
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v

    def zero(self):
        """Replaces optimizer.zero_grad(): one fill of the flat buffer, gradients stay views into it."""
        self.attach()
        self.flat.zero_()

    def active(self):
        return dist.is_initialized() and dist.get_world_size() > 1

    def _adopt(self):
        """Gradients that are no views of the bucket (someone called zero_grad(set_to_none=True) and backward made
        fresh tensors, or left None): copy / zero them into place and re-attach."""
        for p, v in zip(self.params, self.views):
            if p.grad is v:
                continue
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
                self.n_copied += 1
            p.grad = v

    def reduce_async(self, group=None):
        """Start the all-reduce of one parameter group (None: all of them, one collective per group).  Returns at once;
        the collective is ordered behind everything already enqueued on the current stream."""
        if not self.active():
            return
        self._adopt()
        if self.timing and self.flat.is_cuda:
            # (re)recorded by every call: the pair measures from the LAST reduce_async of a step, i.e. from the end of
            # the last backward, to the end of finish() -- the part of the collectives no backward overlaps with
            self._ev0 = torch.cuda.Event(enable_timing=True)
            self._ev0.record()
        for k in (range(len(self.parts)) if group is None else [group]):
            self._work.append(dist.all_reduce(self.parts[k], op=dist.ReduceOp.SUM, async_op=True))
            self.n_reduces += 1

    def finish(self):
        """Wait for the collectives started by reduce_async and turn the sums into means (one kernel)."""
        if not self._work:
            return
        for w in self._work:
            w.wait()
        self._work = []
        self.flat.mul_(1.0 / dist.get_world_size())
        if self._ev0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.events.append((self._ev0, e1))
            self._ev0 = None

    def exposed_ms(self):
        """ms between the last reduce_async of a step and the end of finish() on the compute stream, per step
        (what the collectives add to an optimiser step beyond the backward they overlap with)."""
        return [a.elapsed_time(b) for a, b in self.events]

    def __call__(self, _params=None):
        self.reduce_async(None)
        self.finish()


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
