"""Data-parallel plumbing added by this build (the reference is single-process: no paddle.distributed anywhere,
SURVEY.md section 2): one process per GPU, every rank runs the same per-GPU batch shape on its own shard of clouds
(clouds are independent units; BatchNorm statistics stay per-GPU, as in plain DDP), and ONE all-reduce of a flat
fp32 gradient bucket per step over RCCL/xGMI (``torch.distributed`` backend "nccl" on ROCm; "gloo" in CPU tests).

``FlatParams`` re-homes every trainable parameter (and its .grad) as a view into one contiguous buffer, so the
collective is a single call on 1.47 M floats (5.9 MB) and the optimiser is one kernel (papc_adam_step_f32).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run sets them).
    Returns (rank, world, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("PAPC_FORCE_DIST") == "1"   # 1-rank group: exercises the collective path on a 1-GPU box
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # PAPC_DIST_BACKEND=gloo: (tests) several ranks on ONE GPU -- RCCL refuses two ranks on the same device, gloo does not care
            backend = os.environ.get("PAPC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)   # binds the communicator to this rank's GPU up front
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


class FlatParams:
    """All trainable parameters of ``module`` as views into ``self.data``; their grads as views into ``self.grad``."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, torch.float32
        # every parameter starts on a 16-byte boundary (the kernels' vector loads of weights and constants want it: a parameter whose
        # numel is not a multiple of 4 must not misalign everything laid out behind it); the gaps stay zero in data, grad and moments
        self._offsets = []
        n = 0
        for p in self.params:
            self._offsets.append(n)
            n += p.numel() + ((-p.numel()) % 4)
        self.numel = n
        self.data = torch.zeros(n, device=dev, dtype=dt)
        self.grad = torch.zeros(n, device=dev, dtype=dt)
        with torch.no_grad():
            for p, off in zip(self.params, self._offsets):
                k = p.numel()
                self.data[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.data[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                # opt in to the kernels' in-place gradient accumulation (mlp.grad_targets_of): this container owns the gradient
                # reduction itself (allreduce_grads); do NOT wrap such a module in torch's DistributedDataParallel or hang
                # post-accumulate-grad hooks on its parameters
                p._papc_inplace_grad = True

    def offset_of(self, module):
        """Element offset in the flat buffers of ``module``'s first trainable parameter (parameters are laid out in
        ``module.parameters()`` order, so everything registered after it follows contiguously)."""
        first = next(p for p in module.parameters() if p.requires_grad)
        for p, off in zip(self.params, self._offsets):
            if p is first:
                return off
        raise ValueError("module is not part of this FlatParams")

    def zero_grad(self):
        if self.grad.is_cuda:
            from . import _lib
            _lib.check(_lib.load().papc_fill_f32(self.grad.data_ptr(), self.grad.numel(), 0.0, _lib.stream_ptr()), "papc_fill_f32")
        else:
            self.grad.zero_()
        for p in self.params:   # autograd accumulates in place into the existing views
            if p.grad is None or p.grad.data_ptr() < self.grad.data_ptr():
                raise RuntimeError("a parameter lost its flat .grad view (someone called zero_grad(set_to_none=True))")

    def broadcast(self, src=0):
        """Make every rank start from rank ``src``'s weights."""
        if dist.is_initialized():
            if self.data.is_cuda and dist.get_backend() == "gloo":
                host = self.data.cpu()
                dist.broadcast(host, src=src)
                self.data.copy_(host)
                return
            dist.broadcast(self.data, src=src)

    def allreduce_grads(self, lo=0, hi=None, async_op=False):
        """Sum the flat gradient bucket (or its slice [lo, hi)) over ranks -- ONE collective per call.  Returns the scale that
        turns the sum into the mean (1/world), folded into the optimiser kernel instead of a separate pass; with
        ``async_op=True`` returns (scale, work handle or None).

        Two-bucket use (bench.py, N > 1): the backward produces the gradients of the FC head and of SA3 first -- 1 388 816 of the
        1 469 520 floats of PointNet2_SSG_Clas, registered last, so they are the TAIL of the flat buffer -- and those are
        all-reduced while SA2 / SA1 are still in their backward; only the 80 704-float head of the buffer is reduced after the
        last kernel."""
        if dist.is_initialized():
            buf = self.grad if (lo == 0 and hi is None) else self.grad[lo:hi]
            scale = 1.0 / dist.get_world_size()
            if buf.is_cuda and dist.get_backend() == "gloo":
                # (tests: ranks sharing one GPU) gloo's device path is not part of every ROCm build: stage through the host, in stream order
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                buf.copy_(host)
                return (scale, None) if async_op else scale
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)
            return (scale, work) if async_op else scale
        return (1.0, None) if async_op else 1.0


class FlatAdam:
    """Adam(lr, weight_decay as L2 on the gradient) -- PAPC/train.py:62-65 -- as one HIP kernel over the flat buffer."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3):
        self.flat, self.lr, self.betas, self.eps, self.wd = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.data)
        self.v = torch.zeros_like(flat.data)
        self.t = 0
        self.t_dev = torch.zeros(2, dtype=torch.int64, device=flat.data.device)   # [0] the step count for the graph-capturable form (tick / step_dev), [1] the self-ticking launch's ticket
        # ONE step count, two homes: the host integer (step) and the device word (tick / step_dev, so that the update can sit inside a captured
        # hipGraph).  Whichever path ran last owns the count; switching paths carries it over (device -> host costs one synchronising read, host ->
        # device one fill ahead of the launch -- never inside a stream capture, where a baked-in fill would reset the count at every replay)
        self._owner = "host"

    def _to_dev(self):
        if self._owner == "host":
            if self.t_dev.is_cuda and torch.cuda.is_current_stream_capturing():
                if self.t != 0:
                    raise RuntimeError("FlatAdam: the step count lives on the host (step() ran last); run one eager tick() / step_dev() before capturing "
                                       "the device-counted update into a graph")
            elif self.t != 0:                       # (a fresh optimiser: both counts are zero already)
                self.t_dev[0].fill_(self.t)
            self._owner = "dev"

    def step_count(self):
        """The number of updates applied so far, whichever path applied them (reads the device word when it owns the count: synchronises)."""
        return int(self.t_dev[0].item()) if self._owner == "dev" else self.t

    def state_dict(self):
        return {"step": self.step_count(), "exp_avg": self.m.clone(), "exp_avg_sq": self.v.clone()}

    def load_state_dict(self, sd):
        self.m.copy_(sd["exp_avg"])
        self.v.copy_(sd["exp_avg_sq"])
        self.t = int(sd["step"])
        self.t_dev[0].fill_(self.t)
        self.t_dev[1].fill_(0)
        self._owner = "host"

    def tick(self):
        """Advance the DEVICE step count (one-thread launch).  Enqueue it anywhere earlier in the step than :meth:`step_dev` -- e.g. on the
        sampling branch, off the critical path -- and both can be captured into a hipGraph."""
        from . import _lib
        self._to_dev()
        _lib.check(_lib.load().papc_adam_tick(self.t_dev.data_ptr(), _lib.stream_ptr()), "papc_adam_tick")

    def step_dev(self, grad_scale=1.0, zero_grad=False, self_tick=False):
        """:meth:`step` with the step count read from device memory (advanced by :meth:`tick`): no host scalar changes between steps, so
        the launch can sit inside a captured hipGraph behind the last backward kernel (an eager launch behind a graph replay starts
        8-20 us late).  ``self.t`` (host) is not advanced on this path: :meth:`step_count` reads whichever count is current, and a later :meth:`step`
        picks the device count up.  ``self_tick=True``: no :meth:`tick` launch, the
        kernel advances the count itself (its last-finishing block) -- for a step with no side branch to put the tick on."""
        from . import _lib
        f = self.flat
        self._to_dev()
        if self_tick and f.numel > 64 * 256:
            # the self-ticking form costs one same-address atomic per 256-parameter block: measured +33 us at 3 200 blocks (PointNet-Basic), -4 us
            # at 3 (PillarFeatureNet).  Large buckets take the one-thread tick launch
            self.tick()
            self_tick = False
        _lib.check(_lib.load().papc_adam_step_dev_f32(f.data.data_ptr(), f.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), f.numel, self.lr,
                                                      self.betas[0], self.betas[1], self.eps, self.wd, self.t_dev.data_ptr(), float(grad_scale),
                                                      int(bool(zero_grad)) | (2 if self_tick else 0), _lib.stream_ptr()), "papc_adam_step_dev_f32")

    def step(self, grad_scale=1.0, zero_grad=False):
        """``zero_grad=True``: the kernel also clears the flat gradient bucket behind the update (papc_adam_step_zero_f32), so a
        training loop whose every backward is followed by a step needs no ``FlatParams.zero_grad()`` launch."""
        from . import _lib
        if self._owner == "dev":
            self.t = int(self.t_dev[0].item())     # the device-counted path ran last (synchronises; only on a change of path)
            self._owner = "host"
        self.t += 1
        f = self.flat
        fn = _lib.load().papc_adam_step_zero_f32 if zero_grad else _lib.load().papc_adam_step_f32
        _lib.check(fn(f.data.data_ptr(), f.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                      f.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                      float(grad_scale), _lib.stream_ptr()), "papc_adam_step_f32")
