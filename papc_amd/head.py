"""Classifier head of the PointNet++ classifiers as fused HIP launches (csrc/head.hip).

Reference: PAPC/models/classify/pointnet2/pointnet2.py:17-23, :37-39 (SSG) / :51-57, :71-73 (MSG)
    x = drop1(relu(bn1(fc1(x)))); x = drop2(relu(bn2(fc2(x)))); x = fc3(x)
One launch per layer forward, one per layer backward (+ one for the input gradient); the library-op chain it replaces is ~60
launch-latency-sized kernels per training step.  The ``nn.Linear`` / ``nn.BatchNorm1d`` / ``nn.Dropout`` modules stay the parameter
holders (state_dict unchanged).  ``model.eval()`` runs the same launches with the norms in eval mode (has_bn = 3: running statistics, nothing
updated, no dropout; batches beyond the kernel's 256 rows in chunks, rows being independent there); only TRAIN-mode shapes outside the kernel's
limits (more than 256 rows, widths that are no multiple of 4) fall back to the modules.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .mlp import grad_targets_of

MAX_ROWS = 256
# PAPC_HEAD_CHAIN=1: the layers of a head as phases of ONE launch each way (papc_head_chain_fwd_f32 / _bwd_f32, grid barrier between the
# phases).  Built to remove six dependent launches (>= 4.7 us each inside a replayed graph) and measured SLOWER on MI355X: a phase hand-over
# is an agent-scope atomic round trip + a poll + a coherent read past the L2, ~2 us each, i.e. more than the launch it replaces (forward
# 49.6 -> 56.5 us, backward 42.5 -> 52.0 us per step; DESIGN "measured and not kept").  Default: one launch per layer.  The chain stays
# selectable and tested (bit-identical results).
CHAIN = os.environ.get("PAPC_HEAD_CHAIN", "0") == "1"
# PAPC_HEAD_MERGE=1: the hand-over-free merges only -- the last layer's single workgroup computes the loss from the logits it still holds in LDS
# (no softmax launch) and the backward's two independent first jobs share one launch: 6 launches instead of 8.  Measured on MI355X: the head's
# span shrinks by 5 us under rocprofv3, the step does not (1.473 / 1.473 ms with a fixed plan, 1.588 / 1.579 ms beside the sampling branch):
# opt-in as well.
MERGE = os.environ.get("PAPC_HEAD_MERGE", "0") == "1"
c_p, c_i, c_f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float


class HeadFcLayer(ctypes.Structure):
    """papc_head_fc_layer"""
    _fields_ = [("x", c_p), ("w", c_p), ("bias", c_p), ("gamma", c_p), ("beta", c_p), ("Cin", c_i), ("Cout", c_i), ("has_bn", c_i), ("eps", c_f),
                ("momentum", c_f), ("running_mean", c_p), ("running_var", c_p), ("num_batches_tracked", c_p), ("drop_p", c_f), ("layer_tag", c_i),
                ("y", c_p), ("mean", c_p), ("invstd", c_p), ("keep", c_p), ("out", c_p)]


class HeadBwdJob(ctypes.Structure):
    """papc_head_bwd_job"""
    _fields_ = [("gnext", c_p), ("wnext", c_p), ("Cn", c_i), ("out", c_p), ("y", c_p), ("mean", c_p), ("invstd", c_p), ("gamma", c_p), ("drop_p", c_f),
                ("has_bn", c_i), ("x", c_p), ("Cin", c_i), ("Cout", c_i), ("dy", c_p), ("dw", c_p), ("db", c_p), ("dgamma", c_p), ("dbeta", c_p),
                ("accumulate", c_i), ("phase", c_i)]


def _p(t):
    return ptr(t) if t is not None else None


def _bwd_job(gnext, wnext, Cn, out, y, mean, invstd, gamma, drop_p, has_bn, x, Cin, Cout, dy, dw, db, dgamma, dbeta, acc, phase):
    return HeadBwdJob(_p(gnext), _p(wnext), Cn, _p(out), _p(y), _p(mean), _p(invstd), _p(gamma), float(drop_p), has_bn, _p(x), Cin, Cout, _p(dy), _p(dw),
                      _p(db), _p(dgamma), _p(dbeta), acc, phase)


def _run_bwd_chain(jobs, B, spec, dev, st):
    arr = (HeadBwdJob * len(jobs))(*jobs)
    check(_lib.load().papc_head_chain_bwd_f32(ctypes.addressof(arr), len(jobs), B, ptr(spec.sync(dev)), st), "papc_head_chain_bwd_f32")


class HeadSpec:
    """Per-model state of the fused head: dropout RNG state on the device, export of the dropout masks for tests."""

    def __init__(self):
        self.rng_state = None          # int64 [2] = (seed, counter), device
        self.export_masks = False
        self.masks = None              # (keep1, keep2) uint8 tensors of the last forward when export_masks
        self.grad_targets = None
        self.training = True           # set per forward by the wrappers: False = model.eval() (norms on running statistics, no dropout)
        self.chain = CHAIN             # the layers as phases of one launch each way (False: one launch per layer)
        self.merge = MERGE             # (chain False) the hand-over-free merges: loss with the last layer, the backward's two independent first jobs
        self._sync = {}                # per stream: the two barrier words of the chain launches (zero between launches)

    def sync(self, device):
        """two uint32 words per stream for papc_head_chain_*: zero at first use, the kernels leave them zero"""
        key = (str(device), torch.cuda.current_stream().cuda_stream)
        t = self._sync.get(key)
        if t is None:
            t = torch.zeros(2, dtype=torch.int32, device=device)
            if not _lib._capturing():
                self._sync[key] = t
        return t

    def state(self, device):
        if self.rng_state is None or self.rng_state.device != device:
            # every data-parallel rank seeds torch identically (same initial weights): mix the rank into the dropout stream so
            # the ranks draw different masks
            seed = torch.initial_seed()
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                seed ^= (torch.distributed.get_rank() * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
            self.rng_state = torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        return self.rng_state


def usable(x, fc1, fc2, fc3, training):
    rows_ok = (2 <= x.shape[0] <= MAX_ROWS) if training else x.shape[0] >= 1   # (train: one row -- BatchNorm1d itself refuses; eval: chunks of MAX_ROWS)
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and rows_ok
            and fc1.in_features % 4 == 0 and fc1.out_features % 4 == 0 and fc2.out_features % 4 == 0
            and fc3.out_features % 4 == 0 and fc1.bias is not None and fc2.bias is not None and fc3.bias is not None)


class _Head(torch.autograd.Function):
    """forward(ctx, spec, bns, drops, labels, x0, params...): labels None -> logits; labels given -> (loss, logits) with the mean softmax
    cross-entropy computed by the same launch (chain path) and its gradient kept for the backward."""

    @staticmethod
    def forward(ctx, spec, bns, drops, labels, x0, w1, b1, g1, be1, w2, b2, g2, be2, w3, b3):
        lib = _lib.load()
        x0 = x0.contiguous()
        ev = not spec.training                      # model.eval(): the norms use their running statistics (has_bn = 3), no dropout, nothing updated
        mode = 3 if ev else 1
        if ev:
            drops = (0.0, 0.0)
        B = x0.shape[0]
        dev = x0.device
        st = stream_ptr()
        rng = spec.state(dev)
        saved = []
        x = x0
        keeps = []
        layers = []
        for li, (w, b, g, be, bn, p) in enumerate(((w1, b1, g1, be1, bns[0], drops[0]), (w2, b2, g2, be2, bns[1], drops[1]))):
            cout, cin = w.shape
            y = torch.empty(B, cout, device=dev, dtype=torch.float32)
            out = torch.empty_like(y)
            mean = torch.empty(cout, device=dev, dtype=torch.float32)
            invstd = torch.empty_like(mean)
            keep = torch.empty(B, cout, device=dev, dtype=torch.uint8) if spec.export_masks else None
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            track = bn.track_running_stats and bn.running_mean is not None
            if ev and not track:
                raise _lib.PapcError("eval-mode head: BatchNorm1d without running statistics (track_running_stats=False) has nothing to normalise with")
            rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
            nbt = bn.num_batches_tracked if track and bn.num_batches_tracked is not None and not ev else None
            if spec.chain:
                layers.append(HeadFcLayer(ptr(x), ptr(w), ptr(b), ptr(g), ptr(be), cin, cout, mode, float(bn.eps), mom, _p(rm), _p(rv), _p(nbt), float(p), li + 1,
                                          ptr(y), ptr(mean), ptr(invstd), _p(keep), ptr(out)))
            else:
                check(lib.papc_head_fc_f32(ptr(x), ptr(w), ptr(b), ptr(g), ptr(be), B, cin, cout, mode, float(bn.eps), mom, ptr(rm) if track else 0,
                                           ptr(rv) if track else 0, ptr(nbt) if nbt is not None else 0,
                                           float(p), ptr(rng), li + 1, 0, ptr(y), ptr(mean), ptr(invstd), ptr(keep), ptr(out), st),
                      "papc_head_fc_f32")
            saved += [y, mean, invstd, out]
            keeps.append(keep)
            x = out
        cout, cin = w3.shape
        logits = torch.empty(B, cout, device=dev, dtype=torch.float32)
        loss = dz = None
        if labels is not None:
            labels = labels.contiguous().long().reshape(-1)
            loss = torch.empty((), device=dev, dtype=torch.float32)
            dz = torch.empty_like(logits)
        if spec.chain:
            layers.append(HeadFcLayer(ptr(x), ptr(w3), ptr(b3), None, None, cin, cout, 0, 0.0, 0.0, None, None, None, 0.0, 3, None, None, None, None, ptr(logits)))
            arr = (HeadFcLayer * 3)(*layers)
            check(lib.papc_head_chain_fwd_f32(ctypes.addressof(arr), 3, B, ptr(rng), None if ev else ptr(rng), _p(labels), _p(loss), _p(dz), ptr(spec.sync(dev)), st),
                  "papc_head_chain_fwd_f32")
        elif labels is not None and spec.merge and cout <= 32:
            # one workgroup makes all the logits: it goes on to the loss from its LDS tile (a one-layer chain launch: no barrier, no hand-over)
            one = (HeadFcLayer * 1)(HeadFcLayer(ptr(x), ptr(w3), ptr(b3), None, None, cin, cout, 0, 0.0, 0.0, None, None, None, 0.0, 3, None, None, None, None, ptr(logits)))
            check(lib.papc_head_chain_fwd_f32(ctypes.addressof(one), 1, B, ptr(rng), None if ev else ptr(rng), ptr(labels), ptr(loss), ptr(dz), ptr(spec.sync(dev)), st),
                  "papc_head_chain_fwd_f32")
        else:
            check(lib.papc_head_fc_f32(ptr(x), ptr(w3), ptr(b3), 0, 0, B, cin, cout, 0, 0.0, 0.0, 0, 0, 0, 0.0, 0, 3, 0 if ev else ptr(rng),   # (eval draws no mask: the counter stays)
                                       0, 0, 0, 0, ptr(logits), st), "papc_head_fc_f32")
            if labels is not None:
                check(lib.papc_softmax_xent_f32(ptr(logits), ptr(labels), B, cout, ptr(loss), ptr(dz), st), "papc_softmax_xent_f32")
        spec.masks = tuple(keeps) if spec.export_masks else None
        ctx.spec = spec
        ctx.drops = drops
        ctx.mode = mode
        ctx.with_loss = labels is not None
        ctx.set_materialize_grads(False)             # (the logits that ride along with the loss carry no gradient: no zeros tensor made for them)
        ctx.save_for_backward(x0, w1, g1, w2, g2, w3, *saved, *((dz,) if dz is not None else ()))
        if labels is not None:
            ctx.mark_non_differentiable(logits)
            return loss, logits
        return logits

    @staticmethod
    def backward(ctx, *gs):
        lib = _lib.load()
        if ctx.with_loss:
            x0, w1, g1, w2, g2, w3, y1, mean1, invstd1, x1, y2, mean2, invstd2, x2, dz = ctx.saved_tensors
            glogits = _scale_dz(dz, gs[0])             # upstream gradient of the scalar loss (unit_gradient(): dz itself, no launch)
        else:
            x0, w1, g1, w2, g2, w3, y1, mean1, invstd1, x1, y2, mean2, invstd2, x2 = ctx.saved_tensors
            glogits = gs[0].contiguous().float()
        spec = ctx.spec
        p1, p2 = ctx.drops
        bm = ctx.mode                               # 1: train-mode norms, 3: eval-mode (frozen) norms
        B = x0.shape[0]
        dev = x0.device
        st = stream_ptr()
        tg = spec.grad_targets
        if tg is not None and any(t is None for t in tg):    # (all of the head's parameters or none: a partial set goes through autograd)
            tg = None
        acc = 1 if tg is not None else 0
        if tg is None:
            shapes = [w1.shape, (w1.shape[0],), (w1.shape[0],), (w1.shape[0],), w2.shape, (w2.shape[0],), (w2.shape[0],),
                      (w2.shape[0],), w3.shape, (w3.shape[0],)]
            tg = [torch.empty(s, device=dev, dtype=torch.float32) for s in shapes]
        dw1, db1, dg1, dbe1, dw2, db2, dg2, dbe2, dw3, db3 = tg
        c1, c0 = w1.shape
        c2 = w2.shape[0]
        c3 = w3.shape[0]
        dy2 = torch.empty(B, c2, device=dev, dtype=torch.float32)
        dy1 = torch.empty(B, c1, device=dev, dtype=torch.float32)
        need_dx = ctx.needs_input_grad[4]
        dx0 = torch.empty(B, c0, device=dev, dtype=torch.float32) if need_dx else None
        if spec.chain:
            jobs = [_bwd_job(glogits, None, c3, None, None, None, None, None, 0.0, 0, x2, c2, c3, None, dw3, db3, None, None, acc, 0),
                    _bwd_job(glogits, w3, c3, x2, y2, mean2, invstd2, g2, p2, bm, x1, c1, c2, dy2, dw2, db2, dg2, dbe2, acc, 0),
                    _bwd_job(dy2, w2, c2, x1, y1, mean1, invstd1, g1, p1, bm, x0, c0, c1, dy1, dw1, db1, dg1, dbe1, acc, 1)]
            if need_dx:
                jobs.append(_bwd_job(dy1, w1, c1, None, None, None, None, None, 0.0, 0, None, 0, c0, dx0, None, None, None, None, 0, 2))
            _run_bwd_chain(jobs, B, spec, dev, st)
        else:
            f = lib.papc_head_bwd_f32
            if spec.merge:      # dW of the last layer and the whole backward of the layer below read the same dlogits and nothing of each other: one launch
                _run_bwd_chain([_bwd_job(glogits, None, c3, None, None, None, None, None, 0.0, 0, x2, c2, c3, None, dw3, db3, None, None, acc, 0),
                                _bwd_job(glogits, w3, c3, x2, y2, mean2, invstd2, g2, p2, bm, x1, c1, c2, dy2, dw2, db2, dg2, dbe2, acc, 0)], B, spec, dev, st)
            else:
                check(f(ptr(glogits), 0, c3, 0, 0, 0, 0, 0, 0.0, 0, ptr(x2), B, c2, c3, 0, ptr(dw3), ptr(db3), 0, 0, acc, st), "papc_head_bwd_f32")
                check(f(ptr(glogits), ptr(w3), c3, ptr(x2), ptr(y2), ptr(mean2), ptr(invstd2), ptr(g2), float(p2), bm, ptr(x1), B, c1, c2,
                        ptr(dy2), ptr(dw2), ptr(db2), ptr(dg2), ptr(dbe2), acc, st), "papc_head_bwd_f32")
            check(f(ptr(dy2), ptr(w2), c2, ptr(x1), ptr(y1), ptr(mean1), ptr(invstd1), ptr(g1), float(p1), bm, ptr(x0), B, c0, c1,
                    ptr(dy1), ptr(dw1), ptr(db1), ptr(dg1), ptr(dbe1), acc, st), "papc_head_bwd_f32")
            if need_dx:
                check(f(ptr(dy1), ptr(w1), c1, 0, 0, 0, 0, 0, 0.0, 0, 0, B, 0, c0, ptr(dx0), 0, 0, 0, 0, 0, st), "papc_head_bwd_f32")
        grads = (None,) * 10 if acc else tuple(tg)
        return (None, None, None, None, dx0) + grads


def _row_chunks(x, labels, training):
    """train mode: the whole batch (its statistics couple the rows; usable() bounds it); eval mode: rows are independent -> chunks of MAX_ROWS"""
    if training or x.shape[0] <= MAX_ROWS:
        return [(x, labels)]
    return [(x[i:i + MAX_ROWS], None if labels is None else labels.reshape(-1)[i:i + MAX_ROWS]) for i in range(0, x.shape[0], MAX_ROWS)]


def classifier_head(spec, x, fc1, bn1, drop1, fc2, bn2, drop2, fc3, training=True):
    """logits = fc3(drop2(relu(bn2(fc2(drop1(relu(bn1(fc1(x)))))))))  fused; ``training=False``: the norms in eval mode, no dropout."""
    params = (fc1.weight, fc1.bias, bn1.weight, bn1.bias, fc2.weight, fc2.bias, bn2.weight, bn2.bias, fc3.weight, fc3.bias)
    spec.training = bool(training)
    spec.grad_targets = grad_targets_of(params) if torch.is_grad_enabled() else None   # per forward: see mlp.shared_mlp_max
    outs = [_Head.apply(spec, (bn1, bn2), (float(drop1.p), float(drop2.p)), None, xc, *params) for xc, _ in _row_chunks(x, None, training)]
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def classifier_head_loss(spec, x, labels, fc1, bn1, drop1, fc2, bn2, drop2, fc3, training=True):
    """(loss, logits): the head AND the mean softmax cross-entropy of its logits (train.py:106-109) -- with the chain path one launch forward
    and one backward; ``logits`` is returned for metrics and carries no gradient."""
    params = (fc1.weight, fc1.bias, bn1.weight, bn1.bias, fc2.weight, fc2.bias, bn2.weight, bn2.bias, fc3.weight, fc3.bias)
    spec.training = bool(training)
    spec.grad_targets = grad_targets_of(params) if torch.is_grad_enabled() else None
    if not training and x.shape[0] > MAX_ROWS:      # eval beyond the kernel's rows: logits in chunks, then the loss's own launch over all of them
        logits = classifier_head(spec, x, fc1, bn1, drop1, fc2, bn2, drop2, fc3, training=False)
        return softmax_cross_entropy(logits, labels), logits.detach()
    return _Head.apply(spec, (bn1, bn2), (float(drop1.p), float(drop2.p)), labels, x, *params)


class _HeadPlain(torch.autograd.Function):
    """fc(1024,512) - ReLU - fc(512,256) - ReLU - Dropout(p) - fc(256,classes): the head of PointNet_Basic_Clas
    (/root/reference/PAPC/models/classify/pointnet_base/pointnet_base.py:26-33), no norm layers: the head kernels' ReLU-only mode."""

    @staticmethod
    def forward(ctx, spec, p, x0, w1, b1, w2, b2, w3, b3):
        lib = _lib.load()
        x0 = x0.contiguous()
        B, dev, st = x0.shape[0], x0.device, stream_ptr()
        rng = spec.state(dev)
        outs, keep = [], None
        x = x0
        layers = []
        for li, (w, b, pl) in enumerate(((w1, b1, 0.0), (w2, b2, p))):
            cout, cin = w.shape
            out = torch.empty(B, cout, device=dev, dtype=torch.float32)
            if pl > 0.0 and spec.export_masks:
                keep = torch.empty(B, cout, device=dev, dtype=torch.uint8)
            if spec.chain:
                layers.append(HeadFcLayer(ptr(x), ptr(w), ptr(b), None, None, cin, cout, 2, 0.0, 0.0, None, None, None, float(pl), li + 1, None, None, None,
                                          _p(keep) if pl > 0.0 else None, ptr(out)))
            else:
                check(lib.papc_head_fc_f32(ptr(x), ptr(w), ptr(b), 0, 0, B, cin, cout, 2, 0.0, 0.0, 0, 0, 0, float(pl), ptr(rng), li + 1, 0,
                                           0, 0, 0, ptr(keep) if pl > 0.0 else 0, ptr(out), st), "papc_head_fc_f32")
            outs.append(out)
            x = out
        cout, cin = w3.shape
        logits = torch.empty(B, cout, device=dev, dtype=torch.float32)
        if spec.chain:
            layers.append(HeadFcLayer(ptr(x), ptr(w3), ptr(b3), None, None, cin, cout, 0, 0.0, 0.0, None, None, None, 0.0, 3, None, None, None, None, ptr(logits)))
            arr = (HeadFcLayer * 3)(*layers)
            check(lib.papc_head_chain_fwd_f32(ctypes.addressof(arr), 3, B, ptr(rng), ptr(rng), None, None, None, ptr(spec.sync(dev)), st), "papc_head_chain_fwd_f32")
        else:
            check(lib.papc_head_fc_f32(ptr(x), ptr(w3), ptr(b3), 0, 0, B, cin, cout, 0, 0.0, 0.0, 0, 0, 0, 0.0, 0, 3, ptr(rng),
                                       0, 0, 0, 0, ptr(logits), st), "papc_head_fc_f32")
        spec.masks = (keep,) if spec.export_masks else None
        ctx.spec, ctx.p = spec, p
        ctx.save_for_backward(x0, w1, w2, w3, *outs)
        return logits

    @staticmethod
    def backward(ctx, glogits):
        lib = _lib.load()
        x0, w1, w2, w3, x1, x2 = ctx.saved_tensors
        spec, p = ctx.spec, ctx.p
        B, dev, st = x0.shape[0], x0.device, stream_ptr()
        glogits = glogits.contiguous().float()
        tg = spec.grad_targets
        if tg is not None and any(t is None for t in tg):
            tg = None
        acc = 1 if tg is not None else 0
        if tg is None:
            tg = [torch.empty(s_, device=dev, dtype=torch.float32) for s_ in (w1.shape, (w1.shape[0],), w2.shape, (w2.shape[0],), w3.shape, (w3.shape[0],))]
        dw1, db1, dw2, db2, dw3, db3 = tg
        c1, c0 = w1.shape
        c2, c3 = w2.shape[0], w3.shape[0]
        dy2 = torch.empty(B, c2, device=dev, dtype=torch.float32)
        dy1 = torch.empty(B, c1, device=dev, dtype=torch.float32)
        need_dx = ctx.needs_input_grad[2]
        dx0 = torch.empty(B, c0, device=dev, dtype=torch.float32) if need_dx else None
        if spec.chain:
            jobs = [_bwd_job(glogits, None, c3, None, None, None, None, None, 0.0, 0, x2, c2, c3, None, dw3, db3, None, None, acc, 0),
                    _bwd_job(glogits, w3, c3, x2, None, None, None, None, p, 2, x1, c1, c2, dy2, dw2, db2, None, None, acc, 0),
                    _bwd_job(dy2, w2, c2, x1, None, None, None, None, 0.0, 2, x0, c0, c1, dy1, dw1, db1, None, None, acc, 1)]
            if need_dx:
                jobs.append(_bwd_job(dy1, w1, c1, None, None, None, None, None, 0.0, 0, None, 0, c0, dx0, None, None, None, None, 0, 2))
            _run_bwd_chain(jobs, B, spec, dev, st)
        else:
            f = lib.papc_head_bwd_f32
            check(f(ptr(glogits), 0, c3, 0, 0, 0, 0, 0, 0.0, 0, ptr(x2), B, c2, c3, 0, ptr(dw3), ptr(db3), 0, 0, acc, st), "papc_head_bwd_f32")
            check(f(ptr(glogits), ptr(w3), c3, ptr(x2), 0, 0, 0, 0, float(p), 2, ptr(x1), B, c1, c2, ptr(dy2), ptr(dw2), ptr(db2), 0, 0, acc, st),
                  "papc_head_bwd_f32")
            check(f(ptr(dy2), ptr(w2), c2, ptr(x1), 0, 0, 0, 0, 0.0, 2, ptr(x0), B, c0, c1, ptr(dy1), ptr(dw1), ptr(db1), 0, 0, acc, st),
                  "papc_head_bwd_f32")
            if need_dx:
                check(f(ptr(dy1), ptr(w1), c1, 0, 0, 0, 0, 0, 0.0, 0, 0, B, 0, c0, ptr(dx0), 0, 0, 0, 0, 0, st), "papc_head_bwd_f32")
        grads = (None,) * 6 if acc else tuple(tg)
        return (None, None, dx0) + grads


def plain_usable(x, fc1, fc2, fc3, training):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= x.shape[0] and (x.shape[0] <= MAX_ROWS or not training)
            and fc1.in_features % 4 == 0 and fc1.out_features % 4 == 0 and fc2.out_features % 4 == 0
            and fc3.out_features % 4 == 0 and fc1.bias is not None and fc2.bias is not None and fc3.bias is not None)


def plain_head(spec, x, fc1, fc2, drop, fc3, training=True):
    """logits = fc3(drop(relu(fc2(relu(fc1(x))))))  three launches each way; ``training=False``: no dropout (rows then independent: chunks of MAX_ROWS)."""
    params = (fc1.weight, fc1.bias, fc2.weight, fc2.bias, fc3.weight, fc3.bias)
    spec.grad_targets = grad_targets_of(params) if torch.is_grad_enabled() else None
    p = float(drop.p) if training else 0.0
    outs = [_HeadPlain.apply(spec, p, xc, *params) for xc, _ in _row_chunks(x, None, training)]
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


_UNIT = {}


def unit_gradient(device):
    """The constant 1.0 to seed ``loss.backward(unit_gradient(dev))`` with: d(loss)/d(loss), allocated once per device.  Seeding with THIS
    tensor lets the loss's backward hand its stored gradient on unscaled (no multiply-by-one launch on the step's serial chain); any other
    upstream gradient -- including the ones tensor ``loss.backward()`` makes for itself -- is multiplied in as usual."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0))
    t = _UNIT.get(key)
    if t is None:
        if _lib._capturing():
            return torch.ones((), device=dev)
        t = _UNIT[key] = torch.ones((), device=dev)
    return t


def _scale_dz(dz, g):
    """dz * g for the upstream gradient g of a scalar loss; seeded with unit_gradient() the stored gradient IS the answer (no launch)"""
    if any(g is u for u in _UNIT.values()) or (g.dim() == 0 and any(g.data_ptr() == u.data_ptr() for u in _UNIT.values())):
        return dz
    g = g.contiguous().float()
    out = torch.empty_like(dz)
    check(_lib.load().papc_scale_by_f32(ptr(dz), ptr(g), dz.numel(), ptr(out), stream_ptr()), "papc_scale_by_f32")
    return out


class _SoftmaxXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels):
        logits = logits.contiguous().float()
        labels = labels.contiguous().long()
        B, C = logits.shape
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        dz = torch.empty_like(logits)
        check(_lib.load().papc_softmax_xent_f32(ptr(logits), ptr(labels), B, C, ptr(loss), ptr(dz), stream_ptr()), "papc_softmax_xent_f32")
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return _scale_dz(dz, g), None


def softmax_cross_entropy(logits, labels):
    """Mean softmax cross-entropy of logits [B, C] with int64 labels [B] (paddle ``F.cross_entropy`` / torch default), one launch
    that also produces the gradient."""
    if not logits.is_cuda:
        raise _lib.PapcError("softmax_cross_entropy needs CUDA (ROCm) tensors: there is no CPU fallback")
    return _SoftmaxXent.apply(logits, labels.reshape(-1))
