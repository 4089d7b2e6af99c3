"""Deferred folds: the partial reductions of a training step's backward (dW / db partials of every stack, the strided column blocks of the
gather-add first layer, the split-K partials of the planes path) in ONE launch behind the last backward kernel
(include/papc_hip.h: papc_fold_jobs_f32, papc_sa_grads.defer) instead of 5-6 launch-latency-sized kernels spread over the backward.

A stack's backward (stack.SharedMLPStack.backward) asks :func:`pending` for the list of the running autograd pass; the first request of a
pass registers :func:`flush` as a final callback of the autograd engine, which runs it on the caller's stream once every node of the pass
has been enqueued -- still inside a hipGraph capture, and once per stage of a two-stage backward (bench.py, N > 1).  The backward scratch
buffers the jobs read are kept alive until the fold has been enqueued.  ``PAPC_DEFER_FOLDS=0`` restores the per-stack launches.

Reference: the reference has no counterpart (Paddle's autograd accumulates dense gradients); this only changes WHEN this library's
split reductions are folded, not what they add up to (same summation order per job: tests/test_gpu_cabi.py).
"""
import ctypes
import os
import threading

import torch

from . import _lib

ENABLED = os.environ.get("PAPC_DEFER_FOLDS", "1") != "0"
CAPACITY = 96


class FoldJob(ctypes.Structure):
    """papc_fold_job"""
    _fields_ = [("partial", ctypes.c_void_p), ("n_chunks", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("ld", ctypes.c_int64),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32), ("out", ctypes.c_void_p), ("out_ld", ctypes.c_int64)]


class FoldList(ctypes.Structure):
    """papc_fold_list"""
    _fields_ = [("jobs", ctypes.POINTER(FoldJob)), ("capacity", ctypes.c_int32), ("count", ctypes.c_int32)]


class _Pending:
    def __init__(self):
        self.jobs = (FoldJob * CAPACITY)()
        self.lst = FoldList(ctypes.cast(self.jobs, ctypes.POINTER(FoldJob)), CAPACITY, 0)
        self.keep = []
        self.stream = None
        self.armed = False


_tls = threading.local()


def pending(keep):
    """The fold list of the running backward pass (a ctypes pointer for papc_sa_grads.defer) or None: deferral is off, this is not an
    autograd backward, or the call runs on another stream than the pass's first stack (the MSG layers' branch streams: those fold in
    place).  ``keep`` = objects that must stay alive until the fold has been enqueued (the backward scratch buffer)."""
    if not ENABLED:
        return None
    p = getattr(_tls, "p", None)
    if p is None:
        p = _tls.p = _Pending()
    cur = torch.cuda.current_stream().cuda_stream
    if not p.armed:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(flush)
        except RuntimeError:        # not inside a backward pass of the autograd engine
            return None
        p.armed, p.stream = True, cur
    elif p.stream != cur:
        return None
    if p.lst.count > CAPACITY - 16:     # (a stack appends at most 2 jobs per layer + 2)
        return None
    p.keep.extend(keep)
    return ctypes.pointer(p.lst)


def flush():
    """Fold everything that is pending, on the current stream (the autograd engine calls this once all nodes of the pass are enqueued)."""
    p = getattr(_tls, "p", None)
    if p is None:
        return
    try:
        if p.lst.count:
            _lib.check(_lib.load().papc_fold_jobs_f32(p.jobs, p.lst.count, _lib.stream_ptr()), "papc_fold_jobs_f32")
    finally:
        p.lst.count = 0
        p.keep.clear()
        p.armed, p.stream = False, None
