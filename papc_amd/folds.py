"""Deferred folds: the partial reductions of a training step's backward (dW / db partials of every stack, the strided column blocks of the
gather-add first layer, the split-K partials of the planes path) in ONE launch behind the last backward kernel
(include/papc_hip.h: papc_fold_jobs_f32, papc_sa_grads.defer) instead of 5-6 launch-latency-sized kernels spread over the backward.

A stack's backward (stack.SharedMLPStack.backward) asks :func:`pending` for the list of the running autograd pass (keyed by the engine's
graph task id, so that a pass that raised and never ran its callbacks leaves nothing behind for the next one); the first request of a
pass registers that list's ``flush`` as a final callback of the autograd engine, which runs it once every node of the pass
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
    """The fold list of ONE autograd pass (one graph task of the engine)."""

    def __init__(self, task_id):
        self.task_id = task_id
        self.jobs = (FoldJob * CAPACITY)()
        self.lst = FoldList(ctypes.cast(self.jobs, ctypes.POINTER(FoldJob)), CAPACITY, 0)
        self.keep = []
        self.stream = _cur_stream()       # where the partials are produced
        self.owner = threading.get_ident()
        self.done = False

    def flush(self):
        """Fold everything that is pending (the autograd engine calls this once all nodes of the pass are enqueued -- on whichever
        thread finishes the graph task, which need not be the one the stacks ran on: nothing here reads thread-local state).  The fold
        runs on the calling thread's current stream, ordered behind the stream the partials were produced on when the two differ."""
        with _lock:
            _live.pop(self.task_id, None)
            # a pass of this thread with a LARGER id was nested inside this one and has ended: whatever it left behind is stale
            for k in [k for k, q in _live.items() if q.owner == self.owner and k > self.task_id]:
                _live.pop(k).drop()
        if self.done:
            return
        try:
            if self.lst.count:
                _launch(self)
        finally:
            self.drop()

    def drop(self):
        self.done = True
        self.lst.count = 0
        self.keep.clear()


def _cur_stream():
    return torch.cuda.current_stream()


def _launch(p):
    cur = torch.cuda.current_stream(p.stream.device)
    if cur != p.stream:
        cur.wait_event(p.stream.record_event())
    with torch.cuda.device(p.stream.device):
        _lib.check(_lib.load().papc_fold_jobs_f32(p.jobs, p.lst.count, ctypes.c_void_p(cur.cuda_stream)), "papc_fold_jobs_f32")


# graph task id -> _Pending.  Keyed by the engine's graph task, not by thread: a backward that RAISES never runs its final callbacks
# (its entry stays behind with stale jobs), and the next pass must neither inherit those jobs nor skip its own registration.
_live = {}
_lock = threading.Lock()
MAX_LIVE = 8        # more than this many unfinished passes = leftovers of failed ones (reentrant nesting is never that deep): oldest dropped,
                    # with a warning -- if such an entry WERE a live outer pass its queued folds would be lost, so the drop must not be silent


def pending(keep):
    """The fold list of the running backward pass (a ctypes pointer for papc_sa_grads.defer) or None: deferral is off, this is not an
    autograd backward, or the call runs on another stream than the pass's first stack (the MSG layers' branch streams: those fold in
    place).  ``keep`` = objects that must stay alive until the fold has been enqueued (the backward scratch buffer)."""
    if not ENABLED:
        return None
    tid = torch._C._current_graph_task_id()
    if tid < 0:                         # not inside a backward pass of the autograd engine
        return None
    with _lock:
        p = _live.get(tid)
    if p is None:
        p = _Pending(tid)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(p.flush)
        except RuntimeError:
            return None
        with _lock:
            _live[tid] = p
            while len(_live) > MAX_LIVE:
                old = _live.pop(min(_live))
                if old.lst.count:
                    import warnings
                    warnings.warn("papc_amd.folds: dropping %d deferred fold job(s) of autograd pass %d, which never completed (a backward that raised?); "
                                  "if that pass is in fact still running -- more than %d nested backward passes -- its parameter gradients are incomplete: "
                                  "set PAPC_DEFER_FOLDS=0" % (old.lst.count, old.task_id, MAX_LIVE))
                old.drop()
    elif p.done or p.stream != _cur_stream():
        return None
    if p.lst.count > CAPACITY - 16:     # (a stack appends at most 2 jobs per layer + 2)
        return None
    p.keep.extend(keep)
    return ctypes.pointer(p.lst)
