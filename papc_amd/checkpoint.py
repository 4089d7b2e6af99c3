"""Weight exchange with the reference's checkpoints.

The reference saves ``paddle.save(model.state_dict(), "*.pdparams")`` (PAPC/train.py:118-120,
PAPC/models/detect/pointpillars/libs/tools/checkpoint.py:90): a pickled ``dict`` of parameter name -> numpy array (plus
the bookkeeping key ``StructuredToParameterName@@``), readable without PaddlePaddle.  Layout differences handled here:

  * ``nn.Linear.weight`` is ``[in, out]`` in Paddle, ``[out, in]`` here  -> transposed;
  * BatchNorm running statistics are ``_mean`` / ``_variance`` in Paddle, ``running_mean`` / ``running_var`` here;
  * conv weights (``[out, in, 1]`` / ``[out, in, 1, 1]``) and BatchNorm ``weight`` / ``bias`` are identical.

The reference keeps the set-abstraction / feature-propagation convs and norms in plain Python lists
(pointnet2_basic_layers.py:185-191, :230-241, :287-293), so they are NOT in its state_dict: importing a reference checkpoint
fills the registered layers (FC head, PointNet-Basic convs, PFN) and reports the SA/FP entries as missing; exporting writes them
under this package's names so that a round trip through this module is lossless.
"""
import pickle

import numpy as np
import torch
import torch.nn as nn

_SKIP = ("StructuredToParameterName@@",)


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """A checkpoint is data from somewhere else: only numpy's array reconstruction helpers and plain containers may be
    instantiated while reading it (a stock ``pickle.load`` would run any callable the file names)."""

    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
                ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("collections", "OrderedDict"), ("_codecs", "encode")}   # (protocol-2 pickles carry array bytes as latin-1 text)

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("checkpoint names %s.%s: only numpy arrays and plain containers are accepted" % (module, name))


def load_pdparams(path):
    """*.pdparams -> {name: numpy array} (pickle restricted to numpy arrays; nothing from paddle is imported)."""
    with open(path, "rb") as f:
        obj = _ArraysOnlyUnpickler(f).load()
    if not isinstance(obj, dict):
        raise ValueError("%s does not hold a state dict" % path)
    out = {}
    for k, v in obj.items():
        if k in _SKIP:
            continue
        if isinstance(v, tuple) and len(v) == 2 and isinstance(v[1], np.ndarray):   # older paddle: (name, ndarray)
            v = v[1]
        out[k] = np.asarray(v)
    return out


def _module_of(model, param_name):
    mod = model
    parts = param_name.split(".")[:-1]
    for p in parts:
        mod = getattr(mod, p) if not p.isdigit() else mod[int(p)]
    return mod


def _to_reference_name(name, model=None):
    """this package's state_dict key -> the reference's.  A model may carry ``reference_names`` (own module prefix -> the
    source's, e.g. PointNet_Basic_Clas: ``convs.0`` -> ``mlp_1.0``, pointnet_base.py:7-25) where its containers are laid out
    differently from the source's nn.Sequential holders."""
    ref = getattr(model, "reference_names", None) if model is not None else None
    if ref:
        for own, theirs in ref.items():
            if name.startswith(own + "."):
                name = theirs + name[len(own):]
                break
    return name.replace("running_mean", "_mean").replace("running_var", "_variance")


def export_state(model):
    """This package's model -> {reference-style name: numpy array} (Linear weights transposed to [in, out])."""
    out = {}
    for name, t in model.state_dict().items():
        if name.endswith("num_batches_tracked"):
            continue
        a = t.detach().cpu().numpy()
        if name.endswith(".weight") and isinstance(_module_of(model, name), nn.Linear):
            a = np.ascontiguousarray(a.T)
        out[_to_reference_name(name, model)] = a
    return out


def save_pdparams(model, path):
    with open(path, "wb") as f:
        pickle.dump(export_state(model), f, protocol=2)


def import_state(model, state, strict=False):
    """Copy a reference-style state (from load_pdparams / export_state) into ``model``.  Returns (missing, unexpected):
    parameter names of ``model`` the state did not provide, and state entries ``model`` has no place for."""
    own = {k: v for k, v in model.state_dict().items() if not k.endswith("num_batches_tracked")}
    used = set()
    missing = []
    with torch.no_grad():
        for name, dst in own.items():
            key = _to_reference_name(name, model)
            if key not in state:
                missing.append(name)
                continue
            a = np.asarray(state[key])
            if name.endswith(".weight") and isinstance(_module_of(model, name), nn.Linear):
                a = a.T
            if tuple(a.shape) != tuple(dst.shape):
                if a.size == dst.numel():
                    a = a.reshape(tuple(dst.shape))       # conv kernels stored with / without the trailing 1x1 dims
                else:
                    raise ValueError("%s: shape %s does not fit %s" % (name, a.shape, tuple(dst.shape)))
            dst.copy_(torch.from_numpy(np.ascontiguousarray(a)).to(dst.dtype))
            used.add(key)
    unexpected = [k for k in state if k not in used]
    if strict and (missing or unexpected):
        raise KeyError("missing: %s; unexpected: %s" % (missing, unexpected))
    return missing, unexpected
