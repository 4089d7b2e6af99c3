"""Axis-aligned NMS on the device.

Mirrors ``nms_gpu(dets, nms_overlap_thresh, device_id=0)`` of
PAPC/models/detect/pointpillars/libs/ops/non_max_suppression/nms_gpu.py:130-164 (its pybind11 twin ``nms.non_max_suppression`` in
libs/ops/cc/nms/nms_kernel.cu.cc) -- score sort, 64-bit suppression words, sequential sweep -- with no host round trip but the
final count.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def nms_gpu_tensor(dets, nms_overlap_thresh):
    """dets [N,5] = (x1, y1, x2, y2, score) fp32 CUDA tensor -> int64 tensor of kept original indices, best score first."""
    if not dets.is_cuda:
        raise _lib.PapcError("nms_gpu needs a CUDA (ROCm) tensor: there is no CPU fallback")
    assert dets.dim() == 2 and dets.shape[1] == 5, "dets must be [N, 5]"
    N = int(dets.shape[0])
    if N == 0:
        return torch.empty(0, dtype=torch.int64, device=dets.device)
    dets = dets.contiguous().float()
    lib = _lib.load()
    ws_bytes = int(lib.papc_nms_workspace(N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dets.device)
    keep = torch.empty(N, dtype=torch.int32, device=dets.device)
    num = torch.empty(1, dtype=torch.int32, device=dets.device)
    check(lib.papc_nms_f32(ptr(dets), N, float(nms_overlap_thresh), ptr(keep), ptr(num), ptr(ws), ws_bytes, stream_ptr()), "papc_nms_f32")
    return keep[:int(num.item())].long()


def nms_gpu(dets, nms_overlap_thresh, device_id=0):
    """The reference's signature: numpy or tensor ``dets`` in, python list of kept indices out (nms_gpu.py:130, :164)."""
    if not torch.is_tensor(dets):
        dets = torch.as_tensor(dets, dtype=torch.float32).to("cuda:%d" % device_id)
    return nms_gpu_tensor(dets, nms_overlap_thresh).tolist()
