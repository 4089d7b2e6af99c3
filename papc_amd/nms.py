"""Axis-aligned and rotated NMS / IoU on the device.

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


def rotate_nms_gpu_tensor(dets, nms_overlap_thresh):
    """dets [N,6] = (x, y, x_d, y_d, angle, score) fp32 CUDA tensor -> int64 tensor of kept original indices, best score first."""
    if not dets.is_cuda:
        raise _lib.PapcError("rotate_nms_gpu needs a CUDA (ROCm) tensor: there is no CPU fallback")
    assert dets.dim() == 2 and dets.shape[1] == 6, "dets must be [N, 6]"
    N = int(dets.shape[0])
    if N == 0:
        return torch.empty(0, dtype=torch.int64, device=dets.device)
    dets = dets.contiguous().float()
    lib = _lib.load()
    ws_bytes = int(lib.papc_nms_workspace(N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dets.device)
    keep = torch.empty(N, dtype=torch.int32, device=dets.device)
    num = torch.empty(1, dtype=torch.int32, device=dets.device)
    check(lib.papc_rotate_nms_f32(ptr(dets), N, float(nms_overlap_thresh), ptr(keep), ptr(num), ptr(ws), ws_bytes, stream_ptr()),
          "papc_rotate_nms_f32")
    return keep[:int(num.item())].long()


def rotate_nms_gpu(dets, nms_overlap_thresh, device_id=0):
    """nms_gpu.py:453-488: numpy or tensor ``dets`` [N,6] in, python list of kept indices out."""
    if not torch.is_tensor(dets):
        dets = torch.as_tensor(dets, dtype=torch.float32).to("cuda:%d" % device_id)
    return rotate_nms_gpu_tensor(dets, nms_overlap_thresh).tolist()


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """nms_gpu.py:618-653 (``rotate_iou_gpu`` :524-559 is criterion -1).  boxes [N,5], query_boxes [K,5] = (x, y, x_d, y_d, angle)
    -> iou [N,K] (tensor in -> tensor out, numpy in -> numpy out like the source)."""
    as_np = not torch.is_tensor(boxes)
    dev = "cuda:%d" % device_id
    b = torch.as_tensor(boxes, dtype=torch.float32).to(dev).contiguous() if as_np else boxes.contiguous().float()
    q = torch.as_tensor(query_boxes, dtype=torch.float32).to(b.device).contiguous()
    if not b.is_cuda:
        raise _lib.PapcError("rotate_iou_gpu needs CUDA (ROCm) tensors: there is no CPU fallback")
    N, K = int(b.shape[0]), int(q.shape[0])
    iou = torch.zeros(N, K, dtype=torch.float32, device=b.device)
    if N and K:
        check(_lib.load().papc_rotate_iou_f32(ptr(b), ptr(q), N, K, int(criterion), ptr(iou), stream_ptr()), "papc_rotate_iou_f32")
    return iou.cpu().numpy() if as_np else iou


def rotate_iou_gpu(boxes, query_boxes, device_id=0):
    return rotate_iou_gpu_eval(boxes, query_boxes, -1, device_id)


def rbbox_iou(box_corners, qbox_corners, standup_iou=None, standup_thresh=0.0, device_id=0):
    """``box_ops_cc.rbbox_iou`` of libs/ops/cc/box_ops.h:23-80 (boost::geometry on the host in the reference): box_corners [N,4,2],
    qbox_corners [K,4,2], standup_iou [N,K] or None (then taken from the corners' bounding boxes) -> overlaps [N,K].  numpy in ->
    numpy out, tensor in -> tensor out."""
    as_np = not torch.is_tensor(box_corners)
    dev = "cuda:%d" % device_id
    b = torch.as_tensor(box_corners, dtype=torch.float32).to(dev).contiguous() if as_np else box_corners.contiguous().float()
    q = torch.as_tensor(qbox_corners, dtype=torch.float32).to(b.device).contiguous()
    if not b.is_cuda:
        raise _lib.PapcError("rbbox_iou needs CUDA (ROCm) tensors: there is no CPU fallback")
    assert b.dim() == 3 and tuple(b.shape[1:]) == (4, 2) and q.dim() == 3 and tuple(q.shape[1:]) == (4, 2)
    N, K = int(b.shape[0]), int(q.shape[0])
    su = None if standup_iou is None else torch.as_tensor(standup_iou, dtype=torch.float32).to(b.device).contiguous()
    assert su is None or tuple(su.shape) == (N, K)
    out = torch.zeros(N, K, dtype=torch.float32, device=b.device)
    if N and K:
        check(_lib.load().papc_rbbox_iou_f32(ptr(b), ptr(q), ptr(su), float(standup_thresh), N, K, ptr(out), stream_ptr()), "papc_rbbox_iou_f32")
    return out.cpu().numpy() if as_np else out


def riou_cc(rbboxes, qrbboxes, standup_thresh=0.0, device_id=0):
    """libs/ops/box_np_ops.py:16-27: rotated IoU of (x, y, w, l, angle) boxes behind an axis-aligned pre-test, one launch."""
    as_np = not torch.is_tensor(rbboxes)
    dev = "cuda:%d" % device_id
    b = torch.as_tensor(rbboxes, dtype=torch.float32).to(dev).contiguous() if as_np else rbboxes.contiguous().float()
    q = torch.as_tensor(qrbboxes, dtype=torch.float32).to(b.device).contiguous()
    if not b.is_cuda:
        raise _lib.PapcError("riou_cc needs CUDA (ROCm) tensors: there is no CPU fallback")
    assert b.dim() == 2 and b.shape[1] == 5 and q.dim() == 2 and q.shape[1] == 5
    N, K = int(b.shape[0]), int(q.shape[0])
    out = torch.zeros(N, K, dtype=torch.float32, device=b.device)
    if N and K:
        check(_lib.load().papc_riou_f32(ptr(b), ptr(q), float(standup_thresh), N, K, ptr(out), stream_ptr()), "papc_riou_f32")
    return out.cpu().numpy() if as_np else out
