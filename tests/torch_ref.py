"""Plain-PyTorch float64 reference of the shared MLP stack (autograd gives the reference gradients).
Used only by tests: the HIP backward kernels are floating-point kernels, so their reference is a torch one."""
import torch


def gather_rows(points, idx):
    """points [B,N,C], idx [B,S,K] -> [B,S,K,C]"""
    B, S, K = idx.shape
    C = points.shape[2]
    flat = idx.reshape(B, S * K).long()
    return torch.gather(points, 1, flat[:, :, None].expand(B, S * K, C)).reshape(B, S, K, C)


def group(xyz, new_xyz, feats, idx, xyz_first):
    gx = gather_rows(xyz, idx) - new_xyz[:, :, None, :]
    if feats is None:
        return gx
    gf = gather_rows(feats, idx)
    return torch.cat([gx, gf], -1) if xyz_first else torch.cat([gf, gx], -1)


def stack_max(rows, params, K, eps):
    """rows [M,Cin] -> max over groups of K of relu(bn(conv)) chain; params = [(w,b,gamma,beta)]"""
    x = rows
    for (w, b, g, bt) in params:
        y = x @ w.t()
        if b is not None:
            y = y + b
        mean = y.mean(0)
        var = y.var(0, unbiased=False)
        x = torch.relu((y - mean) / torch.sqrt(var + eps) * g + bt)
    return x.reshape(-1, K, x.shape[1]).max(dim=1).values
