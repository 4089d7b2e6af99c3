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


def stack_routed(rows, params, K, eps, argmax=None, pooled_alive=None, masks=None):
    """stack_max with its DECISIONS pinned.  A max-pooled stack's gradient is a discontinuous function of the activations: which row
    wins a group's max, and on which side of 0 a pre-activation falls.  When an fp32 kernel and a float64 reference are compared at
    2e-4, a decision within fp32 rounding of a tie makes them differ by a whole row's worth -- legitimately, and on any fp32
    implementation.  Pinning the reference to the kernel's own decisions removes exactly that freedom and nothing else: every
    arithmetic term of the backward is still the reference's own.

      argmax        [G, C_L] int: winner row (offset in the group) of every (group, channel); None = the reference's own max
      pooled_alive  [G, C_L] bool: whether the pooled activation is > 0; None = the reference's own
      masks         per non-pooled layer: [M, C_l] bool "pre-activation > 0", or None = the reference's own ReLU
    Returns (pooled [G, C_L], stats) where stats counts the decisions that differ from the reference's own."""
    x = rows
    L = len(params)
    stats = {"relu_flips": 0, "winner_moves": 0, "alive_flips": 0}
    z = None
    for l, (w, b, g, bt) in enumerate(params):
        y = x @ w.t()
        if b is not None:
            y = y + b
        mean = y.mean(0)
        var = y.var(0, unbiased=False)
        z = (y - mean) / torch.sqrt(var + eps) * g + bt
        if l < L - 1:
            m = masks[l] if masks is not None else None
            if m is None:
                x = torch.relu(z)
            else:
                stats["relu_flips"] += int((m != (z.detach() > 0)).sum())
                x = torch.where(m, z, torch.zeros_like(z))
    C = z.shape[1]
    zg = z.reshape(-1, K, C)
    if argmax is None:
        return torch.relu(zg).max(dim=1).values, stats
    route = argmax.long().unsqueeze(1)
    zr = zg.gather(1, route).squeeze(1)
    own = torch.relu(zg.detach())
    tm = own.max(1).values
    tol = 1e-5 * float(tm.abs().max())
    assert bool((torch.relu(zr.detach()) >= tm - tol).all()), "a pinned winner is not within 1e-5 of the group's max"
    stats["winner_moves"] = int((route.squeeze(1) != own.argmax(1)).sum())
    alive = pooled_alive if pooled_alive is not None else (zr.detach() > 0)
    stats["alive_flips"] = int((alive != (zr.detach() > 0)).sum())
    return torch.where(alive, zr, torch.zeros_like(zr)), stats
