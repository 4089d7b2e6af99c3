"""Shared helpers for the parity tests (oracle = checker; papc_amd = product)."""
import numpy as np


def seeded_weights(chans, seed, bias_scale=0.1):
    """[(conv_w [Cout,Cin], conv_b, gamma, beta)] with non-trivial gamma/beta so BN folding is exercised
    (including NEGATIVE gammas: the max must then pick the smallest pre-BN value)."""
    rng = np.random.default_rng(seed)
    ws = []
    for cin, cout in zip(chans[:-1], chans[1:]):
        w = (rng.normal(size=(cout, cin)) * np.sqrt(2.0 / cin)).astype(np.float32)
        b = (rng.normal(size=cout) * bias_scale).astype(np.float32)
        g = rng.uniform(0.5, 1.5, size=cout).astype(np.float32) * rng.choice([1.0, 1.0, 1.0, -1.0], size=cout).astype(np.float32)
        bt = (rng.normal(size=cout) * 0.2).astype(np.float32)
        ws.append((w, b, g, bt))
    return ws


def assert_close(got, ref, rel=1e-5, what=""):
    """|got-ref| <= rel * max|ref| elementwise-max criterion (activations are O(1) after BN); the north-star
    tolerance for MLP activations is 1e-5 relative."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(np.max(np.abs(ref))), 1e-30)
    err = float(np.max(np.abs(got - ref))) / scale
    # second figure, reported next to the max-norm one: the worst ELEMENTWISE relative error among the entries that are not tiny
    # (|ref| >= 1e-3 of the largest) -- a wrong small entry hides under a max-norm bar
    big = np.abs(ref) >= 1e-3 * scale
    elem = float(np.max(np.abs(got - ref)[big] / np.abs(ref)[big])) if big.any() else 0.0
    print("[assert_close] %s: %.2e of max|ref|, elementwise %.2e (entries >= 1e-3 max)" % (what, err, elem))
    assert err <= rel, "%s: max rel err %.3e > %.1e (elementwise %.3e)" % (what, err, rel, elem)
    return err


def seeded_model_state(model, seed):
    """Deterministic parameter values for a torch module from a numpy generator (independent of torch's RNG streams and of
    the torch version): weights ~ N(0, 2/fan_in), biases ~ 0.1 N(0,1), norm weights in +-[0.5, 1.5], norm biases ~ 0.2 N(0,1).
    Returns {name: float32 ndarray}; copy_into_model() loads it."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, p in model.named_parameters():
        shape = tuple(p.shape)
        if p.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            a = rng.normal(size=shape) * np.sqrt(2.0 / fan_in)
        elif "bn" in name or "norm" in name:
            if name.endswith("weight"):
                a = rng.uniform(0.5, 1.5, size=shape) * rng.choice([1.0, 1.0, 1.0, -1.0], size=shape)
            else:
                a = rng.normal(size=shape) * 0.2
        else:
            a = rng.normal(size=shape) * 0.1
        out[name] = a.astype(np.float32)
    return out


def copy_into_model(model, state):
    import torch
    with torch.no_grad():
        for name, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[name]).to(p.device))


def kernel_decisions(out):
    """The decisions the HIP stack behind ``out`` (a shared_mlp_max result that still has its autograd node) took in its forward pass,
    read from the node's saved tensors: (argmax [G, C_L] int32, pooled_alive [G, C_L] bool, masks per non-pooled layer or None).
    A ReLU mask is sign(scale * y + shift) of the kernel's own stored pre-BN output and folded BN constants, evaluated in float64
    (the product of two fp32 numbers is exact there, so the sign is the sign of the kernel's fp32 fma).  Layers whose output the kernel
    never stores (the moment-path first layer, the planes path) return None: their decisions stay the reference's own."""
    fn = out.grad_fn
    for _ in range(4):
        if "MLPMax" in type(fn).__name__ or "MLPStack" in type(fn).__name__:
            break
        fn = fn.next_functions[0][0]
    assert "MLPMax" in type(fn).__name__ or "MLPStack" in type(fn).__name__, type(out.grad_fn).__name__
    saved = fn.saved_tensors
    alive = out.detach() > 0
    if "Planes" in type(fn).__name__:
        return saved[0], alive, None
    L = fn.L
    if "Stack" in type(fn).__name__:       # the library-orchestrated node (papc_amd/stack.py): views into its saved buffer
        from papc_amd.stack import SharedMLPStack
        argmax, ys, consts = SharedMLPStack.views(fn)
    else:
        argmax = saved[5]
        ys, consts = saved[7 + 4 * L: 7 + 5 * L], saved[7 + 5 * L: 7 + 6 * L]
    rowmap = None
    cp = getattr(fn, "compact", None)
    if cp is not None:
        # compacted stack (papc_amd/compact.py): rows are the distinct neighbours; padded slot (g, k) is physical row start[g] + k while k is
        # inside the group's rows and a copy of its first row beyond; argmax holds absolute rows -> offsets in the padded group
        import torch
        start = cp.start.long()
        n = (start[1:] - start[:-1])
        k = torch.arange(cp.K, device=start.device).view(1, -1)
        rowmap = (start[:-1].view(-1, 1) + torch.where(k < n.view(-1, 1), k, torch.zeros_like(k))).reshape(-1)
        argmax = (argmax.long() - start[:-1].view(-1, 1)).to(argmax.dtype)
    masks = []
    for l in range(L - 1):
        y, c = ys[l], consts[l]
        if y is None or y.dim() != 2 or y.shape[1] != c.shape[1]:
            masks.append(None)
        else:
            m = (c[2].double() * y.double() + c[3].double()) > 0
            masks.append(m if rowmap is None else m[rowmap])
    return argmax, alive, masks
