"""GPU parity of the strided batch copy (csrc/bn_ops.hip, papc_amd/copyops.py) against torch.cat / .contiguous(): forward bit-exact (a
copy), backward bit-exact (a split), over the views the layers hand it -- transposes, broadcast (stride 0) inputs, ragged tile edges."""
import pytest
import torch

from papc_amd.copyops import cat_copy, contiguous_copy

pytestmark = pytest.mark.gpu


def _mk(dev, B, R, C, kind, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if kind == "plain":
        t = torch.randn(B, R, C, generator=g)
    elif kind == "transposed":            # [B, C, R] storage viewed as [B, R, C]
        t = torch.randn(B, C, R, generator=g)
    elif kind == "sliced":                # a column slice of a wider tensor
        t = torch.randn(B, R, C + 5, generator=g)
    else:                                 # broadcast along C (the one-hot label expanded over the points)
        t = torch.randn(B, R, 1, generator=g)
    t = t.to(dev).requires_grad_(kind != "broadcast")
    if kind == "transposed":
        v = t.transpose(1, 2)
    elif kind == "sliced":
        v = t[:, :, 2:2 + C]
    elif kind == "broadcast":
        v = t.expand(B, R, C)
    else:
        v = t
    return t, v


@pytest.mark.parametrize("dim,shapes", [
    (2, [(64, "plain"), (128, "plain"), (128, "plain")]),                  # the MSG branches' outputs
    (2, [(131, "transposed"), (37, "plain")]),                             # points1^T next to the interpolated features, ragged widths
    (1, [(16, "broadcast"), (3, "sliced"), (6, "plain")]),                 # one-hot label | xyz | features along the channel index
    (2, [(3, "transposed")]),                                              # a lone transposed copy
])
def test_cat_copy_matches_torch(dev, dim, shapes):
    B, n_other = 5, 77
    leaves, views = [], []
    for i, (n, kind) in enumerate(shapes):
        R, C = (n_other, n) if dim == 2 else (n, n_other)
        t, v = _mk(dev, B, R, C, kind, 10 + i)
        leaves.append(t)
        views.append(v)
    out = cat_copy(views, dim)
    ref = torch.cat([v.detach() for v in views], dim)
    assert out.is_contiguous() and torch.equal(out, ref)
    g = torch.randn_like(out)
    # a non-contiguous upstream gradient (what a transpose in front of the consumer produces)
    gv = g.transpose(1, 2).contiguous().transpose(1, 2)
    out.backward(gv)
    off = 0
    for t, v, (n, kind) in zip(leaves, views, shapes):
        if kind == "broadcast":
            off += n
            continue
        want = torch.zeros_like(t)
        piece = g.narrow(dim, off, n)
        if kind == "plain":
            want = piece
        elif kind == "transposed":
            want = piece.transpose(1, 2)
        else:
            want = torch.zeros_like(t)
            want[:, :, 2:2 + piece.shape[2]] = piece
        assert torch.equal(t.grad, want.contiguous() if kind != "sliced" else want), kind
        off += n


def test_contiguous_copy(dev):
    x = torch.randn(3, 6, 2048, device=dev)
    v = x.transpose(1, 2)
    c = contiguous_copy(v)
    assert c.is_contiguous() and torch.equal(c, v.contiguous())
    assert contiguous_copy(c) is c


@pytest.mark.parametrize("shape,at,pad", [((64, 6), 3, 1), ((32, 6), 6, 2), ((4, 100, 3), 3, 1), ((2, 33, 10), 0, 2)])
def test_pad_cols(dev, shape, at, pad):
    from papc_amd.copyops import pad_cols
    x = torch.randn(*shape, device=dev, requires_grad=True)
    out = pad_cols(x, at, pad)
    z = torch.zeros(*shape[:-1], pad, device=dev)
    ref = torch.cat([x.detach()[..., :at], z, x.detach()[..., at:]], -1)
    assert out.is_contiguous() and torch.equal(out, ref)
    g = torch.randn_like(out)
    out.backward(g)
    assert torch.equal(x.grad, torch.cat([g[..., :at], g[..., at + pad:]], -1))
