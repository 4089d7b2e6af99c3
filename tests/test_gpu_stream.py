"""GPU parity of the row-streaming GEMM (papc_amd/csrc/mlp_stream.hip) against the tiled kernel it replaces on the big
shapes (same split-bf16 products in the same order: conv outputs must agree to the last bit or two), against the f64 torch
reference, and at a BASELINE config-2 layer size through size-independent properties."""
import numpy as np
import pytest
import torch

from papc_amd import _lib
from papc_amd.mlp import StackSpec, shared_mlp_max
from tests import torch_ref
from tests.util import assert_close, seeded_weights

pytestmark = pytest.mark.gpu


def _knob(name, value):
    _lib.check(_lib.load().papc_knob_set(name.encode(), int(value)), "papc_knob_set")


@pytest.fixture
def stream_knobs():
    lib = _lib.load()
    old = {}
    for n in ("PAPC_STREAM", "PAPC_STREAM_MINTILES", "PAPC_STREAM_ASM"):
        v = _lib.ctypes.c_int(0)
        _lib.check(lib.papc_knob_get(n.encode(), _lib.ctypes.byref(v)), "papc_knob_get")
        old[n] = v.value
    yield
    for n, v in old.items():
        _knob(n, v)


def _run(dev, G, K, chans, seed, stream, need_x_grad=False):
    """plain-input stack (rows [M, chans[0]]) -> max over groups of K; returns out, grads"""
    M = G * K
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.normal(size=(M, chans[0])).astype(np.float32)).to(dev).requires_grad_(need_x_grad)
    ws = seeded_weights(chans, seed + 1)
    params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
    gout = torch.from_numpy(rng.normal(size=(G, chans[-1])).astype(np.float32)).to(dev)
    _knob("PAPC_STREAM", 1 if stream else 0)
    spec = StackSpec(1, M, G, K, chans[0] - 3, True)
    xyz = torch.zeros(1, 1, 3, device=dev)
    out = shared_mlp_max(spec, None, xyz, xyz, None, None, params, x_rows=x)
    out.backward(gout)
    torch.cuda.synchronize()
    return x, params, gout, out


@pytest.mark.parametrize("G,K,chans", [
    (128, 32, [64, 64, 64, 128]),      # SA1-shaped widths: PLAIN 64->64, BNRELU 64->64, BNRELU+gmax 64->128; dX MAX 128->64, DENSE 64->64
    (64, 64, [128, 128, 128, 256]),    # SA2-shaped: K = 128 operands, two column blocks, group = two tiles; dX MAX with K = 256
    (24, 128, [32, 64, 128]),          # group = four tiles, 32-wide first layer, ragged number of units per wave
    (100, 32, [64, 128, 64]),          # unit count not a multiple of the wave count
    (96, 64, [64, 64, 96, 128]),       # the MSG branches' 96-channel layers: three column tiles (64 -> 96), six k blocks (96 -> 128)
    (32, 128, [128, 128, 196, 256]),   # the MSG segmenter's 196-channel pair (pointnet2.py:63): ragged n (128 -> 196, dX 256 -> 196), ragged k (196 -> 256 + max, dX 196 -> 128)
    (25, 128, [128, 128, 196, 256]),   # ... with a unit count that does not divide over the waves
])
@pytest.mark.parametrize("asm", [1, 0])
def test_stream_matches_tiled_and_f64(dev, stream_knobs, G, K, chans, asm):
    _knob("PAPC_STREAM_MINTILES", 1)
    _knob("PAPC_STREAM_ASM", asm)
    xs, ps, gout, outs = _run(dev, G, K, chans, 3, True, need_x_grad=True)
    xt, pt, _, outt = _run(dev, G, K, chans, 3, False, need_x_grad=True)
    # identical products in identical order per output element: only the BN statistics (other summation order) differ
    assert_close(outs.detach().cpu().numpy(), outt.detach().cpu().numpy(), 2e-6, "stream vs tiled forward")
    for l in range(len(chans) - 1):
        for j, nm in enumerate(["w", "b", "gamma", "beta"]):
            if j == 1:
                continue
            assert_close(ps[4 * l + j].grad.cpu().numpy(), pt[4 * l + j].grad.cpu().numpy(), 2e-5, "stream vs tiled d%s %d" % (nm, l))
    assert_close(xs.grad.cpu().numpy(), xt.grad.cpu().numpy(), 2e-5, "stream vs tiled dx")
    # float64 torch reference
    p64 = [p.detach().double().requires_grad_(True) for p in ps]
    x64 = xs.detach().double().requires_grad_(True)
    ref = torch_ref.stack_max(x64, [tuple(p64[4 * l:4 * l + 4]) for l in range(len(chans) - 1)], K, 1e-5)
    assert_close(outs.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, "stream forward vs f64")
    ref.backward(gout.double())
    for l in range(len(chans) - 1):
        for j, nm in enumerate(["w", "b", "gamma", "beta"]):
            if j == 1:
                continue
            assert_close(ps[4 * l + j].grad.cpu().numpy(), p64[4 * l + j].grad.cpu().numpy(), 2e-4, "stream d%s %d vs f64" % (nm, l))
    assert_close(xs.grad.cpu().numpy(), x64.grad.cpu().numpy(), 2e-4, "stream dx vs f64")


def test_stream_is_taken_and_deterministic_at_config2_size(dev, stream_knobs):
    """SA2 of BASELINE config 2 (M = 32*128*64 = 262144 rows, 128 -> 128 -> 256): the default dispatch must pick the streaming
    kernel (kernel-family profiler sees the launches either way; here: results equal the tiled kernel's and repeat exactly)."""
    G, K, chans = 4096, 64, [128, 128, 256]
    _, _, _, o1 = _run(dev, G, K, chans, 11, True)
    _, _, _, o2 = _run(dev, G, K, chans, 11, True)
    assert torch.equal(o1, o2)                                  # static tile assignment, no atomics
    _, _, _, o3 = _run(dev, G, K, chans, 11, False)
    assert torch.isfinite(o1).all()
    assert_close(o1.detach().cpu().numpy(), o3.detach().cpu().numpy(), 2e-6, "full-size stream vs tiled")


@pytest.mark.parametrize("chans,K", [([64, 64, 128], 32), ([64, 64, 64], 16), ([64, 128, 128], 64), ([64, 128, 256], 64), ([128, 128, 256, 512], 16)])
def test_dw_row_streaming_matches_staged_kernels(dev, chans, K):
    """dW of layers with a 64-channel input on the row-streaming kernel (PAPC_DW_ROWS=1: dense and max-pooled layers) against the
    LDS-staged kernels (=0): the same exact-split products in another summation order, ragged last chunk included."""
    import ctypes
    from papc_amd import _lib
    from papc_amd.mlp import StackSpec, shared_mlp_max
    lib = _lib.load()
    G = 1000                      # M = G*K rows: not a multiple of the chunk size
    M = G * K
    torch.manual_seed(11)
    x = torch.randn(M, chans[0], device=dev)
    ps = []
    for cin, cout in zip(chans[:-1], chans[1:]):
        ps += [torch.randn(cout, cin, device=dev) * (2.0 / cin) ** 0.5, torch.zeros(cout, device=dev), torch.rand(cout, device=dev) + 0.5,
               torch.randn(cout, device=dev) * 0.1]
    z = torch.zeros(1, 1, 3, device=dev)
    gout = torch.randn(G, chans[-1], device=dev)
    grads = {}
    try:
        for flav in (0, 1):
            _lib.check(lib.papc_knob_set(b"PAPC_DW_ROWS", flav), "knob")
            _lib.check(lib.papc_knob_set(b"PAPC_DW_ROWSX", flav), "knob")
            prm = [p.clone().requires_grad_(True) for p in ps]
            out = shared_mlp_max(StackSpec(1, M, G, K, chans[0] - 3, True), None, z, z, None, None, prm, x_rows=x)
            out.backward(gout)
            grads[flav] = [p.grad.clone() for p in prm]
    finally:
        _lib.check(lib.papc_knob_set(b"PAPC_DW_ROWS", 1), "knob")
        _lib.check(lib.papc_knob_set(b"PAPC_DW_ROWSX", 1), "knob")
    for a, b in zip(grads[0], grads[1]):
        assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()) + 1e-7


@pytest.mark.parametrize("seed", [40, 41, 42, 43])
@pytest.mark.parametrize("chans,K", [([64, 64, 128], 32), ([64, 128, 128], 64), ([64, 64, 64], 16),
                                     ([64, 128, 128, 256], 64),      # 128-channel inputs: dw_rowsx, four waves (128 -> 128 dense) and eight (128 -> 256 under the max)
                                     ([128, 128, 128], 32),          # ... 128 -> 128 under the max; 8 blocks per chunk: ring remainder of two
                                     ([64, 128, 256, 64], 48)])      # ... 128 -> 256 dense; K = 48, 12 / 6 blocks per chunk
def test_dw_row_streaming_vs_f64(dev, chans, K, seed):
    """dw_rows_kernel / dw_rowsx_kernel on their own shapes (64- and 128-channel BN+ReLU inputs; dense layers and the max-pooled last layer)
    straight against float64 torch autograd -- not through the staged kernels: every weight / norm gradient at 2e-4 of max |grad|, on
    ARBITRARY weight seeds: the float64 reference takes the kernel's own max / ReLU decisions (tests/torch_ref.py::stack_routed), so a
    decision within fp32 rounding of a tie -- about every second seed has one among its 8 M -- cannot hide or fake an arithmetic error."""
    from tests.util import kernel_decisions
    lib = _lib.load()
    v = _lib.ctypes.c_int(0)
    _lib.check(lib.papc_knob_get(b"PAPC_DW_ROWS", _lib.ctypes.byref(v)), "papc_knob_get")
    assert v.value == 1, "the row-streaming dW must be the default path"
    G = 1000
    M = G * K
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.normal(size=(M, chans[0])).astype(np.float32)).to(dev)
    ws = seeded_weights(chans, seed)
    ps = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
    z = torch.zeros(1, 1, 3, device=dev)
    gout = torch.from_numpy(rng.normal(size=(G, chans[-1])).astype(np.float32)).to(dev)
    out = shared_mlp_max(StackSpec(1, M, G, K, chans[0] - 3, True), None, z, z, None, None, ps, x_rows=x)
    argmax, alive, masks = kernel_decisions(out)
    out.backward(gout)
    p64 = [p.detach().double().requires_grad_(True) for p in ps]
    ref, stats = torch_ref.stack_routed(x.double(), [tuple(p64[4 * l:4 * l + 4]) for l in range(len(chans) - 1)], K, 1e-5, argmax, alive, masks)
    print("decisions that differ from float64's own:", stats)
    assert_close(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, "forward")
    ref.backward(gout.double())
    for l in range(len(chans) - 1):
        for j, nm in enumerate(["w", "b", "gamma", "beta"]):
            if j != 1:
                assert_close(ps[4 * l + j].grad.cpu().numpy(), p64[4 * l + j].grad.cpu().numpy(), 2e-4, "dw_rows d%s layer %d vs f64" % (nm, l))
