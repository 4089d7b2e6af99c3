"""GPU tests of the callers: PointNet++ SSG / MSG classifiers, PointNet-Basic, golden SA activations, a train step."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import reference_np as R
from papc_amd.distributed import FlatAdam, FlatParams
from papc_amd.layers import PointNetSetAbstraction, PointNetSetAbstractionMsg
from papc_amd.models import PointNet2_MSG_Clas, PointNet2_SSG_Clas, PointNet_Basic_Clas
from papc_amd.synthetic import make_clouds, make_labels, make_start_idx
from tests.util import assert_close, copy_into_model, seeded_model_state, seeded_weights

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load_stack(convs, bns, ws):
    with torch.no_grad():
        for conv, bn, (w, b, g, bt) in zip(convs, bns, ws):
            conv.weight.copy_(torch.from_numpy(w).reshape(conv.weight.shape)); conv.bias.copy_(torch.from_numpy(b))
            bn.weight.copy_(torch.from_numpy(g)); bn.bias.copy_(torch.from_numpy(bt))


def test_golden_sampling_fixtures(dev):
    from papc_amd import functional as F
    for name, B, N, S in [("sampling_b2_n1024.npz", 2, 1024, 128), ("sampling_b1_n4096.npz", 1, 4096, 512)]:
        g = np.load(os.path.join(GOLD, name))
        x = torch.from_numpy(make_clouds(B, N, int(g["seed"]))).to(dev).transpose(1, 2)
        st = torch.from_numpy(g["start_idx"]).to(dev)
        idx, new_xyz = F._fps_raw(x, S, st)
        assert np.array_equal(idx.cpu().numpy(), g["fps_idx"])
        idx2, _ = F._fps_raw(x, S, st, init_dist=1e10)
        assert np.array_equal(idx2.cpu().numpy(), g["fps_idx_init1e10"])
        for r, k in [(0.1, 16), (0.2, 32), (0.4, 64), (0.8, 128)]:
            got = F.query_ball_point(r, k, x, new_xyz).cpu().numpy()
            assert np.array_equal(got, g["bq_r%s_k%d" % (str(r).replace(".", "p"), k)])


def test_golden_sa_chain(dev):
    g = np.load(os.path.join(GOLD, "sa_b2_n1024.npz"))
    x = torch.from_numpy(make_clouds(2, 1024, int(g["seed"]))).to(dev)
    sa1 = PointNetSetAbstraction(128, 0.2, 32, 3, [64, 64, 128], False).to(dev)
    sa2 = PointNetSetAbstraction(32, 0.4, 64, 131, [128, 128, 256], False).to(dev)
    _load_stack(sa1.mlp_convs, sa1.mlp_bns, seeded_weights([3, 64, 64, 128], 1))
    _load_stack(sa2.mlp_convs, sa2.mlp_bns, seeded_weights([131, 128, 128, 256], 2))
    l1_xyz, l1 = sa1(x, None, torch.from_numpy(g["start1"]).to(dev))
    assert np.array_equal(l1_xyz.cpu().numpy(), g["l1_xyz"])
    assert_close(l1.detach().cpu().numpy(), g["l1_points"], 1e-5, "golden SA1")
    # feed the GOLDEN l1 features so SA2 is checked on its own (chained error would still be ~1e-6)
    l2_xyz, l2 = sa2(l1_xyz, torch.from_numpy(g["l1_points"]).to(dev), torch.from_numpy(g["start2"]).to(dev))
    assert np.array_equal(l2_xyz.cpu().numpy(), g["l2_xyz"])
    assert_close(l2.detach().cpu().numpy(), g["l2_points"], 1e-5, "golden SA2")


def test_ssg_model_forward_vs_oracle_chain(dev):
    B, N = 2, 1024
    x = make_clouds(B, N, 3)
    s1, s2 = make_start_idx(B, N, 3), make_start_idx(B, 512, 4)
    torch.manual_seed(0)
    m = PointNet2_SSG_Clas(num_classes=16).to(dev)
    m.eval()                                  # head in eval (dropout off); SA BNs always use batch stats, as the source
    specs = [(512, 0.2, 32, 3, [64, 64, 128], False), (128, 0.4, 64, 131, [128, 128, 256], False), (None, None, None, 259, [256, 512, 1024], True)]
    feats, xyz = None, x
    for sa, (npnt, r, k, cin, mlp, ga), st in zip([m.sa1, m.sa2, m.sa3], specs, [s1, s2, None]):
        ws = [(c.weight.detach().cpu().numpy().reshape(c.weight.shape[0], -1), c.bias.detach().cpu().numpy(),
               bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy()) for c, bn in zip(sa.mlp_convs, sa.mlp_bns)]
        ora = R.PointNetSetAbstraction(npnt, r, k, cin, mlp, ga, ws)
        xyz, feats = ora.forward(xyz, feats, st, f64=True)
        feats = feats.astype(np.float32)
    with torch.no_grad():
        l1_xyz, l1 = m.sa1(torch.from_numpy(x).to(dev), None, torch.from_numpy(s1).to(dev))
        l2_xyz, l2 = m.sa2(l1_xyz, l1, torch.from_numpy(s2).to(dev))
        _, l3 = m.sa3(l2_xyz, l2)
        logits = m(torch.from_numpy(x).to(dev), (torch.from_numpy(s1).to(dev), torch.from_numpy(s2).to(dev)))
    # nine conv+BN layers chained in fp32 vs an all-f64 chain: rounding compounds through the BN divisions (each SA
    # layer on identical inputs is within 1e-5, see test_golden_sa_chain); the chain itself is held to 2e-4
    assert_close(l3.cpu().numpy(), feats, 2e-4, "l3_points (three chained SA layers) vs f64 oracle")
    assert tuple(logits.shape) == (B, 16) and torch.isfinite(logits).all()


def test_msg_and_basic_models_run(dev):
    x = torch.from_numpy(make_clouds(2, 1024, 5)).to(dev)
    m = PointNet2_MSG_Clas(num_classes=16).to(dev)
    out = m(x, (torch.tensor([1, 2], device=dev), torch.tensor([3, 4], device=dev)))
    assert tuple(out.shape) == (2, 16) and torch.isfinite(out).all()
    out.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # PointNet-Basic (BASELINE config 0 shapes: B=8, N=1024) against the oracle's row stack
    xb = make_clouds(8, 1024, 6)
    pb = PointNet_Basic_Clas(num_classes=16).to(dev)
    pb.eval()
    ws = [(c.weight.detach().cpu().numpy().reshape(c.weight.shape[0], -1), c.bias.detach().cpu().numpy(),
           bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy()) for c, bn in zip(pb.convs, pb.bns)]
    rows = np.ascontiguousarray(xb.transpose(0, 2, 1)).reshape(8 * 1024, 3)
    ref = R.mlp_stack_rows(rows, ws, f64=True).reshape(8, 1024, -1).max(1)
    from papc_amd.mlp import StackSpec, shared_mlp_max
    t = torch.from_numpy(xb).to(dev)
    with torch.no_grad():
        ps = []
        for c, bn in zip(pb.convs, pb.bns):
            ps += [c.weight, c.bias, bn.weight, bn.bias]
        feat = shared_mlp_max(StackSpec(8, 1024, 1, 1024, 0, True), None, t.transpose(1, 2), torch.zeros(8, 1, 3, device=dev), None, None, ps)
        logits = pb(t)
    assert_close(feat.cpu().numpy(), ref, 1e-5, "PointNet-Basic global feature vs f64 oracle")
    assert tuple(logits.shape) == (8, 16)


def test_train_step_decreases_loss_and_quirks_mode(dev):
    B, N = 4, 1024
    x = torch.from_numpy(make_clouds(B, N, 1)).to(dev)
    y = torch.from_numpy(make_labels(B, 16, 1)).reshape(-1).to(dev)
    st = (torch.from_numpy(make_start_idx(B, N, 1)).to(dev), torch.from_numpy(make_start_idx(B, 512, 2)).to(dev))
    torch.manual_seed(0)
    m = PointNet2_SSG_Clas().to(dev)
    m.train()
    m.drop1.p = m.drop2.p = 0.0                 # deterministic head for the monotonicity check
    flat = FlatParams(m)
    opt = FlatAdam(flat, lr=1e-3, weight_decay=0.0)
    losses = []
    for _ in range(8):
        flat.zero_grad()
        loss = TF.cross_entropy(m(x, st), y)
        loss.backward()
        opt.step(flat.allreduce_grads())
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses
    assert m.sa1.mlp_convs[0].weight.grad.abs().sum() > 0        # gradient reaches SA1 through both gathers
    q = PointNet2_SSG_Clas(reference_quirks=True).to(dev)        # the source's behaviour: SA params frozen, gathers cut
    TF.cross_entropy(q(x, st), y).backward()
    assert q.sa1.mlp_convs[0].weight.grad is None and q.fc1.weight.grad is not None


def test_sampling_plan_equals_inline(dev):
    """forward(plan=plan_sampling(x)) must equal forward(x) bit for bit (same kernels, only scheduled earlier)."""
    B, N = 2, 1024
    x = torch.from_numpy(make_clouds(B, N, 8)).to(dev)
    st = (torch.from_numpy(make_start_idx(B, N, 8)).to(dev), torch.from_numpy(make_start_idx(B, 512, 9)).to(dev))
    torch.manual_seed(0)
    m = PointNet2_SSG_Clas().to(dev)
    m.eval()
    with torch.no_grad():
        a = m(x, st)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            plan = m.plan_sampling(x, st)
        torch.cuda.current_stream().wait_stream(side)
        b = m(x, st, plan=plan)
    assert torch.equal(a, b)
    assert plan[0][1].dtype == torch.int32 and tuple(plan[0][0].shape) == (B, 512, 3) and tuple(plan[1][1].shape) == (B, 128, 64)


def test_hipgraph_replay_matches_eager(dev):
    """zero_grad + forward + loss + backward captured into one hipGraph (what bench.py replays) gives the eager step's loss
    and gradients (same weights, same batch; float atomics in the scatter-add make the two agree to rounding, not bitwise)."""
    B, N = 4, 1024
    x = torch.from_numpy(make_clouds(B, N, 5)).to(dev)
    y = torch.from_numpy(make_labels(B, 16, 5)).reshape(-1).to(dev)
    st = (torch.from_numpy(make_start_idx(B, N, 5)).to(dev), torch.from_numpy(make_start_idx(B, 512, 6)).to(dev))
    torch.manual_seed(0)
    m = PointNet2_SSG_Clas().to(dev)
    m.train()
    m.drop1.p = m.drop2.p = 0.0
    flat = FlatParams(m)

    def fwd_bwd():
        flat.zero_grad()
        loss = TF.cross_entropy(m(x, st), y)
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            loss_e = fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    grad_e = flat.grad.clone()
    bn_e = m.sa1.mlp_bns[0].running_mean.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()
    flat.grad.fill_(123.0)                                   # the replay must rewrite all of it
    g.replay()
    torch.cuda.synchronize()
    assert abs(float(static_loss) - float(loss_e)) <= 1e-5 * abs(float(loss_e))
    scale = float(grad_e.abs().max())
    assert float((flat.grad - grad_e).abs().max()) <= 2e-4 * scale
    assert not torch.equal(m.sa1.mlp_bns[0].running_mean, bn_e)   # the running statistics keep moving under replay


def test_grad_targets_follow_the_optimizer():
    """In-place gradient accumulation must track the CURRENT .grad tensors: zero_grad(set_to_none=False) keeps them (kernels add in
    place), zero_grad(set_to_none=True) drops them (gradients go back through autograd) -- switching between the two must not
    leave gradients in orphaned buffers."""
    import torch
    from papc_amd.models import PointNet2_SSG_Clas
    from papc_amd.head import softmax_cross_entropy
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = PointNet2_SSG_Clas(num_classes=16).to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    x = torch.randn(4, 3, 512, device=dev)
    y = torch.randint(0, 16, (4,), device=dev)
    for mode in (True, False, False, True, True):
        opt.zero_grad(set_to_none=mode)
        before = {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}
        loss = softmax_cross_entropy(model(x), y)
        loss.backward()
        for n, p in model.named_parameters():
            if not p.requires_grad:
                continue
            assert p.grad is not None and torch.isfinite(p.grad).all(), n
            if n.endswith("weight") and ("conv" in n or "fc" in n or "mlp" in n.lower()):
                assert float(p.grad.abs().max()) > 0, (n, mode)
        opt.step()


def _np_stack(convs, bns):
    return [(c.weight.detach().cpu().numpy().reshape(c.weight.shape[0], -1), c.bias.detach().cpu().numpy(),
             bn.weight.detach().cpu().numpy(), bn.bias.detach().cpu().numpy()) for c, bn in zip(convs, bns)]


def test_msg_clas_model_forward_vs_oracle_layers(dev):
    """PointNet2_MSG_Clas (/root/reference/PAPC/models/classify/pointnet2/pointnet2.py:43-75; nsample lists [16, 32, 128] and
    [32, 64, 128]: the K = 16 branch runs without the fused group-max epilogue, the K = 128 / group_all stacks on the planes kernels):
    every set-abstraction level against the f64 oracle on the oracle's own inputs (1e-5), centroids exact."""
    B, N = 2, 1024
    x = make_clouds(B, N, 21)
    s1, s2 = make_start_idx(B, N, 21), make_start_idx(B, 512, 22)
    torch.manual_seed(3)
    m = PointNet2_MSG_Clas(num_classes=16).to(dev)
    copy_into_model(m, seeded_model_state(m, 17))        # non-trivial norm weights, including negative gammas
    m.eval()
    o1 = R.PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], 0, [[32, 32, 64], [64, 64, 128], [64, 96, 128]],
                                     [_np_stack(m.sa1.conv_blocks[i], m.sa1.bn_blocks[i]) for i in range(3)])
    o2 = R.PointNetSetAbstractionMsg(128, [0.2, 0.4, 0.8], [32, 64, 128], 320, [[64, 64, 128], [128, 128, 256], [128, 128, 256]],
                                     [_np_stack(m.sa2.conv_blocks[i], m.sa2.bn_blocks[i]) for i in range(3)])
    o3 = R.PointNetSetAbstraction(None, None, None, 643, [256, 512, 1024], True, _np_stack(m.sa3.mlp_convs, m.sa3.mlp_bns))
    r1_xyz, r1 = o1.forward(x, None, s1, f64=True)
    r2_xyz, r2 = o2.forward(r1_xyz, r1.astype(np.float32), s2, f64=True)
    _, r3 = o3.forward(r2_xyz, r2.astype(np.float32), f64=True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    with torch.no_grad():
        l1_xyz, l1 = m.sa1(t(x), None, t(s1))
        l2_xyz, l2 = m.sa2(t(r1_xyz), t(r1.astype(np.float32)), t(s2))          # each level on the ORACLE's input
        _, l3 = m.sa3(t(r2_xyz), t(r2.astype(np.float32)))
        logits = m(t(x), (t(s1), t(s2)))
    assert np.array_equal(l1_xyz.cpu().numpy(), r1_xyz) and np.array_equal(l2_xyz.cpu().numpy(), r2_xyz)
    e1 = assert_close(l1.cpu().numpy(), r1, 1e-5, "MSG clas SA1 (K = 16 / 32 / 128) vs f64 oracle")
    e2 = assert_close(l2.cpu().numpy(), r2, 1e-5, "MSG clas SA2 vs f64 oracle")
    e3 = assert_close(l3.cpu().numpy(), r3, 1e-5, "MSG clas SA3 (group_all, 643 channels) vs f64 oracle")
    print("MSG clas rel err: SA1 %.1e SA2 %.1e SA3 %.1e" % (e1, e2, e3))
    assert tuple(logits.shape) == (B, 16) and torch.isfinite(logits).all()


def test_msg_clas_sa1_backward_vs_f64(dev):
    """one backward through the MSG classifier's first level (branches K = 16, 32, 128, feats-first rows :266-267) against float64
    torch autograd on the same neighbour lists: every parameter gradient and the gradient of the input features, 2e-4"""
    from papc_amd import functional as F_
    from tests import torch_ref
    B, N, S, D = 2, 1024, 128, 6
    x = make_clouds(B, N, 33)
    rng = np.random.default_rng(5)
    pts = torch.from_numpy(rng.normal(size=(B, D, N)).astype(np.float32)).to(dev).requires_grad_(True)
    radii, ks, mlps = [0.1, 0.2, 0.4], [16, 32, 128], [[32, 32, 64], [64, 64, 128], [64, 96, 128]]
    layer = PointNetSetAbstractionMsg(S, radii, ks, D, mlps).to(dev)
    copy_into_model(layer, seeded_model_state(layer, 41))
    xt = torch.from_numpy(x).to(dev)
    st = torch.from_numpy(make_start_idx(B, N, 33)).to(dev)
    _, out = layer(xt, pts, st)
    gout = torch.from_numpy(rng.normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
    out.backward(gout)
    # float64 reference on the kernels' own (index-exact, tests/test_gpu_sampling.py) neighbour lists
    xyz = xt.transpose(1, 2).contiguous()
    _, new_xyz = F_._fps_raw(xyz, S, st)
    idxs = F_._ball_query_raw(radii, ks, xyz, new_xyz)
    f64 = pts.detach().double().requires_grad_(True)
    outs, p64s = [], []
    for i, K in enumerate(ks):
        p64 = []
        for c, bn in zip(layer.conv_blocks[i], layer.bn_blocks[i]):
            p64 += [c.weight.detach().double().reshape(c.weight.shape[0], -1).requires_grad_(True), c.bias.detach().double().requires_grad_(True),
                    bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)]
        rows = torch_ref.group(xyz.double(), new_xyz.double(), f64.transpose(1, 2), idxs[i], False).reshape(B * S * K, D + 3)
        outs.append(torch_ref.stack_max(rows, [tuple(p64[4 * l:4 * l + 4]) for l in range(3)], K, 1e-5).reshape(B, S, -1))
        p64s.append(p64)
    ref = torch.cat(outs, 2).transpose(1, 2)
    assert_close(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, "MSG SA1 forward vs f64")
    ref.backward(gout.double())
    for i in range(3):
        got = []
        for c, bn in zip(layer.conv_blocks[i], layer.bn_blocks[i]):
            got += [c.weight.grad.reshape(c.weight.shape[0], -1), c.bias.grad, bn.weight.grad, bn.bias.grad]
        for j, (g, w) in enumerate(zip(got, p64s[i])):
            if j % 4 == 1:
                continue                                   # conv bias under a train-mode BN: true gradient 0
            assert_close(g.cpu().numpy(), w.grad.cpu().numpy(), 2e-4, "MSG SA1 branch %d (K = %d) param %d" % (i, ks[i], j))
    assert_close(pts.grad.cpu().numpy(), f64.grad.cpu().numpy(), 2e-4, "MSG SA1 d(points)")


def test_full_size_config2_sa_vs_oracle(dev):
    """BASELINE configs[1] at its own size (B=32, N=4096): SA1 (M = 524 288 rows, 3 -> 64 -> 64 -> 128: moment-path first layer, row-streaming
    second, no-store max layer) and SA2 (M = 262 144 rows, 131 -> 128 -> 128 -> 256: gather-add first layer, row-streaming + fused group max)
    forward activations against the float64 oracle at 1e-5 -- the kernels and sizes bench.py times, not a scaled-down stand-in.
    pointnet2_basic_layers.py:194-221, classify/pointnet2/pointnet2.py:11-15."""
    B, N = 32, 4096
    x = make_clouds(B, N, 1234)
    s1, s2 = make_start_idx(B, N, 1234), make_start_idx(B, 512, 1235)
    w1, w2 = seeded_weights([3, 64, 64, 128], 31), seeded_weights([131, 128, 128, 256], 32)
    ora1 = R.PointNetSetAbstraction(512, 0.2, 32, 3, [64, 64, 128], False, w1)
    ref_xyz1, ref1 = ora1.forward(x, None, s1, f64=True)
    sa1 = PointNetSetAbstraction(512, 0.2, 32, 3, [64, 64, 128], False).to(dev)
    sa2 = PointNetSetAbstraction(128, 0.4, 64, 131, [128, 128, 256], False).to(dev)
    _load_stack(sa1.mlp_convs, sa1.mlp_bns, w1)
    _load_stack(sa2.mlp_convs, sa2.mlp_bns, w2)
    with torch.no_grad():
        l1_xyz, l1 = sa1(torch.from_numpy(x).to(dev), None, torch.from_numpy(s1).to(dev))
    assert np.array_equal(l1_xyz.cpu().numpy(), ref_xyz1)
    assert_close(l1.cpu().numpy(), ref1, 1e-5, "config-2 SA1 (M = 524288) vs f64 oracle")
    # SA2 on the ORACLE's l1 features, so that it is held on its own
    l1_ref = ref1.astype(np.float32)
    ora2 = R.PointNetSetAbstraction(128, 0.4, 64, 131, [128, 128, 256], False, w2)
    ref_xyz2, ref2 = ora2.forward(ref_xyz1, l1_ref, s2, f64=True)
    with torch.no_grad():
        l2_xyz, l2 = sa2(torch.from_numpy(ref_xyz1).to(dev), torch.from_numpy(l1_ref).to(dev), torch.from_numpy(s2).to(dev))
    assert np.array_equal(l2_xyz.cpu().numpy(), ref_xyz2)
    assert_close(l2.cpu().numpy(), ref2, 1e-5, "config-2 SA2 (M = 262144) vs f64 oracle")


def test_msg_branch_streams_equal_serial_branches(dev):
    """PointNetSetAbstractionMsg (pointnet2_basic_layers.py:264-280) with its radius branches on parallel streams (`layer.branch_streams`,
    on by default) against the serial branch order: the branches are independent until the concatenation, so outputs and parameter gradients
    are bit-identical (the feature gradient is a float-atomic sum inside each branch and an autograd sum over the branches: tolerance) --
    eagerly and inside a captured hipGraph, where the branch streams become parallel graph branches."""
    from papc_amd import layers
    B, N, S = 4, 1024, 256
    radii, ks, mlps = [0.1, 0.2, 0.4], [16, 32, 64], [[32, 32, 64], [64, 64, 128], [64, 96, 128]]
    x = torch.from_numpy(make_clouds(B, N, 21)).to(dev)
    st = torch.from_numpy(make_start_idx(B, N, 21)).to(dev)
    gout = torch.randn(B, 320, S, device=dev, generator=torch.Generator(device=dev).manual_seed(3))

    def run(par, graph):
        if True:
            torch.manual_seed(5)
            layer = PointNetSetAbstractionMsg(S, radii, ks, 3, mlps).to(dev)
            layer.branch_streams = par          # (per-layer switch: no module-level state is touched)
            pts = x.clone().requires_grad_(True)

            def fwd_bwd():
                _, out = layer(x, pts, st)
                out.backward(gout)
                return out

            if graph:
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    fwd_bwd()                                    # (lazily created streams / constants outside the capture)
                torch.cuda.current_stream().wait_stream(s)
                for p in list(layer.parameters()) + [pts]:
                    p.grad = None
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out = fwd_bwd()
                for p in list(layer.parameters()) + [pts]:
                    p.grad.zero_()
                g.replay()
            else:
                out = fwd_bwd()
            torch.cuda.synchronize()
            return out.detach().clone(), [p.grad.clone() for p in layer.parameters()], pts.grad.clone()

    o0, g0, f0 = run(False, False)
    for par, graph in ((True, False), (True, True)):
        o1, g1, f1 = run(par, graph)
        assert torch.equal(o0, o1), "outputs differ (graph=%s)" % graph
        for a, b in zip(g0, g1):
            assert torch.equal(a, b), "a parameter gradient differs (graph=%s)" % graph
        assert float((f0 - f1).abs().max()) <= 1e-5 * float(f0.abs().max())
