"""The compacted form of a grouped stack (csrc/compact.hip, papc_amd/compact.py): distinct neighbours only, one weight per group.

query_ball_point pads every neighbourhood with copies of its first hit (pointnet2_basic_layers.py:118-124); the compacted stack must be the
SAME function with the SAME gradients as the padded one (:214-219) -- held here to the padded HIP path, to float64 torch autograd on the
padded rows, and to a numpy restatement of the layout."""
import numpy as np
import pytest
import torch

from papc_amd import _lib
from papc_amd import compact as C
from papc_amd import functional as F
from papc_amd.layers import PointNetSetAbstraction
from papc_amd.mlp import StackSpec, shared_mlp_max
from papc_amd.synthetic import make_clouds, make_start_idx
from tests import torch_ref
from tests.util import assert_close, seeded_weights

pytestmark = pytest.mark.gpu


def _lists(dev, B, N, S, K, radius, seed):
    x = make_clouds(B, N, seed)
    xyz = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 1))).to(dev)
    st = torch.from_numpy(make_start_idx(B, N, seed)).to(dev)
    _, new_xyz = F._fps_raw(xyz, S, st)
    idx = F._ball_query_raw([radius], [K], xyz, new_xyz)[0]
    return xyz, new_xyz, idx


@pytest.mark.parametrize("B,N,S,K,radius", [(4, 512, 128, 64, 0.4), (2, 1024, 64, 32, 0.2), (1, 256, 16, 8, 0.05), (3, 300, 20, 16, 2.5)])
def test_compact_plan_matches_numpy(dev, B, N, S, K, radius):
    """layout: per group the distinct neighbours in list order, then copies of the first up to a multiple of 8 rows; the last group takes the
    tail up to a multiple of 128; weights sum to nsample per group"""
    _, _, idx = _lists(dev, B, N, S, K, radius, 3)
    cp = C.plan(idx)
    torch.cuda.synchronize()
    ii = idx.cpu().numpy().reshape(B * S, K)
    G = B * S
    cnt = np.array([1 + int((row[1:] != row[0]).sum()) for row in ii])
    c8 = (cnt + 7) // 8 * 8
    start = np.concatenate([[0], np.cumsum(c8)])
    total = int(start[-1])
    rows = (total + 127) // 128 * 128
    assert cp.rows.cpu().tolist() == [rows, total]
    assert np.array_equal(cp.start.cpu().numpy(), start)
    cidx, wrow, seg, coef = cp.cidx.cpu().numpy(), cp.wrow.cpu().numpy(), cp.seg_grp.cpu().numpy(), cp.coef.cpu().numpy()
    for g in range(G):
        n = int(c8[g]) + (rows - total if g == G - 1 else 0)
        s0 = int(start[g])
        want = np.where(np.arange(n) < cnt[g], ii[g, np.minimum(np.arange(n), K - 1)], ii[g, 0])
        assert np.array_equal(cidx[s0:s0 + n], want), g
        assert coef[g] == K - n and wrow[s0] == 1 + K - n and (wrow[s0 + 1:s0 + n] == 1).all()
        assert (seg[s0 // 8:(s0 + n) // 8] == g).all()
    assert abs(float(wrow[:rows].sum()) - G * K) < 0.5           # multiplicities add up to the padded row count
    if radius > 2:                                                # every ball full: nothing to compact
        assert total == G * K


def _sa2_like(dev, B, seed, compact):
    """an SA2-shaped stack (gather-add first layer, 128 -> 128 -> 256, nsample 64) on B clouds of 512 points"""
    N, S, K, D = 512, 128, 64, 128
    xyz, new_xyz, idx = _lists(dev, B, N, S, K, 0.4, seed)
    rng = np.random.default_rng(seed)
    feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev).requires_grad_(True)
    ws = seeded_weights([D + 3, 128, 128, 256], 50 + seed)
    params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
    spec = StackSpec(B, N, S, K, D, True)
    if compact:
        spec.compact = C.plan(idx)
    out = shared_mlp_max(spec, None, xyz, new_xyz, feats, idx, params)
    return xyz, new_xyz, idx, feats, params, spec, out


@pytest.mark.parametrize("B,seed", [(8, 1), (16, 2)])
def test_compacted_stack_equals_padded_stack_and_f64(dev, B, seed):
    """forward 2e-6 / gradients 2e-5 against the padded HIP path (same products per row, other summation order in the statistics), and
    1e-5 / 2e-4 against float64 torch autograd on the PADDED rows routed through the kernel's own decisions"""
    from tests.util import kernel_decisions
    rng = np.random.default_rng(100 + seed)
    res = {}
    for compact in (False, True):
        xyz, new_xyz, idx, feats, params, spec, out = _sa2_like(dev, B, seed, compact)
        node = out.grad_fn
        assert (node.compact is not None) == compact, "the compacted path was %staken" % ("not " if compact else "")
        dec = kernel_decisions(out)
        gout = torch.from_numpy(np.random.default_rng(7).normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
        out.backward(gout)
        res[compact] = (out.detach(), [p.grad.clone() for p in params], feats.grad.clone(), dec, spec)
    o0, g0, f0, _, _ = res[False]
    o1, g1, f1, dec, spec = res[True]
    frac = spec.compact.fraction()
    print("physical rows / padded rows: %.3f" % frac)
    assert frac < 0.8
    assert_close(o1.cpu().numpy(), o0.cpu().numpy(), 2e-6, "compacted vs padded forward")
    names = ["w", "b", "gamma", "beta"]
    for i, (a, b) in enumerate(zip(g1, g0)):
        if i % 4 == 1:
            continue                                                  # (conv bias under a train-mode BN: exactly 0 on both paths)
        assert_close(a.cpu().numpy(), b.cpu().numpy(), 2e-5, "compacted vs padded d%s layer %d" % (names[i % 4], i // 4))
    assert_close(f1.cpu().numpy(), f0.cpu().numpy(), 2e-5, "compacted vs padded dfeats")
    # float64 on the padded rows, routed through the compacted kernel's decisions
    argmax, alive, masks = dec
    xyz, new_xyz, idx, feats, params, _, _ = _sa2_like(dev, B, seed, False)
    p64 = [p.detach().double().requires_grad_(True) for p in params]
    f64 = feats.detach().double().requires_grad_(True)
    S, K, D = 128, 64, 128
    rows = torch_ref.group(xyz.double(), new_xyz.double(), f64, idx, True).reshape(B * S * K, D + 3)
    ref, stats = torch_ref.stack_routed(rows, [tuple(p64[4 * l:4 * l + 4]) for l in range(3)], K, 1e-5, argmax, alive, masks)
    print("decisions that differ from float64's own:", stats)
    assert_close(o1.cpu().numpy(), ref.detach().cpu().numpy(), 1e-5, "compacted forward vs f64")
    gout = torch.from_numpy(np.random.default_rng(7).normal(size=tuple(o1.shape)).astype(np.float32)).to(dev)
    ref.backward(gout.double())
    for i, (a, b) in enumerate(zip(g1, p64)):
        if i % 4 != 1:
            assert_close(a.cpu().numpy(), b.grad.cpu().numpy(), 2e-4, "compacted d%s layer %d vs f64" % (names[i % 4], i // 4))
    assert_close(f1.cpu().numpy(), f64.grad.cpu().numpy(), 2e-4, "compacted dfeats vs f64")


def test_compacted_layer_follows_the_device_side_row_count(dev):
    """one set of plan buffers, two different batches (what a replayed hipGraph does): the kernels take the row count from device memory"""
    B = 8
    layer = PointNetSetAbstraction(128, 0.4, 64, 131, [128, 128, 256], False).to(dev)
    layer.compact = True
    ref = PointNetSetAbstraction(128, 0.4, 64, 131, [128, 128, 256], False).to(dev)
    ref.compact = False
    ref.load_state_dict(layer.state_dict())
    bufs = None
    counts = []
    for seed in (11, 12):
        x = torch.from_numpy(make_clouds(B, 512, seed)).to(dev)
        if seed == 12:
            x = x * 0.6                                   # denser clouds: more neighbours per ball, another row count
        pts = torch.from_numpy(np.random.default_rng(seed).normal(size=(B, 128, 512)).astype(np.float32)).to(dev)
        st = torch.from_numpy(make_start_idx(B, 512, seed)).to(dev)
        plan = layer.sample(x, st, out=bufs)
        assert len(plan) == 12            # (new_xyz, idx) + the 7 compact-plan tensors + the 3 point-list tensors of a training layer
        if bufs is None:
            bufs = plan
        else:
            assert all(a.data_ptr() == b.data_ptr() for a, b in zip(plan, bufs))
        counts.append(int(plan[4][0].item()))
        with torch.no_grad():
            _, got = layer(x, pts, st, sampled=plan)
            _, want = ref(x, pts, st)
        assert_close(got.cpu().numpy(), want.cpu().numpy(), 2e-6, "compacted layer vs padded layer, batch %d" % seed)
    assert counts[0] != counts[1]


def test_auto_policy_measures_once(dev):
    """compact=None: the first sampling outside a capture decides from the data; full balls keep the padded path"""
    B = 8
    x = torch.from_numpy(make_clouds(B, 512, 5)).to(dev)
    st = torch.from_numpy(make_start_idx(B, 512, 5)).to(dev)
    sparse = PointNetSetAbstraction(128, 0.4, 64, 131, [128, 128, 256], False).to(dev)
    assert len(sparse.sample(x, st)) == 12 and sparse._compact_on is True       # compact plan + point lists
    full = PointNetSetAbstraction(128, 3.0, 64, 131, [128, 128, 256], False).to(dev)      # radius covers the cloud: every list is full
    assert len(full.sample(x, st)) == 5 and full._compact_on is False           # padded, with its point lists (compact.LISTS == 2: padded stacks too)
    sparse.eval()
    assert len(sparse.sample(x, st)) == 9                                        # no backward will follow: no lists
    full.eval()
    assert len(full.sample(x, st)) == 2
    other = PointNetSetAbstraction(128, 0.4, 64, 131, [128, 128, 128], False).to(dev)     # widths without a compacted flavour
    assert len(other.sample(x, st)) == 5 and other._compact_on is None


def test_weighted_statistics_equal_the_correction_launches(dev, monkeypatch):
    """A compacted stack's train-mode BatchNorm statistics are those of the PADDED tensor (pointnet2_basic_layers.py:118-124 pads with copies,
    :215-217 normalises over them): by default the producing kernels weigh their sums with the rows' multiplicities themselves
    (papc_mlp_gemm_rows_w_f32, papc_group_src.wstat); PAPC_WSTATS=0 keeps the unweighted sums + papc_bn_stats_corr_f32.  Same numbers up to
    the summation order."""
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PAPC_WSTATS", mode)
        _, _, _, feats, params, _, out = _sa2_like(dev, 8, 3, True)
        assert out.grad_fn.compact is not None
        gout = torch.from_numpy(np.random.default_rng(9).normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
        out.backward(gout)
        res[mode] = (out.detach().cpu().numpy(), [p.grad.cpu().numpy() for p in params], feats.grad.cpu().numpy())
    assert_close(res["1"][0], res["0"][0], 2e-6, "weighted statistics vs correction launches: forward")
    for i, (a, b) in enumerate(zip(res["1"][1], res["0"][1])):
        if i % 4 != 1:
            assert_close(a, b, 2e-5, "weighted statistics vs correction launches: gradient %d" % i)
    assert_close(res["1"][2], res["0"][2], 2e-5, "weighted statistics vs correction launches: dfeats")


def test_streamed_psel_is_bit_identical(dev, monkeypatch):
    """The compacted max layer's dX takes scale * p per (group, channel) from the array the BatchNorm-backward reduction already formed
    (papc_bwd_dy.psel) instead of testing the ReLU and scaling gout per row (PAPC_PSEL=0 / PAPC_SA_NO_PSEL): the same fp32 product either
    way, so the gradients downstream have the same bits (pointnet2_basic_layers.py:215-219 backward)."""
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("PAPC_PSEL", mode)
        _, _, _, feats, params, _, out = _sa2_like(dev, 8, 3, True)
        assert out.grad_fn.compact is not None
        gout = torch.from_numpy(np.random.default_rng(11).normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
        out.backward(gout)
        res[mode] = ([p.grad.cpu().numpy() for p in params], feats.grad.cpu().numpy())
    for i, (a, b) in enumerate(zip(res["1"][0], res["0"][0])):
        if i >= 4:       # layers 2 and 3: everything downstream of the dX in question is deterministic
            assert np.array_equal(a, b), "streamed psel: gradient %d differs" % i
        elif i % 4 != 1:  # the gather-add first layer sums its rows with float atomics: same numbers, not the same bits run to run
            assert_close(a, b, 2e-5, "streamed psel: gradient %d" % i)
    assert_close(res["1"][1], res["0"][1], 2e-5, "streamed psel: dfeats")


def _collapse_copies(ii, N):
    """padded layout as the list builder sees it: a group's padding copies of its first neighbour (rows k > 0 repeating idx[g, 0]) are ONE entry --
    the first copy's row with weight = their number; the others are in no list (point index -1 here)"""
    B, S, K = ii.shape
    flat = ii.reshape(B * S, K).astype(np.int64)
    rows_pt = flat.copy()
    w = np.ones((B * S, K), np.float32)
    for g in range(B * S):
        cp = np.nonzero((flat[g, 1:] == flat[g, 0]) & (flat[g, 0] >= 0) & (flat[g, 0] < N))[0] + 1
        if len(cp):
            rows_pt[g, cp[1:]] = -1
            w[g, cp[0]] = len(cp)
    return rows_pt.reshape(-1), np.repeat(np.arange(B * S), K), w.reshape(-1)


def _numpy_lists(B, N, S, K, xyz, new_xyz, rows_pt, rows_grp, w):
    """expected (prange, prow, pmeta): per (cloud, point) its rows in ascending order; rows_pt / rows_grp = point index / group of every physical
    row (point outside [0, N): in no list)"""
    n = len(rows_pt)
    b = rows_grp // S
    ok = (rows_pt >= 0) & (rows_pt < N)
    key = (b.astype(np.int64) * N + rows_pt)[ok]
    rws = np.arange(n)[ok]
    order = np.lexsort((rws, key))
    key, rws = key[order], rws[order]
    cnt = np.bincount(key, minlength=B * N)
    prange = np.zeros((B * N, 2), np.int64)
    # lists of cloud b start at the cloud's first physical row and are packed in point order
    first_row = np.array([np.min(np.arange(n)[b == c]) if (b == c).any() else 0 for c in range(B)])
    pos = {}
    out_rows = {}
    for c in range(B):
        run = int(first_row[c])
        for p in range(N):
            prange[c * N + p] = (run, run + cnt[c * N + p])
            run += cnt[c * N + p]
    meta = {}
    for k_, r in zip(key, rws):
        c, p = divmod(int(k_), N)
        d = xyz[c, p] - new_xyz.reshape(-1, 3)[rows_grp[r]]
        meta[r] = (d[0], d[1], d[2], w[r])
    return prange, key, rws, meta


@pytest.mark.parametrize("B,N,S,K,radius,compact", [(4, 512, 128, 64, 0.4, True), (4, 512, 128, 64, 0.4, False), (2, 1024, 64, 128, 0.3, True),
                                                    (3, 300, 20, 16, 0.25, False), (1, 256, 16, 8, 0.05, True)])
def test_point_lists_match_numpy(dev, B, N, S, K, radius, compact):
    """papc_point_lists_f32: the inverse of a grouping -- per source point its physical rows ASCENDING (the fixed summation order of the gather-add
    backward), xyz_j - centre and the row's multiplicity weight per entry -- for the compacted and the padded row layout, bit for bit against numpy."""
    xyz, new_xyz, idx = _lists(dev, B, N, S, K, radius, 5)
    cp = C.plan(idx) if compact else None
    pl = C.point_lists(xyz, new_xyz, idx, cp)
    torch.cuda.synchronize()
    if compact:
        rows = int(cp.rows[0])
        rows_pt = cp.cidx.cpu().numpy()[:rows].astype(np.int64)
        rows_grp = np.repeat(cp.seg_grp.cpu().numpy()[:rows // 8], 8).astype(np.int64)
        w = cp.wrow.cpu().numpy()[:rows]
    else:
        rows_pt, rows_grp, w = _collapse_copies(idx.cpu().numpy(), N)
    prange, key, rws, meta = _numpy_lists(B, N, S, K, xyz.cpu().numpy(), new_xyz.cpu().numpy(), rows_pt, rows_grp, w)
    got_range = pl.prange.cpu().numpy()
    assert np.array_equal(got_range, prange)
    prow, pmeta = pl.prow.cpu().numpy(), pl.pmeta.cpu().numpy()
    seen = 0
    for q in range(B * N):
        a, b_ = prange[q]
        want = rws[key == q] if b_ > a else np.zeros(0, np.int64)
        assert np.array_equal(prow[a:b_], want), q
        for e, r in zip(range(a, b_), want):
            assert tuple(pmeta[e]) == tuple(np.float32(v) for v in meta[int(r)]), (q, r)
        seen += b_ - a
    assert seen == int(((rows_pt >= 0) & (rows_pt < N)).sum())          # every valid row (padded: every distinct row + one entry per group's copies) in exactly one list
    assert pl.compact == compact
    # the lists' per-point moments (pmom: sum w | sum w d | sum w d d^T as 00, 01, 02, 11, 12, 22 | 0, 0), what the backward without y reads
    mom = pl.pmom.cpu().numpy()
    assert mom.shape == (B * N, 12) and not mom[:, 10:].any()
    want = np.zeros((B * N, 10))
    for q in range(B * N):
        e = pmeta[prange[q, 0]:prange[q, 1]].astype(np.float64)
        wq, dq = e[:, 3], e[:, :3]
        m2 = np.einsum("e,es,et->st", wq, dq, dq)
        want[q] = [wq.sum(), *(wq[:, None] * dq).sum(0), m2[0, 0], m2[0, 1], m2[0, 2], m2[1, 1], m2[1, 2], m2[2, 2]]
    assert np.abs(mom[:, :10] - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_point_lists_with_repeated_indices_inside_a_step(dev):
    """lists a caller made up: repeated indices that are NOT ball-query padding (the builder's lane-sweep ranking), out-of-range entries (in no
    list) and an index that equals the group's first entry in the middle of the list"""
    B, N, S, K = 2, 64, 4, 64
    rng = np.random.default_rng(3)
    ii = rng.integers(0, 12, size=(B, S, K)).astype(np.int32)          # 12 distinct values over 64 slots: every step is full of repeats
    ii[0, 1, 5] = N                                                      # the no-hit sentinel
    ii[1, 2, 7] = -1
    idx = torch.from_numpy(ii).to(dev)
    xyz = torch.from_numpy(rng.normal(size=(B, N, 3)).astype(np.float32)).to(dev)
    new_xyz = torch.from_numpy(rng.normal(size=(B, S, 3)).astype(np.float32)).to(dev)
    pl = C.point_lists(xyz, new_xyz, idx, None)
    torch.cuda.synchronize()
    rows_pt, rows_grp, w = _collapse_copies(ii, N)      # (repeats of a group's FIRST index are its padding copies: one weighted entry)
    prange, key, rws, meta = _numpy_lists(B, N, S, K, xyz.cpu().numpy(), new_xyz.cpu().numpy(), rows_pt, rows_grp, w)
    assert np.array_equal(pl.prange.cpu().numpy(), prange)
    prow, pmeta = pl.prow.cpu().numpy(), pl.pmeta.cpu().numpy()
    for q in range(B * N):
        a, b_ = prange[q]
        assert np.array_equal(prow[a:b_], rws[key == q]), q
        assert np.array_equal(pmeta[a:b_, 3], w[rws[key == q]]), q


@pytest.mark.parametrize("compact", [True, False])
def test_list_backward_equals_atomic_backward_and_is_bit_reproducible(dev, compact, monkeypatch):
    """The gather-add first layer's backward over the grouping's point lists (a segmented sum in fixed order) against the float-atomic kernel it
    replaces (PAPC_LG_LISTS=0 through papc_knob_set): every gradient of an SA2-shaped stack equal to 2e-6 / 2e-5 (another summation order of the
    same terms) -- and two runs of the list path BIT-identical in every gradient, which the atomic path is not (the only non-deterministic kernel
    of a training step, pointnet2_basic_layers.py:146-153 backward)."""
    lib = _lib.load()

    def run(lists):
        old = _lib.ctypes.c_int(0)
        _lib.check(lib.papc_knob_get(b"PAPC_LG_LISTS", _lib.ctypes.byref(old)), "papc_knob_get")
        _lib.check(lib.papc_knob_set(b"PAPC_LG_LISTS", 1 if lists else 0), "papc_knob_set")
        try:
            N, S, K, D, B = 512, 128, 64, 128, 8
            xyz, new_xyz, idx = _lists(dev, B, N, S, K, 0.4, 4)
            rng = np.random.default_rng(4)
            feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev).requires_grad_(True)
            ws = seeded_weights([D + 3, 128, 128, 256], 54)
            params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
            spec = StackSpec(B, N, S, K, D, True)
            cp = C.plan(idx) if compact else None
            spec.compact = cp
            spec.plists = C.point_lists(xyz, new_xyz, idx, cp)
            out = shared_mlp_max(spec, None, xyz, new_xyz, feats, idx, params)
            assert (out.grad_fn.compact is not None) == compact
            gout = torch.from_numpy(np.random.default_rng(8).normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
            out.backward(gout)
            torch.cuda.synchronize()
            return [p.grad.cpu().numpy() for p in params], feats.grad.cpu().numpy()
        finally:
            _lib.check(lib.papc_knob_set(b"PAPC_LG_LISTS", old.value), "papc_knob_set")

    ga, fa = run(False)
    g1, f1 = run(True)
    g2, f2 = run(True)
    for i, (a, b) in enumerate(zip(g1, g2)):
        assert np.array_equal(a, b), "list backward: gradient %d differs between two runs" % i
    assert np.array_equal(f1, f2), "list backward: dfeats differs between two runs"
    for i, (a, b) in enumerate(zip(g1, ga)):
        if i % 4 != 1:
            assert_close(a, b, 2e-5, "list vs atomic backward: gradient %d" % i)
    assert_close(f1, fa, 2e-5, "list vs atomic backward: dfeats")


def test_point_lists_of_the_other_row_layout_are_ignored(dev):
    """Lists index ONE row layout (papc_point_lists.compact).  A stack that runs the other layout -- here: lists of the compacted rows handed to a
    stack that runs padded -- must ignore them and take the atomic path (a mismatch would gather arbitrary rows): gradients equal the padded
    stack's without lists."""
    N, S, K, D, B = 512, 128, 64, 128, 8
    res = []
    for with_lists in (False, True):
        xyz, new_xyz, idx = _lists(dev, B, N, S, K, 0.4, 6)
        rng = np.random.default_rng(6)
        feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev).requires_grad_(True)
        ws = seeded_weights([D + 3, 128, 128, 256], 56)
        params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
        spec = StackSpec(B, N, S, K, D, True)
        if with_lists:
            spec.plists = C.point_lists(xyz, new_xyz, idx, C.plan(idx))      # compacted numbering ...
            assert spec.plists.compact
        out = shared_mlp_max(spec, None, xyz, new_xyz, feats, idx, params)   # ... for a stack that runs padded (spec.compact is None)
        assert out.grad_fn.compact is None
        gout = torch.from_numpy(np.random.default_rng(8).normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
        out.backward(gout)
        torch.cuda.synchronize()
        res.append(([p.grad.cpu().numpy() for p in params], feats.grad.cpu().numpy()))
    for i, (a, b) in enumerate(zip(res[1][0], res[0][0])):
        if i % 4 != 1:
            assert_close(a, b, 2e-5, "padded stack with mismatching lists vs without: gradient %d" % i)
    assert_close(res[1][1], res[0][1], 2e-5, "padded stack with mismatching lists vs without: dfeats")


def test_msg_layer_compacted_branch_uses_point_lists(dev):
    """PointNetSetAbstractionMsg (pointnet2_basic_layers.py:224-281): a compacted radius branch of a training layer carries its point lists in the
    sampling plan (7 compact tensors + 3 list tensors) and its gather-add backward sums over them -- two runs give the same BITS in every gradient,
    through sample() + forward(sampled=) and through the in-line path alike, and they agree with the float-atomic backward to 2e-5."""
    from papc_amd.layers import PointNetSetAbstractionMsg
    lib = _lib.load()
    B, N, S, D = 8, 512, 128, 128
    x = torch.from_numpy(make_clouds(B, N, 21)).to(dev)
    pts = torch.from_numpy(np.random.default_rng(21).normal(size=(B, D, N)).astype(np.float32)).to(dev)
    st = torch.from_numpy(make_start_idx(B, N, 21)).to(dev)

    def run(lists, planned):
        old = _lib.ctypes.c_int(0)
        _lib.check(lib.papc_knob_get(b"PAPC_LG_LISTS", _lib.ctypes.byref(old)), "papc_knob_get")
        _lib.check(lib.papc_knob_set(b"PAPC_LG_LISTS", 1 if lists else 0), "papc_knob_set")
        try:
            torch.manual_seed(3)
            layer = PointNetSetAbstractionMsg(S, [0.4, 0.8], [64, 128], D, [[128, 128, 256], [128, 196, 256]]).to(dev).train()
            layer.compact = True
            p = pts.clone().requires_grad_(True)
            plan = layer.sample(x, st) if planned else None
            if planned:
                # branch 0 ([128, 128, 256], K = 64) has the compacted flavour, branch 1 (196 channels) has not: 1 + 2 + 7 + 3 tensors
                assert len(plan) == 13 and tuple(plan[10].shape) == (B * N, 2) and plan[12].shape[1] == 4
            _, out = layer(x, p, st, sampled=plan)
            out.backward(torch.from_numpy(np.random.default_rng(5).normal(size=tuple(out.shape)).astype(np.float32)).to(dev))
            torch.cuda.synchronize()
            return {n: q.grad.cpu().numpy() for n, q in layer.named_parameters()}, p.grad.cpu().numpy()
        finally:
            _lib.check(lib.papc_knob_set(b"PAPC_LG_LISTS", old.value), "papc_knob_set")

    ga, fa = run(False, True)
    g1, f1 = run(True, True)
    g2, f2 = run(True, True)
    g3, f3 = run(True, False)
    for n in g1:
        if ".0." in n[:14]:      # conv_blocks.0.* / bn_blocks.0.*: the compacted branch -- no atomics left in its backward
            assert np.array_equal(g1[n], g2[n]), "MSG layer, list backward: %s differs between two runs" % n
            assert np.array_equal(g1[n], g3[n]), "MSG layer: planned and in-line sampling give different %s" % n
        if np.abs(ga[n]).max() > 0:       # (the 196-channel branch stays padded: float atomics, same numbers up to their order; so does dfeats, the sum of both)
            assert_close(g1[n], ga[n], 2e-5, "MSG layer, list vs atomic backward: %s" % n, elem=1.0)
    assert_close(f1, fa, 2e-5, "MSG layer, list vs atomic backward: dfeats")
    assert_close(f1, f3, 2e-5, "MSG layer, planned vs in-line: dfeats")


@pytest.mark.parametrize("c0", [32, 64, 256])
def test_list_backward_other_first_layer_widths(dev, c0):
    """The list backward maps CQ = C / 4 lanes to a row and 64 / CQ list entries to a wave pass: first-layer widths 32 (8 entries per pass), 64 (4)
    and 256 (1) besides SA2's 128 (2), on a padded grouping (lists of the padded rows), against the float-atomic kernel."""
    lib = _lib.load()
    N, S, K, D, B = 256, 32, 16, 32, 4

    def run(lists):
        old = _lib.ctypes.c_int(0)
        _lib.check(lib.papc_knob_get(b"PAPC_LG_LISTS", _lib.ctypes.byref(old)), "papc_knob_get")
        _lib.check(lib.papc_knob_set(b"PAPC_LG_LISTS", 1 if lists else 0), "papc_knob_set")
        try:
            xyz, new_xyz, idx = _lists(dev, B, N, S, K, 0.3, 9)
            rng = np.random.default_rng(9)
            feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev).requires_grad_(True)
            ws = seeded_weights([D + 3, c0, 64], 59)
            params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
            spec = StackSpec(B, N, S, K, D, True)
            spec.plists = C.point_lists(xyz, new_xyz, idx, None)
            out = shared_mlp_max(spec, None, xyz, new_xyz, feats, idx, params)
            assert out.grad_fn.lin0, "the gather-add first layer was not taken"
            out.backward(torch.from_numpy(np.random.default_rng(8).normal(size=tuple(out.shape)).astype(np.float32)).to(dev))
            torch.cuda.synchronize()
            return [p.grad.cpu().numpy() for p in params], feats.grad.cpu().numpy()
        finally:
            _lib.check(lib.papc_knob_set(b"PAPC_LG_LISTS", old.value), "papc_knob_set")

    ga, fa = run(False)
    g1, f1 = run(True)
    g2, f2 = run(True)
    assert all(np.array_equal(a, b) for a, b in zip(g1, g2)) and np.array_equal(f1, f2)
    for i, (a, b) in enumerate(zip(g1, ga)):
        if i % 4 != 1:
            assert_close(a, b, 2e-5, "c0 = %d, list vs atomic backward: gradient %d" % (c0, i), elem=1.0)
    assert_close(f1, fa, 2e-5, "c0 = %d, list vs atomic backward: dfeats" % c0, elem=1.0)


@pytest.mark.parametrize("compact,c0,radius", [(True, 128, 0.4), (False, 128, 0.4), (False, 64, 0.4), (False, 256, 0.4), (True, 128, 0.1), (False, 128, 0.1)])
def test_list_backward_without_y_equals_the_two_stream_list_backward(dev, compact, c0, radius):
    """papc_lingather_bwd_pp_f32 (the default where the point lists exist): the dX launch above stores dz already masked (papc_bwd_red.store_masked)
    and the gather-add backward gathers that ONE stream -- y's share of the BatchNorm backward comes from P[j] and the lists' per-point moments
    (sum w, sum w d, sum w d d^T).  Against the list backward that gathers y and dz (PAPC_LG_PP=0): every gradient of an SA2-shaped stack equal to
    2e-5 (the closed form does not re-read the forward's own rounding of y), bit-reproducible over two runs, on both row layouts and for the three
    first-layer widths the kernel is built for (one, two and four channels per lane); radius 0.1: most source points in no group at all (empty
    lists: their G rows must come out as exact zeros) and groups that are mostly padding copies."""
    lib = _lib.load()

    def run(pp):
        old = _lib.ctypes.c_int(0)
        _lib.check(lib.papc_knob_get(b"PAPC_LG_PP", _lib.ctypes.byref(old)), "papc_knob_get")
        _lib.check(lib.papc_knob_set(b"PAPC_LG_PP", 1 if pp else 0), "papc_knob_set")
        try:
            N, S, K, D, B = 512, 128, 64, 128, 8
            xyz, new_xyz, idx = _lists(dev, B, N, S, K, radius, 12)
            rng = np.random.default_rng(12)
            feats = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev).requires_grad_(True)
            ws = seeded_weights([D + 3, c0, 128, 256], 61)
            params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
            spec = StackSpec(B, N, S, K, D, True)
            cp = C.plan(idx) if compact else None
            spec.compact = cp
            spec.plists = C.point_lists(xyz, new_xyz, idx, cp)
            out = shared_mlp_max(spec, None, xyz, new_xyz, feats, idx, params)
            assert (out.grad_fn.compact is not None) == compact
            out.backward(torch.from_numpy(np.random.default_rng(8).normal(size=tuple(out.shape)).astype(np.float32)).to(dev))
            torch.cuda.synchronize()
            return [p.grad.cpu().numpy() for p in params], feats.grad.cpu().numpy()
        finally:
            _lib.check(lib.papc_knob_set(b"PAPC_LG_PP", old.value), "papc_knob_set")

    g0, f0 = run(False)
    g1, f1 = run(True)
    g2, f2 = run(True)
    assert all(np.array_equal(a, b) for a, b in zip(g1, g2)) and np.array_equal(f1, f2)
    names = ["w", "b", "gamma", "beta"]
    for i, (a, b) in enumerate(zip(g1, g0)):
        if i % 4 != 1:
            assert_close(a, b, 2e-5, "list backward without y vs with y: d%s layer %d" % (names[i % 4], i // 4), elem=1.0)
        if i >= 4:      # layers 2 and 3 do not depend on how layer 1's backward is formed: the masked store changes nothing they read
            assert np.array_equal(a, b), "masked dz store changed gradient %d" % i
    assert_close(f1, f0, 2e-5, "list backward without y vs with y: dfeats", elem=1.0)
    if radius < 0.2:      # points no group gathered: zero gradient, exactly
        iq = _lists(dev, 8, 512, 128, 64, radius, 12)[2].cpu().numpy()
        used = np.zeros((8, 512), bool)
        for b_ in range(8):
            v = iq[b_][(iq[b_] >= 0) & (iq[b_] < 512)]
            used[b_, v] = True
        assert (~used).sum() > 100 and not f1[~used].any()


@pytest.mark.parametrize("xyz_first", [True, False])
def test_feature_block_prepared_with_the_weight_transposes_changes_nothing(dev, xyz_first):
    """mlp.precompute_wt(feat_blocks=...): the gather-add first layer's feature block W_f made contiguous in the launch that transposes the step's
    weights (papc_transpose_batch_ld_f32, copy = 1 -> papc_sa_io.wfeat) instead of by a launch of the stack's own.  Same bits out and in every
    gradient, for both column orders (SSG: coordinates first; MSG: features first); a block of the wrong order is ignored."""
    from papc_amd.mlp import precompute_wt
    N, S, K, D, B = 512, 128, 64, 128, 8          # (fewer clouds run padded, i.e. on the float-atomic backward: no two runs alike)
    xyz, new_xyz, idx = _lists(dev, B, N, S, K, 0.4, 21)
    rng = np.random.default_rng(21)
    fe = rng.normal(size=(B, N, D)).astype(np.float32)
    ws = seeded_weights([D + 3, 128, 128, 256], 77)
    gout = None

    def run(table_order, poison=False):
        nonlocal gout
        feats = torch.from_numpy(fe).to(dev).requires_grad_(True)
        params = [torch.from_numpy(a).to(dev).requires_grad_(True) for tup in ws for a in tup]
        spec = StackSpec(B, N, S, K, D, xyz_first)
        cp = C.plan(idx)
        spec.compact = cp
        spec.plists = C.point_lists(xyz, new_xyz, idx, cp)
        spec.wt_table = precompute_wt([params[4], params[8]])      # (the reference run: the transposes alone)
        if table_order is not None:
            spec.wt_table = precompute_wt([params[4], params[8]], feat_blocks=[(params[0], table_order)])
            blk, order = spec.wt_table[("f", params[0].data_ptr())]
            want = params[0].detach()[:, 3:] if table_order else params[0].detach()[:, :D]
            torch.cuda.synchronize()
            assert order == table_order and torch.equal(blk, want)
            if poison:
                blk.zero_()
        out = shared_mlp_max(spec, None, xyz, new_xyz, feats, idx, params)
        assert out.grad_fn.compact is not None
        if gout is None:
            gout = torch.from_numpy(np.random.default_rng(5).normal(size=tuple(out.shape)).astype(np.float32)).to(dev)
        out.backward(gout)
        torch.cuda.synchronize()
        return out.detach().cpu().numpy(), [p.grad.cpu().numpy() for p in params], feats.grad.cpu().numpy()

    o0, g0, f0 = run(None)
    for order in (xyz_first, not xyz_first):        # the right block is used, the wrong one ignored: the same bits either way
        o1, g1, f1 = run(order)
        assert np.array_equal(o0, o1) and np.array_equal(f0, f1) and all(np.array_equal(a, b) for a, b in zip(g0, g1))
    # ... and it IS the prepared block the forward multiplies by: zeroed, the output moves; a zeroed block of the other order moves nothing
    assert not np.array_equal(o0, run(xyz_first, poison=True)[0])
    assert np.array_equal(o0, run(not xyz_first, poison=True)[0])
