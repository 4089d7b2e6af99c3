"""CPU tests of the host-side logic (no GPU compute)."""
import pytest
import numpy as np
import torch

from papc_amd import functional as F
from papc_amd.layers import PointNetSetAbstraction, PointNetSetAbstractionMsg
from papc_amd.mlp import StackSpec, _dw_rows_per_chunk
from papc_amd.models import PointNet2_MSG_Clas, PointNet2_SSG_Clas, PointNet_Basic_Clas
from papc_amd.pillars import PillarFeatureNet
from papc_amd.synthetic import make_clouds, make_labels, make_pillars, make_start_idx


def test_radius_threshold_matches_python_double_square():
    assert F.radius_threshold(0.2) == float(np.float32(0.2 * 0.2))
    assert F.radius_threshold(0.2) != float(np.float32(0.2) * np.float32(0.2))


def test_synthetic_generators_are_deterministic_and_normalised():
    a, b = make_clouds(4, 256, 9), make_clouds(4, 256, 9)
    assert a.dtype == np.float32 and a.shape == (4, 3, 256) and np.array_equal(a, b)
    n = np.sqrt((a ** 2).sum(1)).max(1)
    assert np.allclose(n, 1.0, atol=1e-5)                       # pc_normalize: max norm 1
    assert make_labels(5).shape == (5, 1) and make_labels(5).dtype == np.int64
    assert make_start_idx(6, 100).max() < 100
    v, n_, c = make_pillars(P=32, T=20)
    assert v.shape == (32, 20, 4) and c.shape == (32, 4) and n_.min() >= 1
    assert not v[np.arange(20)[None, :] >= n_[:, None]].any()   # padded rows are zero


def test_dw_chunking():
    for M, co, ci in [(524288, 128, 64), (4096, 1024, 512), (300, 16, 9), (262144, 256, 128)]:
        r = _dw_rows_per_chunk(M, co, ci)
        assert r % 64 == 0 and r >= 256
    assert StackSpec(2, 10, 3, 4, 5, True).M == 24


def test_parameter_inventory_matches_reference_shapes():
    m = PointNet2_SSG_Clas()
    head = sum(p.numel() for n, p in m.named_parameters() if n.split(".")[0] in ("fc1", "bn1", "fc2", "bn2", "fc3"))
    assert head == 661776                                        # the params the reference registers
    total = sum(p.numel() for p in m.parameters())
    assert total == 1469520                                      # all params (SA stacks registered here)
    q = PointNet2_SSG_Clas(reference_quirks=True)
    assert sum(p.numel() for p in q.parameters() if p.requires_grad) == 661776
    assert PointNet2_MSG_Clas().sa1.conv_blocks[2][1].weight.shape == (96, 64, 1, 1)
    assert PointNet_Basic_Clas().convs[4].weight.shape == (1024, 128, 1)
    pfn = PillarFeatureNet(num_filters=(64,), voxel_size=(1, 2, 3), pc_range=(0, -40, -3, 70.4, 40, 1))
    assert (pfn.vx, pfn.vy, pfn.x_offset, pfn.y_offset) == (1, 2, 0.5, -39.0)   # the reference's own wiring
    assert pfn.pfn_layers[0].linear.weight.shape == (64, 9)
    two = PillarFeatureNet()                                      # default (64,128): first layer halves (:18-19)
    assert two.pfn_layers[0].units == 32 and two.pfn_layers[1].linear.weight.shape == (128, 64)


def test_layer_constructors_mirror_reference_signatures():
    sa = PointNetSetAbstraction(npoint=512, radius=0.2, nsample=32, in_channel=3, mlp=[64, 64, 128], group_all=False)
    assert len(sa.mlp_convs) == 3 and sa.mlp_convs[0].weight.shape == (64, 3, 1, 1)
    msg = PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], 0, [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
    assert msg.conv_blocks[0][0].weight.shape == (32, 3, 1, 1)


def test_checkpoint_exchange_with_reference_layout(tmp_path):
    """*.pdparams round trip (plain pickle of name -> ndarray): Linear weights are [in, out] on the reference side, BN
    statistics are _mean / _variance, SA layers are absent from a reference checkpoint."""
    import pickle
    import numpy as np
    import torch
    from papc_amd import checkpoint as C
    from papc_amd.models import PointNet2_SSG_Clas
    torch.manual_seed(0)
    m = PointNet2_SSG_Clas(num_classes=16)
    with torch.no_grad():
        m.bn1.running_mean.uniform_(-1, 1); m.bn1.running_var.uniform_(0.5, 2)
    path = str(tmp_path / "model.pdparams")
    C.save_pdparams(m, path)
    st = C.load_pdparams(path)
    assert st["fc1.weight"].shape == (1024, 512)                       # paddle layout [in, out]
    assert "bn1._mean" in st and "bn1._variance" in st and not any("running" in k for k in st)
    m2 = PointNet2_SSG_Clas(num_classes=16)
    missing, unexpected = C.import_state(m2, st, strict=True)
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(a, b), k
    # a reference-made checkpoint: only the registered head layers, plus paddle's bookkeeping key
    ref = {k: v for k, v in st.items() if not k.startswith("sa")}
    ref["StructuredToParameterName@@"] = {}
    with open(path, "wb") as f:
        pickle.dump(ref, f, protocol=2)
    m3 = PointNet2_SSG_Clas(num_classes=16)
    missing, unexpected = C.import_state(m3, C.load_pdparams(path))
    assert missing and all(k.startswith("sa") for k in missing) and not unexpected
    assert torch.equal(m3.fc2.weight, m.fc2.weight) and torch.equal(m3.bn1.running_var, m.bn1.running_var)


def test_checkpoint_pointnet_basic_reference_names_and_restricted_pickle(tmp_path):
    """A reference PointNet-Basic checkpoint names its conv stack mlp_1.{0,1,3,4} / mlp_2.{0,1,3,4,6,7} (pointnet_base.py:7-25):
    the importer must fill the backbone, not only fc.*; and a pickle that names anything but numpy arrays is refused."""
    import pickle
    import numpy as np
    import pytest
    import torch
    from papc_amd import checkpoint as C
    from papc_amd.models import PointNet_Basic_Clas
    torch.manual_seed(1)
    m = PointNet_Basic_Clas(num_classes=16)
    with torch.no_grad():
        for bn in m.bns:
            bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2)
    st = C.export_state(m)
    for k in ("mlp_1.0.weight", "mlp_1.1._mean", "mlp_1.3.bias", "mlp_1.4._variance", "mlp_2.0.weight", "mlp_2.1.weight", "mlp_2.3.weight",
              "mlp_2.4.bias", "mlp_2.6.weight", "mlp_2.7._variance", "fc.0.weight", "fc.5.bias"):
        assert k in st, k
    assert st["mlp_2.6.weight"].shape[:2] == (1024, 128) and st["fc.0.weight"].shape == (1024, 512)
    assert not any(k.startswith(("convs", "bns")) for k in st)
    path = str(tmp_path / "basic.pdparams")
    with open(path, "wb") as f:
        pickle.dump(dict(st, **{"StructuredToParameterName@@": {}}), f, protocol=2)
    m2 = PointNet_Basic_Clas(num_classes=16)
    missing, unexpected = C.import_state(m2, C.load_pdparams(path), strict=True)
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(a, b), k

    class Evil:
        def __reduce__(self):
            return (print, ("arbitrary code ran",))
    bad = str(tmp_path / "bad.pdparams")
    with open(bad, "wb") as f:
        pickle.dump({"fc.0.weight": Evil()}, f, protocol=2)
    with pytest.raises(pickle.UnpicklingError):
        C.load_pdparams(bad)


def test_deferred_fold_list_survives_a_backward_that_raises(monkeypatch):
    """papc_amd/folds.py: a backward pass that raises never runs the engine's final callbacks; the next pass must get a FRESH list and
    its own callback (advisor finding, round 5: the thread-local state stayed armed with stale jobs and later folds were skipped), and the
    callback must work from a thread that never called pending()."""
    import torch
    from papc_amd import folds
    launched = []
    monkeypatch.setattr(folds, "_cur_stream", lambda: "s0")
    monkeypatch.setattr(folds, "_launch", lambda p: launched.append((p.task_id, p.lst.count)))
    monkeypatch.setattr(folds, "ENABLED", True)
    folds._live.clear()

    class Stack(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fail):
            ctx.fail = fail
            return x * 2

        @staticmethod
        def backward(ctx, g):
            lst = folds.pending(["scratch"])
            assert lst is not None
            lst.contents.count += 2                 # what papc_sa_mlp_bwd does: append jobs
            if ctx.fail:
                raise RuntimeError("boom")
            return g * 2, None

    assert folds.pending([]) is None                # not inside a backward pass
    x = torch.ones(3, requires_grad=True)
    with pytest.raises(RuntimeError):
        Stack.apply(Stack.apply(x, True), False).sum().backward()
    assert launched == [] and len(folds._live) == 1         # the failed pass left its entry behind
    Stack.apply(Stack.apply(x, False), False).sum().backward()
    assert len(launched) == 1 and launched[0][1] == 4       # the next pass folded exactly its own four jobs
    assert len(folds._live) == 1                            # (the stale entry: bounded, dropped below)
    for _ in range(folds.MAX_LIVE + 2):
        with pytest.raises(RuntimeError):
            Stack.apply(x, True).sum().backward()
    assert len(folds._live) <= folds.MAX_LIVE
    stale = list(folds._live.values())
    Stack.apply(x, False).sum().backward()
    assert launched[-1][1] == 2
    assert all(q.keep == ["scratch"] or (q.done and not q.keep) for q in stale)
    folds._live.clear()
