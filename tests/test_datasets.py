"""papc_amd/datasets.py (the ShapeNet-part loaders, /root/reference/PAPC/datasets/pnloader.py:7-106, dataloader.py:5-40)
against the literal restatement oracle/loaders_ref.py on a synthetic in-memory file set: same batches bit for bit for the same
``random.seed`` -- over several epochs (the reference shuffles the previous epoch's order in place), every split, truncation to max_point,
short last batches, and through an on-disk .npz twin of the .h5 names (h5py is not in this image)."""
import os
import random

import numpy as np
import pytest

from oracle import loaders_ref as R
from papc_amd import datasets as D


def _fileset(seed=0, npts=40):
    """{h5 name: {'data' [n, npts, 3] f32, 'label' [n, 1] u8, 'pid' [n, npts] u8}} with a different n per file (as ShapeNet-part has)"""
    rng = np.random.default_rng(seed)
    files = {}
    for k, name in enumerate(D.train_list + D.test_list + D.val_list):
        n = 5 + (3 * k) % 7
        files[name] = {"data": rng.normal(size=(n, npts, 3)).astype(np.float32), "label": rng.integers(0, 16, size=(n, 1)).astype(np.uint8),
                       "pid": rng.integers(0, 50, size=(n, npts)).astype(np.uint8)}
    return files


def _opener(files):
    return lambda path: files[os.path.basename(path)]


def _same(a, b):
    assert type(a) is type(b) or not isinstance(a, list)
    if isinstance(a, (list, tuple)):
        assert len(a) == len(b)
        for u, v in zip(a, b):
            _same(u, v)
        return
    assert a.dtype == b.dtype and a.shape == b.shape, (a.dtype, b.dtype, a.shape, b.shape)
    assert np.array_equal(a, b)
    assert a.flags["C_CONTIGUOUS"]


@pytest.mark.parametrize("mode", ["train", "test", "val", "anything-else"])
@pytest.mark.parametrize("max_point,batch", [(32, 4), (40, 7), (16, 64)])
def test_pn_loaders_match_the_reference_semantics(mode, max_point, batch):
    files = _fileset()
    for mine, ref in ((D.PNClasDataLoader, R.PNClasDataLoader), (D.PNSegDataLoader, R.PNSegDataLoader)):
        random.seed(7)
        g_ref = ref(_opener(files), max_point, batch, "/data", mode)
        want = [list(g_ref()) for _ in range(3)]            # three epochs: the shuffle acts on the previous epoch's order
        random.seed(7)
        g = mine(max_point, batch, "/data", mode, opener=_opener(files))
        got = [list(g()) for _ in range(3)]
        assert [len(e) for e in got] == [len(e) for e in want] and len(got[0]) >= 1
        for eg, ew in zip(got, want):
            for bg, bw in zip(eg, ew):
                _same(list(bg), list(bw))
        n = sum(len(files[f]["data"]) for f in D._files_of(mode))
        first = got[0][0]
        x = first[0][0] if isinstance(first[0], list) else first[0]
        assert x.shape == (min(batch, n), 3, max_point) and x.dtype == np.float32       # [B, 3, N] as the models take it (pnloader.py:43-46)
        assert sum(len(b[1]) for b in got[0]) == n                                       # every sample once per epoch, the last batch short
        if mode == "train":
            assert any(not np.array_equal(a[1] if not isinstance(a[0], list) else a[0][1], b[1] if not isinstance(b[0], list) else b[0][1])
                       for a, b in zip(got[0], got[1])) or n <= 1


def test_dispatcher_and_npz_twin(tmp_path):
    files = _fileset(3)
    for name, arrs in files.items():
        np.savez(tmp_path / (os.path.splitext(name)[0] + ".npz"), **arrs)                # the .npz twin of every .h5 name
    random.seed(1)
    want = list(R.PNClasDataLoader(_opener(files), 24, 5, "/x", "train")())
    random.seed(1)
    got = list(D.DataLoader("pointnet2_ssg", 24, 5, str(tmp_path), "clas", "train")())
    for a, b in zip(got, want):
        _same(list(a), list(b))
    assert len(list(D.DataLoader("pointnet2_msg", 24, 5, str(tmp_path), "seg", "val")())) == -(-len(files[D.val_list[0]]["data"]) // 5)
    for bad in (("kdnet", "clas"), ("nope", "clas"), ("pointnet2_ssg", "detect"), ("pointnet2_ssg", "other"), ("kdunet", "seg")):
        with pytest.raises(SystemExit):
            D.DataLoader(bad[0], 24, 5, str(tmp_path), bad[1], "train")
    with pytest.raises(FileNotFoundError):
        D.PNClasDataLoader(24, 5, str(tmp_path / "missing"), "train")


def test_device_batches_cpu_layout():
    files = _fileset(5)
    g = D.PNSegDataLoader(32, 6, "/d", "test", opener=_opener(files))
    out = list(D.device_batches(g, "cpu", fps_seed=3))
    x, cls, tgt, (s1, s2) = out[0]
    assert tuple(x.shape) == (6, 3, 32) and tuple(cls.shape) == (6,) and tuple(tgt.shape) == (6, 32) and tuple(s1.shape) == (6,)
    assert str(x.dtype) == "torch.float32" and str(tgt.dtype) == "torch.int64" and int(s1.max()) < 32 and int(s2.max()) < 512
