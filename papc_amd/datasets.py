"""ShapeNet-part batch loaders with the reference's semantics (PAPC/datasets/pnloader.py:7-106, dataloader.py:5-40,
file lists datalist.py:1-3) -- the data format on the input side of the hot path.

What the reference does, and what is kept:

* every file of the split is read whole and concatenated in list order: ``f['data'][:, :max_point, :]`` (the FIRST max_point points of each
  cloud, no resampling), ``f['label']`` ([n, 1]), for segmentation also ``f['pid'][:, :max_point]`` (pnloader.py:13-31, 60-81);
* the loader is a generator FACTORY: calling the returned function starts one epoch (pnloader.py:37-50, 89-104);
* ``mode == 'train'`` shuffles the sample order in place with the global ``random`` module at the start of every epoch -- the permutation is
  applied to the previous epoch's order, not to 0..n-1 (pnloader.py:35-39) -- so ``random.seed(s)`` reproduces the reference's batches;
* a batch is ``(data [b, 3, max_point] float32, label [b, 1] int64)`` -- each cloud transposed to channels-first (pnloader.py:43-46) --, for
  segmentation ``([data, label], target [b, max_point, 1] int64)`` (pnloader.py:96-100); the last batch of an epoch may be short
  (pnloader.py:49-50, 103-104).

What is done differently (same batches, bit for bit -- tests/test_datasets.py holds them to a literal restatement, oracle/loaders_ref.py):
the clouds are transposed and converted ONCE at load time into one contiguous ``[n, 3, max_point]`` float32 array (the reference transposes
and converts every sample of every epoch and builds each batch from a Python list), and a batch is one gather along the shuffled order.

Files: ``h5py`` is not part of this image.  ``opener(path)`` returns any mapping with ``['data']`` / ``['label']`` / ``['pid']`` arrays; the
default opener takes ``.h5`` through h5py when it is importable and otherwise (or for ``.npz``) reads a numpy archive with the same keys at
the same stem, so a ShapeNet-part directory converted with ``convert_h5_to_npz`` (on a machine that has h5py) trains here unchanged.
"""
import os
import random

import numpy as np

# PAPC/datasets/datalist.py:1-3
train_list = ['ply_data_train0.h5', 'ply_data_train1.h5', 'ply_data_train2.h5', 'ply_data_train3.h5', 'ply_data_train4.h5', 'ply_data_train5.h5']
test_list = ['ply_data_test0.h5', 'ply_data_test1.h5']
val_list = ['ply_data_val0.h5']


def default_opener(path):
    """``path`` (an .h5 name from the file lists) -> mapping of arrays.  h5py when available, else the .npz twin of the file."""
    stem, ext = os.path.splitext(path)
    if ext == ".h5" and os.path.exists(path):
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            with h5py.File(path, "r") as f:
                return {k: np.asarray(f[k]) for k in f.keys()}
    for cand in (path, stem + ".npz"):
        if cand.endswith(".npz") and os.path.exists(cand):
            with np.load(cand) as z:
                return {k: z[k] for k in z.files}
    raise FileNotFoundError("%s: no readable file (h5py %s; looked for %s.npz too)" %
                            (path, "missing" if ext == ".h5" else "not needed", stem))


def convert_h5_to_npz(src_dir, dst_dir=None):
    """One-off helper for a machine with h5py: write the .npz twin of every file of the three lists."""
    import h5py
    dst_dir = dst_dir or src_dir
    for name in train_list + test_list + val_list:
        p = os.path.join(src_dir, name)
        if os.path.exists(p):
            with h5py.File(p, "r") as f:
                np.savez(os.path.join(dst_dir, os.path.splitext(name)[0] + ".npz"), **{k: np.asarray(f[k]) for k in f.keys()})


def _files_of(mode):
    return train_list if mode == 'train' else (test_list if mode == 'test' else val_list)     # pnloader.py:12-31: anything else = val


def _load(max_point, path, mode, opener, with_pid):
    datas, labels, pids = [], [], []
    for name in _files_of(mode):
        f = opener(os.path.join(path, name))
        d = np.asarray(f['data'])[:, :max_point, :]
        # channels-first float32 once, here (pnloader.py:43 does datas[i].T.astype('float32') per sample and epoch)
        datas.append(np.ascontiguousarray(np.transpose(d, (0, 2, 1)), dtype=np.float32))
        labels.append(np.asarray(f['label']).astype(np.int64))
        if with_pid:
            pids.append(np.asarray(f['pid'])[:, :max_point].astype(np.int64))
    cat = lambda xs: np.concatenate(xs, axis=0) if xs else np.zeros((0,))     # noqa: E731
    return cat(datas), cat(labels), (cat(pids) if with_pid else None)


def PNClasDataLoader(max_point=1024, batchsize=64, path='./data/', mode='train', opener=None):
    """pnloader.py:7-52.  Returns a generator function; each call = one epoch of ``(data [b,3,max_point] f32, label [b,1] i64)``."""
    datas, labels, _ = _load(max_point, path, mode, opener or default_opener, False)
    index_list = list(range(len(datas)))

    def PNClasDataGenerator():
        if mode == 'train':
            random.shuffle(index_list)                 # in place, global RNG: the reference's order for a given random.seed (pnloader.py:38-39)
        for lo in range(0, len(index_list), batchsize):
            sel = index_list[lo:lo + batchsize]
            yield datas[sel], labels[sel]

    return PNClasDataGenerator


def PNSegDataLoader(max_point=1024, batchsize=64, path='./data/', mode='train', opener=None):
    """pnloader.py:54-106.  Each call = one epoch of ``([data [b,3,max_point] f32, label [b,1] i64], target [b,max_point,1] i64)``."""
    datas, labels, pids = _load(max_point, path, mode, opener or default_opener, True)
    # np.reshape(targets[i], [max_point, -1]) (pnloader.py:95): a cloud with fewer than max_point points cannot be reshaped there either
    if len(pids) and pids.shape[1] != max_point:
        raise ValueError("cannot reshape array of size %d into shape (%d, newaxis)" % (pids.shape[1], max_point))
    targets = pids.reshape(len(pids), max_point, -1) if len(pids) else pids
    index_list = list(range(len(datas)))

    def PNSegDataGenerator():
        if mode == 'train':
            random.shuffle(index_list)
        for lo in range(0, len(index_list), batchsize):
            sel = index_list[lo:lo + batchsize]
            yield [datas[sel], labels[sel]], targets[sel]

    return PNSegDataGenerator


_PN_MODELS = ('pointnet_basic', 'pointnet', 'vfe', 'pointnet2_ssg', 'pointnet2_msg')


def DataLoader(model_name, max_point, batchsize, path='./data/', mode1='clas', mode2='train', opener=None):
    """dataloader.py:5-40: the dispatcher train.py calls.  The point-set models (every model this library covers) share the PN loaders; the
    KD-tree and voxel loaders belong to model families outside this library's scope and are reported as such."""
    if mode1 == 'clas':
        if model_name in _PN_MODELS:
            return PNClasDataLoader(max_point, batchsize, path, mode2, opener)
        if model_name in ('voxnet', 'kdnet'):
            raise SystemExit('Error: the %s loader is outside this library (point-set models only)' % model_name)
        raise SystemExit('Error: model is incorrect')
    elif mode1 == 'seg':
        if model_name in _PN_MODELS:
            return PNSegDataLoader(max_point, batchsize, path, mode2, opener)
        if model_name == 'kdunet':
            raise SystemExit('Error: the kdunet loader is outside this library (point-set models only)')
        raise SystemExit('Error: model is incorrect')
    elif mode1 == 'detect':
        raise SystemExit('Error: Sorry, do not have detect model')
    else:
        raise SystemExit('Error: mode should be "clas", "detect" or "seg"')


def device_batches(generator, device, fps_seed=None):
    """Wrap one epoch of a PN loader for the GPU step: yields torch tensors on ``device`` -- ``(x [b,3,N] f32, y [b] i64)`` or
    ``(x, cls [b] i64, target [b,N] i64)`` -- copied through pinned host memory on the current stream.  ``fps_seed`` (optional): also yield the
    explicit FPS start indices this library takes where the reference draws them with paddle.randint (pointnet2_basic_layers.py:76), from a
    numpy generator seeded per epoch: ``(..., (s1 [b], s2 [b]))`` for the two sampling levels of the PointNet++ models."""
    import torch
    rng = np.random.default_rng(fps_seed) if fps_seed is not None else None

    def up(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        if torch.device(device).type == "cuda":
            t = t.pin_memory().to(device, non_blocking=True)
        return t

    for batch in generator():
        if isinstance(batch[0], list):
            (x, cls), tgt = batch
            out = (up(x), up(cls.reshape(-1)), up(tgt.reshape(tgt.shape[0], -1)))
        else:
            x, y = batch
            out = (up(x), up(y.reshape(-1)))
        if rng is not None:
            b, n = x.shape[0], x.shape[2]
            out = out + ((up(rng.integers(0, n, size=b).astype(np.int64)), up(rng.integers(0, 512, size=b).astype(np.int64))),)
        yield out
