"""The data side of the hot path end to end on the GPU: batches from papc_amd/datasets.py (the reference's ShapeNet-part loaders,
/root/reference/PAPC/datasets/pnloader.py:7-106, over an in-memory file set) through ``datasets.device_batches`` into a training step of
PointNet2_SSG_Clas (/root/reference/PAPC/train.py:102-116: forward, cross-entropy, backward, Adam) -- layouts and dtypes as the loop wants them,
the loss finite and falling over a few steps on a learnable toy labelling."""
import os
import random

import numpy as np
import pytest
import torch

from papc_amd import datasets as D
from papc_amd.distributed import FlatAdam, FlatParams
from papc_amd.models import PointNet2_SSG_Clas
from papc_amd.synthetic import make_clouds

pytestmark = pytest.mark.gpu


def test_loader_batches_train_the_classifier(dev):
    rng = np.random.default_rng(0)
    files = {}
    for k, name in enumerate(D.train_list):
        n = 8
        clouds = make_clouds(n, 1024, 300 + k).transpose(0, 2, 1).copy()            # [n, 1024, 3] as the .h5 files hold them
        label = (clouds[:, :, 0].std(axis=1) > np.median(clouds[:, :, 0].std(axis=1))).astype(np.uint8).reshape(n, 1)   # a label the shape determines
        files[name] = {"data": clouds, "label": label, "pid": rng.integers(0, 50, size=(n, 1024)).astype(np.uint8)}
    random.seed(3)
    gen = D.DataLoader("pointnet2_ssg", 1024, 8, "/data", "clas", "train", opener=lambda p: files[os.path.basename(p)])
    torch.manual_seed(0)
    model = PointNet2_SSG_Clas(num_classes=2).to(dev).train()
    model.drop1.p = model.drop2.p = 0.0
    flat = FlatParams(model)
    opt = FlatAdam(flat, lr=1e-3, weight_decay=1e-3)
    losses = []
    for epoch in range(3):
        for x, y, starts in D.device_batches(gen, dev, fps_seed=epoch):
            assert tuple(x.shape) == (8, 3, 1024) and x.dtype == torch.float32 and tuple(y.shape) == (8,) and y.dtype == torch.int64
            flat.zero_grad()
            loss, logits = model(x, starts, labels=y)
            loss.backward()
            opt.step(1.0)
            losses.append(float(loss.detach()))
            assert tuple(logits.shape) == (8, 2)
    assert len(losses) == 18 and all(np.isfinite(losses))
    assert np.mean(losses[-6:]) < np.mean(losses[:6]), losses
