import sys; sys.path.insert(0, '.')
import torch, numpy as np
from tests.test_gpu_compact import _sa2_like
dev = torch.device('cuda:0')
xyz, new_xyz, idx, feats, params, spec, out = _sa2_like(dev, 8, 1, True)
try:
    out.backward(torch.ones_like(out))
    print("backward ok")
except Exception as e:
    print("ERR", e)
