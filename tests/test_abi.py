"""CPU checks of the drop-in boundary: the library loads and exports every symbol include/papc_hip.h declares,
the ctypes table covers the header, and argument validation fails loudly (no compute needs a GPU here)."""
import ctypes
import os
import re

import pytest

from papc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "papc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(papc_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_table_agree():
    syms = _header_symbols()
    assert len(syms) >= 25
    assert sorted(_lib.SIGNATURES) == syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in _header_symbols():
        assert hasattr(lib, s), "libpapc_hip.so does not export " + s
    assert _lib.load().papc_version() >= 100


def test_argument_validation_without_gpu():
    lib = _lib.load()
    # null pointers / bad sizes are rejected on the host before any launch
    assert lib.papc_fps_f32(None, 0, 0, 0, 1, 1, 1, None, 1.0, None, None, None) == -1
    assert b"null" in lib.papc_last_error_string()
    assert lib.papc_ball_query_f32(None, 0, 0, 0, None, 1, 1, 1, 1, None, None, None, 0, None) == -1
    assert lib.papc_mlp_gemm_f32(0, None, 0, None, None, None, None, None, 1, 1, 1, None, None, None, None) == -1
    assert lib.papc_mlp_gemm_gmax_ok(524288, 128, 32) == 1 and lib.papc_mlp_gemm_gmax_ok(524288, 128, 16) == 0 and lib.papc_mlp_gemm_gmax_ok(4096, 1024, 128) == 1
    assert lib.papc_pfn_num_blocks(12000) == 1024 and lib.papc_pfn_num_blocks(8) == 2
    assert lib.papc_mlp_gemm_parts(524288) == 768 and lib.papc_mlp_gemm_parts(300) == 3   # rows of the partial buffers: min(row tiles, 3 x 256 CUs)
    with pytest.raises(_lib.PapcError):
        _lib.check(-1, "x")


def test_no_cpu_fallback():
    import torch
    from papc_amd import functional as F
    with pytest.raises(_lib.PapcError):
        F.farthest_point_sample(torch.zeros(1, 8, 3), 2)
    with pytest.raises(_lib.PapcError):
        F.query_ball_point(0.2, 4, torch.zeros(1, 8, 3), torch.zeros(1, 2, 3))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "papc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
