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


def test_abi_version_and_struct_sizes():
    """Every ctypes mirror of a descriptor struct in papc_amd/ has the size the library was compiled with (papc_abi_sizeof), the header's
    PAPC_ABI_VERSION is what the library and the binding report, and every struct of the header is known to papc_abi_sizeof."""
    import importlib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "papc_hip.h")).read()
    assert int(re.search(r"#define PAPC_ABI_VERSION (\d+)", hdr).group(1)) == lib.papc_abi_version() == _lib.ABI_VERSION
    for name in re.findall(r"^typedef struct (papc_[a-z0-9_]+)", hdr, flags=re.M):
        assert lib.papc_abi_sizeof(name.encode()) > 0, name
    assert lib.papc_abi_sizeof(b"no_such_struct") == -1 and lib.papc_abi_sizeof(None) == -1
    mirrors = {"_lib": {"GroupSrc": "papc_group_src", "BwdDy": "papc_bwd_dy", "GroupMax": "papc_group_max", "BwdRed": "papc_bwd_red",
                        "CopyJob": "papc_copy_job", "ReduceJob": "papc_reduce_job", "ScatterDst": "papc_scatter_dst"},
               "folds": {"FoldJob": "papc_fold_job", "FoldList": "papc_fold_list"},
               "head": {"HeadFcLayer": "papc_head_fc_layer", "HeadBwdJob": "papc_head_bwd_job"},
               "pillars": {"PfnDesc": "papc_pfn_desc", "PfnIo": "papc_pfn_io"},
               "smallm": {"PgWJob": "papc_pg_wjob", "PgPrep": "papc_pg_prep", "PgGemm": "papc_pg_gemm", "PgFoldJob": "papc_pg_fold_job"},
               "stack": {"SaDesc": "papc_sa_desc", "SaLayer": "papc_sa_layer", "CompactSrc": "papc_compact_src", "SaIo": "papc_sa_io",
                         "SaPlan": "papc_sa_plan", "SaGrads": "papc_sa_grads"}}
    for mod, table in mirrors.items():
        m = importlib.import_module("papc_amd." + mod)
        for cls, cname in table.items():
            assert ctypes.sizeof(getattr(m, cls)) == lib.papc_abi_sizeof(cname.encode()), (mod, cls, cname)
