"""CPU-side checks: the C-ABI library loads and exports every symbol include/mesm_gfx950.h
declares (no compute without a GPU), argument validation returns error codes instead of
crashing, and the Python binding mirrors the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mesm_gfx950.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mesm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mesm_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(L, s), s


def test_binding_covers_the_header():
    from mesm_amd import _lib
    assert set(_lib.PROTOTYPES) == set(declared_symbols())
    L = _lib.lib()
    assert L.mesm_abi_version() >= 1
    assert L.mesm_arch() == b"gfx950"


def test_struct_sizes_match_the_c_layout():
    # sizeof computed from the field lists of the header (LP64): guards against a drifting mirror
    from mesm_amd import _lib
    assert ctypes.sizeof(_lib.GemmArgs) == 248
    assert ctypes.sizeof(_lib.AttnArgs) == 256
    assert ctypes.sizeof(_lib.LnArgs) == 168


def test_bad_arguments_are_rejected_without_a_gpu():
    from mesm_amd import _lib
    L = _lib.lib()
    g = _lib.GemmArgs()
    assert L.mesm_gemm_f32(ctypes.byref(g), None) == -1  # MESM_EINVAL: null operands
    a = _lib.AttnArgs()
    assert L.mesm_attn_fwd(ctypes.byref(a), None) == -1
    assert L.mesm_layernorm_fwd(None, None, None, None, None, None, 4, 256, 1e-5, 0.0, 0, None, None) == -1
    assert L.mesm_match(None, None, None, None, None, 1, 1, 1, 1.0, 1.0, 1.0, None, None, None) == -1


def test_product_path_refuses_cpu_tensors():
    import torch
    from mesm_amd import kernels as kn
    with pytest.raises(Exception):
        kn.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8))


def test_limits_are_reported_in_words():
    """shapes the kernels cannot take raise a MesmError that says why (not a bare status code)."""
    from mesm_amd import _lib
    from mesm_amd import kernels as kn
    kn._check_match_limits(10, 5)
    kn._check_match_limits(4, 5)  # more targets than queries is a shape like any other (matcher.py:108-117)
    kn._check_match_limits(64, 64)
    for q, t in ((65, 3), (10, 65)):
        with pytest.raises(_lib.MesmError, match="Hungarian matching kernel"):
            kn._check_match_limits(q, t)
    kn._check_drop_index(1 << 33, 0.0)
    with pytest.raises(_lib.MesmError, match="32-bit"):
        kn._check_drop_index(1 << 32, 0.1)
