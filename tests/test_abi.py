"""CPU: the C-ABI library loads and exports every symbol include/p2t_hip.h declares, the ctypes
struct mirrors match sizeof() in the library, and argument errors surface without a GPU.
No compute entry point is launched here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "p2t_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(p2t_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from p2t_hip import _lib
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/p2t_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in p2t_hip/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert _lib.version() == 103


def test_struct_sizes_match():
    from p2t_hip import _lib
    for i, s in enumerate(_lib._STRUCTS):
        assert _lib.lib.p2t_struct_size(i) == ctypes.sizeof(s)
    assert _lib.lib.p2t_struct_size(99) == 0


def test_argument_errors_without_gpu():
    from p2t_hip import _lib
    with pytest.raises(ValueError, match="null"):
        _lib.call("p2t_gemm_nt", None, 64, None, 64, None, None, 64, None, 8, 16, 64, 0, 0, 0, 0, -1, None, 0, 0, None)
    with pytest.raises(ValueError):
        _lib.call("p2t_layernorm", None, 4, None, None, 1e-5, None, 4, 1, 4, 0, None)
    cfg = _lib.EsmConfigC(n_layers=1, hidden=64, ffn=128, heads=4, head_dim=16, vocab=33, dtype=0)
    assert _lib.call("p2t_esm2_workspace_bytes", ctypes.byref(cfg), 2, 16) > 0
    assert _lib.call("p2t_esm2_workspace_bytes", ctypes.byref(cfg), 0, 16) == 0
