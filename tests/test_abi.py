"""CPU: the C-ABI shared library loads and exports every symbol include/mind_hip.h declares."""
import ctypes
import os
import re

from mind_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mind_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mind_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    syms = declared_symbols()
    assert "mind_predict_batch" in syms and "mind_ilqr_solve_trees" in syms
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mind_hip.h but not exported"


def test_python_binding_covers_header():
    _lib.load()
    assert set(declared_symbols()) == set(_lib.EXPORTS)


def test_struct_sizes_match_c_layout():
    # 13 fields: int + 12 pointers (8-byte aligned) ; 6 pointers
    assert ctypes.sizeof(_lib.SceneBatch) == 8 + 12 * 8
    assert ctypes.sizeof(_lib.PredOut) == 6 * 8
    assert ctypes.sizeof(_lib.TensorDesc) == 24


def test_bad_arguments_fail_cleanly_without_gpu():
    lib = _lib.load()
    assert lib.mind_ctx_destroy(None) != 0
    assert lib.mind_predict_batch(None, None, None) != 0
    assert lib.mind_last_error_string(None) == b"null context"
