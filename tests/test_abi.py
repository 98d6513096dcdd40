"""The drop-in boundary without a GPU: libspam_hip.so loads, exports every symbol include/spam_hip.h
declares, reports errors instead of crashing, and never falls back to a CPU path."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="spam_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spam_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from sparse_matrix_amd import _lib

    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in spam_hip.h but missing from libspam_hip.so"
    assert sorted(_lib.SYMBOLS) == names, "python binding table and header disagree"
    assert lib.spam_version() == 1002


def test_rccl_sibling_exports_its_header():
    """libspam_hip_rccl.so (the multi-GPU exchange, include/spam_hip_rccl.h) loads next to libspam_hip.so, links RCCL
    and exports / binds exactly what its header declares.  No communicator is created without a GPU."""
    from sparse_matrix_amd import _lib

    lib = _lib.load_rccl()
    names = declared_symbols("spam_hip_rccl.h")
    for n in names:
        assert hasattr(lib, n), f"{n} declared in spam_hip_rccl.h but missing from libspam_hip_rccl.so"
    assert sorted(_lib.SYMBOLS_RCCL) == names, "python binding table and header disagree"
    out = os.popen(f"readelf -d {_lib.LIB_RCCL_PATH}").read()
    assert "librccl.so" in out and "libspam_hip.so" in out
    assert "librccl" not in os.popen(f"readelf -d {_lib.LIB_PATH}").read(), "single-GPU hosts must not need RCCL"


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import numpy as np

    import sparse_matrix_amd as spam

    A = spam.CsrMatrix.identity(4)
    with pytest.raises(spam.SpamError):
        A.spmv(np.ones(4))
    with pytest.raises(spam.SpamError):
        A.mul_hash(A)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "sparse_matrix_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text and "spam_oracle" not in text, f


def test_status_codes_match_header():
    from sparse_matrix_amd import _lib

    text = open(os.path.join(ROOT, "include", "spam_hip.h")).read()
    codes = dict(re.findall(r"#define (SPAM_E_[A-Z_]+) (\d+)", text))
    assert {int(v): k for k, v in codes.items()} == _lib.ERR_NAMES


def test_stats_struct_layout_matches_header():
    from sparse_matrix_amd._lib import SpgemmStats

    # 7 + 8 + 8 + 1 + 4 u64, then 5 floats (+4 bytes tail padding)
    assert C.sizeof(SpgemmStats) == (7 + 16 + 1 + 4) * 8 + 5 * 4 + 4
