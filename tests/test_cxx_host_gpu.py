"""The C++ host-side mirror (include/spam_csr.hpp) through the C ABI, compiled with g++ and run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "cxx_host_smoke")
    lib_dir = os.path.join(ROOT, "sparse_matrix_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx_host_smoke.cpp"), "-o", exe, "-L", lib_dir, "-lspam_hip",
                           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cxx_header_compiles_and_links(tmp_path):
    """No GPU needed: the header compiles as plain C++17 and every ABI symbol it uses resolves."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_cxx_host_smoke(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "cxx host smoke ok" in out.stdout, out.stdout + out.stderr
