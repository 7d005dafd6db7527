"""The C++ host mirror (hla-la_amd/host/hlala_host.hpp) compiles with plain g++ and drives the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "test_host_mirror")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "host_cpp", "test_host_mirror.cpp"),
                           "-L", os.path.join(ROOT, "hla-la_amd"), "-lhlala_gpu", "-Wl,-rpath," + os.path.join(ROOT, "hla-la_amd")])
    return exe


def test_host_mirror_compiles_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_host_mirror_extends_chain_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "HOST MIRROR OK ACGTACGTACGT 4..15" in r.stdout, r.stdout + r.stderr
