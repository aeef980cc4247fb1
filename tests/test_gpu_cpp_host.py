"""Runs the C++ host-mirror tests (tests/cpp/test_reference_kats.cpp: the reference's own unit tests restated over
zk_amd/host/zk.hpp and the C ABI, no Python in the data path) on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_reference_kats")


@pytest.mark.gpu
def test_cpp_reference_kats_on_gpu():
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: 12 reference tests + 1 batch test passed" in r.stdout


@pytest.mark.gpu
def test_cpp_gkr_host_on_gpu():
    """the GKR-shaped driver through zk.hpp (tests/cpp/test_gkr_host.cpp)"""
    gkr_bin = os.path.join(ROOT, "tests", "cpp", "test_gkr_host")
    if not os.path.exists(gkr_bin):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    r = subprocess.run([gkr_bin], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok: gkr host tests passed" in r.stdout


def test_cpp_host_mirror_compiles_and_fails_loudly_without_gpu():
    import torch

    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, capture_output=True)
    assert os.path.exists(BIN)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no CPU fallback" in r.stdout


def test_carry_free_multiplier_equals_saturated_on_host():
    """field.cuh on the CPU: fe_mul29(a, prepare(c)) == fe_mul(a, c) for 3 x 200k random + edge pairs"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "test_field_host"], check=True, capture_output=True)
    r = subprocess.run([os.path.join(ROOT, "tests", "cpp", "test_field_host")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all equal" in r.stdout, r.stdout + r.stderr
