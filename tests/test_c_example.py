"""include/softbody.h is a C header: it must compile as C99, and a plain-C program linked against the
engine must run (GPU)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "softbody-webgpu_amd", "csrc")


def test_header_is_valid_c99():
    p = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                        os.path.join(ROOT, "include", "softbody.h")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr


def build_example(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / "c_abi_frame")
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "c_abi_frame.c"), "-o", exe, "-L" + CSRC, "-lsoftbody_hip",
                        "-Wl,-rpath," + CSRC], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_c_example_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_c_example_runs(tmp_path):
    exe = build_example(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "C_ABI_OK" in p.stdout, p.stdout + p.stderr
