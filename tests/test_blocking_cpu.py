"""Host logic of the temporally blocked kernel (csrc/sb_blocking.h: beam-hop rings around every tile, per-substep
prefixes, beam ownership, entry lists) checked on the CPU with a plain g++ build: structural invariants plus a
dependency-exact emulation of the kernel's schedule against k global substeps (tests/blocking_check.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("blocking") / "blocking_check")
    p = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "softbody-webgpu_amd", "csrc"),
                        os.path.join(ROOT, "tests", "blocking_check.cpp"), "-o", out], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return out


@pytest.mark.parametrize("w,h,target,seed,mode,K", [
    (60, 40, 256, 1, 0, 4),     # lattice, several tiles
    (33, 7, 64, 2, 0, 3),       # thin strip, small tiles: regions overlap heavily
    (5, 5, 1024, 3, 0, 4),      # single tile
    (40, 30, 128, 4, 1, 4),     # random graph: long-range, self and parallel beams, NaN/inf positions
    (20, 20, 64, 5, 1, 8),      # K larger than the tiles are wide
    (64, 64, 1024, 6, 0, 1),    # K = 1: the single-substep schedule as a special case
])
def test_blocking_plan(exe, w, h, target, seed, mode, K):
    p = subprocess.run([exe, str(w), str(h), str(target), str(seed), str(mode), str(K)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "BLOCKING_OK" in p.stdout, p.stdout + p.stderr
