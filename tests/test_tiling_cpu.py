"""Host logic of the tiled path (csrc/sb_tiling.h: recursive bisection into tiles, cut-beam duplication,
halo lists, tile-local indices) checked on the CPU with a plain g++ build."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("tiling") / "tiling_check")
    p = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "softbody-webgpu_amd", "csrc"),
                        os.path.join(ROOT, "tests", "tiling_check.cpp"), "-o", out], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return out


@pytest.mark.parametrize("w,h,target,seed,mode", [
    (100, 80, 1024, 1, 0),     # lattice, several tiles
    (33, 7, 64, 2, 0),         # thin strip, small tiles
    (5, 5, 1024, 3, 0),        # single tile
    (60, 50, 256, 4, 1),       # random positions, long-range and self beams, NaN/inf positions
    (300, 300, 1024, 5, 0),    # 90 k particles
])
def test_tiling_invariants(exe, w, h, target, seed, mode):
    p = subprocess.run([exe, str(w), str(h), str(target), str(seed), str(mode)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "TILING_OK" in p.stdout, p.stdout + p.stderr
