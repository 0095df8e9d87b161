"""The Node host (JavaScript mirror of engine.ts / engineMapping.ts / engineWorker.ts + the N-API addon)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_TESTS = os.path.join(ROOT, "softbody-webgpu_amd", "host", "test")

needs_node = pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")


def run_node(script, env=None, timeout=300):
    import __graft_entry__ as ge
    ge.build()
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run(["node", os.path.join(HOST_TESTS, script)], capture_output=True, text=True, env=e, timeout=timeout)
    assert p.returncode == 0, "node %s failed:\n%s\n%s" % (script, p.stdout, p.stderr)
    return json.loads(p.stdout.strip().splitlines()[-1])


@needs_node
def test_js_host_cpu():
    """BufferMapper bytes against the golden snapshots, edit API, lock, addon export surface, and the
    loud failure without a GPU."""
    import torch
    r = run_node("cpu.test.js", {"SOFTBODY_EXPECT_NO_GPU": "0" if torch.cuda.is_available() else "1"})
    assert r["failed"] == 0 and r["passed"] == 12


@needs_node
@pytest.mark.gpu
def test_js_host_gpu_end_to_end():
    """Node -> N-API -> C ABI -> HIP: 2 frames of the default scene equal the oracle golden bit for bit;
    1000 substeps of the config-1 lattice through the worker API."""
    r = run_node("gpu.test.js")
    assert r["ok"] and r["info"]["path"] == 2 and r["info"]["tiles"] >= 4
    assert r["renderedBytes"] > 500 * 500 * 3 and r["colouredPixels"] > 50      # N4: the GPU state rendered, equal to the oracle state's picture


@needs_node
@pytest.mark.gpu
def test_js_bench_small():
    """host/bench.js (config-2 generator + timed stepping through the addon) on a small lattice."""
    import __graft_entry__ as ge
    ge.build()
    p = subprocess.run(["node", os.path.join(ROOT, "softbody-webgpu_amd", "host", "bench.js"), "--width", "64",
                        "--height", "48", "--steps", "64", "--warmup", "8"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["particles"] == 64 * 48 and r["info"]["path"] == 2 and r["particle_steps_per_s_device"] > 0
    assert r["first_particle_y"] < 1000.0


@needs_node
@pytest.mark.gpu
def test_js_two_processes_trade_ghost_zones():
    """Multi-GPU from the product host: two `node` processes, each driving its own engine on its slab of one scene
    (host/halo.js partitionScene + PeerExchanger over the N-API bindings of sb_partition_* / sb_halo_* / sb_peer_*),
    trading ghost zones through IPC-mapped mailboxes on one GPU; together bit-identical to the single engine."""
    r = run_node("halo.gpu.test.js", timeout=600)
    assert r["ok"] and r["particles"] == 1440 and r["exchanges"] == 24 and min(r["ghosts"]) > 0
    # and with beams that break: whole frames through PeerExchanger.frame(), the delete pass agreed between the processes
    r = run_node("halo.gpu.test.js", env={"HALO_FRAMES": "3"}, timeout=600)
    assert r["ok"] and r["particles"] == 1440 and r["frames"] == 3 and r["beamsLeft"] < r["beamsAtStart"] - 100
