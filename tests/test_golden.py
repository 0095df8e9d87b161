"""Golden fixtures (tests/golden, made by make_golden.py): the oracle must keep reproducing them, and
the HIP path must hit them without the oracle in the loop."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def read(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


def test_layout_reproduces_default_scene_snapshots(sb):
    assert sb.scenes.default_buffers(1).create_snapshot() == read("default_scene_v1.snapshot")
    assert sb.scenes.default_buffers(2, 256, 512).create_snapshot() == read("default_scene_v2.snapshot")


def test_oracle_reproduces_goldens(sb, oracle):
    buf = sb.Buffers(1, 65536, 65536)
    assert buf.load_snapshot(read("default_scene_v1.snapshot"))
    ref = oracle.OracleEngine(1000.0, 10.0, 64, 1, oracle.COLLIDE_ALLPAIRS)
    ref.write_buffers(buf)
    ref.frame()
    ref.frame()
    assert ref.load_buffers(buf.copy()).create_snapshot() == read("default_scene_v1_after_2_frames.snapshot")
    z = np.load(os.path.join(GOLDEN, "lattice_8x6_after_100_substeps.npz"))
    lat = sb.scenes.lattice_buffers(8, 6, d=25.0, origin=(100.0, 100.0), jitter=2.0, layout=2, strain_limit=0.5)
    assert np.array_equal(lat.particles, z["particles_in"]) and np.array_equal(lat.beams.view("u1"), z["beams_in"])
    ref = oracle.OracleEngine(1000.0, 10.0, 64, 2, oracle.COLLIDE_OFF)
    ref.write_buffers(lat)
    ref.step(100)
    out = ref.load_buffers(lat.copy())
    assert np.array_equal(out.particles.view("u4"), z["particles_out"].view("u4"))
    assert np.array_equal(out.beams.view("u1"), z["beams_out"])


@pytest.mark.gpu
def test_gpu_hits_goldens_without_oracle(sb):
    buf = sb.Buffers(1, 65536, 65536)
    assert buf.load_snapshot(read("default_scene_v1.snapshot"))
    eng = sb.Engine(layout=1, collision_mode=1)
    eng.write_buffers(buf)
    eng.frame()
    eng.frame()
    assert eng.load_buffers(buf.copy()).create_snapshot() == read("default_scene_v1_after_2_frames.snapshot")
    eng.destroy()
    z = np.load(os.path.join(GOLDEN, "lattice_8x6_after_100_substeps.npz"))
    lat = sb.scenes.lattice_buffers(8, 6, d=25.0, origin=(100.0, 100.0), jitter=2.0, layout=2, strain_limit=0.5)
    for path in (1, 2):
        eng = sb.Engine(layout=2, max_particles=lat.max_particles, max_beams=lat.max_beams, collision_mode=0,
                        path=path, tile_particles=64)
        eng.write_buffers(lat)
        eng.step(100)
        out = eng.load_buffers(lat.copy())
        eng.destroy()
        assert np.array_equal(out.particles.view("u4"), z["particles_out"].view("u4"))
        assert np.array_equal(out.beams.view("u1"), z["beams_out"])


def config1_lattice(sb):
    return sb.scenes.lattice_buffers(32, 32, d=25.0, origin=(100.0, 100.0), spring=50.0, damp=700.0, yield_strain=0.2,
                                     strain_limit=0.5, layout=2)


def test_oracle_reproduces_the_config1_golden(sb, oracle):
    import hashlib
    import json
    gold = json.load(open(os.path.join(GOLDEN, "lattice_32x32_after_1000_substeps.json")))
    lat = config1_lattice(sb)
    ref = oracle.OracleEngine(1000.0, 10.0, 64, 2, oracle.COLLIDE_OFF)
    ref.write_buffers(lat)
    ref.step(1000)
    snap = ref.load_buffers(lat.copy()).create_snapshot()
    assert len(snap) == gold["bytes"] and hashlib.sha256(snap).hexdigest() == gold["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("path,block", [(1, 0), (2, 1), (2, 0)])
def test_gpu_hits_the_config1_golden_without_oracle(sb, path, block):
    """BASELINE config 1 (32 x 32 lattice, 1000 substeps): atomic schedule, single-substep tiles, blocked tiles."""
    import hashlib
    import json
    gold = json.load(open(os.path.join(GOLDEN, "lattice_32x32_after_1000_substeps.json")))
    lat = config1_lattice(sb)
    eng = sb.Engine(layout=2, max_particles=lat.max_particles, max_beams=lat.max_beams, collision_mode=0, path=path,
                    tile_particles=256, block_substeps=block)
    eng.write_buffers(lat)
    eng.step(1000)
    out = eng.load_buffers(lat.copy())
    eng.destroy()
    assert [float(x) for x in out.particles[0]] == gold["first_particle"]
    assert hashlib.sha256(out.create_snapshot()).hexdigest() == gold["sha256"]
