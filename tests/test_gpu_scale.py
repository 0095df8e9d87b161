"""Parity at BASELINE's full single-GPU size (config 2/3: 1000x1000 lattice, 1 M particles / 3 M beams).

Direct comparison with the oracle for a few substeps (it manages ~10 ms/substep with the GPU box's
cores), then size-independent properties for longer runs: schedule independence (tiled == atomic),
collision-mode independence while nothing touches (grid == off), boundedness, momentum drift.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W = H = 1000
BOUNDS = 32000.0


@pytest.fixture(scope="module")
def scene(sb):
    return sb.scenes.lattice_buffers(W, H, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2)


def gpu_run(sb, buf, n, **kw):
    eng = sb.Engine(bounds_size=BOUNDS, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, **kw)
    eng.write_buffers(buf)
    eng.step(n)
    out = eng.load_buffers(buf.copy())
    info = {k: eng.info(k) for k in ("tiles", "beam_copies", "material_mode", "materials")}
    eng.destroy()
    return out, info


def test_config2_bit_exact_vs_oracle(sb, oracle, scene):
    n = 24
    ref = oracle.OracleEngine(BOUNDS, 10.0, 64, 2, oracle.COLLIDE_OFF, threads=16)
    ref.write_buffers(scene)
    ref.step(n)
    exp = ref.load_buffers(scene.copy())
    got, info = gpu_run(sb, scene, n, collision_mode=0, path=2)
    assert info["tiles"] == 1024 and info["material_mode"] == 2 and info["materials"] == 2   # two full rounds of 512 resident tiles
    assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4"))
    assert got.beams.tobytes() == exp.beams.tobytes()


def test_config2_schedule_and_collision_mode_independence(sb, scene):
    n = 200
    tiled, _ = gpu_run(sb, scene, n, collision_mode=0, path=2)
    atomic, _ = gpu_run(sb, scene, n, collision_mode=0, path=1)
    assert np.array_equal(tiled.particles.view("u4"), atomic.particles.view("u4"))
    assert tiled.beams.tobytes() == atomic.beams.tobytes()
    # config 3's broad phase on a scene where nothing is within 2r (spacing 30, jitter 1): a no-op
    grid, _ = gpu_run(sb, scene, n, collision_mode=2, path=2)
    assert np.array_equal(grid.particles.view("u4"), tiled.particles.view("u4"))
    p = tiled.particles
    assert np.isfinite(p).all()
    assert (p[:, :2] >= 10.0).all() and (p[:, :2] <= BOUNDS - 10.0).all()
    assert not np.array_equal(p, scene.particles)


def test_config3_contacts_bit_exact_vs_oracle(sb, oracle):
    """Config 3 at full size WITH contacts: the same 1 M-particle lattice packed at 19.5 < 2r = 20, so every
    particle starts in contact with its four neighbours (eight within list reach) and the blob bursts
    apart: neighbour lists in use everywhere, hash + lists rebuilt several times.  Bit-exact against the
    oracle's grid mode (itself bit-identical to all-pairs, tests/test_oracle_kat.py)."""
    n = 10
    buf = sb.scenes.lattice_buffers(W, H, d=19.5, origin=(1000.0, 1000.0), jitter=0.2, layout=2)
    ref = oracle.OracleEngine(BOUNDS, 10.0, 64, 2, oracle.COLLIDE_GRID, threads=16)
    ref.write_buffers(buf)
    ref.step(n)
    exp = ref.load_buffers(buf.copy())
    assert np.isfinite(exp.particles).all()
    eng = sb.Engine(bounds_size=BOUNDS, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                    collision_mode=2, path=2)
    eng.write_buffers(buf)
    eng.step(n)
    got = eng.load_buffers(buf.copy())
    builds = eng.info("grid_builds")
    eng.destroy()
    assert builds >= 2
    assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4"))
    assert got.beams.tobytes() == exp.beams.tobytes()
    off, _ = gpu_run(sb, buf, n, collision_mode=0, path=2)
    assert (off.particles != got.particles).any(axis=1).mean() > 0.9   # the contacts really acted


def test_momentum_drift_without_external_forces(sb, scene):
    """Beam forces are equal and opposite in fixed point (compute.wgsl:127-130), so with gravity, drag
    and walls out of the picture total momentum only moves by per-particle float rounding."""
    buf = scene.copy()
    buf.set_physics_constants(gravity=(0.0, 0.0), border_elasticity=0.5, border_friction=0.2, elasticity=0.5,
                              friction=0.1, drag_coeff=0.0, drag_exp=2.0)
    out, _ = gpu_run(sb, buf, 128, collision_mode=0, path=2)
    v = out.particles[:, 2:4].astype(np.float64)
    assert np.abs(v).max() > 1e-3                      # the jittered lattice is really moving
    drift = np.abs(v.sum(axis=0)) / np.abs(v).sum(axis=0)
    assert (drift < 1e-4).all(), drift


def settle_config3(sb, particles):
    """The pile exactly as bench.py's `extra.config3` prepares it: uploaded, settled for 48 frames on the GPU (delete
    passes included), read back."""
    buf, bounds = sb.scenes.config3_buffers(particles)
    eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
    eng.write_buffers(buf)
    for _ in range(sb.scenes.CONFIG3_SETTLE_FRAMES):
        eng.frame()
    settled = eng.load_buffers(buf.copy())
    eng.destroy()
    assert np.isfinite(settled.particles[:settled.particle_count]).all()
    return settled, bounds


def same_bits(got, exp):
    P, B = exp.particle_count, exp.beam_count
    assert (got.particle_count, got.beam_count) == (P, B)
    assert np.array_equal(got.particles[:P].view("u4"), exp.particles[:P].view("u4"))
    assert got.beams.tobytes() == exp.beams.tobytes()
    assert np.array_equal(got.mapping, exp.mapping)


def test_config3_pile_as_benched_bit_exact_vs_oracle(sb, oracle):
    """The scene the driver's line reports as config 3 (`scenes.config3_buffers`: ~1 M particles as a pile of the
    reference's 9 x 4 blobs, touching, settled for 48 frames): from the settled state, 10 substeps on the GPU against the
    oracle's grid mode, bit for bit -- and the contacts are really acting (more than a fifth of the particles differ from
    a collision-free run of the same substeps)."""
    n = 10
    settled, bounds = settle_config3(sb, 1_000_000)
    assert settled.particle_count > 990_000
    ref = oracle.OracleEngine(bounds, 10.0, 64, 2, oracle.COLLIDE_GRID, threads=16)
    ref.write_buffers(settled)
    ref.step(n)
    exp = ref.load_buffers(settled.copy())
    outs = {}
    for mode in (2, 0):
        eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=settled.max_particles, max_beams=settled.max_beams,
                        collision_mode=mode)
        eng.write_buffers(settled)
        eng.step(n)
        outs[mode] = eng.load_buffers(settled.copy())
        eng.destroy()
    same_bits(outs[2], exp)
    P = exp.particle_count
    assert (outs[0].particles[:P] != outs[2].particles[:P]).any(axis=1).mean() > 0.2


def test_config3_pile_65k_against_the_all_pairs_scan(sb, oracle):
    """The same pile at the reference's own capacity (65 536 particles), settled on the GPU, then 6 substeps of the spatial
    hash on the GPU against the oracle running the reference's O(P^2) scan (compute.wgsl:142-170) on the same state."""
    n = 6
    settled, bounds = settle_config3(sb, 65_536)
    ref = oracle.OracleEngine(bounds, 10.0, 64, 2, oracle.COLLIDE_ALLPAIRS, threads=16)
    ref.write_buffers(settled)
    ref.step(n)
    exp = ref.load_buffers(settled.copy())
    eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=settled.max_particles, max_beams=settled.max_beams, collision_mode=2)
    eng.write_buffers(settled)
    eng.step(n)
    got = eng.load_buffers(settled.copy())
    eng.destroy()
    same_bits(got, exp)
