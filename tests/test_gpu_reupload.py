"""writeBuffers() again with the SAME topology (engineWorker.ts:497-507,580-597: the reference uploads everything on every edit and
every snapshot load): the engine keeps its plan -- bisection, rings, entry lists, material rows, the hash's arrays -- and only
state travels (csrc/sb_api.hip rewrite_scene_state).  The second run must be, bit for bit, what a new engine would have done:
moved particles, changed target / last lengths, beams that the FIRST run's delete passes removed alive again, flags and masks as
uploaded.  Anything that touches the topology (an endpoint, a rest length, a material, a count, the mapping) plans again."""
import numpy as np
import pytest

from test_gpu_parity import ATOMIC, GRID, OFF, TILED, assert_same

pytestmark = pytest.mark.gpu
BOUNDS = 4000.0


def breaking_lattice(sb, *, distinct_lengths=False, layout=2):
    """A 40 x 30 lattice thrown into the corner: yields, breaks beams, delete passes remove them."""
    buf = sb.scenes.lattice_buffers(40, 30, d=30.0, origin=(30.0, 30.0), spring=50.0, damp=100.0, yield_strain=0.05, strain_limit=0.12,
                                    layout=layout, velocity=(-40.0, -35.0), slack=8, jitter=0.5)
    if distinct_lengths:  # material mode 1: every beam its own rest length (as the reference's editor makes them)
        buf = sb.scenes.rest_at_current_length(buf)
    return buf


def moved(buf, seed):
    """The same scene after an edit that leaves the topology alone: everything nudged and thrown elsewhere, some beams
    pre-stretched (target != length: their tiles start as yielded), stale strain / stress values in the records."""
    out = buf.copy()
    rng = np.random.default_rng(seed)
    P, B = out.particle_count, out.beam_count
    out.particles[:P, :2] += rng.uniform(-1.5, 1.5, (P, 2)).astype("f4") + np.float32(25.0)
    out.particles[:P, 2:4] = np.asarray((-55.0, -20.0), "f4") + rng.uniform(-1.0, 1.0, (P, 2)).astype("f4")
    out.particles[:P, 4:6] = rng.uniform(-0.1, 0.1, (P, 2)).astype("f4")
    pick = rng.random(B) < 0.05
    out.beams["target_length"][:B][pick] *= np.float32(1.01)
    out.beams["last_length"][:B] = rng.uniform(29.0, 31.0, B).astype("f4")
    out.beams["strain"][:B] = rng.uniform(-1.0, 1.0, B).astype("f4")
    out.beams["stress"][:B] = rng.uniform(-1.0, 1.0, B).astype("f4")
    return out


def play(eng, ref, buf, frames=3, n=9):
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    for _ in range(frames):
        eng.frame()
        ref.frame()
    eng.step(n)
    ref.step(n)
    return eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())


@pytest.mark.parametrize("mode,path,distinct", [(OFF, TILED, False), (OFF, TILED, True), (GRID, TILED, False), (GRID, ATOMIC, False),
                                                (OFF, ATOMIC, False)])
def test_same_topology_keeps_the_plan(sb, oracle, mode, path, distinct):
    first = breaking_lattice(sb, distinct_lengths=distinct)
    eng = sb.Engine(bounds_size=BOUNDS, layout=2, max_particles=first.max_particles, max_beams=first.max_beams, collision_mode=mode, path=path)
    ref = oracle.OracleEngine(BOUNDS, 10.0, 64, 2, mode, threads=8)
    got, exp = play(eng, ref, first)
    assert exp.beam_count < first.beam_count, "the first run must break beams"
    assert_same(got, exp, "first upload")
    assert eng.info("uploads_kept") == 0
    if path == TILED and mode == OFF:
        assert eng.info("plan_depth") > 1
    # 1. same topology, everything else different: plan kept, the dead beams of the first run alive again
    second = moved(first, 7)
    got2, exp2 = play(eng, ref, second)
    assert eng.info("uploads_kept") == 1
    assert_same(got2, exp2, "same topology, moved")
    assert not np.array_equal(got2.particles, got.particles)
    # 2. the very same buffers again, and once more: kept each time, same answer each time
    for k in (2, 3):
        again, _ = play(eng, ref, second)
        assert eng.info("uploads_kept") == k
        assert_same(again, exp2, "same upload again")
    # 3. what the first run read back (beams gone, mapping compacted): a different topology, planned again
    got3, exp3 = play(eng, ref, got)
    assert eng.info("uploads_kept") == 3
    assert_same(got3, exp3, "read-back state uploaded")
    # ... and THAT topology twice in a row is kept again
    got4, exp4 = play(eng, ref, moved(got, 9))
    assert eng.info("uploads_kept") == 4
    assert_same(got4, exp4, "read-back topology, moved")
    eng.destroy()


@pytest.mark.parametrize("mode", [OFF, GRID])
def test_same_topology_in_the_reference_layout(sb, oracle, mode):
    """Layout 1 (the reference's own bytes: u16 endpoints packed in one word, u16 mapping): the comparison of the records and the
    state that travels read the other record format."""
    first = breaking_lattice(sb, layout=1)
    eng = sb.Engine(bounds_size=BOUNDS, layout=1, max_particles=first.max_particles, max_beams=first.max_beams, collision_mode=mode)
    ref = oracle.OracleEngine(BOUNDS, 10.0, 64, 1, mode, threads=8)
    got, exp = play(eng, ref, first)
    assert exp.beam_count < first.beam_count
    assert_same(got, exp, "layout 1, first upload")
    second = moved(first, 5)
    got2, exp2 = play(eng, ref, second)
    assert eng.info("uploads_kept") == 1
    assert_same(got2, exp2, "layout 1, same topology")
    third = second.copy()
    third.beams["b"][3] = (int(third.beams["b"][3]) + 2) % third.particle_count   # another endpoint: planned again
    got3, exp3 = play(eng, ref, third)
    assert eng.info("uploads_kept") == 1
    assert_same(got3, exp3, "layout 1, endpoint changed")
    eng.destroy()


@pytest.mark.parametrize("what", ["endpoint", "rest length", "spring", "mapping", "constants only"])
def test_what_counts_as_the_same_topology(sb, oracle, what):
    first = breaking_lattice(sb)
    eng = sb.Engine(bounds_size=BOUNDS, layout=2, max_particles=first.max_particles, max_beams=first.max_beams, collision_mode=OFF)
    ref = oracle.OracleEngine(BOUNDS, 10.0, 64, 2, OFF, threads=8)
    got, exp = play(eng, ref, first, frames=1)
    assert_same(got, exp, "first")
    second = moved(first, 3)
    B, P = second.beam_count, second.particle_count
    kept = 0
    if what == "endpoint":       # one beam now ends on another particle
        second.beams["b"][17] = (second.beams["b"][17] + 2) % P
        assert second.beams["a"][17] != second.beams["b"][17]
    elif what == "rest length":
        second.beams["length"][B // 2] *= np.float32(1.25)
    elif what == "spring":
        second.beams["spring"][B - 1] = np.float32(77.0)
    elif what == "mapping":      # two beam slots trade records: same beams, another slot order
        m = second.mapping
        a, b = second.max_particles + 5, second.max_particles + 900
        m[a], m[b] = m[b], m[a]
    else:                         # gravity and drag changed: not topology
        second.set_physics_constants(gravity=(0.3, -0.8), border_elasticity=0.4, border_friction=0.1, elasticity=0.5, friction=0.1,
                                     drag_coeff=0.002, drag_exp=2.0)
        kept = 1
    got2, exp2 = play(eng, ref, second, frames=2)
    assert eng.info("uploads_kept") == kept
    assert_same(got2, exp2, what)
    eng.destroy()


def test_second_upload_of_config2_is_quick(sb):
    """BASELINE config 2 (1 M particles, 3 M beams): the upload that plans takes ~120 ms, the one that keeps the plan a
    quarter of that (<= 30 ms)."""
    import time
    buf = sb.scenes.lattice_buffers(1000, 1000, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2)
    eng = sb.Engine(bounds_size=32000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=OFF)
    eng.write_buffers(buf)
    eng.step(14)
    a = eng.load_buffers(buf.copy())
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.write_buffers(buf)
        t.append((time.perf_counter() - t0) * 1e3)
    assert eng.info("uploads_kept") == 3
    eng.step(14)
    b = eng.load_buffers(buf.copy())
    eng.destroy()
    assert np.array_equal(a.particles.view("u4"), b.particles.view("u4")) and a.beams.tobytes() == b.beams.tobytes()
    print("re-uploads that keep the plan: %s ms" % ", ".join("%.1f" % x for x in t))
    assert min(t) <= 30.0, t
