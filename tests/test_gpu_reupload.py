"""writeBuffers() again with the SAME topology (engineWorker.ts:497-507,580-597: the reference uploads everything on every edit and
every snapshot load): the engine keeps its plan -- bisection, rings, entry lists, material rows, the hash's arrays -- and only
state travels (csrc/sb_api.hip rewrite_scene_state).  The second run must be, bit for bit, what a new engine would have done:
moved particles, changed target / last lengths, beams that the FIRST run's delete passes removed alive again, flags and masks as
uploaded.  Anything that touches the topology (an endpoint, a rest length, a material, the mapping, more beams) plans again --
except an upload that only REMOVED beams (an editor's cut, a snapshot taken after beams broke: engineMapping.ts:500-518 writes the
beams that are left in their old order), which keeps the plan too: the engine's own slots stay, the removed beams die on the
device like beams a delete pass removed, the caller's slots map onto the engine's."""
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
    # 3. what the first run read back (beams gone, the mapping compacted by the delete passes): the same scene minus a few beams --
    # since r04 that keeps the plan as well (test_removed_beams_keep_the_plan), unless the run broke more than an eighth of them
    edit = first.beam_count - got.beam_count <= first.beam_count // 8
    got3, exp3 = play(eng, ref, got)
    assert eng.info("uploads_kept") == (4 if edit else 3) and eng.info("uploads_edited") == (1 if edit else 0)
    assert_same(got3, exp3, "read-back state uploaded")
    # ... and THAT topology again is kept either way
    got4, exp4 = play(eng, ref, moved(got, 9))
    assert eng.info("uploads_kept") == (5 if edit else 4)
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


def without(buf, keep, *, shuffle_mapping=None):
    """`buf` with only the beams keep[] of its first beam_count slots, renumbered in order (what BufferMapper.writeState leaves of a
    scene after removeBeam calls).  shuffle_mapping: a seed -- the records then sit at permuted data indices."""
    out = buf.copy()
    B, maxP = buf.beam_count, buf.max_particles
    recs = buf.beams[buf.mapping[maxP:maxP + B].astype(np.int64)][keep]
    n = len(recs)
    idx = np.arange(n)
    if shuffle_mapping is not None:
        idx = np.random.default_rng(shuffle_mapping).permutation(buf.max_beams)[:n]
    out.beams[idx] = recs
    out.mapping[maxP:maxP + n] = idx
    out.beam_count = n
    return out


@pytest.mark.parametrize("mode,path,kw", [(OFF, TILED, {}), (OFF, TILED, {"block_substeps": 1}), (GRID, TILED, {}), (GRID, ATOMIC, {}),
                                           (OFF, ATOMIC, {})])
def test_removed_beams_keep_the_plan(sb, oracle, mode, path, kw):
    first = breaking_lattice(sb)
    eng = sb.Engine(bounds_size=BOUNDS, layout=2, max_particles=first.max_particles, max_beams=first.max_beams, collision_mode=mode, path=path, **kw)
    ref = oracle.OracleEngine(BOUNDS, 10.0, 64, 2, mode, threads=8)
    got, exp = play(eng, ref, first, frames=1)
    assert_same(got, exp, "first upload")
    rng = np.random.default_rng(11)
    # 1. an edit that cuts 1 % of the beams (and moves everything): plan kept, the cut beams gone from counts, records and mapping
    B = first.beam_count
    keep = rng.random(B) >= 0.01
    second = without(moved(first, 7), keep)
    assert 0 < B - second.beam_count < B // 8
    got2, exp2 = play(eng, ref, second)
    assert eng.info("uploads_kept") == 1 and eng.info("uploads_edited") == 1
    assert exp2.beam_count < second.beam_count, "the run must break beams on top of the cut ones"
    assert_same(got2, exp2, "1 % of the beams cut")
    assert eng.counts() == (second.particle_count, exp2.beam_count)
    # 2. the same buffers again (same topology as the engine's caller-side view): kept, not an edit
    again, _ = play(eng, ref, second)
    assert eng.info("uploads_kept") == 2 and eng.info("uploads_edited") == 1
    assert_same(again, exp2, "the cut scene again")
    # 3. a second cut on top of the first, the records at shuffled data indices this time
    keep2 = rng.random(second.beam_count) >= 0.02
    keep2[:3] = False
    keep2[-2:] = False
    third = without(moved(second, 3), keep2, shuffle_mapping=5)
    got3, exp3 = play(eng, ref, third)
    assert eng.info("uploads_kept") == 3 and eng.info("uploads_edited") == 2
    assert_same(got3, exp3, "a second cut, shuffled mapping")
    # 4. what that run read back (its own breaks on top): still only removals
    got4, exp4 = play(eng, ref, got3)
    assert eng.info("uploads_kept") == 4 and eng.info("uploads_edited") == 3
    assert_same(got4, exp4, "the read-back of a cut scene")
    # 5. a beam comes back (the first upload again): more beams than the engine's caller-side view -- planned again
    got5, exp5 = play(eng, ref, first, frames=1)
    assert eng.info("uploads_kept") == 4
    assert_same(got5, exp5, "the whole scene again")
    eng.destroy()


def test_removed_beams_in_the_reference_layout(sb, oracle):
    first = breaking_lattice(sb, layout=1)
    eng = sb.Engine(bounds_size=BOUNDS, layout=1, max_particles=first.max_particles, max_beams=first.max_beams, collision_mode=GRID)
    ref = oracle.OracleEngine(BOUNDS, 10.0, 64, 1, GRID, threads=8)
    got, exp = play(eng, ref, first, frames=1)
    assert_same(got, exp, "layout 1, first upload")
    keep = np.random.default_rng(2).random(first.beam_count) >= 0.03
    second = without(moved(first, 4), keep)
    got2, exp2 = play(eng, ref, second)
    assert eng.info("uploads_kept") == 1 and eng.info("uploads_edited") == 1
    assert_same(got2, exp2, "layout 1, 3 % of the beams cut")
    # not a subsequence (two of the beams that are left trade places): planned again
    third = second.copy()
    m, a, b = third.mapping, third.max_particles + 5, third.max_particles + 400
    m[a], m[b] = m[b], m[a]
    third.beam_count = third.beam_count - 1
    got3, exp3 = play(eng, ref, third)
    assert eng.info("uploads_kept") == 1
    assert_same(got3, exp3, "layout 1, cut and reordered")
    eng.destroy()


def test_an_edit_of_config2_is_quick(sb):
    """BASELINE config 2 with 1 % of its 3 M beams cut (<= 30 ms against ~125 ms for an upload that plans); the run
    after it equals the run of a new engine given the same buffers."""
    import time
    buf = sb.scenes.lattice_buffers(1000, 1000, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2)
    eng = sb.Engine(bounds_size=32000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=OFF)
    eng.write_buffers(buf)
    eng.step(14)
    state = eng.load_buffers(buf.copy())
    keep = np.random.default_rng(1).random(buf.beam_count) >= 0.01
    cut = without(state, keep)
    t0 = time.perf_counter()
    eng.write_buffers(cut)
    ms = (time.perf_counter() - t0) * 1e3
    assert eng.info("uploads_kept") == 1 and eng.info("uploads_edited") == 1
    eng.step(21)
    a = eng.load_buffers(cut.copy())
    eng.destroy()
    new = sb.Engine(bounds_size=32000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=OFF)
    new.write_buffers(cut)
    new.step(21)
    b = new.load_buffers(cut.copy())
    new.destroy()
    assert np.array_equal(a.particles.view("u4"), b.particles.view("u4")) and a.beams.tobytes() == b.beams.tobytes()
    assert np.array_equal(a.mapping, b.mapping) and a.beam_count == b.beam_count == cut.beam_count
    print("upload with 1 %% of the beams cut, plan kept: %.1f ms" % ms)
    assert ms <= 30.0, ms


def test_removed_beams_under_blocked_launches_of_the_hash(sb, oracle):
    """A quiet lattice in the default collision mode runs blocked launches under the hash (the hybrid plan beside the tiling); beams
    an upload removed have to die in THAT plan as well (k_hybrid_sync_dead, from the tiled layout's endpoint words)."""
    buf = sb.scenes.lattice_buffers(128, 96, d=30.0, origin=(300.0, 900.0), jitter=1.0, layout=2, velocity=(0.4, -1.0))
    eng = sb.Engine(bounds_size=6000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=GRID)
    ref = oracle.OracleEngine(6000.0, 10.0, 64, 2, GRID, threads=16)
    eng.write_buffers(buf); ref.write_buffers(buf)
    eng.step(150); ref.step(150)
    state = eng.load_buffers(buf.copy())
    assert_same(state, ref.load_buffers(buf.copy()), "before the cut")
    before = eng.info("hybrid_substeps")
    assert before >= 60
    cut = without(state, np.random.default_rng(3).random(buf.beam_count) >= 0.02)
    eng.write_buffers(cut); ref.write_buffers(cut)
    assert eng.info("uploads_kept") == 1 and eng.info("uploads_edited") == 1
    for n in (150, 64):
        eng.step(n); ref.step(n)
    got, exp = eng.load_buffers(cut.copy()), ref.load_buffers(cut.copy())
    assert eng.info("hybrid_substeps") - before >= 100, "the run after the cut must go blocked again"
    assert_same(got, exp, "2 % of the beams cut, hybrid")
    eng.destroy()
