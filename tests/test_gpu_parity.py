"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (DESIGN.md "parity"): bit-exact.  The force accumulation is integer (order-free) and both
sides run the same canonical IEEE arithmetic, so every particle and beam float must match
bit for bit, for both device schedules (atomic / tiled).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOMIC, TILED = 1, 2
OFF, ALLPAIRS, GRID = 0, 1, 2


def run_both(sb, oracle, buf, *, n=None, frames=0, path=0, mode=ALLPAIRS, bounds=1000.0, radius=10.0,
             subticks=64, tile=0, before=None, ref_mode=None, block=0):
    eng = sb.Engine(bounds_size=bounds, particle_radius=radius, subticks=subticks, layout=buf.layout,
                    max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=mode,
                    path=path, tile_particles=tile, block_substeps=block)
    ref = oracle.OracleEngine(bounds, radius, subticks, buf.layout, mode if ref_mode is None else ref_mode,
                              threads=8)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    if before:
        before(eng, ref)
    for _ in range(frames):
        eng.frame()
        ref.frame()
    if n:
        eng.step(n)
        ref.step(n)
    got = eng.load_buffers(buf.copy())
    exp = ref.load_buffers(buf.copy())
    info = dict(path=eng.info("path"), tiles=eng.info("tiles"), copies=eng.info("beam_copies"))
    eng.destroy()
    return got, exp, info


def assert_same(got, exp, what=""):
    P, B = exp.particle_count, exp.beam_count
    assert (got.particle_count, got.beam_count) == (P, B), what
    assert np.array_equal(got.metadata, exp.metadata), what
    assert np.array_equal(got.mapping, exp.mapping), what
    gp, ep = got.particles.view("<u4"), exp.particles.view("<u4")
    bad = np.nonzero((gp != ep).any(axis=1))[0]
    assert bad.size == 0, "%s: %d particles differ, first %d: %s vs %s" % (
        what, bad.size, bad[0], got.particles[bad[0]], exp.particles[bad[0]])
    assert got.beams.tobytes() == exp.beams.tobytes(), what + ": beam state differs"


@pytest.mark.parametrize("layout", [1, 2])
def test_default_scene_frames_allpairs(sb, oracle, layout):
    """The reference's own default scene (main.ts:188-246), 3 frames of 64 substeps with the
    reference's all-pairs collisions; blobs fall, hit the floor and each other."""
    buf = sb.scenes.default_buffers(layout, 256, 512)
    got, exp, info = run_both(sb, oracle, buf, frames=3, mode=ALLPAIRS)
    assert info["path"] == ATOMIC
    assert_same(got, exp, "default scene v%d" % layout)
    assert not np.array_equal(got.particles, buf.particles)


@pytest.mark.parametrize("path", [ATOMIC, TILED])
def test_default_scene_no_collisions_both_paths(sb, oracle, path):
    buf = sb.scenes.default_buffers(1, 256, 512)
    got, exp, info = run_both(sb, oracle, buf, frames=2, n=7, mode=OFF, path=path, tile=64)
    assert info["path"] == path
    if path == TILED:
        assert info["tiles"] >= 2 and info["copies"] > 299   # cut beams are duplicated
    assert_same(got, exp, "default scene path %d" % path)


@pytest.mark.parametrize("path,tile", [(ATOMIC, 0), (TILED, 256), (TILED, 1024)])
def test_lattice_parity_1000_substeps(sb, oracle, path, tile):
    """BASELINE config 1 shape (32x32 lattice, 1000 substeps, dt=1/64) plus jitter so every beam
    carries force; bit-exact after 1000 substeps on both schedules."""
    buf = sb.scenes.lattice_buffers(32, 32, d=25.0, origin=(100.0, 100.0), spring=50.0, damp=700.0,
                                    yield_strain=0.2, strain_limit=0.5, jitter=2.0, layout=2)
    got, exp, info = run_both(sb, oracle, buf, n=1000, mode=OFF, path=path, tile=tile)
    assert_same(got, exp, "lattice path %d tile %d" % (path, tile))


def test_lattice_64k_tiled(sb, oracle):
    """256x256 = 65 536 particles / 195 585 beams, 64 tiles of 1024, falling onto the floor (border
    response active), 128 substeps; then the same with the tile size the engine picks for itself (256 tiles of
    256: a scene this small fills the card's resident slots with smaller tiles)."""
    buf = sb.scenes.lattice_buffers(256, 256, d=30.0, origin=(40.0, 12.0), jitter=1.0, layout=2,
                                    velocity=(0.5, -3.0))
    got, exp, info = run_both(sb, oracle, buf, n=128, mode=OFF, path=TILED, bounds=8000.0, tile=1024)
    assert info["tiles"] == 64
    assert_same(got, exp, "64k lattice")
    got, exp, info = run_both(sb, oracle, buf, n=128, mode=OFF, path=TILED, bounds=8000.0)
    assert info["tiles"] == 256
    assert_same(got, exp, "64k lattice, automatic tile size")
    assert (got.particles[:, 1] == 10.0).any()  # some particles sit on the floor clamp


def test_yield_break_and_delete(sb, oracle):
    """Plastic yield, break flags and the per-frame delete pass (compute.wgsl:113-121, 205-246;
    canonical stable compaction): a lattice thrown hard at the wall."""
    buf = sb.scenes.lattice_buffers(12, 12, d=30.0, origin=(30.0, 30.0), spring=50.0, damp=100.0,
                                    yield_strain=0.05, strain_limit=0.12, layout=1, velocity=(-40.0, -35.0),
                                    slack=8)
    for path, mode in ((ATOMIC, ALLPAIRS), (TILED, OFF), (ATOMIC, OFF)):
        got, exp, info = run_both(sb, oracle, buf, frames=3, mode=mode, path=path, tile=64)
        assert exp.beam_count < buf.beam_count, "scene must break beams"
        assert_same(got, exp, "break path %d" % path)
    assert (exp.beams["target_length"][:exp.beam_count] != exp.beams["length"][:exp.beam_count]).any()


def test_user_input_and_constants(sb, oracle):
    buf = sb.scenes.default_buffers(1, 256, 512)
    buf.user_strength = 1.5

    def before(eng, ref):
        b = buf.copy()
        b.set_user_input(applied_force=(0.3, 0.1), mouse_pos=(200.0, 150.0), mouse_vel=(4.0, 2.0), mouse_active=True)
        ui = b.user_input_bytes()
        eng.write_user_input(ui)
        ref.write_user_input(ui)
        c = np.array([0.1, -0.8, 0.4, 0.3, 0.6, 0.2, 0.002, 2.5], "f4")  # drag_exp 2.5 -> general pow
        eng.set_physics_constants(c)
        ref.set_physics_constants(c)

    got, exp, _ = run_both(sb, oracle, buf, frames=2, mode=ALLPAIRS, before=before)
    assert_same(got, exp, "user input")
    assert got.metadata.view("<f4")[19] == np.float32(2.5)


def test_nonidentity_mapping(sb, oracle):
    """Slots need not equal data indices (engineMapping.ts:336-339): permute both mappings."""
    buf = sb.scenes.default_buffers(2, 256, 512)
    rng = np.random.default_rng(3)
    P, B = buf.particle_count, buf.beam_count
    pp, bp = rng.permutation(P), rng.permutation(B)
    # move particle i to data index pp[i] + 50, beam j to bp[j] + 100, slots shuffled
    newp = np.zeros_like(buf.particles)
    newp[pp + 50] = buf.particles[:P]
    newb = np.zeros_like(buf.beams)
    bb = buf.beams[:B].copy()
    bb["a"] = pp[bb["a"]] + 50
    bb["b"] = pp[bb["b"]] + 50
    newb[bp + 100] = bb
    buf.particles[:] = newp
    buf.beams[:] = newb
    buf.mapping[:P] = rng.permutation(pp + 50)
    buf.mapping[buf.max_particles:buf.max_particles + B] = rng.permutation(bp + 100)
    for path, mode in ((ATOMIC, ALLPAIRS), (TILED, OFF)):
        got, exp, _ = run_both(sb, oracle, buf, frames=1, mode=mode, path=path, tile=64)
        assert_same(got, exp, "permuted mapping path %d" % path)


def test_odd_substep_count_and_restep(sb, oracle):
    buf = sb.scenes.default_buffers(1, 256, 512)
    eng = sb.Engine(layout=1, max_particles=256, max_beams=512, collision_mode=OFF)
    ref = oracle.OracleEngine(1000.0, 10.0, 64, 1, OFF)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    for n in (3, 5, 1):
        eng.step(n)
        ref.step(n)
        got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
        P = exp.particle_count
        assert np.array_equal(got.particles[:P].view("u4"), exp.particles[:P].view("u4"))
    eng.destroy()


def test_errors_are_loud(sb):
    eng = sb.Engine(layout=1, max_particles=16, max_beams=16, collision_mode=OFF)
    with pytest.raises(sb.engine.EngineError) as ei:
        eng.step(1)
    assert ei.value.status == 5
    buf = sb.Buffers(1, 16, 16)
    beams = np.zeros(1, sb.layout.BEAM_DTYPE[1])
    beams[0]["a"], beams[0]["b"], beams[0]["length"] = 0, 9, 10.0   # endpoint 9 is not an active particle
    buf.set_scene(np.zeros((2, 6), "f4"), beams)
    with pytest.raises(sb.engine.EngineError) as ei:
        eng.write_buffers(buf)
    assert ei.value.status == 1 and "references particle" in str(ei.value)
    wrong = sb.Buffers(1, 32, 16)
    with pytest.raises(sb.engine.EngineError):
        eng.write_buffers(wrong)
    eng.destroy()


@pytest.mark.parametrize("mode,path", [(OFF, TILED), (OFF, ATOMIC), (GRID, TILED), (ALLPAIRS, ATOMIC)])
def test_empty_and_tiny_scenes(sb, oracle, mode, path):
    """Edge inputs: no particles at all, one particle, two particles and one beam; frames (substeps + delete
    pass) and read-back must behave exactly like the oracle."""
    for n_p, n_b in ((0, 0), (1, 0), (2, 1)):
        buf = sb.Buffers(1, 8, 8)
        pts = np.zeros((n_p, 6), "f4")
        pts[:, 0] = 500.0 + 25.0 * np.arange(n_p)
        pts[:, 1] = 300.0
        beams = np.zeros(n_b, sb.layout.BEAM_DTYPE[1])
        if n_b:
            beams[0]["a"], beams[0]["b"] = 0, 1
            for f, v in (("length", 30.0), ("target_length", 30.0), ("last_length", 30.0), ("spring", 50.0), ("damp", 700.0),
                         ("yield_strain", 0.2), ("strain_break_limit", 0.5)):
                beams[0][f] = v
        buf.set_scene(pts, beams)
        got, exp, _ = run_both(sb, oracle, buf, frames=2, n=3, mode=mode, path=path)
        assert_same(got, exp, "%d particles %d beams mode %d" % (n_p, n_b, mode))
        assert got.particle_count == n_p and got.beam_count == n_b


def test_v1_layout_at_full_capacity(sb, oracle):
    """The reference's own limits: 65 536 particles and 65 536 beams behind u16 indices (engineMapping.ts:362-363),
    every slot in use; 40 substeps with the floor in play, bit-exact."""
    p, b = sb.scenes.rectangle(40.0, 12.0, 30.0, 256, 256, 50, 700, 0.2, 1e9, anti_diagonal=False, layout=1)
    b = b[:65536]                                    # the first 65 536 beams in emission order
    pts = np.zeros((65536, 6), "f4")
    pts[:, :2] = p
    pts[:, 3] = -8.0
    buf = sb.Buffers(1, 65536, 65536)
    buf.set_scene(pts, b)
    assert buf.particle_count == 65536 and buf.beam_count == 65536
    got, exp, info = run_both(sb, oracle, buf, n=40, mode=OFF, path=TILED, bounds=8000.0)
    assert_same(got, exp, "v1 at capacity")
    assert (got.particles[:, 1] == 10.0).any()


# ---------------------------------------------------------------- spatial hash (SB_COLLIDE_GRID)

@pytest.mark.parametrize("path", [ATOMIC, TILED])
def test_grid_default_scene_equals_allpairs_oracle(sb, oracle, path):
    """The GPU spatial hash must give the bits of the reference's all-pairs scan
    (compute.wgsl:144-170): same pair set, contacts applied in ascending slot order."""
    buf = sb.scenes.default_buffers(1, 256, 512)
    got, exp, info = run_both(sb, oracle, buf, frames=3, mode=GRID, ref_mode=ALLPAIRS, path=path, tile=64)
    assert info["path"] == path
    assert_same(got, exp, "grid default scene path %d" % path)


def test_default_scene_long_run_default_options(sb, oracle):
    """The reference's default scene for 90 frames (5760 substeps, 1.5 s of its wall clock at 60 fps) with the
    engine's DEFAULT collision mode (spatial hash + neighbour lists, tiled path) against the oracle's all-pairs
    scan: blobs land, pile up, yield and break beams, delete passes run; still bit for bit the same."""
    buf = sb.scenes.default_buffers(1, 256, 512)
    eng = sb.Engine(layout=1, max_particles=buf.max_particles, max_beams=buf.max_beams)   # every option at its default
    assert eng.info("path") == TILED
    ref = oracle.OracleEngine(1000.0, 10.0, 64, 1, ALLPAIRS, threads=4)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    for frame in range(90):
        eng.frame()
        ref.frame()
        if frame in (9, 44):
            assert_same(eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy()), "frame %d" % frame)
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    builds = eng.info("grid_builds")
    eng.destroy()
    assert np.isfinite(exp.particles).all()
    assert_same(got, exp, "default scene, 90 frames")
    assert 1 < builds < 5760


@pytest.mark.parametrize("path", [ATOMIC, TILED])
def test_grid_particle_soup(sb, oracle, path):
    """2000 free particles at ~30 % area coverage with random velocities, one coincident pair and a
    few particles outside the box: many simultaneous contacts per particle."""
    rng = np.random.default_rng(11)
    P = 2000
    pts = np.zeros((P, 6), "f4")
    pts[:, :2] = rng.uniform(10, 1390, (P, 2))
    pts[:, 2:4] = rng.uniform(-6, 6, (P, 2))
    pts[7, :2] = pts[3, :2]
    pts[100, :2] = (-50.0, 700.0)
    pts[101, :2] = (1500.0, 1450.0)
    buf = sb.Buffers(2, P, 4)
    buf.set_scene(pts, np.zeros(0, sb.layout.BEAM_DTYPE[2]))
    got, exp, _ = run_both(sb, oracle, buf, n=96, mode=GRID, ref_mode=ALLPAIRS, path=path, bounds=1400.0, tile=128)
    assert_same(got, exp, "soup path %d" % path)
    assert not np.array_equal(got.particles[:, 2:4], pts[:, 2:4])


@pytest.mark.parametrize("path", [ATOMIC, TILED])
def test_grid_dense_pile_overflows_neighbour_lists(sb, oracle, path):
    """80 particles thrown into a 30x30 patch (plus three on one spot): every one of them has far more than
    SB_NL_CAP = 16 others within 2r + 2*skin, so their neighbour lists overflow and the kernels must fall back
    to scanning the cells -- while 400 spread-out particles around them use their lists.  Bits of all-pairs."""
    rng = np.random.default_rng(23)
    P = 480
    pts = np.zeros((P, 6), "f4")
    pts[:80, :2] = rng.uniform(600, 630, (80, 2))
    pts[5, :2] = pts[4, :2] = pts[3, :2]
    pts[80:, :2] = rng.uniform(10, 1390, (P - 80, 2))
    pts[:, 2:4] = rng.uniform(-3, 3, (P, 2))
    buf = sb.Buffers(2, P, 4)
    buf.set_scene(pts, np.zeros(0, sb.layout.BEAM_DTYPE[2]))
    got, exp, _ = run_both(sb, oracle, buf, n=48, mode=GRID, ref_mode=ALLPAIRS, path=path, bounds=1400.0, tile=128)
    assert np.isfinite(exp.particles).all()
    assert_same(got, exp, "dense pile path %d" % path)


@pytest.mark.parametrize("path", [ATOMIC, TILED])
def test_grid_fast_gas_adapts_its_skin(sb, oracle, path):
    """1444 free particles (grid of 36 jittered by 7) flying at up to 60 units/s per axis (1.3 units per substep, contacts kick some to 3) in a 1400 box for 240
    substeps: hashes with the default skin last one or two substeps, so the skin doubles (twice), cells and
    list reach change on the fly -- and the result is still the all-pairs scan's, bit for bit."""
    buf = sb.scenes.soup_buffers(38, 38, d=36.0, origin=(30.0, 30.0), jitter=7.0, speed=60.0, seed=31)  # nobody overlaps at t=0
    P = buf.particle_count
    eng = sb.Engine(bounds_size=1400.0, layout=2, max_particles=P, max_beams=4, collision_mode=GRID, path=path, tile_particles=128)
    ref = oracle.OracleEngine(1400.0, 10.0, 64, 2, ALLPAIRS, threads=8)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    skins = set()
    for _ in range(6):
        eng.step(40)
        ref.step(40)
        skins.add(eng.info("grid_skin_x1000"))
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    builds = eng.info("grid_builds")
    eng.destroy()
    assert np.isfinite(exp.particles).all()
    assert_same(got, exp, "fast gas path %d" % path)
    assert max(skins) >= 8000, (skins, builds)   # the skin grew from its default 4.0
    assert builds < 120, builds                  # and hashes lasted longer than two substeps on average


@pytest.mark.parametrize("path", [ATOMIC, TILED])
def test_grid_follows_a_scene_that_leaves_its_frame(sb, oracle, path):
    """The hash is framed on the uploaded bounding box plus a margin.  Two small blobs and a sheet of rain thrown
    down at 120 units/s from the top of a 6000 box leave that frame within a few frames; the hash must notice (it
    counts the particles it had to clamp) and re-frame on the whole domain, not pile everybody into its edge
    cells.  Bits of the all-pairs scan throughout, the blobs' mid-air collision included."""
    buf = leaving_scene(sb)
    eng = sb.Engine(bounds_size=6000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                    collision_mode=GRID, path=path, tile_particles=64)
    ref = oracle.OracleEngine(6000.0, 10.0, 64, 2, ALLPAIRS, threads=8)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    assert eng.info("grid_wide") == 0
    for frame in range(45):
        eng.frame()
        ref.frame()
        if frame in (5, 20):
            assert_same(eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy()), "frame %d" % frame)
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    wide = eng.info("grid_wide")
    eng.destroy()
    assert np.isfinite(exp.particles).all()
    assert_same(got, exp, "leaving the frame, path %d" % path)
    assert exp.particles[:exp.particle_count, 1].max() < 4500.0    # everybody is below the tight frame (it ended near y = 4750)
    assert wide == 1


@pytest.mark.parametrize("path", [ATOMIC, TILED])
def test_grid_reframes_with_many_workgroups(sb, oracle, path):
    """The same switch of frame, but on 24 000 particles: the hash build then runs as ~24 workgroups of
    k_grid_maintain, which must all take the tight-or-wide decision from ONE published word (a build that bins
    with two different geometries misses contacts and can scatter records out of bounds).  A sheet of rain with
    random sideways speeds falls out of the uploaded frame and its drops collide on the way; bits of the oracle's
    collision scan (its own uniform grid, pinned to the all-pairs loop by tests/test_oracle_kat.py)."""
    w, h = 200, 120
    n = w * h
    u = sb.scenes.hash_uniform(5, 2 * n).reshape(n, 2)
    pv = np.zeros((n, 6), "f4")
    k = np.arange(n)
    pv[:, 0] = 600.0 + 25.0 * (k % w) + 2.0 * u[:, 0]
    pv[:, 1] = 8600.0 + 25.0 * (k // w)
    pv[:, 2] = 40.0 * u[:, 1]
    pv[:, 3] = -150.0
    buf = sb.Buffers(2, n, 4)
    buf.set_scene(pv, np.zeros(0, dtype=sb.layout.BEAM_DTYPE[2]))
    eng = sb.Engine(bounds_size=12000.0, layout=2, max_particles=n, max_beams=4, collision_mode=GRID, path=path)
    ref = oracle.OracleEngine(12000.0, 10.0, 64, 2, GRID, threads=8)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    assert eng.info("grid_wide") == 0
    for frame in range(14):
        eng.frame()
        ref.frame()
        if frame in (4, 8):
            assert_same(eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy()), "frame %d" % frame)
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    wide = eng.info("grid_wide")
    eng.sync()
    eng.destroy()
    assert np.isfinite(exp.particles).all()
    assert_same(got, exp, "re-framing with many workgroups, path %d" % path)
    assert exp.particles[:n, 1].min() < 4000.0   # far below the uploaded frame (it ended near y = 7850)
    assert wide == 1


def leaving_scene(sb):
    parts, beams, base = [], [], 0
    for ox, vx in ((900.0, 15.0), (1500.0, -15.0)):
        p, b = sb.scenes.rectangle(ox, 5200.0, 30.0, 10, 8, 50, 700, 0.2, 1e9, base=base, anti_diagonal=True, layout=2)
        pv = np.zeros((p.shape[0], 6), "f4")
        pv[:, :2] = p
        pv[:, 2], pv[:, 3] = vx, -120.0
        parts.append(pv)
        beams.append(b)
        base += p.shape[0]
    k = np.arange(125)
    rain = np.zeros((125, 6), "f4")
    rain[:, 0] = 850.0 + 40.0 * (k % 25)
    rain[:, 1] = 5600.0 + 40.0 * (k // 25)
    rain[:, 3] = -130.0
    parts.append(rain)
    buf = sb.Buffers(2, base + 125, sum(b.shape[0] for b in beams))
    buf.set_scene(np.concatenate(parts), np.concatenate(beams))
    return buf


def two_blob_scene(sb):
    """Blob A (64x32 lattice) resting on the floor, blob B (48x24) dropped onto it, 300 free
    particles raining in: beams + inter-blob contacts + floor/wall response together."""
    pa, ba = sb.scenes.rectangle(60.0, 11.0, 30.0, 64, 32, 50, 700, 0.2, 1e9, base=0, anti_diagonal=False, layout=2)
    pb, bb = sb.scenes.rectangle(215.0, 11.0 + 31 * 30 + 24.0, 30.0, 48, 24, 50, 700, 0.2, 1e9, base=pa.shape[0],
                                 anti_diagonal=False, layout=2)
    rng = np.random.default_rng(5)
    nf = 300
    P = np.zeros((pa.shape[0] + pb.shape[0] + nf, 6), "f4")
    P[:pa.shape[0], :2] = pa
    P[pa.shape[0]:pa.shape[0] + pb.shape[0], :2] = pb
    P[-nf:, :2] = rng.uniform([100, 1700], [1900, 2100], (nf, 2)).astype("f4")
    P[pa.shape[0]:pa.shape[0] + pb.shape[0], 3] = -3.0
    P[-nf:, 2:4] = rng.uniform(-3, 3, (nf, 2))
    P[-nf:, 3] -= 20
    P[:, :2] += (sb.scenes.hash_uniform(9, P.shape[0] * 2).reshape(-1, 2) * 0.3).astype("f4")
    B = np.concatenate([ba, bb])
    buf = sb.Buffers(2, P.shape[0], B.shape[0])
    buf.set_scene(P, B)
    return buf


@pytest.mark.parametrize("path", [ATOMIC, TILED])
def test_grid_two_blobs_and_rain(sb, oracle, path):
    """BASELINE config 3 in small (3500 particles): grid collisions + beams + border, 256 substeps,
    against the oracle's grid mode (itself bit-identical to all-pairs, tests/test_oracle_kat.py).
    The scene stays finite: NaN sign/payload bits are hardware-specific and not part of parity."""
    buf = two_blob_scene(sb)
    got, exp, info = run_both(sb, oracle, buf, n=256, mode=GRID, path=path, bounds=4000.0, tile=256)
    assert np.isfinite(exp.particles).all()
    assert_same(got, exp, "two blobs path %d" % path)
    off, _, _ = run_both(sb, oracle, buf, n=256, mode=OFF, path=path, bounds=4000.0, tile=256)
    assert (off.particles != got.particles).any(axis=1).sum() > 1000  # collisions really acted


# ---------------------------------------------------------------- material dictionary fallbacks

@pytest.mark.parametrize("vary,expect_mode", [("none", 2), ("length", 1), ("spring", 0)])
def test_material_dictionary_modes(sb, oracle, vary, expect_mode):
    """The tiled kernel dictionary-encodes the static beam parameters (lossless).  Force each of
    its three encodings -- full rows, rows without length, plain per-copy arrays -- and check
    bit-exact parity in all of them."""
    buf = sb.scenes.lattice_buffers(48, 48, d=25.0, origin=(100.0, 100.0), jitter=2.0, layout=2, strain_limit=0.5)
    B = buf.beam_count
    rng = np.random.default_rng(17)
    if vary == "length":   # every beam its own rest length (like an editor-triangulated scene)
        L = (buf.beams["length"][:B] * rng.uniform(0.97, 1.03, B)).astype("f4")
        buf.beams["length"][:B] = L
        buf.beams["target_length"][:B] = L
        buf.beams["last_length"][:B] = L
    if vary == "spring":
        buf.beams["spring"][:B] = rng.uniform(20, 60, B).astype("f4")
    eng = sb.Engine(bounds_size=4000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                    collision_mode=OFF, path=TILED, tile_particles=256)
    eng.write_buffers(buf)
    assert eng.info("material_mode") == expect_mode
    eng.destroy()
    got, exp, _ = run_both(sb, oracle, buf, n=200, mode=OFF, path=TILED, bounds=4000.0, tile=256)
    assert_same(got, exp, "materials vary=%s" % vary)


def test_force_saturation_and_nonfinite(sb, oracle):
    """i32() saturates at +-2^31 and maps NaN to 0 (compute.wgsl:127-130): on the GPU that is one
    v_cvt_i32_f32, in the oracle explicit range checks.  Absurd spring constants push the
    fixed-point force past both limits; one substep (before positions go non-finite) must agree."""
    P = [[100, 100, 0, 0, 0, 0], [220, 100, 0, 0, 0, 0], [100, 300, 0, 0, 0, 0], [100, 420, 0, 0, 0, 0],
         [400, 400, 0, 0, 0, 0], [400, 400, 0, 0, 0, 0]]
    buf = sb.Buffers(2, 8, 8)
    bb = np.zeros(3, sb.layout.BEAM_DTYPE[2])
    # stretched by 20: force = -20*1e9 -> -2e10*65536 saturates; second beam compressed; third zero-length
    bb[0] = (0, 1, 100, 100, 100, 1e9, 0, 5, 50, 0, 0)
    bb[1] = (2, 3, 140, 140, 140, 3e8, 0, 5, 50, 0, 0)
    bb[2] = (4, 5, 50, 50, 50, 1e30, 0, 5, 50, 0, 0)
    buf.set_scene(np.array(P, "f4"), bb)
    buf.set_physics_constants(gravity=(0.0, 0.0), border_elasticity=0.5, border_friction=0.2, elasticity=0.5,
                              friction=0.1, drag_coeff=0.0, drag_exp=2.0)
    for path in (ATOMIC, TILED):
        got, exp, _ = run_both(sb, oracle, buf, n=1, mode=OFF, path=path, tile=64)
        assert_same(got, exp, "saturation path %d" % path)
        v = exp.particles[:4, 2:4]
        assert np.abs(v).max() == np.float32(32768.0 / 64.0)  # (2^31 / 65536) * dt: the saturated force


def test_user_input_changes_between_frames(sb, oracle):
    """The reference uploads user input every frame (engineWorker.ts:636-642) and constants at any time:
    values changed AFTER kernels have already run must reach the next launch (they travel in the kernarg;
    a device-resident copy read through the scalar cache went stale between launches on this stack)."""
    buf = sb.scenes.default_buffers(1, 256, 512)
    eng = sb.Engine(layout=1, max_particles=256, max_beams=512, collision_mode=ALLPAIRS)
    ref = oracle.OracleEngine(1000.0, 10.0, 64, 1, ALLPAIRS)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    b = buf.copy()
    for k in range(4):
        b.set_user_input(applied_force=(0.2 * k, -0.1 * k), mouse_pos=(200.0 + 30 * k, 150.0), mouse_vel=(1.0 * k, 2.0),
                         mouse_active=bool(k % 2))
        eng.write_user_input(b.user_input_bytes())
        ref.write_user_input(b.user_input_bytes())
        c = np.array([0.05 * k, -0.5 - 0.1 * k, 0.5, 0.2, 0.5, 0.1, 0.001 * (k + 1), 2.0], "f4")
        eng.set_physics_constants(c)
        ref.set_physics_constants(c)
        eng.frame()
        ref.frame()
        got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
        assert_same(got, exp, "frame %d" % k)
    eng.destroy()


def test_mixed_stiffness_substepped(sb, oracle):
    """BASELINE config 5 in small: per-beam springs from {1, 3, 50, 500} (8 dictionary rows with the two
    rest lengths), 128 subticks, dropped on the floor; bit-exact over 3 frames on both schedules."""
    buf = sb.scenes.lattice_buffers(40, 40, d=30.0, origin=(60.0, 12.0), jitter=1.0, layout=2, velocity=(0.5, -2.0),
                                    strain_limit=0.8)
    sb.scenes.mix_stiffness(buf, subticks=128)
    assert len(set(buf.beams["spring"][:buf.beam_count].tolist())) == 4
    for path in (ATOMIC, TILED):
        got, exp, _ = run_both(sb, oracle, buf, frames=3, mode=OFF, path=path, bounds=4000.0, subticks=127, tile=256)
        assert np.isfinite(exp.particles).all()
        assert_same(got, exp, "mixed stiffness path %d" % path)


@pytest.mark.parametrize("mode", [GRID, OFF])
def test_reupload_replaces_everything(sb, oracle, mode):
    """writeBuffers() may be called again at any time (engineWorker.ts:536: every snapshot load): the second
    scene must run as if the engine were new -- different particle/beam counts, tiling, materials, grid.  The device
    blocks of the previous scene are handed to the next one from a pool, stale contents and all (collisions off: the
    temporally blocked plan; the same scene twice in a row: every block is reused as it is)."""
    eng = sb.Engine(bounds_size=4000.0, layout=2, max_particles=4000, max_beams=12000, collision_mode=mode, path=TILED,
                    tile_particles=256)
    for w, h, spring in ((30, 30, 50.0), (50, 20, 20.0), (50, 20, 20.0), (10, 10, 5.0), (40, 30, 50.0)):
        p, b = sb.scenes.rectangle(100.0, 11.0, 30.0, w, h, spring, 300.0, 0.2, 0.5, anti_diagonal=True, layout=2)
        pv = np.zeros((p.shape[0], 6), "f4")
        pv[:, :2] = p
        pv[:, 3] = -3.0
        buf = sb.Buffers(2, 4000, 12000)
        buf.set_scene(pv, b)
        ref = oracle.OracleEngine(4000.0, 10.0, 64, 2, mode, threads=8)
        eng.write_buffers(buf)
        ref.write_buffers(buf)
        eng.step(70)
        ref.step(70)
        eng.frame()
        ref.frame()
        assert_same(eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy()), "re-upload %dx%d" % (w, h))
    eng.destroy()


@pytest.mark.parametrize("tile,mode", [(3000, OFF), (2500, GRID)])
def test_tiles_larger_than_the_prefetch_window(sb, oracle, tile, mode):
    """Tiles above 2 x 512 particles take the tail loops of k_substep_tiled (phase 0, phase 2, grid ranges).  (One substep per
    launch asked for: where several run per launch a tile size is an upper bound and is lowered to the blocked kernel's own
    slots -- test_explicit_tile_size_above_the_blocked_kernels_slots_is_lowered.)"""
    buf = sb.scenes.lattice_buffers(80, 75, d=30.0, origin=(60.0, 12.0), jitter=1.0, layout=2, velocity=(0.5, -3.0))
    got, exp, info = run_both(sb, oracle, buf, n=150, mode=mode, path=TILED, bounds=4000.0, tile=tile, block=1)
    assert info["tiles"] in (2, 3)
    assert_same(got, exp, "tile %d" % tile)
    assert (got.particles[:, 1] == 10.0).any()


def test_explicit_tile_size_above_the_blocked_kernels_slots_is_lowered(sb, oracle):
    """sb_options.tile_particles = 3000 with collisions off: until r04 no depth of the blocked plan fitted such tiles (its kernel
    owns at most 1024 particles and 3072 beams per tile) and the engine silently dropped to one substep per launch;
    now the size is an upper bound: smaller tiles, the blocked kernel, the oracle's bits."""
    buf = sb.scenes.lattice_buffers(80, 75, d=30.0, origin=(60.0, 12.0), jitter=1.0, layout=2, velocity=(0.5, -3.0))
    eng = sb.Engine(bounds_size=4000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=OFF, path=TILED,
                    tile_particles=3000)
    ref = oracle.OracleEngine(4000.0, 10.0, 64, 2, OFF, threads=8)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    assert eng.info("substeps_per_launch") > 1 and eng.info("tiles") >= 6
    eng.step(150)
    ref.step(150)
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    eng.destroy()
    assert_same(got, exp, "tile 3000 lowered")


@pytest.mark.parametrize("subticks", [100, 36])
def test_time_steps_that_are_not_powers_of_two(sb, oracle, subticks):
    """dt^2 is a power of two for the default 64 subticks, and the contact response then multiplies by its exact
    reciprocal instead of dividing (compute.wgsl:168); any other subtick count must take the IEEE division."""
    buf = sb.scenes.default_buffers(1, 256, 512)
    got, exp, _ = run_both(sb, oracle, buf, frames=2, mode=GRID, ref_mode=ALLPAIRS, subticks=subticks)
    assert_same(got, exp, "subticks %d" % subticks)
    pile, bounds = sb.scenes.blob_pile_buffers(6, 3, gap=19.7)
    got, exp, _ = run_both(sb, oracle, pile, n=60, mode=GRID, ref_mode=GRID, bounds=bounds, subticks=subticks)
    assert_same(got, exp, "pile, subticks %d" % subticks)
