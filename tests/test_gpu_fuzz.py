"""Seeded fuzzing of the collision path: random mixtures of free particles and small lattice blobs, random
speeds, random physics constants and user input, the engine's spatial hash (both device schedules, small tiles)
against the oracle's ALL-PAIRS scan.  Every case must match bit for bit at every checkpoint.  The generator keeps
the cases finite (NaN sign and payload bits are hardware-specific and outside the parity contract): no overlap
inside the free cloud at t = 0, drag parameters inside the explicit integrator's stable range.  Speeds still reach
hundreds of units per second in some cases (hash rebuilt every substep, skin backing off), a few in others
(skin growing), with beams yielding and breaking in between."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OFF, ALLPAIRS, GRID = 0, 1, 2
# SB_FUZZ_OFFSET=n shifts every seed by n (a soak run over other cases than the committed ones); SB_FUZZ_COUNT widens it
import os
SEED0 = int(os.environ.get("SB_FUZZ_OFFSET", "0"))
NGRID = int(os.environ.get("SB_FUZZ_COUNT", "64"))


def make_case(sb, seed):
    rng = np.random.default_rng(1000 + seed)
    bounds = float(rng.choice([700.0, 1000.0, 1600.0]))
    parts, beams, base = [], [], 0
    for _ in range(int(rng.integers(0, 4))):                      # a few lattice blobs
        w, h = int(rng.integers(2, 9)), int(rng.integers(2, 9))
        d = float(rng.uniform(24.0, 40.0))
        ox, oy = rng.uniform(20, bounds - 20 - w * d), rng.uniform(20, bounds - 20 - h * d)
        p, b = sb.scenes.rectangle(ox, oy, d, w, h, float(rng.choice([1, 3, 50, 500])), float(rng.choice([10, 50, 700])),
                                   float(rng.uniform(0.05, 2.0)), float(rng.choice([0.5, 2.5, 1e9])), base=base,
                                   anti_diagonal=bool(rng.integers(0, 2)), layout=2)
        pv = np.zeros((p.shape[0], 6), "f4")
        pv[:, :2] = p + rng.uniform(-1, 1, p.shape).astype("f4")
        pv[:, 2:4] = rng.uniform(-20, 20, 2).astype("f4")
        parts.append(pv)
        beams.append(b)
        base += p.shape[0]
    # ... and a cloud of free particles on a jittered grid (nobody of the cloud overlaps at t = 0: a random
    # overlap of 10 units is a 320 units/s kick per substep and the case would be all NaN within a frame)
    g = float(rng.uniform(27.0, 45.0))
    side = int((bounds - 40.0) // g)
    cells = rng.permutation(side * side)[: int(rng.integers(50, min(600, side * side)))]
    nfree = cells.size
    pv = np.zeros((nfree, 6), "f4")
    pv[:, 0] = 20.0 + (cells % side) * g + rng.uniform(-3, 3, nfree)
    pv[:, 1] = 20.0 + (cells // side) * g + rng.uniform(-3, 3, nfree)
    pv[:, 2:4] = rng.uniform(-1, 1, (nfree, 2)) * float(rng.choice([5.0, 30.0, 80.0]))
    if rng.integers(0, 3) == 0 and nfree > 4:
        pv[1, :2] = pv[0, :2]                                      # a coincident pair (+-1 shift, compute.wgsl:151-154)
    parts.append(pv)
    P = np.concatenate(parts)
    B = np.concatenate(beams) if beams else np.zeros(0, sb.layout.BEAM_DTYPE[2])
    buf = sb.Buffers(2, P.shape[0] + 7, B.shape[0] + 5)
    buf.set_scene(P, B)
    # drag is integrated explicitly: keep coeff * v^(exp-1) * dt well below 1 or the case oscillates into NaN
    drag_coeff, drag_exp = [(0.0, 2.0), (0.002, 2.0), (0.002, 2.5), (0.00002, 3.0), (0.02, 1.0)][int(rng.integers(0, 5))]
    consts = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-1.0, 0.2), rng.uniform(0, 1), rng.uniform(0, 1),
                       rng.uniform(0, 1), rng.uniform(0, 1), drag_coeff, drag_exp], "f4")
    ui = buf.copy()
    ui.user_strength = float(rng.uniform(0.5, 2.0))
    ui.set_user_input(applied_force=tuple(rng.uniform(-0.3, 0.3, 2)), mouse_pos=tuple(rng.uniform(0, bounds, 2)),
                      mouse_vel=tuple(rng.uniform(-5, 5, 2)), mouse_active=bool(rng.integers(0, 2)))
    return buf, bounds, consts, ui.user_input_bytes(), int(rng.choice([64, 128])), int(rng.integers(3, 8))


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + NGRID))
def test_random_scene_grid_equals_allpairs(sb, oracle, seed):
    buf, bounds, consts, ui, tile, chunks = make_case(sb, seed)
    path = 1 + seed % 2
    eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                    collision_mode=GRID, path=path, tile_particles=tile)
    ref = oracle.OracleEngine(bounds, 10.0, 64, 2, ALLPAIRS, threads=8)
    for e in (eng, ref):
        e.write_buffers(buf)
        e.write_user_input(ui)
        e.set_physics_constants(consts)
    compared = 0
    for k in range(chunks):
        n = 17 + 9 * k                      # odd and even substep counts, frames in between
        eng.step(n)
        ref.step(n)
        if k % 2:
            eng.frame()
            ref.frame()
        got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
        if not np.isfinite(exp.particles[:exp.particle_count]).all():
            break
        assert (got.particle_count, got.beam_count) == (exp.particle_count, exp.beam_count)
        assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")), "seed %d chunk %d" % (seed, k)
        assert got.beams.tobytes() == exp.beams.tobytes(), "seed %d chunk %d beams" % (seed, k)
        assert np.array_equal(got.mapping, exp.mapping)
        compared += 1
    eng.destroy()
    assert compared == chunks, "seed %d went non-finite after %d of %d checkpoints" % (seed, compared, chunks)


def make_beam_case(sb, seed):
    """Collisions off (the temporally blocked kernel's domain): several lattice blobs of different shapes and materials,
    some joined by long beams (rings then grow across blobs), with yield and break limits in reach, random constants."""
    rng = np.random.default_rng(5000 + seed)
    scale = int(os.environ.get("SB_FUZZ_SCALE", "1"))      # (soak runs: blobs of up to 24 * scale particles a side, many tiles each)
    bounds = float(rng.choice([1000.0, 1600.0, 3000.0])) * scale
    parts, beams, base, anchors = [], [], 0, []
    for _ in range(int(rng.integers(2, 7))):
        w, h = int(rng.integers(2, 24 * scale)), int(rng.integers(2, 24 * scale))
        d = float(rng.uniform(12.0, 40.0))
        ox, oy = rng.uniform(20, max(21.0, bounds - 20 - w * d)), rng.uniform(20, max(21.0, bounds - 20 - h * d))
        p, b = sb.scenes.rectangle(ox, oy, d, w, h, float(rng.choice([1, 3, 50, 500])), float(rng.choice([10, 50, 700])),
                                   float(rng.choice([0.02, 0.2, 2.0])), float(rng.choice([0.1, 0.5, 1e9])), base=base,
                                   anti_diagonal=bool(rng.integers(0, 2)), layout=2)
        pv = np.zeros((p.shape[0], 6), "f4")
        pv[:, :2] = p + rng.uniform(-1.5, 1.5, p.shape).astype("f4")
        pv[:, 2:4] = rng.uniform(-30, 30, 2).astype("f4")
        parts.append(pv)
        beams.append(b)
        anchors.append((base, p.shape[0], pv))
        base += p.shape[0]
    extra = np.zeros(int(rng.integers(0, 6)), sb.layout.BEAM_DTYPE[2])   # long beams between blobs, arbitrary rest lengths
    for k in range(extra.size):
        (b0, n0, p0), (b1, n1, p1) = [anchors[i] for i in rng.choice(len(anchors), 2, replace=False)]
        i, j = int(rng.integers(0, n0)), int(rng.integers(0, n1))
        L = np.float32(np.hypot(*(p0[i, :2] - p1[j, :2])) * rng.uniform(0.8, 1.2))
        extra[k] = (b0 + i, b1 + j, L, L, L, rng.choice([1.0, 3.0]), 50.0, 2.0, 1e9, 0.0, 0.0)
    P = np.concatenate(parts)
    B = np.concatenate(beams + [extra])
    buf = sb.Buffers(2, P.shape[0] + 3, B.shape[0] + 9)
    buf.set_scene(P, B)
    drag_coeff, drag_exp = [(0.0, 2.0), (0.002, 2.0), (0.002, 2.5), (0.00002, 3.0), (0.02, 1.0)][int(rng.integers(0, 5))]
    consts = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-1.0, 0.2), rng.uniform(0, 1), rng.uniform(0, 1),
                       rng.uniform(0, 1), rng.uniform(0, 1), drag_coeff, drag_exp], "f4")
    ui = buf.copy()
    ui.user_strength = float(rng.uniform(0.5, 2.0))
    ui.set_user_input(applied_force=tuple(rng.uniform(-0.3, 0.3, 2)), mouse_pos=tuple(rng.uniform(0, bounds, 2)),
                      mouse_vel=tuple(rng.uniform(-5, 5, 2)), mouse_active=bool(rng.integers(0, 2)))
    return buf, bounds, consts, ui.user_input_bytes(), int(rng.choice([64, 160, 512])), int(rng.integers(1, 9))


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + max(NGRID // 2, 1)))
def test_random_beam_scene_blocked_equals_oracle(sb, oracle, seed):
    """Seeded fuzzing of the temporally blocked kernel: random block depth (1-8), tile size, scene and constants, odd
    substep counts with frames (delete passes) in between; bit for bit at every checkpoint, mapping included."""
    buf, bounds, consts, ui, tile, K = make_beam_case(sb, seed)
    eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                    collision_mode=OFF, path=2, tile_particles=tile, block_substeps=K)
    ref = oracle.OracleEngine(bounds, 10.0, 64, 2, OFF, threads=8)
    for e in (eng, ref):
        e.write_buffers(buf)
        e.write_user_input(ui)
        e.set_physics_constants(consts)
    for k in range(4):
        n = 13 + 11 * k
        eng.step(n)
        ref.step(n)
        if k % 2:
            eng.frame()
            ref.frame()
        got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
        assert np.isfinite(exp.particles[:exp.particle_count]).all(), "seed %d is not a finite case" % seed
        assert (got.particle_count, got.beam_count) == (exp.particle_count, exp.beam_count)
        assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")), "seed %d chunk %d" % (seed, k)
        assert got.beams.tobytes() == exp.beams.tobytes(), "seed %d chunk %d beams" % (seed, k)
        assert np.array_equal(got.mapping, exp.mapping)
    eng.destroy()


def make_quiet_case(sb, seed):
    """Grid mode with room to be quiet in: lattice blobs dealt onto the squares of a coarse board (nobody overlaps anybody at
    t = 0), drifting slowly, falling under gravity until they land on the floor, the walls or each other; yield and break
    limits in reach; random tile size, block depth, constants and user input.  The engine's hybrid path goes
    blocked while the closest listed pair stays clear, back to single substeps when somebody is about to touch."""
    rng = np.random.default_rng(9000 + seed)
    scale = float(os.environ.get("SB_FUZZ_SCALE", "1"))      # (soak runs: 6 makes blobs of up to 80 x 80 particles, many tiles each)
    bounds = float(rng.choice([1600.0, 3000.0])) * scale
    square = 420.0 * scale
    side = int((bounds - 40.0) // square)
    squares = rng.permutation(side * side)[: int(rng.integers(2, min(9, side * side)))]
    parts, beams, base = [], [], 0
    for sq in squares:
        d = float(rng.uniform(24.0, 34.0))
        w, h = int(rng.integers(3, int(360.0 * scale // d))), int(rng.integers(3, int(360.0 * scale // d)))
        ox, oy = 30.0 + (sq % side) * square + rng.uniform(0, 20), 30.0 + (sq // side) * square + rng.uniform(0, 20)
        p, b = sb.scenes.rectangle(ox, oy, d, w, h, float(rng.choice([3, 50, 500])), float(rng.choice([50, 700])),
                                   float(rng.choice([0.02, 0.2, 2.0])), float(rng.choice([0.1, 0.5, 1e9])), base=base,
                                   anti_diagonal=bool(rng.integers(0, 2)), layout=2)
        pv = np.zeros((p.shape[0], 6), "f4")
        pv[:, :2] = p + rng.uniform(-0.8, 0.8, p.shape).astype("f4")
        pv[:, 2:4] = rng.uniform(-6, 6, 2).astype("f4")
        parts.append(pv)
        beams.append(b)
        base += p.shape[0]
    P, B = np.concatenate(parts), np.concatenate(beams)
    if rng.integers(0, 2):   # editor-style rest lengths: material mode 1, neighbours listed (28 .. 34 apart) but not touching
        dx = P[B["b"], 0] - P[B["a"], 0]
        dy = P[B["b"], 1] - P[B["a"], 1]
        ln = np.sqrt(dx * dx + dy * dy, dtype=np.float32)
        for f in ("length", "target_length", "last_length"):
            B[f] = ln
    buf = sb.Buffers(2, P.shape[0] + 5, B.shape[0] + 3)
    buf.set_scene(P, B)
    consts = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.8, -0.1), rng.uniform(0, 1), rng.uniform(0, 1),
                       rng.uniform(0, 1), rng.uniform(0, 1), 0.002, 2.0], "f4")
    ui = buf.copy()
    ui.user_strength = float(rng.uniform(0.5, 2.0))
    ui.set_user_input(applied_force=tuple(rng.uniform(-0.1, 0.1, 2)), mouse_pos=tuple(rng.uniform(0, bounds, 2)),
                      mouse_vel=tuple(rng.uniform(-5, 5, 2)), mouse_active=bool(rng.integers(0, 3) == 0))   # (one case in three: the
    return buf, bounds, consts, ui.user_input_bytes(), int(rng.choice([0, 128, 512])), int(rng.choice([0, 0, 3, 5]))  # particle phase with the mouse terms)


BLOCKED_SUBSTEPS = {}


# (1091: found by a soak run -- beams removed after their last substeps ran one by one; a later blocked run's way back into the
# tiled layout overwrote the records they died with)
@pytest.mark.parametrize("seed", list(range(SEED0, SEED0 + max(NGRID // 2, 1))) + ([1091] if SEED0 == 0 else []))
def test_random_quiet_scene_hybrid_equals_oracle(sb, oracle, seed):
    """Seeded fuzzing of the hybrid path against the oracle's grid mode (itself the all-pairs scan, bit for bit): substep runs and
    frames with delete passes, read back and compared at every checkpoint."""
    buf, bounds, consts, ui, tile, K = make_quiet_case(sb, seed)
    eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                    collision_mode=GRID, tile_particles=tile, block_substeps=K)
    ref = oracle.OracleEngine(bounds, 10.0, 64, 2, GRID, threads=8)
    for e in (eng, ref):
        e.write_buffers(buf)
        e.write_user_input(ui)
        e.set_physics_constants(consts)
    for k in range(5):
        n = 90 + 37 * k
        eng.step(n)
        ref.step(n)
        if k % 2:
            for _ in range(2):
                eng.frame()
                ref.frame()
        got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
        assert np.isfinite(exp.particles[:exp.particle_count]).all(), "seed %d is not a finite case" % seed
        assert (got.particle_count, got.beam_count) == (exp.particle_count, exp.beam_count)
        assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")), "seed %d chunk %d" % (seed, k)
        assert got.beams.tobytes() == exp.beams.tobytes(), "seed %d chunk %d beams" % (seed, k)
        assert np.array_equal(got.mapping, exp.mapping)
    BLOCKED_SUBSTEPS[seed] = (eng.info("hybrid_substeps"), eng.info("substeps_done"), eng.info("hybrid_failed"), eng.info("hybrid_launches"))
    eng.destroy()


def test_the_quiet_scenes_really_ran_blocked():
    """(after the cases above) a good part of those substeps went through tracked blocked launches, some were refused and redone"""
    if not BLOCKED_SUBSTEPS:
        pytest.skip("the hybrid fuzz cases did not run in this session")
    blocked = sum(v[0] for v in BLOCKED_SUBSTEPS.values())
    done = sum(v[1] for v in BLOCKED_SUBSTEPS.values())
    engaged = sum(1 for v in BLOCKED_SUBSTEPS.values() if v[0] > 0)
    print("hybrid fuzz: %d of %d substeps in blocked launches, %d of %d cases engaged, %d launches validated, %d refused"
          % (blocked, done, engaged, len(BLOCKED_SUBSTEPS), sum(v[3] for v in BLOCKED_SUBSTEPS.values()),
             sum(v[2] for v in BLOCKED_SUBSTEPS.values())))
    assert engaged * 2 >= len(BLOCKED_SUBSTEPS) and blocked * 10 >= done


@pytest.mark.parametrize("seed", range(SEED0, SEED0 + max(NGRID // 4, 1)))
def test_random_upload_sequences_keep_or_replan(sb, oracle, seed):
    """One engine, a random sequence of uploads: the same topology with everything else changed (the plan is kept), the state just
    read back (beams gone: since r04 an edit that keeps the plan too), a cut of a few per cent of the beams, another scene of the same capacity, the same buffers twice -- each followed by substeps
    and frames and compared with a fresh oracle.  Collisions off (blocked plan) or on (tiling + hybrid), tiled or atomic path."""
    rng = np.random.default_rng(7000 + seed)
    mode = GRID if rng.integers(0, 2) else OFF
    path = 2 if rng.integers(0, 4) else 1
    make = make_quiet_case if mode == GRID else make_beam_case
    scenes = [make(sb, 10 * seed + j) for j in range(2)]
    cap_p = max(s[0].max_particles for s in scenes) + 4
    cap_b = max(s[0].max_beams for s in scenes) + 4
    bounds = max(s[1] for s in scenes)

    def fit(b):   # the same scene in buffers of the engine's capacity
        out = sb.Buffers(2, cap_p, cap_b)
        out.set_scene(b.particles[:b.particle_count], b.beams[:b.beam_count])
        return out

    eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=cap_p, max_beams=cap_b, collision_mode=mode, path=path,
                    tile_particles=int(rng.choice([0, 64, 256])))
    cur = fit(scenes[0][0])
    last_read = None
    kept_expected = 0
    for step in range(6):
        what = int(rng.integers(0, 5)) if step else 0
        if what == 4 and cur.beam_count > 8:                   # an edit that cuts up to 6 % of the beams (r04: the plan is kept)
            src = last_read if (last_read is not None and rng.integers(0, 2)) else cur
            B, maxP = src.beam_count, src.max_particles
            keep = rng.random(B) >= rng.uniform(0.0, 0.06)
            recs = src.beams[src.mapping[maxP:maxP + B].astype(np.int64)][keep]
            cur = src.copy()
            idx = rng.permutation(src.max_beams)[:len(recs)] if rng.integers(0, 2) else np.arange(len(recs))
            cur.beams[idx] = recs
            cur.mapping[maxP:maxP + len(recs)] = idx
            cur.beam_count = len(recs)
        elif what == 1 and last_read is not None:
            cur = last_read                                   # what the engine returned last time (maybe fewer beams)
        elif what == 2:
            cur = fit(scenes[int(rng.integers(0, 2))][0])     # a scene from scratch
        elif what in (3, 4) or (what == 1 and last_read is None):
            cur = cur.copy()                                  # same topology, everything that may move moved
            P, B = cur.particle_count, cur.beam_count
            cur.particles[:P, :2] += rng.uniform(-0.4, 0.4, (P, 2)).astype("f4")
            cur.particles[:P, 2:4] += rng.uniform(-2, 2, (P, 2)).astype("f4")
            m = cur.mapping[cur.max_particles:cur.max_particles + B].astype(np.int64)
            cur.beams["last_length"][m] *= rng.uniform(0.99, 1.01, B).astype("f4")
            pick = rng.random(B) < 0.1
            cur.beams["target_length"][m[pick]] *= np.float32(1.005)
        ref = oracle.OracleEngine(bounds, 10.0, 64, 2, mode, threads=8)
        consts, ui = scenes[0][2], scenes[0][3]
        for e in (eng, ref):
            e.write_buffers(cur)
            e.write_user_input(ui)
            e.set_physics_constants(consts)
        n = int(rng.integers(5, 60))
        eng.step(n); ref.step(n)
        if rng.integers(0, 2):
            eng.frame(); ref.frame()
        got, exp = eng.load_buffers(cur.copy()), ref.load_buffers(cur.copy())
        if not np.isfinite(exp.particles[:exp.particle_count]).all():
            break
        assert (got.particle_count, got.beam_count) == (exp.particle_count, exp.beam_count), "seed %d upload %d" % (seed, step)
        assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")), "seed %d upload %d (kind %d)" % (seed, step, what)
        assert got.beams.tobytes() == exp.beams.tobytes(), "seed %d upload %d (kind %d) beams" % (seed, step, what)
        assert np.array_equal(got.mapping, exp.mapping)
        last_read = got
    UPLOADS_KEPT[seed] = eng.info("uploads_kept")
    UPLOADS_EDITED[seed] = eng.info("uploads_edited")
    eng.destroy()


UPLOADS_KEPT = {}
UPLOADS_EDITED = {}


def test_some_of_those_uploads_kept_their_plan():
    if not UPLOADS_KEPT:
        pytest.skip("the upload-sequence cases did not run in this session")
    print("upload sequences: %d uploads kept the plan in %d cases, %d of them with beams removed" % (sum(UPLOADS_KEPT.values()), len(UPLOADS_KEPT),
                                                                                             sum(UPLOADS_EDITED.values())))
    assert sum(UPLOADS_KEPT.values()) >= len(UPLOADS_KEPT) // 2
    assert len(UPLOADS_KEPT) < 8 or sum(UPLOADS_EDITED.values()) >= 2
