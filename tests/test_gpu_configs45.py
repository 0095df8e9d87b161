"""BASELINE configs 4 and 5 at ONE GPU's share of the 8-GPU runs (a gpurun box has a single MI355X; the 8-GPU run
itself is the driver's): config 4 = two neighbouring ranks of the 16 M-particle lattice (500 x 4000 owned columns
each plus ghost zones 24 columns deep) trading ghost zones through sb_peer_* on one card; config 5 = one rank's
1000 x 8000 slab with springs drawn from {1, 3, 50, 500} at dt = 1/128.  Bit-exact against the oracle for a few
substeps, schedule independence (tiled == atomic), ghosts equal to their owners after three refreshes, the two
ranks together equal to one engine holding both slabs, everything finite and inside the box."""
import numpy as np
import pytest

from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


def engine(sb, buf, bounds, subticks=64, path=2):
    e = sb.Engine(bounds_size=bounds, subticks=subticks, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                  collision_mode=0, path=path)
    e.write_buffers(buf)
    return e


def test_config4_two_ranks_at_share_size(sb, oracle):
    halo = sb.halo
    W, H, depth, world = 500, 4000, 24, 2
    bounds = float(4000 * 30.0 + 2000.0)
    kw = dict(d=30.0, origin=(1000.0, 1000.0), jitter=1.0)
    made = []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        made.append((buf, plan, engine(sb, buf, bounds)))
    assert made[0][0].particle_count == (W + depth) * H == 2_096_000

    # (a) rank 0's scene on its own, 8 substeps (no refresh falls inside): GPU == oracle, ghosts included
    buf0 = made[0][0]
    ref = oracle.OracleEngine(bounds, 10.0, 64, 2, oracle.COLLIDE_OFF, threads=16)
    ref.write_buffers(buf0)
    ref.step(8)
    solo = engine(sb, buf0, bounds)
    solo.step(8)
    assert_same(solo.load_buffers(buf0.copy()), ref.load_buffers(buf0.copy()), "config 4 share vs oracle")
    # (b) schedule independence at this size: 16 more substeps, blocked/tiled == atomic
    solo.step(16)
    atomic = engine(sb, buf0, bounds, path=1)
    atomic.step(24)
    a, b = solo.load_buffers(buf0.copy()), atomic.load_buffers(buf0.copy())
    solo.destroy()
    atomic.destroy()
    assert np.array_equal(a.particles.view("u4"), b.particles.view("u4")) and a.beams.tobytes() == b.beams.tobytes()

    # (c) the two ranks wired by sb_peer_* (mailboxes by pointer), three refresh periods
    exs = [halo.PeerExchanger(e, plan, timeout_ms=5000) for _, plan, e in made]
    cards = [ex.card for ex in exs]
    for ex in exs:
        ex.connect(cards)
    for _ in range(3):
        for ex in exs:
            ex.step(depth)
    outs = []
    for buf, plan, e in made:
        e.sync()
        outs.append(e.load_buffers(buf.copy()))
        e.destroy()
    # right after a refresh every ghost record equals its owner's record
    bad = 0
    for r, (buf, plan, _) in enumerate(made):
        for p in plan.peers:
            theirs_plan, theirs = made[p.rank][1], outs[p.rank]
            gid_of = {int(g): i for i, g in enumerate(theirs_plan.global_particle_id[theirs_plan.owned_particles])}
            idx = theirs_plan.owned_particles[[gid_of[int(g)] for g in plan.global_particle_id[p.ghost_p]]]
            bad += int((outs[r].particles[p.ghost_p].view("u4") != theirs.particles[idx].view("u4")).any(axis=1).sum())
    assert bad == 0, "%d ghost particles differ from their owners after three refreshes" % bad
    # (d) ... and the two ranks together are the single engine that holds both slabs
    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    whole = engine(sb, gbuf, bounds)
    whole.step(3 * depth)
    want = whole.load_buffers(gbuf.copy())
    whole.destroy()
    parts = np.zeros_like(want.particles)
    for (buf, plan, _), out in zip(made, outs):
        gid, prt, _, _ = halo.gather_owned(plan, out)
        parts[gid] = prt
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    assert np.isfinite(parts).all() and (parts[:, :2] >= 10.0).all() and (parts[:, :2] <= bounds - 10.0).all()
    assert not np.array_equal(parts, gbuf.particles)


def test_config5_one_rank_at_share_size(sb, oracle):
    """1000 x 8000 slab (8 M particles / 24 M beams), springs {1, 3, 50, 500} with capped damping, subticks 128."""
    W, H = 1000, 8000
    bounds = float(8000 * 30.0 + 2000.0)
    buf = sb.scenes.lattice_buffers(W, H, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2)
    sb.scenes.mix_stiffness(buf, subticks=128)
    assert len(np.unique(buf.beams["spring"][:buf.beam_count])) == 4
    ref = oracle.OracleEngine(bounds, 10.0, 128, 2, oracle.COLLIDE_OFF, threads=16)
    ref.write_buffers(buf)
    ref.step(4)
    exp = ref.load_buffers(buf.copy())
    eng = engine(sb, buf, bounds, subticks=128)
    assert eng.info("materials") >= 8 and eng.info("substeps_per_launch") > 1
    eng.step(4)
    got = eng.load_buffers(buf.copy())
    assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")) and got.beams.tobytes() == exp.beams.tobytes()
    del exp, ref
    eng.step(124)                        # one whole frame of 128 substeps in all
    tiled = eng.load_buffers(buf.copy())
    eng.destroy()
    atomic = engine(sb, buf, bounds, subticks=128, path=1)
    atomic.step(128)
    b = atomic.load_buffers(buf.copy())
    atomic.destroy()
    assert np.array_equal(tiled.particles.view("u4"), b.particles.view("u4")) and tiled.beams.tobytes() == b.beams.tobytes()
    p = tiled.particles[:buf.particle_count]
    assert np.isfinite(p).all() and (p[:, :2] >= 10.0).all() and (p[:, :2] <= bounds - 10.0).all()
    assert np.abs(p[:, 2:4]).max() < 50.0 and not np.array_equal(p, buf.particles[:buf.particle_count])
