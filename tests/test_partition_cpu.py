"""The generic x-slab partitioner of the C library (csrc/sb_partition.cpp, sb_partition_* in include/softbody.h) on the
CPU: a partitioned scene stepped rank by rank with the oracle and the ghost refresh of halo.Exchanger must equal the
unpartitioned oracle run bit for bit -- for the reference's own default scene (main.ts:188-246), with permuted
mappings, and with collisions across the slab faces."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def run_partitioned(sb, oracle, buf, world, depth, steps, *, mode=0, reach=0.0, bounds=1000.0):
    from halo_oracle import LocalBus, OracleRank, step_all
    halo = sb.halo
    ref = OracleRank(oracle, buf, bounds, mode=mode)
    ref.step(steps)
    want = ref.load(buf)
    made = halo.partition_scene(buf, world, depth, contact_reach=reach)
    bus, exs, engines = LocalBus(), [], []
    for r, (lbuf, plan) in enumerate(made):
        eng = OracleRank(oracle, lbuf, bounds, mode=mode)
        tr = bus.transport(r, lambda a, b: (np.zeros(max(a, 1), "f4"), np.zeros(max(b, 1), "f4")), lambda t: t)
        exs.append(halo.Exchanger(eng, plan, tr))
        engines.append(eng)
    step_all(exs, bus, steps, lambda dst, src: dst.__setitem__(slice(None), src))
    P, B = buf.particle_count, buf.beam_count
    got_p = np.zeros_like(want.particles)
    seen_p = np.zeros(buf.max_particles, bool)
    got_b = {}
    for (lbuf, plan), eng in zip(made, engines):
        out = eng.load(lbuf)
        gid = plan.global_particle_id[plan.owned_particles]
        assert not seen_p[gid].any()
        seen_p[gid] = True
        got_p[gid] = out.particles[plan.owned_particles]
        for k, rec in zip(plan.global_beam_key[plan.owned_beams], out.beams[plan.owned_beams]):
            assert int(k) not in got_b
            got_b[int(k)] = rec
    active_p = buf.mapping[:P].astype(np.int64)
    active_b = buf.mapping[buf.max_particles:buf.max_particles + B].astype(np.int64)
    assert seen_p[active_p].all() and seen_p.sum() == P and sorted(got_b) == sorted(active_b.tolist())
    assert np.array_equal(got_p[active_p].view("u4"), want.particles[active_p].view("u4"))
    for k in active_b:
        assert got_b[int(k)].tobytes()[8 if buf.layout == 2 else 4:] == want.beams[k].tobytes()[8 if buf.layout == 2 else 4:], "beam %d" % k
    return made, want


@pytest.mark.parametrize("layout,world,depth", [(1, 2, 3), (2, 3, 2), (2, 2, 1)])
def test_partitioned_default_scene_equals_the_unpartitioned_run(sb, oracle, layout, world, depth):
    """The reference's default scene (119 particles / 299 beams), beams only: 2 and 3 slabs, ghost zones 1-3 hops deep,
    80 substeps (so exchanges fall mid-call), bit-exact against the single oracle run."""
    buf = sb.scenes.default_buffers(layout, 256, 512)
    made, want = run_partitioned(sb, oracle, buf, world, depth, 80)
    owned = sum(plan.n_owned for _, plan in made)
    assert owned == 119 and all(plan.n_owned in (39, 40, 59, 60) for _, plan in made)
    assert not np.array_equal(want.particles, buf.particles)
    for lbuf, plan in made:
        for peer in plan.peers:
            obuf, oplan = made[peer.rank]
            back = [p for p in oplan.peers if p.rank == plan.rank][0]
            assert np.array_equal(plan.global_particle_id[peer.ghost_p], oplan.global_particle_id[back.send_p])
            assert np.array_equal(plan.global_beam_key[peer.ghost_b], oplan.global_beam_key[back.send_b])
            assert np.array_equal(lbuf.particles[peer.ghost_p], obuf.particles[back.send_p])


def test_partition_keeps_slot_and_index_order_of_a_permuted_scene(sb, oracle):
    """Slots need not equal data indices (engineMapping.ts:336-339): the rank scenes keep both orders (monotone maps)."""
    buf = sb.scenes.default_buffers(2, 256, 512)
    rng = np.random.default_rng(3)
    P, B = buf.particle_count, buf.beam_count
    pp, bp = rng.permutation(P), rng.permutation(B)
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
    made, _ = run_partitioned(sb, oracle, buf, 2, 2, 40)
    for lbuf, plan in made:
        n = lbuf.particle_count
        gslot = {int(d): s for s, d in enumerate(buf.mapping[:P])}
        slots = [gslot[int(plan.global_particle_id[int(d)])] for d in lbuf.mapping[:n]]
        assert slots == sorted(slots)                                    # local slot order = global slot order
        assert (np.diff(plan.global_particle_id) > 0).all()              # local data order = global data order


def test_partitioned_pile_with_contacts_across_the_faces(sb, oracle):
    """Blobs resting on each other, cut into three slabs through the middle of blobs and between blobs in contact;
    ghosts = 2 beam hops around own particles and around a contact band of 2 x (longest beam + 2r): the oracle's
    collision loop on every rank's own scene reproduces the unpartitioned run bit for bit."""
    buf, bounds = sb.scenes.blob_pile_buffers(9, 3, gap=19.6)   # neighbouring blobs start in contact
    reach = 2 * (30.0 * 2 ** 0.5 + 20.0 + 2.0)
    made, want = run_partitioned(sb, oracle, buf, 3, 2, 48, mode=oracle.COLLIDE_GRID, reach=reach, bounds=bounds)
    off = oracle.OracleEngine(bounds, 10.0, 64, 2, oracle.COLLIDE_OFF)
    off.write_buffers(buf)
    off.step(48)
    assert (off.load_buffers(buf.copy()).particles != want.particles).any(axis=1).mean() > 0.05   # contacts really acted
    assert all(0 < plan.n_owned < buf.particle_count for _, plan in made)
    assert any(lbuf.particle_count < buf.particle_count for lbuf, _ in made)                      # and it is a real split


def two_clouds(sb):
    """two clouds of free particles flying at each other: 200 apart at the start, through each other by frame 9"""
    rng = np.random.default_rng(5)
    pts = []
    for cx, vx in ((250.0, 20.0), (750.0, -20.0)):
        for i in range(10):
            for j in range(12):
                pts.append((cx - 150 + 30.0 * i + rng.uniform(-3, 3), 40.0 + 30.0 * j + rng.uniform(-3, 3), vx + rng.uniform(-2, 2),
                            rng.uniform(-2, 2), 0, 0))
    buf = sb.Buffers(2, len(pts) + 8, 16)
    buf.set_scene(np.array(pts, "f4"), np.zeros(0, sb.layout.BEAM_DTYPE[2]))
    return buf


@pytest.mark.parametrize("again", [True, False])
def test_repartition_lets_ownership_follow_the_particles(sb, oracle, again):
    """Ghost zones are those of the partition: two clouds that start 200 apart share no ghosts (contact band 80), so on
    their own ranks they fly through each other.  halo.repartition() between frames -- gather the owned state, partition
    again -- keeps the run bit-identical to the single engine through the whole collision; without it the run diverges
    (the negative control)."""
    from halo_oracle import LocalBus, OracleRank, frame_all
    halo = sb.halo
    world, depth, reach, frames, mode = 2, 2, 80.0, 9, oracle.COLLIDE_GRID
    gbuf = two_clouds(sb)
    ref = OracleRank(oracle, gbuf, 1000.0, mode=mode)
    for _ in range(frames):
        ref.ref.frame()
    want = ref.load(gbuf)

    def build(made):
        bus, exs, engs = LocalBus(), [], []
        for r, (lbuf, plan) in enumerate(made):
            eng = OracleRank(oracle, lbuf, 1000.0, mode=mode)
            tr = bus.transport(r, lambda a, b: (np.zeros(max(a, 1), "f4"), np.zeros(max(b, 1), "f4")), lambda t: t)
            exs.append(halo.Exchanger(eng, plan, tr))
            engs.append(eng)
        return bus, exs, engs

    made = halo.partition_scene(gbuf, world, depth, contact_reach=reach)
    assert all(not p.ghost_p.size for _, plan in made for p in plan.peers)        # nothing in common at the start
    bus, exs, engs = build(made)
    for _ in range(frames):
        frame_all(exs, bus, lambda dst, src: dst.__setitem__(slice(None), src))
        if again:
            states = [halo.owned_state(plan, eng.load(lbuf)) for (lbuf, plan), eng in zip(made, engs)]
            made = halo.repartition(gbuf, states, world, depth, reach)
            bus, exs, engs = build(made)
    got = np.zeros_like(want.particles)
    for (lbuf, plan), eng in zip(made, engs):
        out = eng.load(lbuf)
        got[plan.global_particle_id[plan.owned_particles]] = out.particles[plan.owned_particles]
    P = gbuf.particle_count
    same = np.array_equal(got[:P].view("u4"), want.particles[:P].view("u4"))
    if again:
        assert same and sum(p.ghost_p.size for _, plan in made for p in plan.peers) > 40   # the clouds are in each other's zones now
    else:
        assert not same and np.abs(got[:P, :2] - want.particles[:P, :2]).max() > 10.0


def test_repartition_carries_beam_state_and_removed_beams(sb, oracle):
    """A lattice whose beams yield and break, partitioned in three, re-partitioned after every frame: the new scenes carry
    target / last lengths and leave the removed beams out of the mapping (stable compaction, as compute_delete does), and the
    merged run stays bit-identical to the single engine."""
    from halo_oracle import LocalBus, OracleRank, frame_all
    halo = sb.halo
    world, depth, frames = 3, 3, 4
    gbuf = sb.scenes.lattice_buffers(18, 7, d=25.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), layout=2, strain_limit=0.02)
    ref = OracleRank(oracle, gbuf, 1000.0)
    for _ in range(frames):
        ref.ref.frame()
    want = ref.load(gbuf)
    assert want.beam_count < gbuf.beam_count - 50

    def build(made):
        bus, exs, engs = LocalBus(), [], []
        for r, (lbuf, plan) in enumerate(made):
            eng = OracleRank(oracle, lbuf, 1000.0)
            tr = bus.transport(r, lambda a, b: (np.zeros(max(a, 1), "f4"), np.zeros(max(b, 1), "f4")), lambda t: t)
            exs.append(halo.Exchanger(eng, plan, tr))
            engs.append(eng)
        return bus, exs, engs

    cur = gbuf.copy()
    made = halo.partition_scene(cur, world, depth)
    bus, exs, engs = build(made)
    for _ in range(frames):
        frame_all(exs, bus, lambda dst, src: dst.__setitem__(slice(None), src))
        states = [halo.owned_state(plan, eng.load(lbuf)) for (lbuf, plan), eng in zip(made, engs)]
        made = halo.repartition(cur, states, world, depth)
        bus, exs, engs = build(made)
    # `cur` is the gathered scene now: same particles, same surviving beams in the same slot order, same beam records
    P = gbuf.particle_count
    assert np.array_equal(cur.particles[:P].view("u4"), want.particles[:P].view("u4"))
    assert cur.beam_count == want.beam_count
    P0 = gbuf.max_particles
    assert np.array_equal(cur.mapping[P0:P0 + cur.beam_count], want.mapping[P0:P0 + want.beam_count])
    live = want.mapping[P0:P0 + want.beam_count].astype(np.int64)
    assert cur.beams[live].tobytes() == want.beams[live].tobytes()


def test_partition_argument_checks(sb):
    buf = sb.scenes.default_buffers(2, 256, 512)
    with pytest.raises(sb.engine.EngineError, match="depth 0"):
        sb.halo.partition_scene(buf, 2, 0)
    bad = buf.copy()
    bad.beams["a"][0] = 255       # an endpoint no slot maps to
    with pytest.raises(sb.engine.EngineError, match="references particle"):
        sb.halo.partition_scene(bad, 2, 2)
    one = sb.halo.partition_scene(buf, 1, 0)
    assert len(one) == 1 and not one[0][1].peers and one[0][0].particle_count == 119
    assert np.array_equal(one[0][0].particles[:119], buf.particles[:119])
