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
