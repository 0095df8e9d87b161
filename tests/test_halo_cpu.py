"""Multi-rank sharding on the CPU: plan consistency, simulated ranks with the oracle, and a real
world_size-2 torch.distributed (gloo) run.  The GPU counterpart is tests/test_gpu_halo.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def merged_state(plans_bufs, P, B_keys):
    """owned particles / beams of every rank -> global arrays keyed by global id / beam key"""
    parts = np.zeros((P, 6), "f4")
    seen = np.zeros(P, bool)
    beams = {}
    for plan, buf in plans_bufs:
        op = plan.owned_particles
        gid = plan.global_particle_id[op]
        assert not seen[gid].any()
        seen[gid] = True
        parts[gid] = buf.particles[op]
        for k, rec in zip(plan.global_beam_key[plan.owned_beams], buf.beams[plan.owned_beams]):
            beams[int(k)] = rec
    assert seen.all()
    return parts, beams


def test_plan_lists_agree_between_neighbours(sb):
    halo = sb.halo
    world, W, H, depth = 3, 6, 5, 2
    made = [halo.slab_scene(sb, r, world, W, H, jitter=1.0, depth=depth) for r in range(world)]
    for r in range(world):
        buf, plan = made[r]
        assert plan.n_owned == W * H
        for peer in plan.peers:
            obuf, oplan = made[peer.rank]
            back = [p for p in oplan.peers if p.rank == r][0]
            # what I expect as ghosts is exactly what the owner packs, in the same order
            assert np.array_equal(plan.global_particle_id[peer.ghost_p], oplan.global_particle_id[back.send_p])
            assert np.array_equal(plan.global_beam_key[peer.ghost_b], oplan.global_beam_key[back.send_b])
            # and the ghost copies start out bit-identical to the owner's data
            assert np.array_equal(buf.particles[peer.ghost_p], obuf.particles[back.send_p])
            assert peer.ghost_p.size == depth * H and peer.ghost_b.size > 0
    # world == 1 is the plain lattice scene
    one, plan1 = halo.slab_scene(sb, 0, 1, 7, 5, jitter=1.0, depth=2)
    ref = sb.scenes.lattice_buffers(7, 5, jitter=1.0)
    assert np.array_equal(one.particles, ref.particles) and one.beams.tobytes() == ref.beams.tobytes()
    assert not plan1.peers
    with pytest.raises(ValueError):
        halo.slab_scene(sb, 0, 2, 4, 5, depth=5)


@pytest.mark.parametrize("world,depth,steps", [(2, 4, 37), (3, 1, 24), (3, 3, 30)])
def test_simulated_ranks_equal_single_run(sb, oracle, world, depth, steps):
    """N oracle-backed ranks with deep ghost zones reproduce the single-engine run bit for bit,
    including exchanges that fall mid-way and a ghost depth of 1 (exchange every substep)."""
    from halo_oracle import LocalBus, OracleRank, step_all
    halo = sb.halo
    W, H = 6, 7
    kw = dict(d=25.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.5)
    bounds = 1000.0
    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = OracleRank(oracle, gbuf, bounds)
    ref.step(steps)
    want = ref.load(gbuf)
    bus = LocalBus()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = OracleRank(oracle, buf, bounds)
        tr = bus.transport(r, lambda a, b: (np.zeros(max(a, 1), "f4"), np.zeros(max(b, 1), "f4")), lambda t: t)
        exs.append(halo.Exchanger(eng, plan, tr))
        made.append((buf, plan, eng))

    def copy(dst, src):
        dst[:] = src

    step_all(exs, bus, steps, copy)
    parts, beams = merged_state([(plan, eng.load(buf)) for buf, plan, eng in made], W * world * H, None)
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)].tobytes()[8:] == rec.tobytes()[8:], "beam %d" % k  # endpoints are local indices
    assert (want.particles[:, 1] == 10.0).any()  # the lattice reached the floor


def live_beam_keys(plan, buf, owned_only=True):
    """global keys of the beams still in the mapping (engineMapping.ts: the first beam_count beam slots)"""
    live = buf.mapping[buf.max_particles:buf.max_particles + buf.beam_count].astype(np.int64)
    if owned_only:
        own = np.zeros(buf.max_beams, bool)
        own[plan.owned_beams] = True
        live = live[own[live]]
    return set(int(k) for k in plan.global_beam_key[live])


@pytest.mark.parametrize("world,depth,limit,velocity", [(2, 4, 0.05, (0.3, -4.0)), (3, 3, 0.02, (0.3, -4.0)), (3, 5, 0.02, (6.0, -8.0))])
def test_beams_that_break_are_deleted_on_every_rank_in_the_same_pass(sb, oracle, world, depth, limit, velocity):
    """Frames with delete passes across ranks (include/softbody.h, sb_halo_delete_ghosts): the owner removes its broken
    beams, the refresh right after carries the deaths, the neighbours' ghost copies follow -- the merged state equals the
    single-engine run bit for bit, with the same beams gone, while ghost copies near the outer edge of a zone (whose inputs
    are stale) flag themselves spuriously or not at all."""
    from halo_oracle import LocalBus, OracleRank, frame_all
    halo = sb.halo
    W, H, frames = 6, 7, 3
    kw = dict(d=25.0, origin=(100.0, 11.5), jitter=1.0, velocity=velocity, strain_limit=limit)
    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = OracleRank(oracle, gbuf, 1000.0)
    for _ in range(frames):
        ref.ref.frame()
    want = ref.load(gbuf)
    assert want.beam_count < gbuf.beam_count - 10, "the scene is meant to break beams"
    bus = LocalBus()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = OracleRank(oracle, buf, 1000.0)
        tr = bus.transport(r, lambda a, b: (np.zeros(max(a, 1), "f4"), np.zeros(max(b, 1), "f4")), lambda t: t)
        exs.append(halo.Exchanger(eng, plan, tr))
        made.append((buf, plan, eng))

    def copy(dst, src):
        dst[:] = src

    for _ in range(frames):
        frame_all(exs, bus, copy)
    for ex in exs:
        ex.verify()
    loaded = [(plan, eng.load(buf)) for buf, plan, eng in made]
    parts, beams = merged_state(loaded, W * world * H, None)
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    live = set()
    for plan, buf in loaded:
        live |= live_beam_keys(plan, buf)
    assert live == live_beam_keys(gplan, want)
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)].tobytes()[8:] == rec.tobytes()[8:], "beam %d" % k   # (the OWNER's record, removed beams included)
    # and every rank agrees with the owners about its ghost copies too: none outlives its owner's beam, none died on its own
    for plan, buf in loaded:
        here = set(int(k) for k in plan.global_beam_key)      # every beam this rank holds, owned or ghost
        assert live_beam_keys(plan, buf, owned_only=False) == here & live


@pytest.mark.parametrize("world,depth", [(2, 3), (4, 2)])
def test_peer_exchanger_routing_equals_single_engine(sb, oracle, world, depth):
    """PeerExchanger's wiring (whose mailbox, which flag slot, which destination offset) on oracle-backed
    doubles of sb_peer_*: merged result is bit-identical to the unsharded run."""
    from halo_oracle import OracleRank, step_all_peer
    halo = sb.halo
    W, H, steps = 6, 7, 40
    kw = dict(d=25.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.5)
    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = OracleRank(oracle, gbuf, 1000.0)
    ref.step(steps)
    want = ref.load(gbuf)
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = OracleRank(oracle, buf, 1000.0)
        exs.append(halo.PeerExchanger(eng, plan))
        made.append((buf, plan, eng))
    cards = [ex.card for ex in exs]
    for ex in exs:
        ex.connect(cards)
    assert [p["slot"] for p in exs[1].engine.peers] == ([0] if world == 2 else [0, 0])
    step_all_peer(exs, steps)
    parts, beams = merged_state([(plan, eng.load(buf)) for buf, plan, eng in made], W * world * H, None)
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)].tobytes()[8:] == rec.tobytes()[8:], "beam %d" % k


def test_mixed_stiffness_slabs_equal_single_engine(sb, oracle):
    """BASELINE config 5 across ranks: springs drawn per GLOBAL beam key, so ghost copies carry their owners'
    parameters and the sharded run stays bit-identical to the unsharded one."""
    from halo_oracle import LocalBus, OracleRank, step_all
    halo = sb.halo
    W, H, world, depth, steps = 6, 7, 3, 2, 30
    kw = dict(d=25.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0))
    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    halo.mix_stiffness(gbuf, gplan, subticks=64)
    assert len(set(gbuf.beams["spring"][:gbuf.beam_count].tolist())) == 4
    ref = OracleRank(oracle, gbuf, 1000.0)
    ref.step(steps)
    want = ref.load(gbuf)
    bus = LocalBus()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        halo.mix_stiffness(buf, plan, subticks=64)
        eng = OracleRank(oracle, buf, 1000.0)
        tr = bus.transport(r, lambda a, b: (np.zeros(max(a, 1), "f4"), np.zeros(max(b, 1), "f4")), lambda t: t)
        exs.append(halo.Exchanger(eng, plan, tr))
        made.append((buf, plan, eng))

    def copy(dst, src):
        dst[:] = src

    step_all(exs, bus, steps, copy)
    parts, beams = merged_state([(plan, eng.load(buf)) for buf, plan, eng in made], W * world * H, None)
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)].tobytes()[8:] == rec.tobytes()[8:], "beam %d" % k


def test_without_refresh_ghost_zone_goes_stale(sb, oracle):
    """Negative control: skipping the exchange must change the owned result (the test above is
    sensitive to the halo logic)."""
    from halo_oracle import OracleRank
    halo = sb.halo
    kw = dict(d=25.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0))
    gbuf, _ = halo.slab_scene(sb, 0, 1, 12, 7, depth=2, **kw)
    ref = OracleRank(oracle, gbuf, 1000.0)
    ref.step(20)
    want = ref.load(gbuf)
    buf, plan = halo.slab_scene(sb, 0, 2, 6, 7, depth=2, **kw)
    eng = OracleRank(oracle, buf, 1000.0)
    eng.step(20)
    got = eng.load(buf)
    gid = plan.global_particle_id[plan.owned_particles]
    assert not np.array_equal(got.particles[plan.owned_particles], want.particles[gid])


@pytest.mark.parametrize("frames", [0, 3])
def test_gloo_world_size_2(sb, oracle, frames):
    """Real torch.distributed ranks (gloo, 127.0.0.1), the same Exchanger/TorchTransport the GPU bench
    uses, oracle-backed engines on CPU tensors.  frames > 0: beams break and every rank runs Exchanger.frame()."""
    port = 29500 + (os.getpid() + 7 * frames) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "halo_gloo_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, HALO_FRAMES=str(frames)))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "HALO_GLOO_OK" in p.stdout
