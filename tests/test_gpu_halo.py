"""GPU halo path: several engines on ONE GPU play the ranks (a gpurun box has a single MI355X);
pack/unpack are the real HIP kernels, the wire is a device-to-device copy.  The merged result must
equal the single-engine run bit for bit.  (Real multi-process RCCL runs are the driver's.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,depth,path,yield_strain", [(2, 4, 2, 0.2), (3, 8, 2, 0.2), (2, 2, 1, 0.2), (2, 5, 2, 0.004), (3, 7, 2, 0.004)])
def test_simulated_gpu_ranks_equal_single_engine(sb, world, depth, path, yield_strain):
    """(yield_strain 0.004: beams yield all over the lattice when it lands, ghost copies receive their owners' new targets with every
    refresh -- the tiles they sit in stop being "never yielded" for the blocked kernel, sb_kernels.hip k_halo_unpack)"""
    import torch
    from halo_oracle import LocalBus, step_all
    halo = sb.halo
    W, H, steps = 40, 48, 100
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.5, yield_strain=yield_strain)
    bounds = 8000.0

    def engine_for(buf):
        e = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                      collision_mode=0, path=path, tile_particles=256)
        e.write_buffers(buf)
        return e

    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = engine_for(gbuf)
    ref.step(steps)
    want = ref.load_buffers(gbuf.copy())
    ref.destroy()
    if yield_strain < 0.1:
        B = want.beam_count
        assert (want.beams["target_length"][:B] != want.beams["length"][:B]).mean() > 0.02, "the scene is meant to yield"

    dev = torch.device("cuda", 0)
    bus = LocalBus()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = engine_for(buf)
        tr = bus.transport(r, lambda a, b: (torch.zeros(max(a, 1), device=dev), torch.zeros(max(b, 1), device=dev)),
                           lambda t: t.data_ptr())
        exs.append(halo.Exchanger(eng, plan, tr))
        made.append((buf, plan, eng))

    def sync():
        for _, _, e in made:
            e.sync()
        torch.cuda.synchronize()

    step_all(exs, bus, steps, lambda dst, src: dst.copy_(src), sync)
    parts = np.zeros_like(want.particles)
    beams = {}
    for buf, plan, eng in made:
        out = eng.load_buffers(buf.copy())
        gid, prt, bkey, brec = halo.gather_owned(plan, out)
        parts[gid] = prt
        for k, rec in zip(bkey, brec):
            beams[int(k)] = rec.tobytes()[8:]
        eng.destroy()
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)] == rec.tobytes()[8:]
    assert (want.particles[:, 1] == 10.0).any()


@pytest.mark.parametrize("world,depth,yield_strain", [(2, 12, 0.2), (3, 16, 0.004)])
def test_simulated_gpu_ranks_with_the_hash_on_run_blocked(sb, world, depth, yield_strain):
    """The engine's DEFAULT collision mode on every rank (spatial hash; the reference always collides, compute.wgsl:142-170) on a
    falling lattice in which nothing is within reach: between two ghost refreshes each rank runs blocked launches beside its
    tiled layout -- until r04 an engine with ghost zones stayed on single substeps.  Every refresh orders a fresh
    hash (ghosts jump), so a period is one single substep, a look, then tracked launches.  Owned particles and beams of the ranks
    together = the single engine's, bit for bit; and the ranks did run blocked."""
    import torch
    from halo_oracle import LocalBus, step_all
    halo = sb.halo
    W, H = 36, 40
    steps = 4 * depth + 5
    kw = dict(d=30.0, origin=(100.0, 700.0), jitter=1.0, velocity=(0.3, -1.0), strain_limit=0.5, yield_strain=yield_strain)
    bounds = 8000.0

    def engine_for(buf):
        e = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2,
                      tile_particles=256)
        e.write_buffers(buf)
        return e

    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = engine_for(gbuf)
    ref.step(steps)
    want = ref.load_buffers(gbuf.copy())
    assert ref.info("hybrid_substeps") > 0
    ref.destroy()
    dev = torch.device("cuda", 0)
    bus = LocalBus()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = engine_for(buf)
        tr = bus.transport(r, lambda a, b: (torch.zeros(max(a, 1), device=dev), torch.zeros(max(b, 1), device=dev)),
                           lambda t: t.data_ptr())
        exs.append(halo.Exchanger(eng, plan, tr))
        made.append((buf, plan, eng))

    def sync():
        for _, _, e in made:
            e.sync()
        torch.cuda.synchronize()

    step_all(exs, bus, steps, lambda dst, src: dst.copy_(src), sync)
    parts = np.zeros_like(want.particles)
    beams = {}
    blocked = []
    for buf, plan, eng in made:
        out = eng.load_buffers(buf.copy())
        gid, prt, bkey, brec = halo.gather_owned(plan, out)
        parts[gid] = prt
        for k, rec in zip(bkey, brec):
            beams[int(k)] = rec.tobytes()[8:]
        blocked.append((eng.info("hybrid_substeps"), eng.info("substeps_done"), eng.info("hybrid_failed")))
        eng.destroy()
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)] == rec.tobytes()[8:]
    assert all(b[0] >= b[1] // 2 for b in blocked), blocked   # more than half of every rank's substeps ran in blocked launches


@pytest.mark.parametrize("world,depth,path,block", [(2, 4, 2, 0), (3, 6, 2, 0), (2, 5, 2, 1), (2, 3, 1, 0)])
def test_beams_that_break_across_simulated_gpu_ranks(sb, world, depth, path, block):
    """Frames with delete passes across ranks on the real kernels (blocked plan, single-substep tiling, atomic path): the
    owner's delete pass, the refresh that carries the deaths (a NaN payload in last_length), sb_halo_delete_ghosts on the
    neighbours.  Same particles, same owner records and the same beams gone as the single engine stepping whole frames."""
    import torch
    from halo_oracle import LocalBus, frame_all
    halo = sb.halo
    W, H, frames = 40, 48, 3
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.02)
    bounds = 8000.0

    def engine_for(buf):
        e = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                      collision_mode=0, path=path, tile_particles=256, block_substeps=block)
        e.write_buffers(buf)
        return e

    def live_keys(plan, out, owned_only=True):
        live = out.mapping[out.max_particles:out.max_particles + out.beam_count].astype(np.int64)
        if owned_only:
            own = np.zeros(out.max_beams, bool)
            own[plan.owned_beams] = True
            live = live[own[live]]
        return set(int(k) for k in plan.global_beam_key[live])

    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = engine_for(gbuf)
    for _ in range(frames):
        ref.frame()
    want = ref.load_buffers(gbuf.copy())
    ref.destroy()
    assert want.beam_count < gbuf.beam_count - 100, "the scene is meant to break beams"

    dev = torch.device("cuda", 0)
    bus = LocalBus()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = engine_for(buf)
        tr = bus.transport(r, lambda a, b: (torch.zeros(max(a, 1), device=dev), torch.zeros(max(b, 1), device=dev)),
                           lambda t: t.data_ptr())
        exs.append(halo.Exchanger(eng, plan, tr))
        made.append((buf, plan, eng))

    def sync():
        for _, _, e in made:
            e.sync()
        torch.cuda.synchronize()

    for _ in range(frames):
        frame_all(exs, bus, lambda dst, src: dst.copy_(src), sync)
    for ex in exs:
        ex.verify()
    parts = np.zeros_like(want.particles)
    beams, live = {}, set()
    for buf, plan, eng in made:
        out = eng.load_buffers(buf.copy())
        gid, prt, bkey, brec = halo.gather_owned(plan, out)
        parts[gid] = prt
        for k, rec in zip(bkey, brec):
            beams[int(k)] = rec.tobytes()[8:]
        live |= live_keys(plan, out)
        # ghost copies: none outlives its owner's beam, none died on its own (stale inputs at the edge of the zone)
        assert live_keys(plan, out, owned_only=False) == set(int(k) for k in plan.global_beam_key) & live_keys(gplan, want)
        eng.destroy()
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    assert live == live_keys(gplan, want)
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)] == rec.tobytes()[8:]


@pytest.mark.parametrize("world,depth", [(2, 4), (3, 8)])
def test_peer_exchange_in_process_equals_single_engine(sb, world, depth):
    """sb_peer_* between engines of one process (each on its own stream, mailboxes passed by pointer):
    pack-into-neighbour, flag handshake and unpack, all device-side; merged result bit-identical."""
    halo = sb.halo
    W, H, steps = 40, 48, 100
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.5)

    def engine_for(buf):
        e = sb.Engine(bounds_size=8000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                      collision_mode=0, path=2, tile_particles=256)
        e.write_buffers(buf)
        return e

    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref = engine_for(gbuf)
    ref.step(steps)
    want = ref.load_buffers(gbuf.copy())
    ref.destroy()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = engine_for(buf)
        exs.append(halo.PeerExchanger(eng, plan, timeout_ms=3000))
        made.append((buf, plan, eng))
    cards = [ex.card for ex in exs]
    for ex in exs:
        ex.connect(cards)
    # one exchange period at a time per rank: every wait kernel's counterpart is already enqueued on
    # another stream before the host moves on, and nothing here blocks the host
    done = 0
    while done < steps:
        m = min(depth, steps - done)
        for ex in exs:
            ex.step(m)
        done += m
    for _, _, eng in made:
        eng.sync()          # raises if a wait gave up
    parts = np.zeros_like(want.particles)
    beams = {}
    for buf, plan, eng in made:
        out = eng.load_buffers(buf.copy())
        gid, prt, bkey, brec = halo.gather_owned(plan, out)
        parts[gid] = prt
        for k, rec in zip(bkey, brec):
            beams[int(k)] = rec.tobytes()[8:]
        eng.destroy()
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)] == rec.tobytes()[8:]


@pytest.mark.parametrize("world,depth", [(2, 4), (3, 6)])
def test_collisions_across_slab_faces(sb, world, depth):
    """Spatial-hash collisions with sharding: ghost particles are ordinary particles of the rank's scene, so a
    contact between an owned particle and a ghost is computed on both ranks from the same inputs, in the same
    ascending-slot order (local data indices are a monotone map of the global ones).  Lattice at spacing 30
    with jitter 5.5 (some neighbours start closer than 2r = 20, also across the faces), thrown at the floor;
    the scene stays gentle (a lattice PACKED below 2r bursts, and a burst outruns any ghost zone)."""
    halo = sb.halo
    W, H, steps = 24, 32, 72
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=5.5, velocity=(0.2, -6.0), strain_limit=1e9)

    def engine_for(buf, mode):
        e = sb.Engine(bounds_size=4000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                      collision_mode=mode, path=2, tile_particles=256)
        e.write_buffers(buf)
        return e

    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    want = {}
    for mode in (0, 2):
        ref = engine_for(gbuf, mode)
        ref.step(steps)
        want[mode] = ref.load_buffers(gbuf.copy())
        ref.destroy()
    assert np.isfinite(want[2].particles).all()
    assert (want[0].particles != want[2].particles).any(axis=1).mean() > 0.5   # the contacts really acted
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng = engine_for(buf, 2)
        exs.append(halo.PeerExchanger(eng, plan, timeout_ms=3000))
        made.append((buf, plan, eng))
    for ex in exs:
        ex.connect([x.card for x in exs])
    done = 0
    while done < steps:
        m = min(depth, steps - done)
        for ex in exs:
            ex.step(m)
        done += m
    parts = np.zeros_like(want[2].particles)
    beams = {}
    for buf, plan, eng in made:
        eng.sync()
        out = eng.load_buffers(buf.copy())
        gid, prt, bkey, brec = halo.gather_owned(plan, out)
        parts[gid] = prt
        for k, rec in zip(bkey, brec):
            beams[int(k)] = rec.tobytes()[8:]
        eng.destroy()
    assert np.array_equal(parts.view("u4"), want[2].particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want[2].beams):
        assert beams[int(k)] == rec.tobytes()[8:]


def test_peer_wait_gives_up_instead_of_hanging(sb):
    """A neighbour that never posts: the bounded wait ends, sb_sync reports it, the engine stays usable."""
    halo = sb.halo
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=1.0)
    exs = []
    for r in range(2):
        buf, plan = halo.slab_scene(sb, r, 2, 8, 8, depth=2, **kw)
        e = sb.Engine(bounds_size=8000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams,
                      collision_mode=0)
        e.write_buffers(buf)
        exs.append(halo.PeerExchanger(e, plan, timeout_ms=200))
    for ex in exs:
        ex.connect([x.card for x in exs])
    exs[0].engine.peer_exchange()       # rank 1 never calls it
    with pytest.raises(sb.EngineError, match="did not signal"):
        exs[0].engine.sync()
    exs[0].engine.step(2)
    exs[0].engine.sync()
    for ex in exs:
        ex.engine.destroy()


@pytest.mark.parametrize("frames", [0, 3])
def test_two_processes_peer_exchange(sb, frames):
    """IPC-mapped mailboxes between two OS processes on this GPU (tests/halo_peer_worker.py).  frames > 0: beams break and
    both processes run PeerExchanger.frame() -- delete pass, a refresh that carries the deaths, removal of the ghost copies."""
    import subprocess
    port = 29500 + (os.getpid() + 7 * frames) % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "halo_peer_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(os.environ, HALO_FRAMES=str(frames)))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "HALO_PEER_OK" in p.stdout


def test_halo_and_peer_argument_checks(sb):
    """The exchange entry points refuse what they cannot honour, with a status and a message (no crash, no
    silent corruption): odd offsets, use before configuration, segments that do not fit, too many neighbours."""
    halo = sb.halo
    buf, plan = halo.slab_scene(sb, 0, 2, 8, 8, depth=2, d=30.0, origin=(100.0, 11.5), jitter=1.0)
    e = sb.Engine(bounds_size=8000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=0)
    with pytest.raises(sb.EngineError, match="before sb_write_buffers"):
        e.halo_configure([0], [1])
    e.write_buffers(buf)
    with pytest.raises(sb.EngineError, match="before sb_peer_connect"):
        e.peer_exchange()
    gp, sp, gb, sb_ = plan.lists()
    with pytest.raises(sb.EngineError, match="not active"):
        e.halo_configure([buf.particle_count + 5], sp)
    e.halo_configure(gp, sp, gb, sb_)
    segs, n_send, n_recv, offsets = plan.segments()
    bad = [o.copy() for o in offsets]
    bad[0][0] += 1
    with pytest.raises(sb.EngineError, match="even"):
        e.halo_set_layout(*bad)
    e.halo_set_layout(*offsets)
    box, handle, nbytes = e.peer_mailbox()
    assert nbytes >= 256 + 8 * n_recv and len(handle) == 64
    with pytest.raises(sb.EngineError, match="after sb_peer_mailbox"):
        e.halo_set_layout(*offsets)
    with pytest.raises(sb.EngineError, match="exceeds the packed send layout"):
        e.peer_connect([box], [n_recv], [0], [n_send + 2], [0], [0])
    with pytest.raises(sb.EngineError, match="does not fit"):
        e.peer_connect([box], [n_send - 2], [0], [n_send], [0], [0])
    with pytest.raises(sb.EngineError, match="at most"):
        e.peer_connect([box] * 9, [n_recv] * 9, [0] * 9, [2] * 9, [0] * 9, list(range(9)))
    e.step(4)
    e.sync()
    e.destroy()



@pytest.mark.parametrize("world,mode", [(2, 2), (3, 0)])
def test_partitioned_scene_on_engines_equals_single_engine(sb, world, mode):
    """halo.partition_scene (C library sb_partition_*) on real engines wired by sb_peer_*: a pile of blobs in contact
    cut into slabs (spatial-hash collisions, contact band + 2 beam hops of ghosts), and the reference's default scene cut
    in three (beams only) -- the owned particles and beams of all ranks together equal the single-engine run bit for bit."""
    halo = sb.halo
    if mode == 2:
        buf, bounds = sb.scenes.blob_pile_buffers(12, 4, gap=19.7)
        depth, reach, steps = 2, 2 * (30.0 * 2 ** 0.5 + 22.0), 60
    else:
        buf, bounds = sb.scenes.default_buffers(2, 256, 512), 1000.0
        depth, reach, steps = 3, 0.0, 96

    def engine_for(b):
        e = sb.Engine(bounds_size=bounds, layout=2, max_particles=b.max_particles, max_beams=b.max_beams, collision_mode=mode,
                      tile_particles=256)
        e.write_buffers(b)
        return e

    ref = engine_for(buf)
    ref.step(steps)
    want = ref.load_buffers(buf.copy())
    ref.destroy()
    made = [(lb, plan, engine_for(lb)) for lb, plan in halo.partition_scene(buf, world, depth, contact_reach=reach)]
    assert all(0 < plan.n_owned < buf.particle_count for _, plan, _ in made)
    exs = [halo.PeerExchanger(e, plan, timeout_ms=4000) for _, plan, e in made]
    cards = [ex.card for ex in exs]
    for ex in exs:
        ex.connect(cards)
    done = 0
    while done < steps:
        m = min(depth, steps - done)
        for ex in exs:
            ex.step(m)
        done += m
    seen = 0
    for (lb, plan, e), ex in zip(made, exs):
        ex.verify()
        out = e.load_buffers(lb.copy())
        e.destroy()
        gid = plan.global_particle_id[plan.owned_particles]
        assert np.array_equal(out.particles[plan.owned_particles].view("u4"), want.particles[gid].view("u4"))
        for k, rec in zip(plan.global_beam_key[plan.owned_beams], out.beams[plan.owned_beams]):
            assert rec.tobytes()[8:] == want.beams[int(k)].tobytes()[8:]
        seen += gid.size
    assert seen == buf.particle_count and not np.array_equal(want.particles, buf.particles)


def test_repartition_on_engines_follows_two_clouds_through_each_other(sb):
    """halo.repartition() with real engines (spatial-hash collisions): two clouds of free particles that share no ghosts at
    the start fly through each other; re-partitioned after every frame (gather the owned state, partition, upload), the two
    ranks reproduce the single engine bit for bit."""
    import torch
    from halo_oracle import LocalBus, frame_all
    from test_partition_cpu import two_clouds
    halo = sb.halo
    world, depth, reach, frames = 2, 2, 80.0, 9
    gbuf = two_clouds(sb)

    def engine_for(buf):
        e = sb.Engine(bounds_size=1000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
        e.write_buffers(buf)
        return e

    ref = engine_for(gbuf)
    for _ in range(frames):
        ref.frame()
    want = ref.load_buffers(gbuf.copy())
    ref.destroy()
    dev = torch.device("cuda", 0)

    def build(made):
        bus, exs, engs = LocalBus(), [], []
        for r, (lbuf, plan) in enumerate(made):
            eng = engine_for(lbuf)
            tr = bus.transport(r, lambda a, b: (torch.zeros(max(a, 1), device=dev), torch.zeros(max(b, 1), device=dev)),
                               lambda t: t.data_ptr())
            exs.append(halo.Exchanger(eng, plan, tr))
            engs.append(eng)
        return bus, exs, engs

    made = halo.partition_scene(gbuf, world, depth, contact_reach=reach)
    bus, exs, engs = build(made)
    for f in range(frames):
        def sync():
            for e in engs:
                e.sync()
            torch.cuda.synchronize()
        frame_all(exs, bus, lambda dst, src: dst.copy_(src), sync)
        states = [halo.owned_state(plan, eng.load_buffers(lbuf.copy())) for (lbuf, plan), eng in zip(made, engs)]
        for e in engs:
            e.destroy()
        made = halo.repartition(gbuf, states, world, depth, reach)
        if f + 1 < frames:
            bus, exs, engs = build(made)
    P = gbuf.particle_count
    assert np.array_equal(gbuf.particles[:P].view("u4"), want.particles[:P].view("u4"))       # gbuf IS the gathered state now
    assert sum(p.ghost_p.size for _, plan in made for p in plan.peers) > 40



@pytest.mark.parametrize("world,depth,mode", [(2, 5, 0), (2, 6, 2)])
def test_ghost_zones_on_engines_that_kept_their_plan_through_a_cut(sb, world, depth, mode):
    """r04: an upload that only removed beams keeps the plan, and the caller's beam slots map onto the engine's (sb_engine.h
    h_user_slot).  sb_halo_configure names beams by DATA index: here every rank first holds its slab plus forty extra beams in the
    middle of the slot order, then receives the slab itself (the forty cut, another beam mapping), and only then its ghost lists --
    the merged run must still equal the single engine's, bit for bit."""
    import torch
    from halo_oracle import LocalBus, step_all
    halo = sb.halo
    W, H, steps, extra = 40, 48, 60, 40
    kw = dict(d=30.0, origin=(100.0, 11.5), jitter=1.0, velocity=(0.3, -4.0), strain_limit=0.5, yield_strain=0.2)
    bounds = 8000.0

    def engine_for(buf, cut_first):
        big = sb.Buffers(2, buf.max_particles, buf.max_beams + extra)
        P, B = buf.particle_count, buf.beam_count
        big.set_scene(buf.particles[:P], buf.beams[:B])                 # identity mapping: data index == slot, as in `buf`
        big.metadata[:] = buf.metadata
        big.metadata[11] = big.max_beams                                 # (MD_MAX_BEAMS)
        e = sb.Engine(bounds_size=bounds, layout=2, max_particles=big.max_particles, max_beams=big.max_beams, collision_mode=mode,
                      tile_particles=256)
        if cut_first:
            sup = big.copy()
            sup.beams[B:B + extra] = big.beams[:extra]                   # forty more beams (copies of the first forty, softer) ...
            sup.beams["spring"][B:B + extra] *= np.float32(0.5)
            m, maxP, at = sup.mapping, sup.max_particles, B // 2
            m[maxP:maxP + B + extra] = np.concatenate([np.arange(at), np.arange(B, B + extra), np.arange(at, B)])   # ... in the middle of the slot order
            sup.beam_count = B + extra
            e.write_buffers(sup)
            e.step(3)
        e.write_buffers(big)
        assert e.info("uploads_edited") == (1 if cut_first else 0)
        return e, big

    gbuf, gplan = halo.slab_scene(sb, 0, 1, W * world, H, depth=depth, **kw)
    ref, gbig = engine_for(gbuf, False)
    ref.step(steps)
    want = ref.load_buffers(gbig.copy())
    ref.destroy()

    dev = torch.device("cuda", 0)
    bus = LocalBus()
    exs, made = [], []
    for r in range(world):
        buf, plan = halo.slab_scene(sb, r, world, W, H, depth=depth, **kw)
        eng, big = engine_for(buf, True)
        tr = bus.transport(r, lambda a, b: (torch.zeros(max(a, 1), device=dev), torch.zeros(max(b, 1), device=dev)),
                           lambda t: t.data_ptr())
        exs.append(halo.Exchanger(eng, plan, tr))
        made.append((big, plan, eng))

    def sync():
        for _, _, e in made:
            e.sync()
        torch.cuda.synchronize()

    step_all(exs, bus, steps, lambda dst, src: dst.copy_(src), sync)
    parts = np.zeros_like(want.particles)
    beams = {}
    for big, plan, eng in made:
        out = eng.load_buffers(big.copy())
        gid, prt, bkey, brec = halo.gather_owned(plan, out)
        parts[gid] = prt
        for k, rec in zip(bkey, brec):
            beams[int(k)] = rec.tobytes()[8:]
        eng.destroy()
    assert np.array_equal(parts.view("u4"), want.particles.view("u4"))
    for k, rec in zip(gplan.global_beam_key, want.beams):
        assert beams[int(k)] == rec.tobytes()[8:]
