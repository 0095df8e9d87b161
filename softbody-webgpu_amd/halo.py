"""Multi-GPU sharding of the physics step: x-slabs with deep ghost zones.

One process / one engine per GPU.  Rank r owns W lattice columns; its engine also holds a ghost
zone `depth` columns deep on each interior side, copied from the neighbour that owns them.  Ghost
particles and ghost beams are stepped redundantly (deterministic arithmetic: bit-identical to the
owner's copy while their inputs are valid).  Invalid data enters at the outer edge of a ghost zone
and moves one beam hop per substep, so the owned slab sees only valid neighbours for `depth`
substeps; then the owners refresh the whole zone: p,v,a of every ghost particle and
target_length/last_length of every ghost beam.  One exchange per `depth` substeps instead of one
per substep: the exchange is latency-bound (a few hundred KB over 2 of the 7 xGMI links), so
amortising it is what keeps weak scaling near-linear; the price is depth/W redundant work.

The only exchange on the data path is with the two neighbours: direct stores into their IPC-mapped
mailboxes (PeerExchanger, sb_peer_*), or RCCL send/recv through torch.distributed ordered on the
engine's own HIP stream (Exchanger + TorchTransport); there is no all-reduce.  Collisions across slab
faces need nothing extra: ghosts are ordinary particles of the rank's scene (tests/test_gpu_halo.py).
"""
import numpy as np

from . import scenes
from .layout import LAYOUT_V2, Buffers


def hash_at(seed, idx):
    """scenes.hash_uniform evaluated at arbitrary indices (same splitmix64 stream)."""
    x = (np.uint64(seed) << np.uint64(32)) + np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 52) - 1.0


class Peer:
    def __init__(self, rank, ghost_p, send_p, ghost_b, send_b):
        self.rank = rank
        self.ghost_p, self.send_p, self.ghost_b, self.send_b = ghost_p, send_p, ghost_b, send_b


class HaloPlan:
    """Who owns what on one rank, and what it trades with each neighbour."""

    def __init__(self, rank, world, depth, n_local, owned_particles, owned_beams, peers, global_particle_id,
                 global_beam_key):
        self.rank, self.world, self.depth = rank, world, depth
        self.n_local = n_local
        self.owned_particles = owned_particles          # local data indices
        self.owned_beams = owned_beams                  # local beam data indices (endpoint A owned)
        self.n_owned = int(owned_particles.size)
        self.peers = peers
        self.global_particle_id = global_particle_id    # per local particle
        self.global_beam_key = global_beam_key          # per local beam: (global id of A) * 4 + kind

    def lists(self):
        cat = lambda xs: np.concatenate(xs).astype("<u4") if xs else np.zeros(0, "<u4")  # noqa: E731
        return (cat([p.ghost_p for p in self.peers]), cat([p.send_p for p in self.peers]),
                cat([p.ghost_b for p in self.peers]), cat([p.send_b for p in self.peers]))

    def segments(self):
        """Packed buffer layout with ONE contiguous segment per peer and direction:
        [peer 0: particles x6 floats, beams x2 floats][peer 1: ...].  Returns (per-peer segments,
        send floats, recv floats, (send_p_off, send_b_off, ghost_p_off, ghost_b_off)) where the
        offset arrays follow the concatenated list order of `lists()`."""
        out, offs = [], [[], [], [], []]
        so = ro = 0
        for p in self.peers:
            ns, nr = 6 * p.send_p.size + 2 * p.send_b.size, 6 * p.ghost_p.size + 2 * p.ghost_b.size
            out.append(dict(rank=p.rank, send=[(so, ns)], recv=[(ro, nr)]))
            offs[0].append(so + 6 * np.arange(p.send_p.size))
            offs[1].append(so + 6 * p.send_p.size + 2 * np.arange(p.send_b.size))
            offs[2].append(ro + 6 * np.arange(p.ghost_p.size))
            offs[3].append(ro + 6 * p.ghost_p.size + 2 * np.arange(p.ghost_b.size))
            so += ns
            ro += nr
        cat = lambda xs: np.concatenate(xs).astype("<u4") if xs else np.zeros(0, "<u4")  # noqa: E731
        return out, so, ro, tuple(cat(x) for x in offs)


def slab_scene(sb, rank, world, W, H, d=30.0, origin=(1000.0, 1000.0), jitter=0.0, depth=8, seed=1,
               spring=50.0, damp=700.0, yield_strain=0.2, strain_limit=1.0e9, velocity=None):
    """Rank `rank`'s share of one (W*world) x H lattice blob (3-beam topology, BASELINE configs 2/4):
    W owned columns plus `depth` ghost columns per interior side.  With world == 1 this is exactly
    scenes.lattice_buffers(W, H, ...).  Returns (Buffers, HaloPlan)."""
    if world > 1 and depth > W:
        raise ValueError("ghost depth %d exceeds the slab width %d" % (depth, W))
    kL = depth if rank > 0 else 0
    kR = depth if rank < world - 1 else 0
    c0 = rank * W - kL                      # global column of local column 0
    ncol = kL + W + kR
    p, beams = scenes.rectangle(origin[0] + c0 * float(d), origin[1], d, ncol, H, spring, damp, yield_strain,
                                strain_limit, anti_diagonal=False, layout=LAYOUT_V2)
    # positions must be bit-identical to the single-blob scene: recompute from GLOBAL columns
    xl = np.repeat(np.arange(ncol), H)
    yl = np.tile(np.arange(H), ncol)
    gcol = xl + c0
    gid = gcol.astype(np.int64) * H + yl
    pv = np.zeros((ncol * H, 6), dtype="<f4")
    pv[:, 0] = (gcol * float(d) + float(origin[0])).astype("<f4")
    pv[:, 1] = (yl * float(d) + float(origin[1])).astype("<f4")
    if jitter:
        jx = hash_at(seed, 2 * gid).astype("<f4") * np.float32(jitter)
        jy = hash_at(seed, 2 * gid + 1).astype("<f4") * np.float32(jitter)
        pv[:, 0] += jx
        pv[:, 1] += jy
    if velocity is not None:
        pv[:, 2:4] = np.asarray(velocity, dtype="<f4")
    buf = Buffers(LAYOUT_V2, pv.shape[0], beams.shape[0])
    buf.set_scene(pv, beams)

    col_a = beams["a"].astype(np.int64) // H
    col_b = beams["b"].astype(np.int64) // H
    a_loc = beams["a"].astype(np.int64)
    kind = np.where(beams["b"] == beams["a"] + 1, 0, np.where(beams["b"] == beams["a"] + H, 1, 2))
    beam_key = gid[a_loc] * 4 + kind
    pcol = xl
    owned_p = np.nonzero((pcol >= kL) & (pcol < kL + W))[0].astype("<u4")
    owned_b = np.nonzero((col_a >= kL) & (col_a < kL + W))[0].astype("<u4")

    def band_particles(lo, hi):
        return np.nonzero((pcol >= lo) & (pcol < hi))[0].astype("<u4")

    def band_beams(lo, hi):
        return np.nonzero((col_a >= lo) & (col_a < hi) & (col_b >= lo) & (col_b < hi))[0].astype("<u4")

    peers = []
    if kL:
        peers.append(Peer(rank - 1, band_particles(0, kL), band_particles(kL, kL + depth),
                          band_beams(0, kL), band_beams(kL, kL + depth)))
    if kR:
        peers.append(Peer(rank + 1, band_particles(kL + W, ncol), band_particles(kL + W - depth, kL + W),
                          band_beams(kL + W, ncol), band_beams(kL + W - depth, kL + W)))
    plan = HaloPlan(rank, world, depth if world > 1 else 0, pv.shape[0], owned_p, owned_b, peers, gid, beam_key)
    return buf, plan


def partition_scene(buf, world, depth, contact_reach=0.0, ranks=None):
    """Split ANY scene (a layout.Buffers: the default scene, a loaded snapshot, something a BufferMapper built) into
    x-slabs with ghost zones `depth` beam hops deep: the C library's sb_partition_* (csrc/sb_partition.cpp), shared
    with the Node host.  contact_reach > 0 also makes every particle within that distance (in x) of a rank's own
    particles a ghost, so that contacts across slab faces are computed on both sides.  Returns [(Buffers, HaloPlan)]
    for `ranks` (default: all).  global_particle_id / global_beam_key of the plans are GLOBAL DATA INDICES."""
    import ctypes
    from .engine import EngineError, _ptr, load_library
    L = load_library()
    h = ctypes.c_void_p()

    def check(st):
        if st != 0:
            raise EngineError(st, L.sb_last_error(None).decode())

    check(L.sb_partition_create(buf.layout, buf.max_particles, buf.max_beams, _ptr(buf.metadata), _ptr(buf.mapping),
                                _ptr(buf.particles), _ptr(buf.beams), world, depth, float(contact_reach), ctypes.byref(h)))
    try:
        out = []
        for r in (range(world) if ranks is None else ranks):
            c = (ctypes.c_uint32 * 8)()
            check(L.sb_partition_rank_counts(h, r, ctypes.byref(c)))
            nP, nB, n_peers = c[0], c[1], c[4]
            local = Buffers(buf.layout, max(nP, 1), max(nB, 1))
            check(L.sb_partition_rank_scene(h, r, local.max_particles, local.max_beams, _ptr(local.metadata), _ptr(local.mapping),
                                            _ptr(local.particles), _ptr(local.beams)))
            pg, po = np.zeros(nP, "<u4"), np.zeros(nP, "u1")
            bg, bo = np.zeros(nB, "<u4"), np.zeros(nB, "u1")
            check(L.sb_partition_rank_ids(h, r, _ptr(pg), _ptr(po), _ptr(bg), _ptr(bo)))
            peers = []
            for j in range(n_peers):
                pr, pc = ctypes.c_uint32(), (ctypes.c_uint32 * 4)()
                check(L.sb_partition_peer_counts(h, r, j, ctypes.byref(pr), ctypes.byref(pc)))
                lists = [np.zeros(pc[k], "<u4") for k in range(4)]
                check(L.sb_partition_peer_lists(h, r, j, *[_ptr(x) for x in lists]))
                peers.append(Peer(pr.value, *lists))
            plan = HaloPlan(r, world, depth if world > 1 else 0, nP, np.nonzero(po)[0].astype("<u4"),
                            np.nonzero(bo)[0].astype("<u4"), peers, pg.astype(np.int64), bg.astype(np.int64))
            out.append((local, plan))
        return out
    finally:
        L.sb_partition_destroy(h)


def mix_stiffness(buf, plan, seed=1, subticks=128):
    """scenes.mix_stiffness for slab scenes (BASELINE config 5 across ranks): the draw is keyed by the GLOBAL
    beam key, so a ghost copy of a beam gets exactly its owner's spring and damping."""
    B = buf.beam_count
    u = hash_at(seed + 77, plan.global_beam_key[:B])
    pick = ((u + 1.0) * 2.0).astype(np.int64).clip(0, 3)
    springs = np.array([1.0, 3.0, 50.0, 500.0], dtype="<f4")[pick]
    damp_cap = np.float32(0.5 / 6.0 * subticks * subticks)
    buf.beams["spring"][:B] = springs
    buf.beams["damp"][:B] = np.minimum(springs * np.float32(14.0), damp_cap).astype("<f4")
    return buf


class StepTimer:
    """Device time of a rank's substep launches and of its ghost refreshes, span by span, with sb_mark: events recorded on the
    engine's own stream, nothing waits until totals() is read (bench.py: `roofline` and the exchange cost of the N > 1 line)."""

    def __init__(self, engine, max_marks=4096):
        self.engine, self.max_marks = engine, max_marks
        self.n, self.spans, self.dropped = 0, [], 0

    def prepare(self, spans):
        """Create the HIP events of the first `spans` spans now (sb_mark makes an event the first time its slot is used: ~20 us
        each, which would otherwise land inside the caller's timed region)."""
        for slot in range(min(2 * spans, self.max_marks)):
            self.engine.mark(slot)

    def run(self, kind, fn, *args):
        if self.n + 2 > self.max_marks:          # out of events: the span still runs, untimed (and says so)
            self.dropped += 1
            return fn(*args)
        a = self.n
        self.n += 2
        self.engine.mark(a)
        out = fn(*args)
        self.engine.mark(a + 1)
        self.spans.append((kind, a))
        return out

    def totals(self):
        """{kind: (spans, milliseconds)} -- waits for the last mark of each span."""
        out = {}
        for kind, a in self.spans:
            n, ms = out.get(kind, (0, 0.0))
            out[kind] = (n + 1, ms + self.engine.mark_elapsed(a, a + 1))
        return out


class Exchanger:
    """Steps one rank's engine and refreshes its ghost zone every `plan.depth` substeps.

    `engine` needs step(n), halo_pack(ptr), halo_unpack(ptr); `transport` moves the packed
    buffers: TorchTransport for real ranks (torch.distributed P2P, nccl=RCCL on GPU tensors, gloo on
    CPU tensors), or a test double."""

    def __init__(self, engine, plan, transport):
        self.engine, self.plan, self.transport = engine, plan, transport
        self.since = 0
        self.timer = None            # a StepTimer while somebody wants the device times of step() (bench.py)
        self.exchanges = 0
        self.beams_after_frames = engine.counts()[1]
        gp, sp, gb, sb_ = plan.lists()
        engine.halo_configure(gp, sp, gb, sb_)
        self.segs, n_send, n_recv, offsets = plan.segments()
        engine.halo_set_layout(*offsets)
        self.send, self.recv = transport.allocate(n_send, n_recv)

    def verify(self):
        """What a halo run must not have done (call after stepping; it drains the engine's stream): removed beams by any
        other route than frame() below.  A beam that breaks keeps acting until the delete pass at the end of its frame
        (compute.wgsl:205-246), so flagged beams are harmless between passes, and frame() makes the pass agree between ranks
        (the owner decides, its neighbours' ghost copies follow: include/softbody.h, sb_halo_delete_ghosts).  A plain
        engine.delete_pass() / engine.frame() on one rank is what would make the ranks diverge, silently.
        Also stated, not checkable here: a contact between particles of two NON-adjacent slabs of a folded body is
        missed (each rank only knows its neighbours' ghost zones)."""
        _, beams = self.engine.counts()
        if beams != self.beams_after_frames:
            raise RuntimeError("halo run in which rank %d lost %d beams outside Exchanger.frame(): delete passes must go "
                               "through frame() on every rank" % (self.plan.rank, self.beams_after_frames - beams))

    def frame(self):
        """One frame on this rank: engine.subticks substeps with the usual refreshes, the delete pass of owned beams, one
        more refresh that carries the deaths, the removal of the ghost copies.  Every rank calls it at the same time.
        (It ends by reading the beam count, which waits for this rank's stream: ranks that live in ONE process and trade
        through sb_peer_* must therefore be driven in lock step, phase by phase -- tests/halo_oracle.py frame_all -- or the
        first rank waits for a neighbour whose kernels have not been enqueued yet.  One process per rank has no such issue.)"""
        self.step(self.engine.subticks)
        if not self.plan.peers:
            self.engine.delete_pass()
        else:
            self.engine.delete_pass()
            self.exchange()
            self.since = 0
            self.engine.halo_delete_ghosts()
        self.beams_after_frames = self.engine.counts()[1]

    def exchange(self):
        if not self.plan.peers:
            return
        self.engine.halo_pack(self.transport.pointer(self.send))
        self.transport.exchange(self.send, self.recv, self.segs, self.engine)
        self.engine.halo_unpack(self.transport.pointer(self.recv))

    def refresh_now(self):
        """An exchange ahead of schedule (always allowed: ghosts are refreshed with valid data), so that the next one falls
        exactly plan.depth substeps from here -- bench.py aligns its timed region with it."""
        self.exchange()
        self.exchanges += 1
        self.since = 0

    def step(self, n):
        k = self.plan.depth
        t = self.timer
        if not self.plan.peers or k <= 0:
            if t is not None:
                t.run("step", self.engine.step, n)
            else:
                self.engine.step(n)
            return
        while n > 0:
            m = min(n, k - self.since)
            if t is not None:
                t.run("step", self.engine.step, m)
            else:
                self.engine.step(m)
            self.since += m
            n -= m
            if self.since == k:
                if t is not None:
                    t.run("exchange", self.exchange)
                else:
                    self.exchange()
                self.exchanges += 1
                self.since = 0


class PeerExchanger(Exchanger):
    """Exchanger whose refresh is sb_peer_exchange: the pack kernel stores straight into the neighbours'
    mailboxes (IPC-mapped device memory, over xGMI between GPUs), a one-wave kernel trades sequence
    flags, the unpack kernel reads this rank's own mailbox.  Three launches on the engine stream, no
    collective library and no host synchronisation per exchange.

    Two-phase setup because every rank's mailbox must exist before anyone connects:
        ex = PeerExchanger(engine, plan);  cards = all_gather(ex.card);  ex.connect(cards)
    `cards` is indexable by rank."""

    def __init__(self, engine, plan, timeout_ms=10000):
        self.engine, self.plan, self.transport = engine, plan, None
        self.since = 0
        self.timer = None
        self.exchanges = 0
        self.timeout_ms = timeout_ms
        self.beams_after_frames = engine.counts()[1]
        gp, sp, gb, sb_ = plan.lists()
        engine.halo_configure(gp, sp, gb, sb_)
        self.segs, n_send, n_recv, offsets = plan.segments()
        engine.halo_set_layout(*offsets)
        ptr, handle, _ = engine.peer_mailbox()
        import os
        self.card = dict(rank=plan.rank, pid=os.getpid(), pointer=ptr, handle=handle, recv_floats=n_recv,
                         recv=[(s["rank"],) + s["recv"][0] for s in self.segs])
        self.connected = False

    def connect(self, cards):
        # HIP multiplexes a process's streams onto 4 hardware queues and a polling kernel blocks whatever is queued
        # behind it: with more than 3 engines of ONE process wired together a wait kernel can end up in front of the
        # kernel it waits for (gpurun_out/slabs.log, round 1).  Refused here, with the reason, instead of a timeout later.
        same = [c for c in cards if c is not None and c["pid"] == self.card["pid"]]
        if len(same) > 3 and hasattr(self.engine, "stream"):   # (CPU test doubles have no streams and no such limit)
            raise ValueError("%d engines of one process wired by sb_peer_*: at most 3 (HIP's 4 hardware queues); "
                             "use one process per engine, the deployment shape" % len(same))
        boxes, rfl, sbeg, slen, dbeg, slot = [], [], [], [], [], []
        for s in self.segs:
            them = cards[s["rank"]]
            mine = [k for k, (r, _, _) in enumerate(them["recv"]) if r == self.plan.rank]
            if len(mine) != 1:
                raise ValueError("rank %d does not list rank %d as a neighbour" % (s["rank"], self.plan.rank))
            _, ro, rn = them["recv"][mine[0]]
            (so, sn), = s["send"]
            if sn != rn:
                raise ValueError("rank %d sends %d floats to rank %d, which expects %d" % (self.plan.rank, sn, s["rank"], rn))
            same_process = them["pid"] == self.card["pid"]
            boxes.append(them["pointer"] if same_process else self.engine.peer_map(them["handle"]))
            rfl.append(them["recv_floats"])
            sbeg.append(so)
            slen.append(sn)
            dbeg.append(ro)
            slot.append(mine[0])
        self.engine.peer_connect(boxes, rfl, sbeg, slen, dbeg, slot, self.timeout_ms)
        self.connected = True

    def exchange(self):
        if self.plan.peers:
            self.engine.peer_exchange()


class TorchTransport:
    """Neighbour exchange over torch.distributed.  On GPU (`ordered=True`) the P2P ops are issued while
    the engine's own HIP stream is torch's current stream, so RCCL waits for the pack kernel and the
    unpack kernel waits for RCCL, all device-side: no host synchronisation per exchange.  If that
    stream-ordered form raises on the first exchange, the transport drops to the conservative form
    (drain the engine stream, exchange on torch's stream, drain it) and says so in `self.mode`."""

    def __init__(self, torch, dist, device, stream_ptr=None, ordered=True):
        self.torch, self.dist, self.device = torch, dist, device
        self.stream = None
        self.mode = "host-synchronised"
        self.exchanges = 0
        if ordered and stream_ptr is not None and device.type == "cuda":
            self.stream = torch.cuda.ExternalStream(stream_ptr, device=device)
            self.mode = "stream-ordered"

    def allocate(self, n_send, n_recv):
        t = self.torch
        return (t.zeros(max(n_send, 1), dtype=t.float32, device=self.device),
                t.zeros(max(n_recv, 1), dtype=t.float32, device=self.device))

    def pointer(self, tensor):
        return tensor.data_ptr()

    def _ops(self, send, recv, segs):
        dist = self.dist
        ops = []
        for s in segs:
            for off, n in s["send"]:
                if n:
                    ops.append(dist.P2POp(dist.isend, send[off:off + n], s["rank"]))
            for off, n in s["recv"]:
                if n:
                    ops.append(dist.P2POp(dist.irecv, recv[off:off + n], s["rank"]))
        return ops

    def exchange(self, send, recv, segs, engine):
        ops = self._ops(send, recv, segs)
        if not ops:
            return
        if self.stream is not None:
            try:
                with self.torch.cuda.stream(self.stream):
                    for r in self.dist.batch_isend_irecv(ops):
                        r.wait()
                self.exchanges += 1
                return
            except Exception as exc:  # first use only: later failures are real errors
                if self.exchanges:
                    raise
                self.stream = None
                self.mode = "host-synchronised (stream-ordered exchange failed: %s)" % type(exc).__name__
                ops = self._ops(send, recv, segs)
        if self.device.type == "cuda":
            engine.sync()                       # pack kernel done
        for r in self.dist.batch_isend_irecv(ops):
            r.wait()
        if self.device.type == "cuda":
            self.torch.cuda.synchronize(self.device)   # data landed before the unpack kernel is enqueued
        self.exchanges += 1


def ghost_mismatches(plan, buf, dist, torch, device):
    """Right after a refresh every ghost record must equal its owner's current record.  `buf` is this rank's
    state read back from the engine; the owners' records travel over torch.distributed (tensors on `device`:
    cuda for nccl, cpu for gloo), independently of the transport being checked.  Returns how many ghost
    records differ (bitwise) on this rank."""
    ops, pending = [], []
    for p in plan.peers:
        mine = np.concatenate([buf.particles[p.send_p].reshape(-1),
                               buf.beams["target_length"][p.send_b], buf.beams["last_length"][p.send_b]]).astype("<f4")
        t_send = torch.from_numpy(mine.view("<i4").copy()).to(device)
        t_recv = torch.empty(6 * p.ghost_p.size + 2 * p.ghost_b.size, dtype=torch.int32, device=device)
        ops += [dist.P2POp(dist.isend, t_send, p.rank), dist.P2POp(dist.irecv, t_recv, p.rank)]
        pending.append((p, t_recv, t_send))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    bad = 0
    for p, t_recv, _ in pending:
        theirs = t_recv.cpu().numpy().view("<u4")
        n = p.ghost_p.size
        have = np.concatenate([buf.particles[p.ghost_p].reshape(-1), buf.beams["target_length"][p.ghost_b],
                               buf.beams["last_length"][p.ghost_b]]).astype("<f4").view("<u4")
        diff = theirs != have
        bad += int(diff[:6 * n].reshape(-1, 6).any(axis=1).sum()) + int(diff[6 * n:].reshape(2, -1).any(axis=0).sum())
    return bad


def gather_owned(plan, buf):
    """(global particle ids, particle rows, global beam keys, beam records) of what this rank owns."""
    op, ob = plan.owned_particles, plan.owned_beams
    return (plan.global_particle_id[op], buf.particles[op], plan.global_beam_key[ob], buf.beams[ob])


def owned_state(plan, buf):
    """What a rank contributes to the gathered scene: (global particle ids, particle rows, global beam keys, their dynamic
    fields, which of them are still in the mapping)."""
    op, ob = plan.owned_particles, plan.owned_beams
    live = np.zeros(buf.max_beams, bool)
    live[buf.mapping[buf.max_particles:buf.max_particles + buf.beam_count].astype(np.int64)] = True
    dyn = np.stack([buf.beams[f][ob] for f in ("target_length", "last_length", "strain", "stress")], axis=1).astype("<f4")
    return (plan.global_particle_id[op], buf.particles[op].copy(), plan.global_beam_key[ob], dyn, live[ob])


def repartition(gbuf, states, world, depth, contact_reach=0.0, ranks=None):
    """Ownership does not migrate by itself: the slabs and ghost zones are those of the partition, so free particles that
    wander into another rank's territory stop being seen by that rank's owners.  The remedy is to partition
    again from the current state: `gbuf` is the global scene the run was partitioned from (layout.Buffers; brought up to
    date IN PLACE -- particle rows, the beams' dynamic fields, removed beams taken out of the mapping by stable
    compaction, as compute_delete does), `states` = owned_state(plan, loaded buffers) of EVERY rank (an all-gather in a real
    run: a few tens of bytes per particle and beam).  Returns partition_scene() of the updated scene.  Call it BETWEEN
    frames: beams flagged since the last delete pass are engine state that an upload does not carry.  Every rank then
    uploads its new scene and builds a new Exchanger (a new upload starts a new mailbox)."""
    seen = np.zeros(gbuf.max_particles, bool)
    dead = np.zeros(gbuf.max_beams, bool)
    for gid, rows, bkey, dyn, live in states:
        assert not seen[gid].any(), "two ranks own the same particle"
        seen[gid] = True
        gbuf.particles[gid] = rows
        for k, f in enumerate(("target_length", "last_length", "strain", "stress")):
            gbuf.beams[f][bkey] = dyn[:, k]
        dead[bkey[~np.asarray(live, bool)]] = True
    P0, n = gbuf.max_particles, gbuf.beam_count
    slots = gbuf.mapping[P0:P0 + n]
    keep = slots[~dead[slots.astype(np.int64)]]
    gbuf.mapping[P0:P0 + keep.size] = keep
    gbuf.beam_count = int(keep.size)
    return partition_scene(gbuf, world, depth, contact_reach, ranks)

