"""Test doubles for the multi-rank path: an oracle-backed rank engine with the halo_* methods of
sb.Engine (TEST INFRASTRUCTURE: the product never imports this), and an in-process transport that
wires several simulated ranks together."""
import numpy as np


class OracleRank:
    """oracle.OracleEngine + halo_configure/pack/unpack on numpy buffers (what sb_halo_* do on the GPU)."""

    def __init__(self, oracle, buf, bounds, mode=0, threads=1):
        self.ref = oracle.OracleEngine(bounds, 10.0, 64, buf.layout, mode, threads=threads)
        self.ref.write_buffers(buf)

        self.subticks = self.ref.subticks

    def step(self, n):
        self.ref.step(n)

    # ---- beams that break (include/softbody.h, sb_halo_delete_ghosts): the same protocol on the oracle's buffers.
    # The oracle compacts the beam slots in place, so a beam is found through the live part of the mapping.
    DEAD = np.uint32(0x7FC0DEAD)

    def _beam_slots(self):
        """data index -> current mapping slot of every live beam"""
        md = self.ref.metadata.view("<u4")
        live = self.ref.mapping[self.ref.max_particles:self.ref.max_particles + int(md[6])].astype(np.int64)
        slot_of = np.full(self.ref.max_beams, -1, np.int64)
        slot_of[live] = np.arange(live.size)
        return slot_of

    def _set_flags(self, data_idx, on):
        slot_of = self._beam_slots()
        for s in slot_of[np.asarray(data_idx, np.int64)]:
            if s < 0:
                continue
            bit = self.ref.max_particles + int(s)
            if on:
                self.ref.delete[bit // 32] |= np.uint32(1 << (bit % 32))
            else:
                self.ref.delete[bit // 32] &= np.uint32(~(1 << (bit % 32)) & 0xFFFFFFFF)

    def delete_pass(self):
        if getattr(self, "gb", None) is not None and self.gb.size:
            self._set_flags(self.gb, False)       # ghost copies do not decide for themselves
        self.ref.delete_pass()

    def halo_delete_ghosts(self):
        self.ref.delete_pass()

    def counts(self):
        md = self.ref.metadata.view("<u4")
        return int(md[1]), int(md[6])       # particle_i_c, beam_i_c (engineMapping.ts:255-256)

    def info(self, key):
        assert key == "beams_flagged"
        return int(sum(bin(int(w)).count("1") for w in self.ref.delete))

    def halo_configure(self, gp, sp, gb, sb_):
        self.gp, self.sp, self.gb, self.sb = [np.asarray(x, dtype=np.int64) for x in (gp, sp, gb, sb_)]
        self.halo_set_layout(6 * np.arange(self.sp.size), 6 * self.sp.size + 2 * np.arange(self.sb.size),
                             6 * np.arange(self.gp.size), 6 * self.gp.size + 2 * np.arange(self.gb.size))

    def halo_set_layout(self, sp_off, sb_off, gp_off, gb_off):
        self.sp_off, self.sb_off, self.gp_off, self.gb_off = [np.asarray(x, dtype=np.int64) for x in
                                                              (sp_off, sb_off, gp_off, gb_off)]

    def _cur(self):
        return self.ref.particles_b if self.ref.final_in_b else self.ref.particles_a

    def halo_pack(self, dst):
        dst = np.asarray(dst)
        if self.sp.size:
            dst[self.sp_off[:, None] + np.arange(6)] = self._cur()[self.sp]
        b = self.ref.beams
        if self.sb.size:
            dead = self._beam_slots()[self.sb] < 0
            dst[self.sb_off] = np.where(dead, np.float32(0.0), b["target_length"][self.sb])
            dst.view("<u4")[self.sb_off + 1] = np.where(dead, OracleRank.DEAD, b["last_length"][self.sb].view("<u4"))

    def halo_unpack(self, src):
        src = np.asarray(src)
        if self.gp.size:
            self._cur()[self.gp] = src[self.gp_off[:, None] + np.arange(6)]
        if self.gb.size:
            dead = src.view("<u4")[self.gb_off + 1] == OracleRank.DEAD
            keep = ~dead
            self.ref.beams["target_length"][self.gb[keep]] = src[self.gb_off[keep]]
            self.ref.beams["last_length"][self.gb[keep]] = src[self.gb_off[keep] + 1]
            if dead.any():
                self._set_flags(self.gb[dead], True)   # (already removed copies have no slot any more: skipped)

    # ---- sb_peer_* double: mailboxes are numpy arrays in a registry keyed by a fake pointer; the exchange is
    # split in two (post / collect) because simulated ranks run one after another, not concurrently
    MAILBOXES = {}

    def peer_mailbox(self):
        n = int(max((self.gp_off + 6).max() if self.gp.size else 0, (self.gb_off + 2).max() if self.gb.size else 0))
        self.box = dict(flags=np.zeros(64, "u4"), bufs=[np.zeros(n, "f4"), np.zeros(n, "f4")])
        ptr = 0x1000 + len(OracleRank.MAILBOXES)
        OracleRank.MAILBOXES[ptr] = self.box
        return ptr, b"\0" * 64, 256 + 8 * n

    def peer_map(self, handle):
        raise AssertionError("simulated ranks share one process: connect() must use the local pointer")

    def peer_connect(self, boxes, recv_floats, send_begin, send_len, dst_begin, their_slot, timeout_ms=0):
        self.peers = [dict(box=OracleRank.MAILBOXES[b], begin=int(sb), len=int(sl), dst=int(db), slot=int(ts))
                      for b, sb, sl, db, ts in zip(boxes, send_begin, send_len, dst_begin, their_slot)]
        for p, n in zip(self.peers, recv_floats):
            assert p["box"]["bufs"][0].size == n
        self.seq = 0

    def peer_post(self):
        self.seq += 1
        n = int(max((self.sp_off + 6).max() if self.sp.size else 0, (self.sb_off + 2).max() if self.sb.size else 0))
        packed = np.zeros(n, "f4")
        self.halo_pack(packed)
        for p in self.peers:
            p["box"]["bufs"][self.seq & 1][p["dst"]:p["dst"] + p["len"]] = packed[p["begin"]:p["begin"] + p["len"]]
            p["box"]["flags"][p["slot"]] = self.seq

    def peer_collect(self):
        for j in range(len(self.peers)):
            assert self.box["flags"][j] == self.seq, "neighbour %d has not posted exchange %d" % (j, self.seq)
        self.halo_unpack(self.box["bufs"][self.seq & 1])

    def load(self, buf):
        return self.ref.load_buffers(buf.copy())


class CpuTorchTransport:
    """halo.TorchTransport for CPU tensors whose `pointer` is a numpy view (gloo tests)."""

    def __init__(self, torch, dist):
        from importlib import import_module
        self.inner = import_module("softbody_webgpu_amd.halo").TorchTransport(torch, dist, torch.device("cpu"))

    def allocate(self, a, b):
        return self.inner.allocate(a, b)

    def pointer(self, t):
        return t.numpy()

    def exchange(self, send, recv, segs, engine):
        self.inner.exchange(send, recv, segs, engine)


class LocalBus:
    """Lock-step exchange between simulated ranks living in one process."""

    def __init__(self):
        self.ranks = {}

    def transport(self, rank, make_buffers, pointer, sync=None):
        bus = self

        class T:
            def allocate(self, n_send, n_recv):
                s, r = make_buffers(n_send, n_recv)
                bus.ranks[rank] = dict(send=s, recv=r)
                return s, r

            def pointer(self, t):
                return pointer(t)

            def exchange(self, send, recv, segs, engine):
                bus.ranks[rank]["segs"] = segs  # the copy happens in LocalBus.flush once every rank packed

        return T()

    def flush(self, copy):
        """copy(dst_tensor_slice, src_tensor_slice) for every (sender -> receiver) segment pair."""
        for r, me in self.ranks.items():
            for seg in me.get("segs", []):
                peer = self.ranks[seg["rank"]]
                back = [s for s in peer["segs"] if s["rank"] == r][0]
                for (so, sn), (ro, rn) in zip(back["send"], seg["recv"]):
                    assert sn == rn, "send/recv segment sizes disagree"
                    if rn:
                        copy(me["recv"][ro:ro + rn], peer["send"][so:so + sn])


def step_all(exchangers, bus, n, copy, sync=lambda: None):
    """Advance every simulated rank n substeps, exchanging ghost zones every `depth` substeps."""
    k = exchangers[0].plan.depth
    since = 0
    while n > 0:
        m = min(n, k - since) if k > 0 else n
        for ex in exchangers:
            ex.engine.step(m)
        since += m
        n -= m
        if k > 0 and since == k:
            for ex in exchangers:
                ex.engine.halo_pack(ex.transport.pointer(ex.send))
                ex.transport.exchange(ex.send, ex.recv, ex.segs, ex.engine)
            sync()
            bus.flush(copy)
            sync()
            for ex in exchangers:
                ex.engine.halo_unpack(ex.transport.pointer(ex.recv))
            since = 0


def frame_all(exchangers, bus, copy, sync=lambda: None):
    """Exchanger.frame() for simulated ranks in lock step: the substeps of a frame, every rank's delete pass of its OWN beams,
    a refresh that carries the deaths, the removal of the ghost copies."""
    step_all(exchangers, bus, exchangers[0].engine.subticks, copy, sync)
    for ex in exchangers:
        ex.engine.delete_pass()
    if exchangers[0].plan.depth > 0 and any(ex.plan.peers for ex in exchangers):
        for ex in exchangers:
            ex.engine.halo_pack(ex.transport.pointer(ex.send))
            ex.transport.exchange(ex.send, ex.recv, ex.segs, ex.engine)
        sync()
        bus.flush(copy)
        sync()
        for ex in exchangers:
            ex.engine.halo_unpack(ex.transport.pointer(ex.recv))
            ex.engine.halo_delete_ghosts()
    for ex in exchangers:
        ex.beams_after_frames = ex.engine.counts()[1]


def step_all_peer(exchangers, n):
    """step_all for PeerExchangers over OracleRank doubles (post everywhere, then collect everywhere)."""
    k = exchangers[0].plan.depth
    since = 0
    while n > 0:
        m = min(n, k - since) if k > 0 else n
        for ex in exchangers:
            ex.engine.step(m)
        since += m
        n -= m
        if k > 0 and since == k:
            for ex in exchangers:
                if ex.plan.peers:
                    ex.engine.peer_post()
            for ex in exchangers:
                if ex.plan.peers:
                    ex.engine.peer_collect()
            since = 0
