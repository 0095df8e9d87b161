'use strict';
/**
 * Multi-GPU sharding from the Node host: x-slabs with deep ghost zones (SURVEY.md 8(e)).
 *
 * The reference runs one scene on one GPU (engineWorker.ts:646-665); here one Node process per GPU each drives its
 * own engine (WGPUSoftbodyEngineWorker) on ITS slab of the scene plus a ghost zone `depth` beam hops deep, and every
 * `depth` substeps the owners refresh their neighbours' ghost zones by storing straight into the neighbours'
 * IPC-mapped device mailboxes (sb_peer_* in include/softbody.h; no collective library, no host copy).
 *   partitionScene(mapperOrBuffers, world, depth, contactReach)   any scene -> per-rank scenes + plans (C library: sb_partition_*)
 *   HaloPlan                                                       who owns what on a rank, what it trades with whom
 *   PeerExchanger                                                  mailbox set-up, connect, step(n) with a refresh every depth substeps, frame()
 *   ownedState / repartition                                       ownership follows the particles: partition again from the gathered state
 * The same three pieces exist for the Python harness in softbody-webgpu_amd/halo.py; both call the same C code.
 */
const { native } = require('./native');

function concatU32(list) {
    let n = 0;
    for (const a of list) n += a.length;
    const out = new Uint32Array(n);
    let o = 0;
    for (const a of list) { out.set(a, o); o += a.length; }
    return out;
}

class HaloPlan {
    constructor(rank, world, depth, nLocal, ownedParticles, ownedBeams, peers, globalParticleId, globalBeamKey) {
        Object.assign(this, { rank, world, depth, nLocal, ownedParticles, ownedBeams, peers, globalParticleId, globalBeamKey });
        this.nOwned = ownedParticles.length;
    }
    /** [ghost particles, sent particles, ghost beams, sent beams]: local data indices, peers concatenated */
    lists() {
        return ['ghostP', 'sendP', 'ghostB', 'sendB'].map((k) => concatU32(this.peers.map((p) => p[k])));
    }
    /**
     * Packed buffer layout with ONE contiguous segment per peer and direction:
     * [peer 0: particles x6 floats, beams x2 floats][peer 1: ...].  Returns {segments, sendFloats, recvFloats,
     * offsets: [sendPOff, sendBOff, ghostPOff, ghostBOff]} in the order of lists().
     */
    segments() {
        const segs = [], offs = [[], [], [], []];
        let so = 0, ro = 0;
        const ramp = (base, n, stride) => { const a = new Uint32Array(n); for (let k = 0; k < n; k++) a[k] = base + stride * k; return a; };
        for (const p of this.peers) {
            const ns = 6 * p.sendP.length + 2 * p.sendB.length, nr = 6 * p.ghostP.length + 2 * p.ghostB.length;
            segs.push({ rank: p.rank, send: [so, ns], recv: [ro, nr] });
            offs[0].push(ramp(so, p.sendP.length, 6));
            offs[1].push(ramp(so + 6 * p.sendP.length, p.sendB.length, 2));
            offs[2].push(ramp(ro, p.ghostP.length, 6));
            offs[3].push(ramp(ro + 6 * p.ghostP.length, p.ghostB.length, 2));
            so += ns;
            ro += nr;
        }
        return { segments: segs, sendFloats: so, recvFloats: ro, offsets: offs.map(concatU32) };
    }
}

/**
 * Split any scene into `world` x-slabs with ghost zones.  `scene` is a BufferMapper (after writeState()) or
 * {layout, maxParticles, maxBeams, metadata, mapping, particleData, beamData}.  contactReach > 0 also ghosts every
 * particle within that x-distance of a rank's own particles (collisions across slab faces; choose at least
 * depth * max(2r + motion per substep, longest beam)).  Returns [{rank, maxParticles, maxBeams, metadata, mapping,
 * particleData, beamData, plan}], the buffers in the scene's layout, sized to the rank's share.
 */
function partitionScene(scene, world, depth, contactReach, ranks) {
    const addon = native();
    const layout = typeof scene.layout === 'object' ? scene.layout.id : scene.layout;
    const part = addon.partitionCreate(layout, scene.maxParticles, scene.maxBeams, scene.metadata, scene.mapping,
        scene.particleData, scene.beamData, world, depth, contactReach || 0);
    try {
        const out = [];
        const which = ranks || Array.from({ length: world }, (_, r) => r);
        const beamStride = layout === 1 ? 40 : 44, indexBytes = layout === 1 ? 2 : 4;
        for (const r of which) {
            const c = addon.partitionRankCounts(part, r);
            const maxParticles = Math.max(c[0], 1), maxBeams = Math.max(c[1], 1);
            const local = {
                rank: r, layout, maxParticles, maxBeams,
                metadata: new ArrayBuffer(112), mapping: new ArrayBuffer(indexBytes * (maxParticles + maxBeams)),
                particleData: new ArrayBuffer(24 * maxParticles), beamData: new ArrayBuffer(beamStride * maxBeams)
            };
            addon.partitionRankScene(part, r, maxParticles, maxBeams, local.metadata, local.mapping, local.particleData, local.beamData);
            const ids = addon.partitionRankIds(part, r);
            const owned = (flags) => { const a = []; flags.forEach((f, i) => { if (f) a.push(i); }); return Uint32Array.from(a); };
            const peers = [];
            for (let j = 0; j < c[4]; j++) peers.push(addon.partitionPeer(part, r, j));
            local.plan = new HaloPlan(r, world, world > 1 ? depth : 0, c[0], owned(ids.particleOwned), owned(ids.beamOwned), peers,
                ids.particleGlobal, ids.beamGlobal);
            out.push(local);
        }
        return out;
    } finally {
        addon.partitionDestroy(part);
    }
}

/**
 * Steps one rank's engine and refreshes its ghost zone every plan.depth substeps through sb_peer_exchange: three
 * launches on the engine's own stream (pack straight into the neighbours' mailboxes over xGMI, flag handshake, unpack).
 * Two-phase set-up, because every rank's mailbox must exist before anyone connects:
 *     const ex = new PeerExchanger(handle, plan);   // ... every rank publishes ex.card ...
 *     ex.connect(cardsByRank);                      // cards of all ranks, indexable by rank
 * A card is plain data (the 64-byte IPC handle as an array of bytes), so it survives JSON / process.send.
 */
class PeerExchanger {
    constructor(handle, plan, timeoutMs) {
        this.addon = native();
        this.handle = handle;
        this.plan = plan;
        this.timeoutMs = timeoutMs || 10000;
        this.since = 0;
        this.connected = false;
        this.beamsAfterFrames = this.addon.getCounts(handle).beams;
        this.addon.haloConfigure(handle, ...plan.lists());
        const lay = plan.segments();
        this.segs = lay.segments;
        this.addon.haloSetLayout(handle, ...lay.offsets);
        const box = this.addon.peerMailbox(handle);
        this.card = {
            rank: plan.rank, pid: process.pid, pointer: box.pointer, handle: Array.from(new Uint8Array(box.handle)),
            recvFloats: lay.recvFloats, recv: this.segs.map((s) => [s.rank, s.recv[0], s.recv[1]])
        };
    }

    connect(cards) {
        const boxes = [], rfl = [], sbeg = [], slen = [], dbeg = [], slot = [];
        const same = cards.filter((c) => c && c.pid === this.card.pid);
        if (same.length > 3) throw new RangeError(same.length + ' engines of one process wired by sb_peer_*: at most 3 (HIP\'s 4 hardware queues); use one process per engine');
        for (const s of this.segs) {
            const them = cards[s.rank];
            if (!them) throw new Error('no card for rank ' + s.rank);
            const mine = them.recv.map((r, k) => [r, k]).filter(([r]) => r[0] === this.plan.rank);
            if (mine.length !== 1) throw new Error('rank ' + s.rank + ' does not list rank ' + this.plan.rank + ' as a neighbour');
            const [[, ro, rn], k] = mine[0];
            if (s.send[1] !== rn) throw new Error('rank ' + this.plan.rank + ' sends ' + s.send[1] + ' floats to rank ' + s.rank + ', which expects ' + rn);
            boxes.push(them.pid === this.card.pid ? them.pointer : this.addon.peerMap(this.handle, Uint8Array.from(them.handle)));
            rfl.push(them.recvFloats);
            sbeg.push(s.send[0]);
            slen.push(s.send[1]);
            dbeg.push(ro);
            slot.push(k);
        }
        this.addon.peerConnect(this.handle, boxes, Uint32Array.from(rfl), Uint32Array.from(sbeg), Uint32Array.from(slen),
            Uint32Array.from(dbeg), Uint32Array.from(slot), this.timeoutMs);
        this.connected = true;
    }

    exchange() {
        if (this.plan.peers.length) this.addon.peerExchange(this.handle);
    }

    /** n substeps, with a ghost refresh after every plan.depth of them (counted across calls) */
    step(n) {
        const k = this.plan.depth;
        if (!this.plan.peers.length || k <= 0) { this.addon.step(this.handle, n); return; }
        while (n > 0) {
            const m = Math.min(n, k - this.since);
            this.addon.step(this.handle, m);
            this.since += m;
            n -= m;
            if (this.since === k) { this.exchange(); this.since = 0; }
        }
    }

    /**
     * One frame on this rank (every rank calls it at the same time): `subticks` substeps with the usual refreshes, the
     * delete pass of the beams this rank OWNS, one more refresh that carries the deaths, removal of the ghost copies
     * (include/softbody.h, sb_halo_delete_ghosts; halo.py Exchanger.frame).  subticks: what the engine was created with.
     */
    frame(subticks) {
        this.step(subticks);
        this.addon.deletePass(this.handle);
        if (this.plan.peers.length) {
            this.exchange();
            this.since = 0;
            this.addon.haloDeleteGhosts(this.handle);
        }
        this.beamsAfterFrames = this.addon.getCounts(this.handle).beams;
    }

    /** beams may only disappear through frame() above: a plain deletePass()/frame() of one rank's engine makes the ranks diverge */
    verify() {
        this.addon.sync(this.handle);
        const left = this.addon.getCounts(this.handle).beams;
        if (left !== this.beamsAfterFrames)
            throw new Error('halo run in which rank ' + this.plan.rank + ' lost ' + (this.beamsAfterFrames - left) + ' beams outside PeerExchanger.frame()');
    }
}

const layoutOf = (scene) => {
    const id = typeof scene.layout === 'object' ? scene.layout.id : scene.layout;
    return { id, beamStride: id === 1 ? 40 : 44, floats: id === 1 ? 4 : 8, indexBytes: id === 1 ? 2 : 4 };
};
const DYN = [1, 2, 7, 8]; // target_length, last_length, strain, stress among a beam record's nine floats

/**
 * What a rank contributes to the gathered scene (halo.py owned_state): the rows of the particles it owns, the dynamic
 * fields of the beams it owns, and which of those beams are still in the mapping.  `local` is the rank's scene (a RankScene
 * of partitionScene) with its four buffers as read back from the engine; plain typed arrays, so the result survives
 * structured clone / JSON (Array.from) on its way to the other processes.
 */
function ownedState(plan, local) {
    const L = layoutOf(local), f = new Float32Array(local.particleData), md = new DataView(local.metadata);
    const beamCount = md.getUint32(24, true);
    const map = L.indexBytes === 2 ? new Uint16Array(local.mapping) : new Uint32Array(local.mapping);
    const alive = new Uint8Array(local.maxBeams);
    for (let s = 0; s < beamCount; s++) alive[map[local.maxParticles + s]] = 1;
    const op = plan.ownedParticles, ob = plan.ownedBeams, bv = new DataView(local.beamData);
    const particleIds = new Uint32Array(op.length), particleRows = new Float32Array(6 * op.length);
    op.forEach((i, k) => { particleIds[k] = plan.globalParticleId[i]; particleRows.set(f.subarray(6 * i, 6 * i + 6), 6 * k); });
    const beamKeys = new Uint32Array(ob.length), beamDyn = new Float32Array(4 * ob.length), live = new Uint8Array(ob.length);
    ob.forEach((j, k) => {
        beamKeys[k] = plan.globalBeamKey[j];
        DYN.forEach((w, q) => { beamDyn[4 * k + q] = bv.getFloat32(j * L.beamStride + L.floats + 4 * w, true); });
        live[k] = alive[j];
    });
    return { particleIds, particleRows, beamKeys, beamDyn, live };
}

/**
 * Ownership does not migrate by itself (DESIGN.md 5): partition again from the current state.  `scene` is the GLOBAL scene the
 * run was partitioned from (brought up to date in place: particle rows, the beams' dynamic fields, removed beams taken out
 * of the mapping by stable compaction, as compute_delete does); `states` = ownedState() of EVERY rank.  Call it between
 * frames; every rank then uploads its new scene and builds a new PeerExchanger.  (halo.py repartition.)
 */
function repartition(scene, states, world, depth, contactReach, ranks) {
    const L = layoutOf(scene), f = new Float32Array(scene.particleData), bv = new DataView(scene.beamData), md = new DataView(scene.metadata);
    const dead = new Uint8Array(scene.maxBeams), seen = new Uint8Array(scene.maxParticles);
    for (const st of states) {
        st.particleIds.forEach((g, k) => {
            if (seen[g]) throw new Error('two ranks own particle ' + g);
            seen[g] = 1;
            for (let q = 0; q < 6; q++) f[6 * g + q] = st.particleRows[6 * k + q];
        });
        st.beamKeys.forEach((g, k) => {
            DYN.forEach((w, q) => bv.setFloat32(g * L.beamStride + L.floats + 4 * w, st.beamDyn[4 * k + q], true));
            if (!st.live[k]) dead[g] = 1;
        });
    }
    const map = L.indexBytes === 2 ? new Uint16Array(scene.mapping) : new Uint32Array(scene.mapping);
    const n = md.getUint32(24, true);
    let w = 0;
    for (let s = 0; s < n; s++) {
        const idx = map[scene.maxParticles + s];
        if (!dead[idx]) map[scene.maxParticles + w++] = idx;
    }
    md.setUint32(24, w, true);
    return partitionScene(scene, world, depth, contactReach, ranks);
}

module.exports = { HaloPlan, PeerExchanger, partitionScene, ownedState, repartition };
