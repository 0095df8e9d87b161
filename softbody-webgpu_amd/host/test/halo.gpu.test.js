'use strict';
// GPU test of the multi-GPU path FROM THE NODE HOST: two `node` processes, each with its own engine on its x-slab of one
// scene (host/halo.js partitionScene -> C library sb_partition_*), trading ghost zones through IPC-mapped device mailboxes
// (PeerExchanger -> N-API -> sb_peer_*) on the one GPU of the test box; the owned particles of both ranks together must
// equal the single-engine run bit for bit.  The parent is the third process: it runs the unpartitioned scene.
const assert = require('assert');
const { fork } = require('child_process');
const h = require('..');

const BOUNDS = 4000, DEPTH = 4, STEPS = 96, WORLD = 2;
// HALO_FRAMES=n: beams break (strain limit 0.02) and every rank runs n whole frames with PeerExchanger.frame() -- the delete pass
// agreed between ranks (the owner decides, the ghost copies follow); the single engine steps n frames
const FRAMES = Number(process.env.HALO_FRAMES || 0), SUBTICKS = 64;

function buildScene() {
    const m = new h.BufferMapper(1 << 27, { layout: 2, maxParticles: 2048, maxBeams: 8192 });
    h.addRectangle(m, { particleId: 0, beamId: 0 }, 100, 40, 25, 48, 30, 50, 700, 0.2, FRAMES ? 0.02 : 1e9, false);
    m.writeState();
    const f = new Float32Array(m.particleData);
    for (let i = 0; i < m.meta.particleCount; i++) { // a thrown lattice: the floor response and border accelerations act
        f[6 * i] += 0.37 * Math.sin(i * 12.9898);
        f[6 * i + 1] += 0.41 * Math.cos(i * 78.233);
        f[6 * i + 2] = 0.3;
        f[6 * i + 3] = -4.0;
    }
    return m;
}
const engineOpts = (maxParticles, maxBeams) => ({ layout: 2, boundsSize: BOUNDS, maxParticles, maxBeams,
    collisionMode: h.COLLIDE.OFF, path: h.PATH.TILED, tileParticles: 256 });
const copyInto = (dst, src) => new Uint8Array(dst).set(new Uint8Array(src));

async function child(rank) {
    const local = h.partitionScene(buildScene(), WORLD, DEPTH, 0, [rank])[0];
    const w = new h.WGPUSoftbodyEngineWorker(null, engineOpts(local.maxParticles, local.maxBeams));
    const bm = w.bufferMapper;
    copyInto(bm.metadata, local.metadata);
    copyInto(bm.mapping, local.mapping);
    copyInto(bm.particleData, local.particleData);
    copyInto(bm.beamData, local.beamData);
    await w.writeBuffers();
    const ex = new h.PeerExchanger(w.handle, local.plan, 8000);
    const cards = await new Promise((resolve) => {
        process.once('message', (m) => resolve(m.cards));
        process.send({ type: 'card', rank, card: ex.card });
    });
    ex.connect(cards);
    if (FRAMES) for (let k = 0; k < FRAMES; k++) ex.frame(SUBTICKS);
    else ex.step(STEPS);
    ex.verify();
    const beamsLeft = w.addon.getCounts(w.handle).beams;
    await w.loadBuffers();
    const f = new Uint32Array(bm.particleData);
    const owned = Array.from(local.plan.ownedParticles).map((i) => [local.plan.globalParticleId[i], Array.from(f.subarray(6 * i, 6 * i + 6))]);
    const substepsPerLaunch = w.addon.getInfo(w.handle, 'substeps_per_launch');
    await w.destroy();
    process.send({ type: 'result', rank, owned, nLocal: local.plan.nLocal, ghosts: local.plan.nLocal - local.plan.nOwned, substepsPerLaunch, beamsLeft });
    process.disconnect();
}

async function parent() {
    const kids = [], cards = [], results = [];
    const done = new Promise((resolve, reject) => {
        for (let r = 0; r < WORLD; r++) {
            const k = fork(__filename, ['child', String(r)]);
            kids.push(k);
            k.on('message', (m) => {
                if (m.type === 'card') {
                    cards[m.rank] = m.card;
                    if (cards.filter(Boolean).length === WORLD) kids.forEach((q) => q.send({ cards }));
                } else if (m.type === 'result') {
                    results[m.rank] = m;
                    if (results.filter(Boolean).length === WORLD) resolve();
                }
            });
            k.on('exit', (code) => { if (code) reject(new Error('rank ' + r + ' exited with ' + code)); });
        }
    });
    // meanwhile: the single engine
    const m = buildScene();
    const w = new h.WGPUSoftbodyEngineWorker(null, engineOpts(m.maxParticles, m.maxBeams));
    copyInto(w.bufferMapper.metadata, m.metadata);
    copyInto(w.bufferMapper.mapping, m.mapping);
    copyInto(w.bufferMapper.particleData, m.particleData);
    copyInto(w.bufferMapper.beamData, m.beamData);
    await w.writeBuffers();
    if (FRAMES) for (let k = 0; k < FRAMES; k++) { w.addon.step(w.handle, SUBTICKS); w.addon.deletePass(w.handle); }
    else await w.step(STEPS);
    const beamsAtStart = m.meta.beamCount, beamsLeft = w.addon.getCounts(w.handle).beams;
    await w.loadBuffers();
    const want = new Uint32Array(w.bufferMapper.particleData);
    const before = new Uint32Array(m.particleData);
    await w.destroy();
    await done;
    let compared = 0, moved = 0;
    const seen = new Set();
    for (const r of results) {
        assert.ok(r.ghosts > 0 && r.substepsPerLaunch > 1, JSON.stringify({ ghosts: r.ghosts, k: r.substepsPerLaunch }));
        for (const [gid, rec] of r.owned) {
            assert.ok(!seen.has(gid));
            seen.add(gid);
            for (let k = 0; k < 6; k++) assert.strictEqual(rec[k], want[6 * gid + k], 'particle ' + gid + ' word ' + k + ' on rank ' + r.rank);
            if (want[6 * gid + 1] !== before[6 * gid + 1]) moved++;
            compared++;
        }
    }
    assert.strictEqual(compared, 48 * 30);
    assert.ok(moved > 1000);
    if (FRAMES) assert.ok(beamsLeft < beamsAtStart - 100, 'the scene is meant to break beams: ' + beamsLeft + ' of ' + beamsAtStart);
    console.log(JSON.stringify({ ok: true, ranks: WORLD, particles: compared, exchanges: STEPS / DEPTH, frames: FRAMES, beamsAtStart, beamsLeft,
        ghosts: results.map((r) => r.ghosts), substepsPerLaunch: results[0].substepsPerLaunch }));
}

(process.argv[2] === 'child' ? child(Number(process.argv[3])) : parent()).catch((e) => { console.error(e); process.exit(1); });
