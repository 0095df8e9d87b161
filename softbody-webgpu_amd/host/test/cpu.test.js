'use strict';
// CPU-side tests of the JavaScript host mirror (run by tests/test_node_host.py with Node 12).
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const h = require('..');

const GOLDEN = path.resolve(__dirname, '..', '..', '..', 'tests', 'golden');
const results = [];
function test(name, fn) {
    try {
        const r = fn();
        if (r && r.then) return r.then(() => results.push(name), (e) => { console.error('FAIL', name, e); process.exitCode = 1; });
        results.push(name);
    } catch (e) {
        console.error('FAIL', name, e);
        process.exitCode = 1;
    }
    return null;
}
const same = (ab, file) => Buffer.compare(Buffer.from(ab), fs.readFileSync(path.join(GOLDEN, file))) === 0;

(async () => {
    test('default scene snapshot bytes, layout 1 (engineMapping.ts:377-401, main.ts:188-246)', () => {
        const m = h.defaultScene(new h.BufferMapper(1 << 27));
        assert.strictEqual(m.maxParticles, 65536);
        assert.strictEqual(m.meta.particleCount, 0);
        const snap = m.createSnapshotBuffer();
        assert.strictEqual(m.meta.particleCount, 119);
        assert.strictEqual(m.meta.beamCount, 299);
        assert.ok(same(snap, 'default_scene_v1.snapshot'));
    });
    test('default scene snapshot bytes, layout 2', () => {
        const m = h.defaultScene(new h.BufferMapper(1 << 27, { layout: 2, maxParticles: 256, maxBeams: 512 }));
        assert.ok(same(m.createSnapshotBuffer(), 'default_scene_v2.snapshot'));
    });
    test('snapshot load -> object model -> snapshot is the identity', () => {
        const golden = fs.readFileSync(path.join(GOLDEN, 'default_scene_v1_after_2_frames.snapshot'));
        const ab = golden.buffer.slice(golden.byteOffset, golden.byteOffset + golden.byteLength);
        const m = new h.BufferMapper(1 << 27);
        assert.strictEqual(m.loadSnapshotbuffer(ab), true);
        assert.strictEqual(m.particleSet.size, 119);
        assert.strictEqual(m.beamSet.size, 299);
        const b0 = m.findBeam(0);
        assert.strictEqual(b0.a, 0);
        assert.strictEqual(b0.b, 1);
        assert.ok(same(m.createSnapshotBuffer(), 'default_scene_v1_after_2_frames.snapshot'));
    });
    test('record bytes of Particle.to / Beam.to (engineMapping.ts:118-124,178-194)', () => {
        const m = new h.BufferMapper(4096, { maxParticles: 8, maxBeams: 8 });
        m.addParticle(new h.Particle(5, new h.Vector2D(1, 2), new h.Vector2D(3, 4), new h.Vector2D(5, 6)));
        m.addParticle(new h.Particle(9, new h.Vector2D(7, 8)));
        m.addBeam(new h.Beam(3, 9, 5, 30, 50, 700, 0.2, 0.5, 29, 31));
        m.writeState();
        assert.deepStrictEqual(Array.from(new Float32Array(m.particleData, 0, 6)), [1, 2, 3, 4, 5, 6]);
        assert.deepStrictEqual(Array.from(new Uint16Array(m.mapping, 0, 2)), [0, 1]);
        assert.deepStrictEqual(Array.from(new Uint16Array(m.beamData, 0, 2)), [1, 0]); // ids 9,5 -> indices 1,0
        const f = new Float32Array(m.beamData, 4, 7);
        assert.deepStrictEqual(Array.from(f).map((x) => +x.toFixed(4)), [30, 29, 31, 50, 700, 0.2, 0.5]);
        const md = new Uint32Array(m.metadata);
        assert.deepStrictEqual([md[0], md[1], md[5], md[6], md[10], md[11]], [3, 2, 2, 1, 8, 8]);
        assert.strictEqual(new Float32Array(m.metadata)[13], -0.5);
    });
    test('edit API (engineMapping.ts:432-495)', () => {
        const m = new h.BufferMapper(4096, { maxParticles: 4, maxBeams: 4 });
        assert.strictEqual(m.firstEmptyParticleId, 0);
        assert.ok(m.addParticle(new h.Particle(0)));
        assert.ok(!m.addParticle(new h.Particle(0)));
        assert.ok(m.addParticle(new h.Particle(2)));
        assert.strictEqual(m.firstEmptyParticleId, 1);
        const b = new h.Beam(0, 0, 2, 10, 1, 1, 1, 1);
        assert.ok(m.addBeam(b));
        assert.strictEqual(m.getConnectedBeams(2).size, 1);
        assert.ok(m.removeBeam(0));
        assert.strictEqual(m.getConnectedBeams(2).size, 0);
        assert.strictEqual(m.findBeam(0), null);
        assert.ok(m.removeParticle(2));
        assert.strictEqual(m.particleSet.size, 1);
    });
    test('Vector2D helpers', () => {
        const v = new h.Vector2D(3, 4);
        assert.strictEqual(v.magnitude, 5);
        assert.strictEqual(v.norm().x, 3 * (1 / 5));
        assert.strictEqual(v.sub(new h.Vector2D(1, 1)).dot(new h.Vector2D(1, 0)), 2);
        assert.strictEqual(h.Vector2D.turnDirection(new h.Vector2D(0, 0), new h.Vector2D(1, 0), new h.Vector2D(0, 1)), -1);
        assert.strictEqual(h.Vector2D.clamp(new h.Vector2D(5, -5), h.Vector2D.zero, new h.Vector2D(1, 1)).toString(), 'Vector2D<1, 0>');
    });
    await test('AsyncLock is FIFO (lock.ts:4-19)', async () => {
        const lock = new h.AsyncLock();
        const order = [];
        await lock.acquire();
        const a = lock.acquire().then(() => { order.push('a'); lock.release(); });
        const b = lock.acquire().then(() => { order.push('b'); lock.release(); });
        order.push('first');
        lock.release();
        await Promise.all([a, b]);
        assert.deepStrictEqual(order, ['first', 'a', 'b']);
        await lock.acquire(); // free again
        lock.release();
    });
    test('layout 1 rejects more than 65536 elements; message enum matches engine.ts:3-14', () => {
        assert.throws(() => new h.BufferMapper(1 << 30, { maxParticles: 70000 }), RangeError);
        assert.strictEqual(h.WGPUSoftbodyEngineMessageTypes.SNAPSHOT_LOAD, 7);
        assert.strictEqual(h.WGPUSoftbodyEngineMessageTypes.CORRUPT_BUFFERS, 9);
    });
    test('addon loads, exports the C-ABI wrappers, and has no CPU fallback', () => {
        const a = h.native();
        for (const f of ['create', 'destroy', 'writeBuffers', 'loadBuffers', 'writeUserInput', 'setPhysicsConstants',
            'frame', 'step', 'sync', 'stepTimed', 'mark', 'markElapsed', 'getCounts', 'getInfo', 'deletePass'])
            assert.strictEqual(typeof a[f], 'function', f);
        if (process.env.SOFTBODY_EXPECT_NO_GPU === '1') {
            assert.throws(() => new h.WGPUSoftbodyEngine({}), /no CPU fallback/);
        }
    });
    test('headless renderer draws discs and stress-coloured beams (render.wgsl:42-54,82)', () => {
        const m = h.defaultScene(new h.BufferMapper(1 << 27));
        m.writeState();
        const ppm = h.renderPPM(m, { resolution: 500 });
        assert.strictEqual(ppm.slice(0, 15).toString('ascii'), 'P6\n500 500\n255\n');
        const px = (x, y) => { const k = 15 + ((499 - y) * 500 + x) * 3; return [ppm[k], ppm[k + 1], ppm[k + 2]]; };
        // free particle at (925, 10) (main.ts:222), 2 px per unit: inner colour slightly off the centre
        assert.deepStrictEqual(px(463, 6), [0, 89, 128]);
        // beam 0 joins particles (185,10)-(185,70): stress = strain = 0 -> (1,1,1) on the line between the discs
        assert.deepStrictEqual(px(92, 20), [255, 255, 255]);
        assert.deepStrictEqual(px(300, 300), [0, 0, 0]);
    });
    test('partitionScene: the default scene in two and three x-slabs (C library sb_partition_*, host/halo.js)', () => {
        const m = h.defaultScene(new h.BufferMapper(1 << 27, { layout: 2, maxParticles: 256, maxBeams: 512 }));
        m.writeState();
        for (const world of [2, 3]) {
            const ranks = h.partitionScene(m, world, 2, 0);
            assert.strictEqual(ranks.length, world);
            assert.strictEqual(ranks.reduce((n, r) => n + r.plan.nOwned, 0), 119);
            assert.strictEqual(ranks.reduce((n, r) => n + r.plan.ownedBeams.length, 0), 299);
            const seen = new Set();
            for (const r of ranks) {
                for (const i of r.plan.ownedParticles) { assert.ok(!seen.has(r.plan.globalParticleId[i])); seen.add(r.plan.globalParticleId[i]); }
                const md = new Uint32Array(r.metadata);
                assert.strictEqual(md[1], r.plan.nLocal);
                assert.strictEqual(md[10], r.maxParticles);
                for (const peer of r.plan.peers) {
                    const back = ranks[peer.rank].plan.peers.filter((q) => q.rank === r.rank)[0];
                    // what I hold as ghosts is what the owner sends, in the same order, and starts out bit-identical
                    assert.deepStrictEqual(Array.from(peer.ghostP).map((i) => r.plan.globalParticleId[i]),
                        Array.from(back.sendP).map((i) => ranks[peer.rank].plan.globalParticleId[i]));
                    assert.deepStrictEqual(Array.from(peer.ghostB).map((i) => r.plan.globalBeamKey[i]),
                        Array.from(back.sendB).map((i) => ranks[peer.rank].plan.globalBeamKey[i]));
                    const mine = new Float32Array(r.particleData), theirs = new Float32Array(ranks[peer.rank].particleData);
                    peer.ghostP.forEach((i, k) => assert.strictEqual(mine[6 * i], theirs[6 * back.sendP[k]]));
                }
                const lay = r.plan.segments();
                assert.strictEqual(lay.offsets[2].length, r.plan.lists()[0].length);
            }
            assert.strictEqual(seen.size, 119);
        }
        assert.throws(() => h.partitionScene(m, 2, 0, 0), /depth 0/);
        const a = h.native();
        {
            // buffers sized for the v1 records (2-byte indices, 40-byte beams) handed in for a v2 partition: sb_partition_rank_scene
            // would write 4-byte indices and 44-byte beams past their ends, so the addon refuses them by the PARTITION's layout
            const part = a.partitionCreate(2, m.maxParticles, m.maxBeams, m.metadata, m.mapping, m.particleData, m.beamData, 2, 2, 0);
            try {
                const c = a.partitionRankCounts(part, 0), P = Math.max(c[0], 1), B = Math.max(c[1], 1);
                const md = new ArrayBuffer(112), pd = new ArrayBuffer(24 * P);
                assert.throws(() => a.partitionRankScene(part, 0, P, B, md, new ArrayBuffer(2 * (P + B)), pd, new ArrayBuffer(44 * B)), RangeError);
                assert.throws(() => a.partitionRankScene(part, 0, P, B, md, new ArrayBuffer(4 * (P + B)), pd, new ArrayBuffer(40 * B)), RangeError);
                a.partitionRankScene(part, 0, P, B, md, new ArrayBuffer(4 * (P + B)), pd, new ArrayBuffer(44 * B));
            } finally {
                a.partitionDestroy(part);
            }
        }
        for (const f of ['haloConfigure', 'haloSetLayout', 'peerMailbox', 'peerMap', 'peerConnect', 'peerExchange', 'getStream'])
            assert.strictEqual(typeof a[f], 'function', f);
    });
    test('repartition: gathered state back into the global scene, removed beams compacted out, same partition again', () => {
        const m = h.defaultScene(new h.BufferMapper(1 << 27, { layout: 2, maxParticles: 256, maxBeams: 512 }));
        m.writeState();
        const before = { particles: new Uint8Array(m.particleData).slice(), beams: new Uint8Array(m.beamData).slice() };
        const ranks = h.partitionScene(m, 3, 2, 0);
        // every rank moves its own particles by its rank number and "loses" the beam with the lowest key it owns
        const gone = [];
        const states = ranks.map((r) => {
            const f = new Float32Array(r.particleData);
            for (const i of r.plan.ownedParticles) f[6 * i] += 1 + r.rank;
            const st = h.ownedState(r.plan, r);
            st.live[0] = 0;
            gone.push(st.beamKeys[0]);
            st.beamDyn[1] = 123.5; // last_length of that beam
            return st;
        });
        const again = h.repartition(m, states, 3, 2, 0);
        assert.strictEqual(new DataView(m.metadata).getUint32(24, true), 299 - 3);
        const map = new Uint32Array(m.mapping), left = Array.from(map.subarray(m.maxParticles, m.maxParticles + 296));
        for (const g of gone) assert.ok(!left.includes(g));
        assert.deepStrictEqual(left, Array.from({ length: 299 }, (_, k) => k).filter((k) => !gone.includes(k)));   // stable
        const f = new Float32Array(m.particleData), f0 = new Float32Array(before.particles.buffer);
        for (const r of ranks) for (const i of r.plan.ownedParticles) {
            const g = r.plan.globalParticleId[i];
            assert.strictEqual(f[6 * g], Math.fround(f0[6 * g] + 1 + r.rank));
            assert.strictEqual(f[6 * g + 1], f0[6 * g + 1]);
        }
        assert.strictEqual(new DataView(m.beamData).getFloat32(gone[0] * 44 + 8 + 8, true), 123.5);
        assert.strictEqual(again.length, 3);
        assert.strictEqual(again.reduce((n, r) => n + r.plan.ownedBeams.length, 0), 296);
        assert.strictEqual(again.reduce((n, r) => n + r.plan.nOwned, 0), 119);
        assert.throws(() => h.repartition(m, [states[0], states[0]], 3, 2, 0), /two ranks own/);
    });
    console.log(JSON.stringify({ passed: results.length, failed: process.exitCode ? 1 : 0, names: results }));
})();
