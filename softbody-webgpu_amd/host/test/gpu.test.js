'use strict';
// GPU test of the full Node path: JS host -> N-API addon -> C ABI -> HIP kernels.
// Loads the web app's default scene, runs 2 frames through the WGPUSoftbodyEngine façade and
// compares the saved snapshot with the golden produced by the CPU oracle (bit-exact).
const assert = require('assert');
const fs = require('fs');
const path = require('path');
const h = require('..');

const GOLDEN = path.resolve(__dirname, '..', '..', '..', 'tests', 'golden');
const rd = (f) => { const b = fs.readFileSync(path.join(GOLDEN, f)); return b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength); };

(async () => {
    const engine = new h.WGPUSoftbodyEngine(null, 0, { particleRadius: 10, subticks: 64 });
    assert.strictEqual(await engine.loadSnapshot(rd('default_scene_v1.snapshot')), true);
    await engine.setPhysicsConstants(h.Metadata.defaultConstants());
    const c = await engine.getPhysicsConstants();
    assert.strictEqual(c.gravity.y, -0.5);
    await engine.run(2);
    const snap = await engine.saveSnapshot();
    const want = Buffer.from(rd('default_scene_v1_after_2_frames.snapshot'));
    const got = Buffer.from(snap);
    assert.strictEqual(got.length, want.length);
    let diff = 0;
    for (let i = 0; i < got.length; i++) if (got[i] !== want[i]) diff++;
    assert.strictEqual(diff, 0, diff + ' bytes differ from the oracle golden');

    // N4, on GPU-produced state (VERDICT r03): the picture render.wgsl:33-89 draws -- discs with a white ring, beams coloured by
    // the device's stress / strain outputs -- of the state the GPU frames left, against the picture of the oracle's state
    const picture = (snapshot) => {
        const m = new h.BufferMapper(1 << 22, { layout: 1 });
        assert.strictEqual(m.loadSnapshotbuffer(snapshot), true);
        return h.renderPPM(m, { resolution: 500 });
    };
    const ppmGpu = picture(snap), ppmOracle = picture(rd('default_scene_v1_after_2_frames.snapshot'));
    assert.ok(ppmGpu.equals(ppmOracle), 'the render of the GPU state differs from the render of the oracle state');
    const ppmStart = picture(rd('default_scene_v1.snapshot'));
    assert.ok(!ppmGpu.equals(ppmStart), 'two frames must have moved the picture');
    let coloured = 0; // beams under load are not white / plain green: some pixel has a red channel between the extremes
    for (let i = 15; i + 2 < ppmGpu.length; i += 3) if (ppmGpu[i] > 0 && ppmGpu[i] < 255 && ppmGpu[i + 2] !== ppmGpu[i]) coloured++;
    assert.ok(coloured > 50, 'stress-coloured beam pixels: ' + coloured);

    // wide layout + tiled path through the worker API directly, 1000 substeps (BASELINE config 1 shape)
    const w = new h.WGPUSoftbodyEngineWorker(null, { layout: 2, maxParticles: 2048, maxBeams: 8192, collisionMode: h.COLLIDE.OFF,
        path: h.PATH.TILED, tileParticles: 256 });
    let ids = { particleId: 0, beamId: 0 };
    ids = h.addRectangle(w.bufferMapper, ids, 100, 100, 25, 32, 32, 50, 700, 0.2, 0.5, false);
    w.bufferMapper.writeState();
    await w.writeBuffers();
    const ms = await w.step(1000);
    await w.loadBuffers();
    w.bufferMapper.loadState();
    assert.strictEqual(w.bufferMapper.particleSet.size, 1024);
    assert.strictEqual(w.bufferMapper.beamSet.size, 2945);
    const p0 = w.bufferMapper.findParticle(0);
    assert.ok(p0.position.y < 100 && p0.position.y >= 10, 'lattice fell: ' + p0.position.y);
    // ... and byte for byte the oracle's state after the same 1000 substeps (tests/golden/make_golden.py)
    const gold = JSON.parse(fs.readFileSync(path.join(GOLDEN, 'lattice_32x32_after_1000_substeps.json'), 'utf8'));
    const lsnap = Buffer.from(w.bufferMapper.createSnapshotBuffer());
    assert.strictEqual(lsnap.length, gold.bytes);
    assert.deepStrictEqual([p0.position.x, p0.position.y, p0.velocity.x, p0.velocity.y], gold.first_particle.slice(0, 4));
    assert.strictEqual(require('crypto').createHash('sha256').update(lsnap).digest('hex'), gold.sha256, 'config-1 lattice after 1000 substeps differs from the oracle golden');
    const info = { tiles: w.addon.getInfo(w.handle, 'tiles'), path: w.addon.getInfo(w.handle, 'path') };
    await w.destroy();
    await engine.destroy();
    assert.strictEqual(engine.destroyed, true);
    console.log(JSON.stringify({ ok: true, frames: 2, substeps1000_ms: ms, info, renderedBytes: ppmGpu.length, colouredPixels: coloured }));
})().catch((e) => { console.error(e); process.exit(1); });
