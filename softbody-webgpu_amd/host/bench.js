'use strict';
// Node-side benchmark: BASELINE config 2 driven from JavaScript through the N-API addon.
//   node softbody-webgpu_amd/host/bench.js [--width 1000] [--height 1000] [--steps 1000] [--warmup 64]
// Prints one JSON line: particle-steps/s by device time (HIP events inside sb_step_timed) and by host
// wall clock around the addon call.
const h = require('.');

function arg(name, dflt) {
    const i = process.argv.indexOf('--' + name);
    return i > 0 ? Number(process.argv[i + 1]) : dflt;
}

(async () => {
    const W = arg('width', 1000), H = arg('height', 1000), steps = arg('steps', 1000), warmup = arg('warmup', 64);
    const P = W * H, B = (H - 1) * W + (W - 1) * H + (W - 1) * (H - 1);
    const t0 = Date.now();
    const worker = new h.WGPUSoftbodyEngineWorker(null, {
        layout: 2, maxParticles: P, maxBeams: B, boundsSize: Math.max(W, H) * 30 + 2000, collisionMode: h.COLLIDE.OFF,
        maxByteLength: 2 ** 31
    });
    // splitmix-free cheap jitter: deterministic, +-1
    const jitter = (k) => ((Math.imul(k + 1, 2654435761) >>> 0) / 4294967296) * 2 - 1;
    h.fillLattice(worker.bufferMapper, 1000, 1000, 30, W, H, 50, 700, 0.2, 1e9, jitter);
    const tScene = Date.now();
    await worker.writeBuffers();
    const tUpload = Date.now();
    await worker.step(warmup);
    const w0 = process.hrtime.bigint();
    const ms = await worker.step(steps);
    const wallMs = Number(process.hrtime.bigint() - w0) / 1e6;
    await worker.loadBuffers();
    const y0 = new Float32Array(worker.bufferMapper.particleData, 0, 6)[1];
    const info = { path: worker.addon.getInfo(worker.handle, 'path'), tiles: worker.addon.getInfo(worker.handle, 'tiles') };
    await worker.destroy();
    console.log(JSON.stringify({
        host: 'node ' + process.version + ' -> N-API -> C ABI -> HIP', particles: P, beams: B, steps,
        device_ms_per_step: ms / steps, wall_ms_per_step: wallMs / steps,
        particle_steps_per_s_device: P * steps / (ms / 1e3), particle_steps_per_s_wall: P * steps / (wallMs / 1e3),
        scene_build_ms: tScene - t0, upload_ms: tUpload - tScene, first_particle_y: y0, info
    }));
})().catch((e) => { console.error(e); process.exit(1); });
