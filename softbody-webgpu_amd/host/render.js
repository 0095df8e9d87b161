'use strict';
/**
 * Headless software renderer (debugging / visual regression only; SURVEY.md 8(f) N4).
 * Draws what /root/reference/src/shaders/render.wgsl draws, from the same buffers the reference binds
 * as vertex/index buffers (engineWorker.ts:675-683): one disc per particle slot (inner colour
 * (0,0.7,1)*0.5 inside 0.8 r, white ring up to r; render.wgsl:42-54) and one line per beam slot coloured
 * by the device outputs stress/strain (render.wgsl:82): r = clamp(stress+1), g = clamp(1-stress),
 * b = max(0, 1-|strain|).  Particles first, beams over them, y up.  Output: binary PPM (P6).
 */
const { LAYOUTS } = require('./engineMapping');

function clamp01(v) { return Math.max(0, Math.min(1, v)); }

function renderPPM(mapper, opts) {
    const o = opts || {};
    const S = o.boundsSize !== undefined ? o.boundsSize : 1000;
    const r = o.particleRadius !== undefined ? o.particleRadius : 10;
    const res = o.resolution || 512;
    const lay = mapper.layout || LAYOUTS[1];
    const img = new Float32Array(res * res * 3); // cleared to black (engineWorker.ts:672)
    const toPx = (v) => v / S * res;
    const put = (px, py, c) => {
        if (px < 0 || py < 0 || px >= res || py >= res) return;
        const k = ((res - 1 - py) * res + px) * 3;
        img[k] = c[0]; img[k + 1] = c[1]; img[k + 2] = c[2];
    };
    const map = new DataView(mapper.mapping);
    const idx = lay.indexBytes === 2 ? (s) => map.getUint16(s * 2, true) : (s) => map.getUint32(s * 4, true);
    const pf = new Float32Array(mapper.particleData);
    const P = mapper.meta.particleCount, B = mapper.meta.beamCount;
    const inner = [0, 0.35, 0.5], ring = [1, 1, 1];
    for (let s = 0; s < P; s++) {
        const i = idx(s), cx = pf[i * 6], cy = pf[i * 6 + 1];
        const x0 = Math.floor(toPx(cx - r)), x1 = Math.ceil(toPx(cx + r));
        const y0 = Math.floor(toPx(cy - r)), y1 = Math.ceil(toPx(cy + r));
        for (let py = y0; py <= y1; py++) for (let px = x0; px <= x1; px++) {
            const wx = (px + 0.5) / res * S, wy = (py + 0.5) / res * S;
            const d = Math.hypot(wx - cx, wy - cy);
            if (d < r * 0.8) put(px, py, inner);
            else if (d < r) put(px, py, ring);
        }
    }
    const bd = new DataView(mapper.beamData);
    for (let s = 0; s < B; s++) {
        const o0 = idx(mapper.maxParticles + s) * lay.beamStride;
        const a = lay.indexBytes === 2 ? bd.getUint16(o0, true) : bd.getUint32(o0, true);
        const b = lay.indexBytes === 2 ? bd.getUint16(o0 + 2, true) : bd.getUint32(o0 + 4, true);
        const strain = bd.getFloat32(o0 + lay.beamFloatBase + 28, true), stress = bd.getFloat32(o0 + lay.beamFloatBase + 32, true);
        const col = [clamp01(stress + 1), clamp01(1 - stress), Math.max(0, 1 - Math.abs(strain))];
        const ax = toPx(pf[a * 6]), ay = toPx(pf[a * 6 + 1]), bx = toPx(pf[b * 6]), by = toPx(pf[b * 6 + 1]);
        const n = Math.max(1, Math.ceil(Math.max(Math.abs(bx - ax), Math.abs(by - ay))));
        for (let k = 0; k <= n; k++) put(Math.floor(ax + (bx - ax) * k / n), Math.floor(ay + (by - ay) * k / n), col);
    }
    const head = Buffer.from('P6\n' + res + ' ' + res + '\n255\n', 'ascii');
    const body = Buffer.alloc(res * res * 3);
    for (let k = 0; k < body.length; k++) body[k] = Math.round(clamp01(img[k]) * 255);
    return Buffer.concat([head, body]);
}

module.exports = { renderPPM };
