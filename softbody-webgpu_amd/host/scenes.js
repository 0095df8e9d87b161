'use strict';
/** Scene builders on top of BufferMapper: the lattice helper and default scene of src/main.ts:188-246. */
const { Beam, Particle, Vector2D } = require('./engineMapping');

/**
 * w x h lattice at (ox, oy) with spacing d: particle (x, y) has id base + x*h + y; beams to +y, +x and
 * the two diagonals of each cell (main.ts:203-214).  Returns the next free {particleId, beamId}.
 */
function addRectangle(mapper, ids, ox, oy, d, w, h, spring, damp, yieldStrain, strainLimit, antiDiagonal) {
    let i = ids.particleId, j = ids.beamId;
    const diag = Math.SQRT2 * d;
    const beam = (a, b, len) => mapper.addBeam(new Beam(j++, a, b, len, spring, damp, yieldStrain, strainLimit));
    for (let x = 0; x < w; x++) {
        for (let y = 0; y < h; y++) {
            const self = i++;
            mapper.addParticle(new Particle(self, new Vector2D(x * d + ox, y * d + oy)));
            const up = y < h - 1, right = x < w - 1;
            if (up) beam(self, self + 1, d);
            if (right) beam(self, self + h, d);
            if (up && right) beam(self, self + h + 1, diag);
            if (antiDiagonal !== false && y > 0 && right) beam(self, self + h - 1, diag);
        }
    }
    return { particleId: i, beamId: j };
}

/** the web app's initial scene: 119 particles / 299 beams (main.ts:218-241) */
function defaultScene(mapper) {
    mapper.clear();
    let ids = { particleId: 0, beamId: 0 };
    const rect = (...a) => { ids = addRectangle(mapper, ids, ...a); };
    const free = (x, y) => { mapper.addParticle(new Particle(ids.particleId++, new Vector2D(x, y))); };
    rect(185, 10, 60, 2, 2, 1, 50, 1, 2.5);
    rect(35, 10, 60, 2, 2, 1, 50, 1, 2.5);
    rect(20, 120, 30, 9, 4, 50, 700, 0.2, 0.5);
    free(445, 10);
    free(925, 10);
    rect(400, 40, 30, 20, 2, 500, 800, 0.1, 0.5);
    rect(700, 400, 40, 5, 5, 3, 50, 2, 5);
    rect(20, 900, 50, 2, 2, 0.05, 10, 2, 3);
    rect(20, 700, 50, 2, 2, 0.1, 10, 2, 3);
    return mapper;
}

module.exports = { addRectangle, defaultScene };

/**
 * Fast path for big lattices: fills the BufferMapper's ArrayBuffers directly (identity mapping, what
 * writeState() would produce) without creating millions of Particle/Beam objects.  Same topology and
 * emission order as addRectangle with antiDiagonal=false (BASELINE config 2: ~3 beams per particle).
 * `jitter(k)` returns a displacement for coordinate k = 2*index + {0,1} (default: none).
 */
function fillLattice(mapper, ox, oy, d, w, h, spring, damp, yieldStrain, strainLimit, jitter) {
    const P = w * h, B = (h - 1) * w + (w - 1) * h + (w - 1) * (h - 1);
    if (P > mapper.maxParticles || B > mapper.maxBeams) throw new RangeError('lattice exceeds the mapper capacity');
    const lay = mapper.layout;
    const pf = new Float32Array(mapper.particleData);
    const map = new DataView(mapper.mapping);
    const bd = new DataView(mapper.beamData);
    const setIndex = lay.indexBytes === 2 ? (o, v) => map.setUint16(o * 2, v, true) : (o, v) => map.setUint32(o * 4, v, true);
    const diag = Math.SQRT2 * d;
    let j = 0;
    const beam = (a, b, len) => {
        const o = j * lay.beamStride;
        if (lay.indexBytes === 2) { bd.setUint16(o, a, true); bd.setUint16(o + 2, b, true); } else { bd.setUint32(o, a, true); bd.setUint32(o + 4, b, true); }
        const f = o + lay.beamFloatBase;
        bd.setFloat32(f, len, true); bd.setFloat32(f + 4, len, true); bd.setFloat32(f + 8, len, true);
        bd.setFloat32(f + 12, spring, true); bd.setFloat32(f + 16, damp, true);
        bd.setFloat32(f + 20, yieldStrain, true); bd.setFloat32(f + 24, strainLimit, true);
        setIndex(mapper.maxParticles + j, j);
        j++;
    };
    for (let x = 0; x < w; x++) {
        for (let y = 0; y < h; y++) {
            const i = x * h + y;
            pf[i * 6] = x * d + ox + (jitter ? jitter(2 * i) : 0);
            pf[i * 6 + 1] = y * d + oy + (jitter ? jitter(2 * i + 1) : 0);
            setIndex(i, i);
            const up = y < h - 1, right = x < w - 1;
            if (up) beam(i, i + 1, d);
            if (right) beam(i, i + h, d);
            if (up && right) beam(i, i + h + 1, diag);
        }
    }
    mapper.meta.particleCount = P;
    mapper.meta.beamCount = B;
    return { particles: P, beams: B };
}

module.exports.fillLattice = fillLattice;
