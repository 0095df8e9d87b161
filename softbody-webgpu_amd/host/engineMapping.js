'use strict';
/**
 * Host-side state mapper: the JavaScript emit of engineMapping (types in index.d.ts).
 *
 * Public surface mirrors /root/reference/src/engineMapping.ts -- Vector2D, Particle, Beam,
 * Metadata, BufferMapper with the same method names and the same bytes in the four
 * ArrayBuffers -- because those buffers ARE the drop-in boundary (SURVEY.md 8(b)).  The
 * implementation is table-driven so one code path serves two record layouts:
 *   layout 1  the reference's: u16 mapping, beam = u16 a, u16 b, 9 x f32 (40 B)
 *   layout 2  wide:            u32 mapping, beam = u32 a, u32 b, 9 x f32 (44 B)
 * ES2019 CommonJS (Node 12 has no `??`/`?.` and there is no TypeScript compiler in the image).
 */

const LE = true;

/** Record geometry per layout (reference values: engineMapping.ts:103,151,239,355). */
const LAYOUTS = {
    1: { id: 1, indexBytes: 2, beamStride: 40, beamFloatBase: 4, maxCount: 65536, IndexArray: Uint16Array },
    2: { id: 2, indexBytes: 4, beamStride: 44, beamFloatBase: 8, maxCount: 0x7fffffff, IndexArray: Uint32Array }
};
const PARTICLE_STRIDE = 24;
const METADATA_BYTES = 112;
/** float slots of a beam record after the endpoints (engineMapping.ts:187-193 + strain/stress) */
const BEAM_FIELD = { length: 0, targetLen: 1, lastLen: 2, spring: 3, damp: 4, yieldStrain: 5, strainLimit: 6, strain: 7, stress: 8 };
/** metadata byte offsets (engineMapping.ts:254-262, compute.wgsl:29-54) */
const MD = { particleVertexCount: 0, particleCount: 4, beamVertexCount: 20, beamCount: 24, maxParticles: 40, maxBeams: 44, constants: 48, userInput: 80 };

function either(value, fallback) {
    return value === undefined || value === null ? fallback : value;
}

class Vector2D {
    constructor(x, y) {
        this.x = x;
        this.y = y;
        this.magnitude = Math.sqrt(x * x + y * y);
        Object.freeze(this);
    }
    translate(dx, dy) { return new Vector2D(this.x + dx, this.y + dy); }
    mult(s) { return new Vector2D(this.x * s, this.y * s); }
    norm() { return this.mult(1 / this.magnitude); }
    negate() { return new Vector2D(-this.x, -this.y); }
    add(o) { return new Vector2D(this.x + o.x, this.y + o.y); }
    sub(o) { return new Vector2D(this.x - o.x, this.y - o.y); }
    dot(o) { return this.x * o.x + this.y * o.y; }
    cross(o) { return this.x * o.y - this.y * o.x; }
    static min(u, v) { return new Vector2D(Math.min(u.x, v.x), Math.min(u.y, v.y)); }
    static max(u, v) { return new Vector2D(Math.max(u.x, v.x), Math.max(u.y, v.y)); }
    static clamp(vec, lo, hi) { return Vector2D.max(lo, Vector2D.min(vec, hi)); }
    /** sign of the 3x3 determinant |1 1 1; P R Q| -- 0 colinear, 1 right turn, -1 left turn */
    static turnDirection(p, q, r) {
        return Math.sign(p.x * (r.y - q.y) + r.x * (q.y - p.y) + q.x * (p.y - r.y));
    }
    toString() { return 'Vector2D<' + this.x + ', ' + this.y + '>'; }
    /** store into any typed array at element offset */
    to(buffer, offset) {
        buffer[offset] = this.x;
        buffer[offset + 1] = this.y;
    }
    static from(buffer, offset) { return new Vector2D(buffer[offset], buffer[offset + 1]); }
    toObject() { return { x: this.x, y: this.y }; }
    static fromObject(obj) { return new Vector2D(obj.x, obj.y); }
}
Vector2D.zero = new Vector2D(0, 0);
Vector2D.i = new Vector2D(1, 0);
Vector2D.j = new Vector2D(0, 1);

function readIndex(view, layout, slot) {
    return layout.indexBytes === 2 ? view.getUint16(slot * 2, LE) : view.getUint32(slot * 4, LE);
}
function writeIndex(view, layout, slot, value) {
    if (layout.indexBytes === 2) view.setUint16(slot * 2, value, LE);
    else view.setUint32(slot * 4, value, LE);
}
/** mapping argument may be an ArrayBuffer, a typed array (the reference passes a Uint16Array) or a DataView */
function mappingView(m) {
    if (m instanceof DataView) return m;
    if (ArrayBuffer.isView(m)) return new DataView(m.buffer, m.byteOffset, m.byteLength);
    return new DataView(m);
}
function layoutOfMapping(m, layout) {
    if (layout) return layout;
    return ArrayBuffer.isView(m) && m.BYTES_PER_ELEMENT === 4 ? LAYOUTS[2] : LAYOUTS[1];
}

class Particle {
    /** ids are transient: writeState() renumbers them (engineMapping.ts:105) */
    constructor(id, position, velocity, acceleration) {
        this.id = id;
        this.position = either(position, Vector2D.zero);
        this.velocity = either(velocity, Vector2D.zero);
        this.acceleration = either(acceleration, Vector2D.zero);
    }
    /** record = [px, py, vx, vy, ax, ay] f32 at index*24; mapping[id] = index */
    to(pBuf, mBuf, index, layout) {
        const lay = layoutOfMapping(mBuf, layout);
        writeIndex(mappingView(mBuf), lay, this.id, index);
        const rec = new DataView(pBuf, index * PARTICLE_STRIDE, PARTICLE_STRIDE);
        const parts = [this.position, this.velocity, this.acceleration];
        for (let k = 0; k < 3; k++) {
            rec.setFloat32(k * 8, parts[k].x, LE);
            rec.setFloat32(k * 8 + 4, parts[k].y, LE);
        }
    }
    static from(pBuf, mBuf, id, layout) {
        const lay = layoutOfMapping(mBuf, layout);
        const index = readIndex(mappingView(mBuf), lay, id);
        const rec = new DataView(pBuf, index * PARTICLE_STRIDE, PARTICLE_STRIDE);
        const vec = (k) => new Vector2D(rec.getFloat32(k * 8, LE), rec.getFloat32(k * 8 + 4, LE));
        return new Particle(id, vec(0), vec(1), vec(2));
    }
}
Particle.stride = PARTICLE_STRIDE;

function idOf(ref) {
    return typeof ref === 'number' ? ref : ref.id;
}

class Beam {
    /**
     * a, b: particle ids (or Particle objects).  length = rest length; targetLen drifts with plastic
     * yield; lastLen = length in the previous substep (damping input); spring/damp constants;
     * yieldStrain / strainLimit as fractions of `length` (engineMapping.ts:157-163).
     */
    constructor(id, a, b, length, spring, damp, yieldStrain, strainLimit, targetLen, lastLen) {
        this.id = id;
        this.a = a;
        this.b = b;
        this.length = length;
        this.targetLen = either(targetLen, length);
        this.lastLen = either(lastLen, length);
        this.spring = spring;
        this.damp = damp;
        this.yieldStrain = yieldStrain;
        this.strainLimit = strainLimit;
    }
    /** endpoints are stored as particle DATA INDICES looked up through the mapping (engineMapping.ts:180-186) */
    to(bBuf, mBuf, index, mBufOffset, layout) {
        const lay = layoutOfMapping(mBuf, layout);
        const map = mappingView(mBuf);
        writeIndex(map, lay, mBufOffset + this.id, index);
        const rec = new DataView(bBuf, index * lay.beamStride, lay.beamStride);
        writeIndex(rec, lay, 0, readIndex(map, lay, idOf(this.a)));
        writeIndex(rec, lay, 1, readIndex(map, lay, idOf(this.b)));
        const f = lay.beamFloatBase;
        rec.setFloat32(f + 4 * BEAM_FIELD.length, this.length, LE);
        rec.setFloat32(f + 4 * BEAM_FIELD.targetLen, this.targetLen, LE);
        rec.setFloat32(f + 4 * BEAM_FIELD.lastLen, this.lastLen, LE);
        rec.setFloat32(f + 4 * BEAM_FIELD.spring, this.spring, LE);
        rec.setFloat32(f + 4 * BEAM_FIELD.damp, this.damp, LE);
        rec.setFloat32(f + 4 * BEAM_FIELD.yieldStrain, this.yieldStrain, LE);
        rec.setFloat32(f + 4 * BEAM_FIELD.strainLimit, this.strainLimit, LE);
        // strain / stress are device outputs; the host never writes them (engineMapping.ts:187-193)
    }
    /**
     * idLookup(index) -> particle id.  The reference searches the whole mapping with indexOf
     * (engineMapping.ts:201-202); callers here pass an inverse table built once per loadState.
     */
    static from(bBuf, mBuf, id, mBufOffset, layout, idLookup) {
        const lay = layoutOfMapping(mBuf, layout);
        const map = mappingView(mBuf);
        const index = readIndex(map, lay, mBufOffset + id);
        const rec = new DataView(bBuf, index * lay.beamStride, lay.beamStride);
        const lookup = idLookup || ((dataIndex) => {
            for (let s = 0; s < map.byteLength / lay.indexBytes; s++) if (readIndex(map, lay, s) === dataIndex) return s;
            return -1;
        });
        const f = lay.beamFloatBase;
        const g = (name) => rec.getFloat32(f + 4 * BEAM_FIELD[name], LE);
        return new Beam(id, lookup(readIndex(rec, lay, 0)), lookup(readIndex(rec, lay, 1)), g('length'), g('spring'),
            g('damp'), g('yieldStrain'), g('strainLimit'), g('targetLen'), g('lastLen'));
    }
    /** device outputs of the last substep (render inputs in the reference, render.wgsl:82) */
    static readStrainStress(bBuf, index, layout) {
        const lay = layout || LAYOUTS[1];
        const rec = new DataView(bBuf, index * lay.beamStride, lay.beamStride);
        return { strain: rec.getFloat32(lay.beamFloatBase + 28, LE), stress: rec.getFloat32(lay.beamFloatBase + 32, LE) };
    }
}
Beam.stride = 40;
Beam.strideOf = (layoutId) => LAYOUTS[layoutId].beamStride;

/**
 * 112-byte metadata block: 2 indirect-draw argument blocks (instance counts = live particle/beam
 * counts), capacities, 8 physics constants, 8 words of user input (engineMapping.ts:211-262).
 */
class Metadata {
    constructor(buf, maxParticles, maxBeams) {
        this.buffer = buf;
        this._v = new DataView(buf, 0, METADATA_BYTES);
        this._v.setUint32(MD.particleVertexCount, 3, LE);
        this._v.setUint32(MD.beamVertexCount, 2, LE);
        this._v.setUint32(MD.maxParticles, maxParticles, LE);
        this._v.setUint32(MD.maxBeams, maxBeams, LE);
        this.userStrength = 1;
        this.setPhysicsConstants(Metadata.defaultConstants());
    }
    static defaultConstants() {
        return { gravity: new Vector2D(0, -0.5), borderElasticity: 0.5, borderFriction: 0.2, elasticity: 0.5,
            friction: 0.1, dragCoeff: 0.001, dragExp: 2 };
    }
    get particleCount() { return this._v.getUint32(MD.particleCount, LE); }
    set particleCount(c) { this._v.setUint32(MD.particleCount, c, LE); }
    get beamCount() { return this._v.getUint32(MD.beamCount, LE); }
    set beamCount(c) { this._v.setUint32(MD.beamCount, c, LE); }
    setPhysicsConstants(c) {
        const vals = [c.gravity.x, c.gravity.y, c.borderElasticity, c.borderFriction, c.elasticity, c.friction,
            c.dragCoeff, c.dragExp];
        vals.forEach((val, k) => this._v.setFloat32(MD.constants + 4 * k, val, LE));
    }
    getPhysicsConstants() {
        const g = (k) => this._v.getFloat32(MD.constants + 4 * k, LE);
        return { gravity: new Vector2D(g(0), g(1)), borderElasticity: g(2), borderFriction: g(3), elasticity: g(4),
            friction: g(5), dragCoeff: g(6), dragExp: g(7) };
    }
    /** the 8 constants as the Float32Array the C ABI takes (sb_set_physics_constants) */
    physicsConstantsArray() { return new Float32Array(this.buffer.slice(MD.constants, MD.constants + 32)); }
    get userStrength() { return this._v.getFloat32(MD.userInput, LE); }
    set userStrength(s) { this._v.setFloat32(MD.userInput, s, LE); }
    setUserInput(appliedForce, mousePos, mouseVel, mouseActive) {
        this._v.setUint32(MD.userInput + 4, mouseActive ? 1 : 0, LE);
        [mousePos, mouseVel, appliedForce].forEach((vec, k) => {
            this._v.setFloat32(MD.userInput + 8 + 8 * k, vec.x, LE);
            this._v.setFloat32(MD.userInput + 12 + 8 * k, vec.y, LE);
        });
    }
    /** the 32 bytes at offset 80 that the reference uploads every frame */
    userInputBytes() { return new Uint8Array(this.buffer, MD.userInput, 32); }
    /**
     * Reference signature (queue, buffer) wrote through WebGPU (engineMapping.ts:323-325).  Here `queue`
     * is anything with writeUserInput(bytes32): the engine worker passes the native engine.
     */
    writeUserInput(queue, buffer) {
        queue.writeUserInput(this.userInputBytes(), buffer);
    }
}
Metadata.byteLength = METADATA_BYTES;

const SNAPSHOT_V2_MAGIC = 0x32574253; // 'SBW2'

/**
 * Owns the four host ArrayBuffers (metadata, particleData, beamData, mapping) plus an editable
 * object model of the scene; writeState()/loadState() move between the two
 * (engineMapping.ts:341-528).  Elements never move inside the data buffers; the mapping maps
 * slot/id -> data index, particle slots first then beam slots at offset maxParticles.
 */
class BufferMapper {
    /**
     * @param maxByteLength cap on any one buffer (the reference passes maxStorageBufferBindingSize)
     * @param opts optional {layout: 1|2, maxParticles, maxBeams}
     */
    constructor(maxByteLength, opts) {
        const o = opts || {};
        this.layout = LAYOUTS[either(o.layout, 1)];
        if (!this.layout) throw new RangeError('unknown layout ' + o.layout);
        const byCapacity = (stride) => Math.floor(Math.min(this.layout.maxCount, maxByteLength / this.layout.indexBytes / 2,
            Math.floor(maxByteLength / stride)));
        // the reference sizes maxBeams by the PARTICLE stride too (engineMapping.ts:363); kept for layout 1
        this.maxParticles = either(o.maxParticles, byCapacity(PARTICLE_STRIDE));
        this.maxBeams = either(o.maxBeams, byCapacity(this.layout.id === 1 ? PARTICLE_STRIDE : this.layout.beamStride));
        if (this.maxParticles > this.layout.maxCount || this.maxBeams > this.layout.maxCount)
            throw new RangeError('layout ' + this.layout.id + ' holds at most ' + this.layout.maxCount + ' elements');
        this.metadata = new ArrayBuffer(METADATA_BYTES);
        this.particleData = new ArrayBuffer(PARTICLE_STRIDE * this.maxParticles);
        this.beamData = new ArrayBuffer(this.layout.beamStride * this.maxBeams);
        this.mapping = new ArrayBuffer(this.layout.indexBytes * (this.maxParticles + this.maxBeams));
        this.meta = new Metadata(this.metadata, this.maxParticles, this.maxBeams);
        this._map = new DataView(this.mapping);
        this._particles = new Map();
        this._beams = new Map();
        this._beamsOf = new Map();
    }

    // ---- snapshots (checkpoint/resume; engineMapping.ts:377-430)
    /**
     * Layout 1 writes the reference's format byte for byte, including its u16 size fields, which
     * wrap above 2730 particles / 1638 beams exactly like the reference's do (SURVEY.md section 5).
     * Layout 2 writes 'SBW2' + five u32 sizes, then the same sections.
     */
    createSnapshotBuffer() {
        this.writeState();
        const P = this.meta.particleCount, B = this.meta.beamCount, ib = this.layout.indexBytes;
        const sizes = [ib * P, PARTICLE_STRIDE * P, ib * B, this.layout.beamStride * B, 32];
        const head = this.layout.id === 1 ? 12 : 24;
        const out = new ArrayBuffer(head + sizes[4] + sizes[0] + sizes[1] + sizes[2] + sizes[3]);
        const hv = new DataView(out);
        if (this.layout.id === 1) sizes.forEach((s, k) => hv.setUint16(2 * k, s & 0xffff, LE));
        else {
            hv.setUint32(0, SNAPSHOT_V2_MAGIC, LE);
            sizes.forEach((s, k) => hv.setUint32(4 + 4 * k, s, LE));
        }
        const bytes = new Uint8Array(out);
        let at = head;
        const put = (src, off, len) => { bytes.set(new Uint8Array(src, off, len), at); at += len; };
        put(this.metadata, MD.constants, 32);
        put(this.mapping, 0, sizes[0]);
        put(this.particleData, 0, sizes[1]);
        put(this.mapping, ib * this.maxParticles, sizes[2]);
        put(this.beamData, 0, sizes[3]);
        return out;
    }
    /** @returns false when the snapshot cannot fit (engineMapping.ts:418) */
    loadSnapshotbuffer(buf) {
        const hv = new DataView(buf);
        let sizes, head;
        if (this.layout.id === 1) {
            sizes = [0, 1, 2, 3, 4].map((k) => hv.getUint16(2 * k, LE));
            head = 12;
            // the reference compares BYTE sizes with element capacities here; kept
            if (sizes[0] > this.maxParticles || sizes[2] > this.maxBeams) return false;
        } else {
            if (buf.byteLength < 24 || hv.getUint32(0, LE) !== SNAPSHOT_V2_MAGIC) return false;
            sizes = [0, 1, 2, 3, 4].map((k) => hv.getUint32(4 + 4 * k, LE));
            head = 24;
            if (sizes[0] > this.layout.indexBytes * this.maxParticles || sizes[2] > this.layout.indexBytes * this.maxBeams) return false;
        }
        const src = new Uint8Array(buf);
        let at = head;
        const take = (dst, off, len) => { new Uint8Array(dst).set(src.subarray(at, at + len), off); at += len; };
        take(this.metadata, MD.constants, sizes[4]);
        take(this.mapping, 0, sizes[0]);
        take(this.particleData, 0, sizes[1]);
        take(this.mapping, this.layout.indexBytes * this.maxParticles, sizes[2]);
        take(this.beamData, 0, sizes[3]);
        this.meta.particleCount = sizes[0] / this.layout.indexBytes;
        this.meta.beamCount = sizes[2] / this.layout.indexBytes;
        this.loadState();
        return true;
    }

    // ---- edit API (engineMapping.ts:432-495)
    addParticle(p) {
        if (this._particles.size === this.maxParticles || this._particles.has(p.id)) return false;
        this._particles.set(p.id, p);
        return true;
    }
    addBeam(b) {
        if (this._beams.size === this.maxBeams || this._beams.has(b.id)) return false;
        this._beams.set(b.id, b);
        for (const end of [idOf(b.a), idOf(b.b)]) {
            if (!this._beamsOf.has(end)) this._beamsOf.set(end, new Set());
            this._beamsOf.get(end).add(b);
        }
        return true;
    }
    removeParticle(p) { return this._particles.delete(idOf(p)); }
    removeBeam(b) {
        const beam = typeof b === 'number' ? this._beams.get(b) : b;
        if (beam === undefined || !this._beams.delete(beam.id)) return false;
        for (const end of [idOf(beam.a), idOf(beam.b)]) {
            const set = this._beamsOf.get(end);
            if (set) set.delete(beam);
        }
        return true;
    }
    findParticle(id) { return this._particles.has(id) ? this._particles.get(id) : null; }
    findBeam(id) { return this._beams.has(id) ? this._beams.get(id) : null; }
    getConnectedBeams(p) { return new Set(this._beamsOf.get(idOf(p))); }
    _firstFree(map, cap) {
        if (map.size === cap) return -1;
        for (let id = 0; id < cap; id++) if (!map.has(id)) return id;
        return -1;
    }
    get firstEmptyParticleId() { return this._firstFree(this._particles, this.maxParticles); }
    get firstEmptyBeamId() { return this._firstFree(this._beams, this.maxBeams); }
    get particleSet() { return new Set(this._particles.values()); }
    get beamSet() { return new Set(this._beams.values()); }
    clear() {
        this._particles.clear();
        this._beams.clear();
        this._beamsOf.clear();
    }

    /**
     * Object model -> buffers.  Elements are renumbered densely in insertion order: the i-th
     * particle gets id i AND data index i, so the mapping is the identity afterwards
     * (engineMapping.ts:500-517).
     */
    writeState() {
        this.meta.particleCount = this._particles.size;
        this.meta.beamCount = this._beams.size;
        const newId = new Map();
        let n = 0;
        for (const p of this._particles.values()) {
            newId.set(p.id, n);
            new Particle(n, p.position, p.velocity, p.acceleration).to(this.particleData, this._map, n, this.layout);
            n++;
        }
        const renumber = (end) => {
            const id = idOf(end);
            return newId.has(id) ? newId.get(id) : end;
        };
        n = 0;
        for (const b of this._beams.values()) {
            new Beam(n, renumber(b.a), renumber(b.b), b.length, b.spring, b.damp, b.yieldStrain, b.strainLimit,
                b.targetLen, b.lastLen).to(this.beamData, this._map, n, this.maxParticles, this.layout);
            n++;
        }
    }
    /** buffers -> object model (engineMapping.ts:521-527) */
    loadState() {
        this.clear();
        const P = this.meta.particleCount, B = this.meta.beamCount;
        const idOfIndex = new Map();
        for (let id = 0; id < P; id++) {
            const idx = readIndex(this._map, this.layout, id);
            if (!idOfIndex.has(idx)) idOfIndex.set(idx, id);
            this.addParticle(Particle.from(this.particleData, this._map, id, this.layout));
        }
        const lookup = (dataIndex) => (idOfIndex.has(dataIndex) ? idOfIndex.get(dataIndex) : -1);
        for (let id = 0; id < B; id++) this.addBeam(Beam.from(this.beamData, this._map, id, this.maxParticles, this.layout, lookup));
    }
}

module.exports = { Vector2D, Particle, Beam, Metadata, BufferMapper, LAYOUTS };
