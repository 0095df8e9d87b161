'use strict';
/**
 * Host driver of the native engine: the counterpart of /root/reference/src/engineWorker.ts
 * without the canvas / render pass / Worker thread.  Same state, same message protocol
 * (engineWorker.ts:490-545), same three locked operations:
 *   writeBuffers()  engineWorker.ts:580-597  ->  sb_write_buffers
 *   loadBuffers()   engineWorker.ts:548-579  ->  sb_load_buffers
 *   frame()         engineWorker.ts:626-695  ->  sb_write_user_input + sb_frame
 */
const { WGPUSoftbodyEngineMessageTypes: MSG } = require('./messages');
const { BufferMapper, Vector2D } = require('./engineMapping');
const { AsyncLock } = require('./lock');
const { native, COLLIDE, PATH } = require('./native');

class WGPUSoftbodyEngineWorker {
    static create(canvas, opts, post) {
        WGPUSoftbodyEngineWorker.sInstance = new WGPUSoftbodyEngineWorker(canvas, opts, post);
        return WGPUSoftbodyEngineWorker.sInstance;
    }
    static instance() { return WGPUSoftbodyEngineWorker.sInstance; }

    /**
     * @param canvas ignored (kept for signature compatibility; there is no render pass)
     * @param opts {particleRadius, subticks} as in the reference, plus {boundsSize, layout,
     *        maxParticles, maxBeams, collisionMode, path, tileParticles, device, gridSkin, blockSubsteps}
     * @param post function(message) receiving {type, data} replies
     */
    constructor(canvas, opts, post) {
        const o = opts || {};
        this.boundsSize = o.boundsSize !== undefined ? o.boundsSize : 1000;        // engineWorker.ts:39
        this.particleRadius = o.particleRadius !== undefined ? o.particleRadius : 10; // :40,89
        this.subticks = Math.ceil((o.subticks !== undefined ? o.subticks : 64) / 2) * 2; // :41,90
        this.layout = o.layout !== undefined ? o.layout : 1;
        this.lock = new AsyncLock();
        this.post = post || (() => {});
        this.addon = native();
        // capacity: the reference asks the adapter (engineWorker.ts:118-122); here the caller may size it
        const cap = o.maxByteLength !== undefined ? o.maxByteLength : (this.layout === 1 ? 1 << 27 : 1 << 31);
        this.bufferMapper = new BufferMapper(cap, { layout: this.layout, maxParticles: o.maxParticles, maxBeams: o.maxBeams });
        this.handle = this.addon.create({
            boundsSize: this.boundsSize, particleRadius: this.particleRadius, subticks: this.subticks,
            maxParticles: this.bufferMapper.maxParticles, maxBeams: this.bufferMapper.maxBeams, layout: this.layout,
            collisionMode: o.collisionMode !== undefined ? o.collisionMode : COLLIDE.GRID, // same bits as ALLPAIRS
            path: o.path !== undefined ? o.path : PATH.AUTO, tileParticles: o.tileParticles || 0, device: o.device || 0,
            gridSkin: o.gridSkin || 0, blockSubsteps: o.blockSubsteps || 0
        });
        this.running = true;
        this.visible = true;
        this.uploaded = false;
        this.frameTimes = [];
        this.userInput = { appliedForce: Vector2D.zero, mousePos: Vector2D.zero, lastMouse: Vector2D.zero,
            mouseActive: false, lastFrame: Date.now() };
    }

    get currentFps() { return this.frameTimes.length; }

    _buffers() {
        const m = this.bufferMapper;
        return [m.metadata, m.mapping, m.particleData, m.beamData];
    }

    /** device -> host ArrayBuffers of the BufferMapper (engineWorker.ts:548-579) */
    async loadBuffers() {
        await this.lock.run(() => {
            if (this.uploaded) this.addon.loadBuffers(this.handle, ...this._buffers());
        });
    }
    /** host ArrayBuffers -> device; also resets accumulators and the second particle buffer (:580-597) */
    async writeBuffers() {
        await this.lock.run(() => {
            this.addon.writeBuffers(this.handle, ...this._buffers());
            this.uploaded = true;
        });
    }

    /** one frame = user input upload + `subticks` substeps + one delete pass (:626-695, minus rendering) */
    async frame() {
        await this.lock.run(() => {
            if (!this.uploaded) return;
            const now = Date.now();
            const ui = this.userInput;
            const meta = this.bufferMapper.meta;
            meta.setUserInput(
                ui.appliedForce,
                ui.mousePos.mult(this.boundsSize),
                ui.mousePos.sub(ui.lastMouse).mult(this.currentFps * (now - ui.lastFrame) / 1000 * this.boundsSize),
                ui.mouseActive);
            meta.writeUserInput({ writeUserInput: (bytes) => this.addon.writeUserInput(this.handle, bytes) }, null);
            ui.lastMouse = ui.mousePos;
            ui.lastFrame = now;
            this.addon.frame(this.handle);
            this.addon.sync(this.handle); // device.queue.onSubmittedWorkDone(), :687
        });
        const t = Date.now();
        this.frameTimes.push(t);
        while (this.frameTimes[0] + 1000 < t) this.frameTimes.shift();
        this.post({ type: MSG.FRAMERATE, data: this.currentFps });
    }

    /** benchmark granularity: n substeps, no delete pass; returns device milliseconds */
    async step(n) {
        return this.lock.run(() => this.addon.stepTimed(this.handle, n));
    }

    /** the message protocol of engineWorker.ts:490-545 */
    async onMessage(msg) {
        const mapper = this.bufferMapper;
        switch (msg.type) {
            case MSG.DESTROY:
                await this.destroy();
                break;
            case MSG.PHYSICS_CONSTANTS: {
                const c = Object.assign({}, msg.data, { gravity: Vector2D.fromObject(msg.data.gravity) });
                mapper.meta.setPhysicsConstants(c);
                // the reference reads back and re-uploads the whole scene here (:499-504); the C ABI
                // updates the 32 bytes in place, which leaves the same device state
                await this.lock.run(() => {
                    if (this.uploaded) this.addon.setPhysicsConstants(this.handle, mapper.meta.physicsConstantsArray());
                });
                this.post({ type: MSG.PHYSICS_CONSTANTS, data: mapper.meta.getPhysicsConstants() });
                break;
            }
            case MSG.GET_PHYSICS_CONSTANTS:
                await this.loadBuffers();
                this.post({ type: MSG.PHYSICS_CONSTANTS, data: mapper.meta.getPhysicsConstants() });
                break;
            case MSG.INPUT:
                this.userInput.appliedForce = Vector2D.fromObject(msg.data[0]);
                this.userInput.mousePos = Vector2D.fromObject(msg.data[1]);
                this.userInput.mouseActive = !!msg.data[2];
                this.post({ type: MSG.INPUT });
                break;
            case MSG.VISIBILITY_CHANGE:
                this.visible = !msg.data;
                break;
            case MSG.SNAPSHOT_SAVE:
                await this.loadBuffers();
                mapper.loadState();
                this.post({ type: MSG.SNAPSHOT_SAVE, data: mapper.createSnapshotBuffer() });
                break;
            case MSG.SNAPSHOT_LOAD: {
                const ok = mapper.loadSnapshotbuffer(msg.data);
                if (ok) await this.writeBuffers();
                this.post({ type: MSG.SNAPSHOT_LOAD, data: ok });
                break;
            }
            case MSG.CORRUPT_BUFFERS:
                // fault-injection toy of the web app (engineWorker.ts:599-617); not part of the physics path
                throw new Error('CORRUPT_BUFFERS is not supported by the native engine');
            default:
                break;
        }
    }

    async destroy() {
        if (!this.running) return;
        this.running = false;
        await this.lock.run(() => this.addon.destroy(this.handle));
        this.handle = null;
        this.post({ type: MSG.DESTROY });
        if (WGPUSoftbodyEngineWorker.sInstance === this) WGPUSoftbodyEngineWorker.sInstance = null;
    }
}
WGPUSoftbodyEngineWorker.sInstance = null;

module.exports = { WGPUSoftbodyEngineWorker };
