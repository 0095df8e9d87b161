'use strict';
/**
 * Headless façade with the public surface of /root/reference/src/engine.ts: same class name,
 * same async methods (setPhysicsConstants, getPhysicsConstants, saveSnapshot, loadSnapshot,
 * destroy, destroyed, keyboardForce), same message protocol to the worker object.  DOM input
 * listeners, the OffscreenCanvas blit and the Worker thread are dropped (out of scope: SURVEY.md
 * section 2); input arrives through setInput(), and frames are driven by frame() / run().
 */
const { WGPUSoftbodyEngineMessageTypes: MSG } = require('./messages');
const { WGPUSoftbodyEngineWorker } = require('./engineWorker');
const { Vector2D } = require('./engineMapping');

class WGPUSoftbodyEngine {
    /**
     * Reference signature is (canvas, resolution, opts); canvas/resolution are accepted and ignored.
     * `new WGPUSoftbodyEngine(opts)` also works.
     */
    constructor(canvas, resolution, opts) {
        if (resolution === undefined && opts === undefined && canvas && typeof canvas === 'object' && !('getContext' in canvas)) {
            opts = canvas;
            canvas = null;
        }
        this.canvas = canvas || null;
        this.resolution = resolution || 0;
        this.keyboardForce = 1;
        this.fps = 0;
        this.running = true;
        this.listeners = [];
        this.userInput = { appliedForce: Vector2D.zero, rawMousePos: Vector2D.zero, mouseActive: false };
        this.worker = WGPUSoftbodyEngineWorker.create(null, opts, (m) => this._onMessage(m));
    }

    _onMessage(m) {
        if (m.type === MSG.FRAMERATE) this.fps = m.data;
        if (m.type === MSG.DESTROY) this.running = false;
        for (const l of this.listeners.slice()) l(m);
    }
    /** post a message and resolve with the data of the first reply of `responseType` (engine.ts:159-171) */
    _ask(type, responseType, data) {
        const want = responseType === undefined ? type : responseType;
        return new Promise((resolve, reject) => {
            const l = (m) => {
                if (m.type !== want) return;
                this.listeners.splice(this.listeners.indexOf(l), 1);
                resolve(m.data);
            };
            this.listeners.push(l);
            this.worker.onMessage({ type, data }).catch((err) => {
                this.listeners.splice(this.listeners.indexOf(l), 1);
                reject(err);
            });
        });
    }

    async setPhysicsConstants(constants) {
        const plain = Object.assign({}, constants, { gravity: { x: constants.gravity.x, y: constants.gravity.y } });
        await this._ask(MSG.PHYSICS_CONSTANTS, undefined, plain);
    }
    async getPhysicsConstants() { return this._ask(MSG.GET_PHYSICS_CONSTANTS, MSG.PHYSICS_CONSTANTS); }
    async saveSnapshot() { return this._ask(MSG.SNAPSHOT_SAVE); }
    async loadSnapshot(buf) { return this._ask(MSG.SNAPSHOT_LOAD, undefined, buf); }
    corruptBuffers() { return this.worker.onMessage({ type: MSG.CORRUPT_BUFFERS }); }

    /** what the DOM listeners of engine.ts:46-125 would have sent: force in [-1,1]^2 * keyboardForce, mouse in [0,1]^2 */
    async setInput(appliedForce, rawMousePos, mouseActive) {
        this.userInput = { appliedForce, rawMousePos, mouseActive };
        await this._ask(MSG.INPUT, undefined, [appliedForce.toObject(), rawMousePos.toObject(), !!mouseActive]);
    }

    /** one simulated frame (the rAF callback of engineWorker.ts:699-709) */
    async frame() { await this.worker.frame(); }
    /** n frames back to back */
    async run(frames) { for (let i = 0; i < frames && this.running; i++) await this.frame(); }

    destroy() {
        this.running = false;
        return this.worker.onMessage({ type: MSG.DESTROY });
    }
    get destroyed() { return !this.running; }
}

module.exports = { WGPUSoftbodyEngine, WGPUSoftbodyEngineMessageTypes: MSG };
