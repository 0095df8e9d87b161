'use strict';
/**
 * FIFO promise mutex.  The reference serialises frame / loadBuffers / writeBuffers with one
 * (src/lock.ts:4-19; engineWorker.ts:553,584,632): that is the threading contract of the
 * boundary -- strictly one engine call at a time.
 */
class AsyncLock {
    constructor() {
        this._held = false;
        this._waiters = [];
    }
    acquire() {
        if (!this._held) {
            this._held = true;
            return Promise.resolve();
        }
        return new Promise((resolve) => this._waiters.push(resolve));
    }
    release() {
        const next = this._waiters.shift();
        if (next) next(); // ownership passes straight to the next waiter
        else this._held = false;
    }
    /** run fn under the lock */
    async run(fn) {
        await this.acquire();
        try {
            return await fn();
        } finally {
            this.release();
        }
    }
}
module.exports = { AsyncLock };
