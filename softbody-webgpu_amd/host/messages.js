'use strict';
/** Message enum of the engine <-> worker protocol (src/engine.ts:3-14); numeric values match the TS enum. */
const WGPUSoftbodyEngineMessageTypes = Object.freeze({
    INIT: 0, DESTROY: 1, PHYSICS_CONSTANTS: 2, GET_PHYSICS_CONSTANTS: 3, INPUT: 4, VISIBILITY_CHANGE: 5,
    SNAPSHOT_SAVE: 6, SNAPSHOT_LOAD: 7, FRAMERATE: 8, CORRUPT_BUFFERS: 9
});
module.exports = { WGPUSoftbodyEngineMessageTypes };
