'use strict';
/**
 * Loads the N-API addon (csrc/softbody_napi.node) and points it at the HIP engine
 * (csrc/libsoftbody_hip.so).  There is no JavaScript or CPU fallback: if either file is missing
 * this throws, and if the machine has no MI355X the first create() throws.
 */
const path = require('path');
const fs = require('fs');

const CSRC = path.resolve(__dirname, '..', 'csrc');
let addon = null;

function native() {
    if (addon) return addon;
    const node = process.env.SOFTBODY_NAPI || path.join(CSRC, 'softbody_napi.node');
    const lib = process.env.SOFTBODY_HIP_LIB || path.join(CSRC, 'libsoftbody_hip.so');
    for (const f of [node, lib]) {
        if (!fs.existsSync(f)) throw new Error('softbody: ' + f + ' is missing; build it with `make -C ' + CSRC + '`');
    }
    const a = require(node);
    const abi = a.load(lib);
    if (abi !== 1) throw new Error('softbody: unsupported C ABI version ' + abi);
    addon = a;
    return a;
}

const COLLIDE = { OFF: 0, ALLPAIRS: 1, GRID: 2 };
const PATH = { AUTO: 0, ATOMIC: 1, TILED: 2 };

module.exports = { native, COLLIDE, PATH };
