// Type surface of the host mirror (TypeScript declarations for the ES2019 JavaScript beside them).
// Names and member signatures follow /root/reference/src/engine.ts:3-29,187-238 and
// /root/reference/src/engineMapping.ts; DOM/canvas members are dropped (headless), and a few
// members are added for the wide layout and for driving frames explicitly.

export type TypedArray = Uint8Array | Int8Array | Uint16Array | Int16Array | Uint32Array | Int32Array | Float32Array | Float64Array;

export class Vector2D {
    readonly x: number;
    readonly y: number;
    readonly magnitude: number;
    constructor(x: number, y: number);
    translate(x: number, y: number): Vector2D;
    mult(s: number): Vector2D;
    norm(): Vector2D;
    negate(): Vector2D;
    add(o: Vector2D): Vector2D;
    sub(o: Vector2D): Vector2D;
    dot(o: Vector2D): number;
    cross(o: Vector2D): number;
    static min(u: Vector2D, v: Vector2D): Vector2D;
    static max(u: Vector2D, v: Vector2D): Vector2D;
    static clamp(vec: Vector2D, min: Vector2D, max: Vector2D): Vector2D;
    static turnDirection(p: Vector2D, q: Vector2D, r: Vector2D): number;
    toString(): string;
    static readonly zero: Vector2D;
    static readonly i: Vector2D;
    static readonly j: Vector2D;
    to(buffer: TypedArray, offset: number): void;
    static from(buffer: TypedArray, offset: number): Vector2D;
    toObject(): { x: number, y: number };
    static fromObject(obj: { x: number, y: number }): Vector2D;
}

export interface LayoutInfo {
    readonly id: 1 | 2;
    readonly indexBytes: 2 | 4;
    readonly beamStride: 40 | 44;
    readonly beamFloatBase: 4 | 8;
    readonly maxCount: number;
}
export const LAYOUTS: { readonly 1: LayoutInfo, readonly 2: LayoutInfo };

type MappingArg = ArrayBuffer | Uint16Array | Uint32Array | DataView;

export class Particle {
    static readonly stride: 24;
    readonly id: number;
    position: Vector2D;
    velocity: Vector2D;
    acceleration: Vector2D;
    constructor(id: number, position?: Vector2D, velocity?: Vector2D, acceleration?: Vector2D);
    to(pBuf: ArrayBuffer, mBuf: MappingArg, index: number, layout?: LayoutInfo): void;
    static from(pBuf: ArrayBuffer, mBuf: MappingArg, id: number, layout?: LayoutInfo): Particle;
}

export class Beam {
    static readonly stride: 40;
    static strideOf(layoutId: 1 | 2): number;
    readonly id: number;
    readonly a: number | Particle;
    readonly b: number | Particle;
    length: number;
    targetLen: number;
    lastLen: number;
    spring: number;
    damp: number;
    yieldStrain: number;
    strainLimit: number;
    constructor(id: number, a: number | Particle, b: number | Particle, length: number, spring: number, damp: number,
        yieldStrain: number, strainLimit: number, targetLen?: number, lastLen?: number);
    to(bBuf: ArrayBuffer, mBuf: MappingArg, index: number, mBufOffset: number, layout?: LayoutInfo): void;
    static from(bBuf: ArrayBuffer, mBuf: MappingArg, id: number, mBufOffset: number, layout?: LayoutInfo,
        idLookup?: (dataIndex: number) => number): Beam;
    static readStrainStress(bBuf: ArrayBuffer, index: number, layout?: LayoutInfo): { strain: number, stress: number };
}

export type WGPUSoftbodyEnginePhysicsConstants = {
    readonly gravity: Vector2D
    readonly borderElasticity: number
    readonly borderFriction: number
    readonly elasticity: number
    readonly friction: number
    readonly dragCoeff: number
    readonly dragExp: number
};

export class Metadata {
    static readonly byteLength: 112;
    readonly buffer: ArrayBuffer;
    constructor(buf: ArrayBuffer, maxParticles: number, maxBeams: number);
    static defaultConstants(): WGPUSoftbodyEnginePhysicsConstants;
    particleCount: number;
    beamCount: number;
    setPhysicsConstants(constants: WGPUSoftbodyEnginePhysicsConstants): void;
    getPhysicsConstants(): WGPUSoftbodyEnginePhysicsConstants;
    physicsConstantsArray(): Float32Array;
    userStrength: number;
    setUserInput(appliedForce: Vector2D, mousePos: Vector2D, mouseVel: Vector2D, mouseActive: boolean): void;
    userInputBytes(): Uint8Array;
    writeUserInput(queue: { writeUserInput(bytes32: Uint8Array, buffer?: unknown): void }, buffer?: unknown): void;
}

export class BufferMapper {
    readonly metadata: ArrayBuffer;
    readonly particleData: ArrayBuffer;
    readonly beamData: ArrayBuffer;
    readonly mapping: ArrayBuffer;
    readonly meta: Metadata;
    readonly maxParticles: number;
    readonly maxBeams: number;
    readonly layout: LayoutInfo;
    constructor(maxByteLength: number, opts?: { layout?: 1 | 2, maxParticles?: number, maxBeams?: number });
    createSnapshotBuffer(): ArrayBuffer;
    loadSnapshotbuffer(buf: ArrayBuffer): boolean;
    addParticle(p: Particle): boolean;
    addBeam(b: Beam): boolean;
    removeParticle(p: Particle | number): boolean;
    removeBeam(b: Beam | number): boolean;
    findParticle(id: number): Particle | null;
    findBeam(id: number): Beam | null;
    getConnectedBeams(p: Particle | number): Set<Beam>;
    readonly firstEmptyParticleId: number;
    readonly firstEmptyBeamId: number;
    readonly particleSet: Set<Particle>;
    readonly beamSet: Set<Beam>;
    clear(): void;
    writeState(): void;
    loadState(): void;
}

export enum WGPUSoftbodyEngineMessageTypes {
    INIT, DESTROY, PHYSICS_CONSTANTS, GET_PHYSICS_CONSTANTS, INPUT, VISIBILITY_CHANGE, SNAPSHOT_SAVE, SNAPSHOT_LOAD,
    FRAMERATE, CORRUPT_BUFFERS
}

export type WGPUSoftbodyEngineOptions = {
    readonly particleRadius: number
    readonly subticks: number
};

/** options the native engine adds to the reference's two */
export type NativeEngineOptions = Partial<WGPUSoftbodyEngineOptions> & {
    readonly boundsSize?: number        // fixed 1000 in the reference (engineWorker.ts:39)
    readonly layout?: 1 | 2
    readonly maxParticles?: number
    readonly maxBeams?: number
    readonly maxByteLength?: number
    readonly collisionMode?: 0 | 1 | 2  // COLLIDE.OFF | ALLPAIRS | GRID (default GRID: the bits of ALLPAIRS)
    readonly path?: 0 | 1 | 2           // PATH.AUTO | ATOMIC | TILED
    readonly tileParticles?: number
    readonly device?: number
    readonly gridSkin?: number          // spatial-hash reuse margin; 0/undefined = adaptive (0.4 r .. 1.6 r), > 0 = fixed
    readonly blockSubsteps?: number     // collisions off: substeps per launch (0/undefined = 6, 1 = one launch per substep)
};

export class WGPUSoftbodyEngine {
    readonly resolution: number;
    readonly canvas: unknown | null;
    keyboardForce: number;
    constructor(canvas: unknown | null, resolution?: number, opts?: NativeEngineOptions);
    constructor(opts?: NativeEngineOptions);
    setPhysicsConstants(constants: WGPUSoftbodyEnginePhysicsConstants): Promise<void>;
    getPhysicsConstants(): Promise<WGPUSoftbodyEnginePhysicsConstants>;
    saveSnapshot(): Promise<ArrayBuffer>;
    loadSnapshot(buf: ArrayBuffer): Promise<boolean>;
    corruptBuffers(): Promise<void>;
    setInput(appliedForce: Vector2D, rawMousePos: Vector2D, mouseActive: boolean): Promise<void>;
    frame(): Promise<void>;
    run(frames: number): Promise<void>;
    destroy(): Promise<void>;
    readonly destroyed: boolean;
}

export class WGPUSoftbodyEngineWorker {
    static create(canvas: unknown | null, opts?: NativeEngineOptions, post?: (m: { type: number, data?: unknown }) => void): WGPUSoftbodyEngineWorker;
    static instance(): WGPUSoftbodyEngineWorker | null;
    readonly boundsSize: number;
    readonly particleRadius: number;
    readonly subticks: number;
    readonly bufferMapper: BufferMapper;
    readonly currentFps: number;
    loadBuffers(): Promise<void>;
    writeBuffers(): Promise<void>;
    frame(): Promise<void>;
    step(nSubsteps: number): Promise<number>;
    onMessage(msg: { type: WGPUSoftbodyEngineMessageTypes, data?: unknown }): Promise<void>;
    destroy(): Promise<void>;
}

export class AsyncLock {
    acquire(): Promise<void>;
    release(): void;
    run<T>(fn: () => T | Promise<T>): Promise<T>;
}

export function addRectangle(mapper: BufferMapper, ids: { particleId: number, beamId: number }, ox: number, oy: number,
    d: number, w: number, h: number, spring: number, damp: number, yieldStrain: number, strainLimit: number,
    antiDiagonal?: boolean): { particleId: number, beamId: number };
export function defaultScene(mapper: BufferMapper): BufferMapper;
/** fills the mapper's ArrayBuffers directly (no object model): w x h lattice, ~3 beams per particle */
export function fillLattice(mapper: BufferMapper, ox: number, oy: number, d: number, w: number, h: number, spring: number,
    damp: number, yieldStrain: number, strainLimit: number, jitter?: ((k: number) => number) | null): { particles: number, beams: number };

export const COLLIDE: { readonly OFF: 0, readonly ALLPAIRS: 1, readonly GRID: 2 };
export const PATH: { readonly AUTO: 0, readonly ATOMIC: 1, readonly TILED: 2 };
/** the raw N-API addon (csrc/sb_napi.c); throws if the addon or the HIP library is missing */
export function native(): Record<string, (...args: unknown[]) => unknown>;

/** headless software renderer of render.wgsl's picture (debugging only): binary PPM (P6) */
export function renderPPM(mapper: BufferMapper, opts?: { boundsSize?: number, particleRadius?: number, resolution?: number }): Buffer;

/** multi-GPU sharding from the Node host (host/halo.js; C library sb_partition_*, sb_halo_*, sb_peer_*) */
export type HaloPeer = { readonly rank: number, readonly ghostP: Uint32Array, readonly sendP: Uint32Array,
    readonly ghostB: Uint32Array, readonly sendB: Uint32Array };
export class HaloPlan {
    readonly rank: number; readonly world: number; readonly depth: number;
    readonly nLocal: number; readonly nOwned: number;
    readonly ownedParticles: Uint32Array; readonly ownedBeams: Uint32Array;
    readonly peers: HaloPeer[];
    readonly globalParticleId: Uint32Array; readonly globalBeamKey: Uint32Array;
    lists(): [Uint32Array, Uint32Array, Uint32Array, Uint32Array];
    segments(): { segments: { rank: number, send: [number, number], recv: [number, number] }[], sendFloats: number,
        recvFloats: number, offsets: [Uint32Array, Uint32Array, Uint32Array, Uint32Array] };
}
export type RankScene = { readonly rank: number, readonly layout: 1 | 2, readonly maxParticles: number, readonly maxBeams: number,
    readonly metadata: ArrayBuffer, readonly mapping: ArrayBuffer, readonly particleData: ArrayBuffer, readonly beamData: ArrayBuffer,
    readonly plan: HaloPlan };
/** any scene (a BufferMapper after writeState(), or its four buffers) -> x-slabs with ghost zones `depth` beam hops deep */
export function partitionScene(scene: BufferMapper | { layout: 1 | 2, maxParticles: number, maxBeams: number, metadata: ArrayBuffer,
    mapping: ArrayBuffer, particleData: ArrayBuffer, beamData: ArrayBuffer }, world: number, depth: number, contactReach?: number,
    ranks?: number[]): RankScene[];
/** a rank's share of the gathered scene: rows of its own particles, dynamic fields of its own beams, which of them are left */
export type OwnedState = { readonly particleIds: Uint32Array, readonly particleRows: Float32Array, readonly beamKeys: Uint32Array,
    readonly beamDyn: Float32Array, readonly live: Uint8Array };
export function ownedState(plan: HaloPlan, local: RankScene): OwnedState;
/** partition again from the current state (between frames): updates `scene` in place from every rank's ownedState(), then partitionScene() */
export function repartition(scene: Parameters<typeof partitionScene>[0], states: OwnedState[], world: number, depth: number,
    contactReach?: number, ranks?: number[]): RankScene[];
export type PeerCard = { readonly rank: number, readonly pid: number, readonly pointer: number, readonly handle: number[],
    readonly recvFloats: number, readonly recv: [number, number, number][] };
export class PeerExchanger {
    constructor(handle: unknown, plan: HaloPlan, timeoutMs?: number);
    readonly card: PeerCard;
    readonly connected: boolean;
    connect(cardsByRank: (PeerCard | undefined)[]): void;
    exchange(): void;
    step(nSubsteps: number): void;
    /** one frame with the delete pass agreed between ranks (owner decides, ghost copies follow); every rank calls it */
    frame(subticks: number): void;
    verify(): void;
}
