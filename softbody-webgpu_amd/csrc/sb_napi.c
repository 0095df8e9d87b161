/*
 * sb_napi.c -- thin N-API addon over the C ABI of include/softbody.h.
 *
 * This is the binding a maintainer of spsquared/softbody-webgpu would load from a Node host
 * in place of the WebGPU calls of src/engineWorker.ts (device.createBuffer / queue.writeBuffer /
 * dispatchWorkgroups / mapAsync).  Plain C, built with `gcc -shared -I/usr/include/node`; the
 * HIP library is dlopen'ed at load() time, so the addon itself has no link-time dependency.
 * Every non-zero sb_status becomes a thrown JS Error carrying sb_last_error() (the reference
 * throws TypeError on init failure, engineWorker.ts:86,93,98).
 *
 * JS surface (all synchronous, one call at a time per engine, like the reference's AsyncLock):
 *   load(path)                        -> abi version
 *   create({boundsSize, particleRadius, subticks, maxParticles, maxBeams, layout,
 *           collisionMode, path, tileParticles, device, gridSkin, blockSubsteps}) -> handle
 *   destroy(h)
 *   writeBuffers(h, metadata, mapping, particles, beams)     ArrayBuffers, engineWorker.ts:580-597
 *   loadBuffers(h, metadata, mapping, particles, beams)      ArrayBuffers, engineWorker.ts:548-579
 *   writeUserInput(h, bytes32)        ArrayBuffer/TypedArray, engineMapping.ts:323-325
 *   setPhysicsConstants(h, Float32Array(8)) / getPhysicsConstants(h) -> Float32Array(8)
 *   frame(h) / step(h, n) / deletePass(h) / haloDeleteGhosts(h) / sync(h) / stepTimed(h, n) -> ms
 *   mark(h, slot) / markElapsed(h, a, b) -> ms
 *   getCounts(h) -> {particles, beams} / getInfo(h, key) -> number
 * multi-GPU (include/softbody.h "multi-GPU halo exchange", "direct neighbour exchange", "generic x-slab partition"):
 *   haloConfigure(h, ghostP, sendP, ghostB, sendB) / haloSetLayout(h, sendPOff, sendBOff, ghostPOff, ghostBOff)   Uint32Arrays
 *   haloPack(h, devicePtr) / haloUnpack(h, devicePtr) / getStream(h) -> number
 *   peerMailbox(h) -> {pointer, handle: ArrayBuffer(64), bytes} / peerMap(h, handle) -> pointer
 *   peerConnect(h, mailboxes[], recvFloats, sendBegin, sendLen, dstBegin, theirSlot, timeoutMs) / peerExchange(h)
 *   partitionCreate(layout, maxP, maxB, metadata, mapping, particles, beams, world, depth, contactReach) -> partition
 *   partitionRankCounts(p, rank) -> Uint32Array(8) / partitionRankScene(p, rank, maxP, maxB, metadata, mapping, particles, beams)
 *   partitionRankIds(p, rank) -> {particleGlobal, particleOwned, beamGlobal, beamOwned}
 *   partitionPeer(p, rank, j) -> {rank, ghostP, sendP, ghostB, sendB} / partitionDestroy(p)
 * Device pointers travel as JS numbers (they are < 2^53).  An engine handle is a small box around the sb_engine
 * pointer: destroy() empties the box, every later call with that handle throws instead of touching freed memory, and an
 * engine whose handle is garbage-collected without destroy() is destroyed by the finalizer.
 */
#define NAPI_VERSION 4
#include <dlfcn.h>
#include <node_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/softbody.h"

static struct {
    void *lib;
    void (*default_options)(sb_options *);
    sb_status (*create)(const sb_options *, sb_engine **);
    sb_status (*destroy)(sb_engine *);
    sb_status (*write_buffers)(sb_engine *, const void *, size_t, const void *, size_t, const void *, size_t,
                               const void *, size_t);
    sb_status (*load_buffers)(sb_engine *, void *, size_t, void *, size_t, void *, size_t, void *, size_t);
    sb_status (*write_user_input)(sb_engine *, const void *);
    sb_status (*set_physics_constants)(sb_engine *, const float *);
    sb_status (*get_physics_constants)(sb_engine *, float *);
    sb_status (*frame)(sb_engine *);
    sb_status (*step)(sb_engine *, uint32_t);
    sb_status (*delete_pass)(sb_engine *);
    sb_status (*halo_delete_ghosts)(sb_engine *);
    sb_status (*sync)(sb_engine *);
    sb_status (*step_timed)(sb_engine *, uint32_t, float *);
    sb_status (*mark)(sb_engine *, uint32_t);
    sb_status (*mark_elapsed)(sb_engine *, uint32_t, uint32_t, float *);
    sb_status (*get_counts)(sb_engine *, uint32_t *, uint32_t *);
    sb_status (*get_info)(sb_engine *, const char *, uint64_t *);
    sb_status (*halo_configure)(sb_engine *, const uint32_t *, uint32_t, const uint32_t *, uint32_t, const uint32_t *, uint32_t,
                                const uint32_t *, uint32_t);
    sb_status (*halo_set_layout)(sb_engine *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *);
    sb_status (*halo_pack)(sb_engine *, void *);
    sb_status (*halo_unpack)(sb_engine *, const void *);
    sb_status (*peer_mailbox)(sb_engine *, void **, void *, uint64_t *);
    sb_status (*peer_map)(sb_engine *, const void *, void **);
    sb_status (*peer_connect)(sb_engine *, uint32_t, void *const *, const uint32_t *, const uint32_t *, const uint32_t *,
                              const uint32_t *, const uint32_t *, uint32_t);
    sb_status (*peer_exchange)(sb_engine *);
    sb_status (*get_stream)(sb_engine *, void **);
    sb_status (*partition_create)(uint32_t, uint32_t, uint32_t, const void *, const void *, const void *, const void *, uint32_t,
                                  uint32_t, float, sb_partition **);
    sb_status (*partition_destroy)(sb_partition *);
    sb_status (*partition_rank_counts)(const sb_partition *, uint32_t, uint32_t *);
    sb_status (*partition_layout)(const sb_partition *, uint32_t *);
    sb_status (*partition_rank_scene)(const sb_partition *, uint32_t, uint32_t, uint32_t, void *, void *, void *, void *);
    sb_status (*partition_rank_ids)(const sb_partition *, uint32_t, uint32_t *, uint8_t *, uint32_t *, uint8_t *);
    sb_status (*partition_peer_counts)(const sb_partition *, uint32_t, uint32_t, uint32_t *, uint32_t *);
    sb_status (*partition_peer_lists)(const sb_partition *, uint32_t, uint32_t, uint32_t *, uint32_t *, uint32_t *, uint32_t *);
    const char *(*last_error)(const sb_engine *);
    uint32_t (*abi_version)(void);
} sb;

typedef struct { sb_engine *e; } engine_box;

#define CHECK_NAPI(call)                                          \
    do {                                                          \
        if ((call) != napi_ok) {                                  \
            napi_throw_error(env, NULL, "N-API call failed: " #call); \
            return NULL;                                          \
        }                                                         \
    } while (0)

static napi_value throw_status(napi_env env, sb_engine *e, sb_status st, const char *what)
{
    char msg[640], code[16];
    snprintf(code, sizeof code, "SB_%d", (int)st);
    snprintf(msg, sizeof msg, "%s failed (status %d): %s", what, (int)st, sb.last_error ? sb.last_error(e) : "?");
    napi_throw_error(env, code, msg);
    return NULL;
}

static int need_lib(napi_env env)
{
    if (!sb.lib) {
        napi_throw_error(env, NULL, "softbody addon: call load(pathToLibsoftbodyHip) first");
        return 0;
    }
    return 1;
}

static napi_value js_load(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    char path[4096];
    size_t len = 0;
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], path, sizeof path, &len) != napi_ok) {
        napi_throw_type_error(env, NULL, "load(path): path string expected");
        return NULL;
    }
    void *lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        char msg[4300];
        snprintf(msg, sizeof msg, "cannot load the HIP engine '%s': %s (there is no CPU fallback)", path, dlerror());
        napi_throw_error(env, NULL, msg);
        return NULL;
    }
#define SYM(field, name)                                                     \
    do {                                                                     \
        *(void **)(&sb.field) = dlsym(lib, name);                            \
        if (!sb.field) {                                                     \
            napi_throw_error(env, NULL, "HIP engine lacks symbol " name);    \
            dlclose(lib);                                                    \
            return NULL;                                                     \
        }                                                                    \
    } while (0)
    SYM(default_options, "sb_default_options");
    SYM(create, "sb_create");
    SYM(destroy, "sb_destroy");
    SYM(write_buffers, "sb_write_buffers");
    SYM(load_buffers, "sb_load_buffers");
    SYM(write_user_input, "sb_write_user_input");
    SYM(set_physics_constants, "sb_set_physics_constants");
    SYM(get_physics_constants, "sb_get_physics_constants");
    SYM(frame, "sb_frame");
    SYM(step, "sb_step");
    SYM(delete_pass, "sb_delete_pass");
    SYM(halo_delete_ghosts, "sb_halo_delete_ghosts");
    SYM(sync, "sb_sync");
    SYM(step_timed, "sb_step_timed");
    SYM(mark, "sb_mark");
    SYM(mark_elapsed, "sb_mark_elapsed");
    SYM(get_counts, "sb_get_counts");
    SYM(get_info, "sb_get_info");
    SYM(halo_configure, "sb_halo_configure");
    SYM(halo_set_layout, "sb_halo_set_layout");
    SYM(halo_pack, "sb_halo_pack");
    SYM(halo_unpack, "sb_halo_unpack");
    SYM(peer_mailbox, "sb_peer_mailbox");
    SYM(peer_map, "sb_peer_map");
    SYM(peer_connect, "sb_peer_connect");
    SYM(peer_exchange, "sb_peer_exchange");
    SYM(get_stream, "sb_get_stream");
    SYM(partition_create, "sb_partition_create");
    SYM(partition_destroy, "sb_partition_destroy");
    SYM(partition_rank_counts, "sb_partition_rank_counts");
    SYM(partition_layout, "sb_partition_layout");
    SYM(partition_rank_scene, "sb_partition_rank_scene");
    SYM(partition_rank_ids, "sb_partition_rank_ids");
    SYM(partition_peer_counts, "sb_partition_peer_counts");
    SYM(partition_peer_lists, "sb_partition_peer_lists");
    SYM(last_error, "sb_last_error");
    SYM(abi_version, "sb_abi_version");
#undef SYM
    sb.lib = lib;
    napi_value v;
    CHECK_NAPI(napi_create_uint32(env, sb.abi_version(), &v));
    return v;
}

static int opt_number(napi_env env, napi_value obj, const char *name, double *out)
{
    bool has = false;
    napi_value v;
    napi_valuetype t;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return 0;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_number) return 0;
    return napi_get_value_double(env, v, out) == napi_ok;
}

static void box_finalize(napi_env env, void *data, void *hint)
{
    (void)env;
    (void)hint;
    engine_box *box = (engine_box *)data;
    if (box->e && sb.destroy) sb.destroy(box->e); /* a handle dropped without destroy(): release the device */
    free(box);
}

static napi_value js_create(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 1;
    napi_value argv[1];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_options o;
    sb.default_options(&o);
    if (argc >= 1) {
        napi_valuetype t;
        CHECK_NAPI(napi_typeof(env, argv[0], &t));
        if (t == napi_object) {
            double d;
            if (opt_number(env, argv[0], "boundsSize", &d)) o.bounds_size = (float)d;
            if (opt_number(env, argv[0], "particleRadius", &d)) o.particle_radius = (float)d;
            if (opt_number(env, argv[0], "subticks", &d)) o.subticks = (uint32_t)d;
            if (opt_number(env, argv[0], "maxParticles", &d)) o.max_particles = (uint32_t)d;
            if (opt_number(env, argv[0], "maxBeams", &d)) o.max_beams = (uint32_t)d;
            if (opt_number(env, argv[0], "layout", &d)) o.layout = (uint32_t)d;
            if (opt_number(env, argv[0], "collisionMode", &d)) o.collision_mode = (uint32_t)d;
            if (opt_number(env, argv[0], "path", &d)) o.path = (uint32_t)d;
            if (opt_number(env, argv[0], "tileParticles", &d)) o.tile_particles = (uint32_t)d;
            if (opt_number(env, argv[0], "device", &d)) o.device_ordinal = (int32_t)d;
            if (opt_number(env, argv[0], "gridSkin", &d)) o.grid_skin = (float)d;
            if (opt_number(env, argv[0], "blockSubsteps", &d)) o.block_substeps = (uint32_t)d;
        }
    }
    sb_engine *e = NULL;
    sb_status st = sb.create(&o, &e);
    if (st != SB_OK) return throw_status(env, NULL, st, "sb_create");
    engine_box *box = (engine_box *)malloc(sizeof *box);
    if (!box) {
        sb.destroy(e);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    box->e = e;
    napi_value ext;
    if (napi_create_external(env, box, box_finalize, NULL, &ext) != napi_ok) {
        sb.destroy(e);
        free(box);
        napi_throw_error(env, NULL, "napi_create_external failed");
        return NULL;
    }
    return ext;
}

static engine_box *get_box(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "engine handle expected");
        return NULL;
    }
    return (engine_box *)p;
}

static sb_engine *get_engine(napi_env env, napi_value v)
{
    engine_box *box = get_box(env, v);
    if (!box) return NULL;
    if (!box->e) {
        napi_throw_error(env, NULL, "this engine has been destroyed");
        return NULL;
    }
    return box->e;
}

/* ArrayBuffer or TypedArray/DataView -> pointer + byte length */
static int get_bytes(napi_env env, napi_value v, void **data, size_t *len)
{
    bool is = false;
    if (napi_is_arraybuffer(env, v, &is) == napi_ok && is) return napi_get_arraybuffer_info(env, v, data, len) == napi_ok;
    if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
        napi_typedarray_type tt;
        size_t n, off;
        napi_value ab;
        if (napi_get_typedarray_info(env, v, &tt, &n, data, &ab, &off) != napi_ok) return 0;
        static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
        if ((size_t)tt >= sizeof esz / sizeof esz[0]) return 0; /* an element type newer than this table (Float16Array ...) */
        *len = n * esz[tt];
        return 1;
    }
    if (napi_is_dataview(env, v, &is) == napi_ok && is) {
        napi_value ab;
        size_t off;
        return napi_get_dataview_info(env, v, len, data, &ab, &off) == napi_ok;
    }
    return 0;
}

static napi_value js_buffers(napi_env env, napi_callback_info info, int write)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 5;
    napi_value argv[5];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc < 5) {
        napi_throw_type_error(env, NULL, "(handle, metadata, mapping, particles, beams) expected");
        return NULL;
    }
    sb_engine *e = get_engine(env, argv[0]);
    if (!e) return NULL;
    void *p[4];
    size_t n[4];
    for (int i = 0; i < 4; i++) {
        if (!get_bytes(env, argv[i + 1], &p[i], &n[i])) {
            napi_throw_type_error(env, NULL, "ArrayBuffer or TypedArray expected");
            return NULL;
        }
    }
    sb_status st = write ? sb.write_buffers(e, p[0], n[0], p[1], n[1], p[2], n[2], p[3], n[3])
                         : sb.load_buffers(e, p[0], n[0], p[1], n[1], p[2], n[2], p[3], n[3]);
    if (st != SB_OK) return throw_status(env, e, st, write ? "sb_write_buffers" : "sb_load_buffers");
    return NULL;
}
static napi_value js_write_buffers(napi_env env, napi_callback_info info) { return js_buffers(env, info, 1); }
static napi_value js_load_buffers(napi_env env, napi_callback_info info) { return js_buffers(env, info, 0); }

static napi_value js_write_user_input(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    void *p;
    size_t n;
    if (argc < 2 || !get_bytes(env, argv[1], &p, &n) || n < SB_USER_INPUT_BYTES) {
        napi_throw_type_error(env, NULL, "writeUserInput(handle, 32 bytes) expected");
        return NULL;
    }
    sb_status st = sb.write_user_input(e, p);
    if (st != SB_OK) return throw_status(env, e, st, "sb_write_user_input");
    return NULL;
}

static napi_value js_set_constants(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    void *p;
    size_t n;
    if (argc < 2 || !get_bytes(env, argv[1], &p, &n) || n < 32) {
        napi_throw_type_error(env, NULL, "setPhysicsConstants(handle, Float32Array(8)) expected");
        return NULL;
    }
    sb_status st = sb.set_physics_constants(e, (const float *)p);
    if (st != SB_OK) return throw_status(env, e, st, "sb_set_physics_constants");
    return NULL;
}

static napi_value js_get_constants(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 1;
    napi_value argv[1];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    void *data;
    napi_value ab, ta;
    CHECK_NAPI(napi_create_arraybuffer(env, 32, &data, &ab));
    sb_status st = sb.get_physics_constants(e, (float *)data);
    if (st != SB_OK) return throw_status(env, e, st, "sb_get_physics_constants");
    CHECK_NAPI(napi_create_typedarray(env, napi_float32_array, 8, ab, 0, &ta));
    return ta;
}

typedef sb_status (*simple_fn)(sb_engine *);
static napi_value js_simple(napi_env env, napi_callback_info info, int which, const char *name)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 1;
    napi_value argv[1];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    simple_fn f = which == 0 ? sb.frame : which == 1 ? sb.delete_pass : which == 2 ? sb.sync : which == 4 ? sb.peer_exchange : which == 5 ? sb.halo_delete_ghosts : sb.destroy;
    if (which == 3) get_box(env, argv[0])->e = NULL; /* whatever sb_destroy says, the pointer is gone */
    sb_status st = f(e);
    if (st != SB_OK) return throw_status(env, which == 3 ? NULL : e, st, name);
    return NULL;
}
static napi_value js_frame(napi_env env, napi_callback_info info) { return js_simple(env, info, 0, "sb_frame"); }
static napi_value js_delete_pass(napi_env env, napi_callback_info info) { return js_simple(env, info, 1, "sb_delete_pass"); }
static napi_value js_sync(napi_env env, napi_callback_info info) { return js_simple(env, info, 2, "sb_sync"); }
static napi_value js_destroy(napi_env env, napi_callback_info info) { return js_simple(env, info, 3, "sb_destroy"); }
static napi_value js_peer_exchange(napi_env env, napi_callback_info info) { return js_simple(env, info, 4, "sb_peer_exchange"); }
static napi_value js_halo_delete_ghosts(napi_env env, napi_callback_info info) { return js_simple(env, info, 5, "sb_halo_delete_ghosts"); }

static napi_value js_step(napi_env env, napi_callback_info info, int timed)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    uint32_t n = 0;
    if (argc < 2 || napi_get_value_uint32(env, argv[1], &n) != napi_ok) {
        napi_throw_type_error(env, NULL, "step(handle, nSubsteps) expected");
        return NULL;
    }
    if (!timed) {
        sb_status st = sb.step(e, n);
        if (st != SB_OK) return throw_status(env, e, st, "sb_step");
        return NULL;
    }
    float ms = 0.f;
    sb_status st = sb.step_timed(e, n, &ms);
    if (st != SB_OK) return throw_status(env, e, st, "sb_step_timed");
    napi_value v;
    CHECK_NAPI(napi_create_double(env, (double)ms, &v));
    return v;
}
static napi_value js_step_plain(napi_env env, napi_callback_info info) { return js_step(env, info, 0); }
static napi_value js_step_timed(napi_env env, napi_callback_info info) { return js_step(env, info, 1); }

/* mark(handle, slot); markElapsed(handle, a, b) -> ms */
static napi_value js_mark(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    uint32_t slot = 0;
    if (argc < 2 || napi_get_value_uint32(env, argv[1], &slot) != napi_ok) {
        napi_throw_type_error(env, NULL, "mark(handle, slot) expected");
        return NULL;
    }
    sb_status st = sb.mark(e, slot);
    if (st != SB_OK) return throw_status(env, e, st, "sb_mark");
    return NULL;
}

static napi_value js_mark_elapsed(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 3;
    napi_value argv[3];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    uint32_t a = 0, b = 0;
    if (argc < 3 || napi_get_value_uint32(env, argv[1], &a) != napi_ok || napi_get_value_uint32(env, argv[2], &b) != napi_ok) {
        napi_throw_type_error(env, NULL, "markElapsed(handle, a, b) expected");
        return NULL;
    }
    float ms = 0.f;
    sb_status st = sb.mark_elapsed(e, a, b, &ms);
    if (st != SB_OK) return throw_status(env, e, st, "sb_mark_elapsed");
    napi_value v;
    CHECK_NAPI(napi_create_double(env, (double)ms, &v));
    return v;
}

static napi_value js_get_counts(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 1;
    napi_value argv[1];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    uint32_t p = 0, b = 0;
    sb_status st = sb.get_counts(e, &p, &b);
    if (st != SB_OK) return throw_status(env, e, st, "sb_get_counts");
    napi_value obj, vp, vb;
    CHECK_NAPI(napi_create_object(env, &obj));
    CHECK_NAPI(napi_create_uint32(env, p, &vp));
    CHECK_NAPI(napi_create_uint32(env, b, &vb));
    CHECK_NAPI(napi_set_named_property(env, obj, "particles", vp));
    CHECK_NAPI(napi_set_named_property(env, obj, "beams", vb));
    return obj;
}

static napi_value js_get_info(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    char key[64];
    size_t len;
    if (argc < 2 || napi_get_value_string_utf8(env, argv[1], key, sizeof key, &len) != napi_ok) {
        napi_throw_type_error(env, NULL, "getInfo(handle, key) expected");
        return NULL;
    }
    uint64_t v = 0;
    sb_status st = sb.get_info(e, key, &v);
    if (st != SB_OK) return throw_status(env, e, st, "sb_get_info");
    napi_value out;
    CHECK_NAPI(napi_create_double(env, (double)v, &out));
    return out;
}

/* ---------------------------------------------------------------- multi-GPU: halo lists, peer mailboxes, partitioner */

/* Uint32Array (or empty / undefined) -> pointer + element count */
static int get_u32(napi_env env, napi_value v, const uint32_t **data, uint32_t *n)
{
    napi_valuetype t;
    *data = NULL;
    *n = 0;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_undefined || t == napi_null) return 1;
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) return 0;
    napi_typedarray_type tt;
    size_t len, off;
    void *p;
    napi_value ab;
    if (napi_get_typedarray_info(env, v, &tt, &len, &p, &ab, &off) != napi_ok || tt != napi_uint32_array) return 0;
    *data = (const uint32_t *)p;
    *n = (uint32_t)len;
    return 1;
}

static napi_value make_typed(napi_env env, napi_typedarray_type tt, size_t n, size_t esz, void **data)
{
    napi_value ab, ta;
    if (napi_create_arraybuffer(env, n * esz, data, &ab) != napi_ok) return NULL;
    if (napi_create_typedarray(env, tt, n, ab, 0, &ta) != napi_ok) return NULL;
    return ta;
}

static napi_value ptr_value(napi_env env, const void *p)
{
    napi_value v;
    if (napi_create_double(env, (double)(uintptr_t)p, &v) != napi_ok) return NULL;
    return v;
}
static int value_ptr(napi_env env, napi_value v, void **p)
{
    double d;
    if (napi_get_value_double(env, v, &d) != napi_ok || d < 0 || d >= 9007199254740992.0) return 0;
    *p = (void *)(uintptr_t)d;
    return 1;
}

static napi_value js_halo_lists(napi_env env, napi_callback_info info, int layout)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 5;
    napi_value argv[5];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    const uint32_t *a[4] = {0};
    uint32_t n[4] = {0};
    for (int k = 0; k < 4; k++)
        if (argc < (size_t)k + 2 || !get_u32(env, argv[k + 1], &a[k], &n[k])) {
            napi_throw_type_error(env, NULL, "four Uint32Arrays expected after the handle");
            return NULL;
        }
    sb_status st = layout ? sb.halo_set_layout(e, a[0], a[1], a[2], a[3])
                          : sb.halo_configure(e, a[0], n[0], a[1], n[1], a[2], n[2], a[3], n[3]);
    if (st != SB_OK) return throw_status(env, e, st, layout ? "sb_halo_set_layout" : "sb_halo_configure");
    return NULL;
}
static napi_value js_halo_configure(napi_env env, napi_callback_info info) { return js_halo_lists(env, info, 0); }
static napi_value js_halo_set_layout(napi_env env, napi_callback_info info) { return js_halo_lists(env, info, 1); }

static napi_value js_halo_move(napi_env env, napi_callback_info info, int unpack)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    void *p = NULL;
    if (argc < 2 || !value_ptr(env, argv[1], &p)) {
        napi_throw_type_error(env, NULL, "(handle, devicePointer) expected");
        return NULL;
    }
    sb_status st = unpack ? sb.halo_unpack(e, p) : sb.halo_pack(e, p);
    if (st != SB_OK) return throw_status(env, e, st, unpack ? "sb_halo_unpack" : "sb_halo_pack");
    return NULL;
}
static napi_value js_halo_pack(napi_env env, napi_callback_info info) { return js_halo_move(env, info, 0); }
static napi_value js_halo_unpack(napi_env env, napi_callback_info info) { return js_halo_move(env, info, 1); }

static napi_value js_get_stream(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 1;
    napi_value argv[1];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    void *s = NULL;
    sb_status st = sb.get_stream(e, &s);
    if (st != SB_OK) return throw_status(env, e, st, "sb_get_stream");
    return ptr_value(env, s);
}

static napi_value js_peer_mailbox(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 1;
    napi_value argv[1];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    void *box = NULL, *hdata = NULL;
    uint64_t bytes = 0;
    napi_value hab, obj, vb;
    CHECK_NAPI(napi_create_arraybuffer(env, 64, &hdata, &hab));
    sb_status st = sb.peer_mailbox(e, &box, hdata, &bytes);
    if (st != SB_OK) return throw_status(env, e, st, "sb_peer_mailbox");
    CHECK_NAPI(napi_create_object(env, &obj));
    CHECK_NAPI(napi_set_named_property(env, obj, "pointer", ptr_value(env, box)));
    CHECK_NAPI(napi_set_named_property(env, obj, "handle", hab));
    CHECK_NAPI(napi_create_double(env, (double)bytes, &vb));
    CHECK_NAPI(napi_set_named_property(env, obj, "bytes", vb));
    return obj;
}

static napi_value js_peer_map(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    void *data = NULL, *mapped = NULL;
    size_t len = 0;
    if (argc < 2 || !get_bytes(env, argv[1], &data, &len) || len < 64) {
        napi_throw_type_error(env, NULL, "peerMap(handle, ipcHandle): 64 bytes expected");
        return NULL;
    }
    sb_status st = sb.peer_map(e, data, &mapped);
    if (st != SB_OK) return throw_status(env, e, st, "sb_peer_map");
    return ptr_value(env, mapped);
}

static napi_value js_peer_connect(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 8;
    napi_value argv[8];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_engine *e = argc >= 1 ? get_engine(env, argv[0]) : NULL;
    if (!e) return NULL;
    bool is_arr = false;
    uint32_t nbox = 0;
    if (argc < 7 || napi_is_array(env, argv[1], &is_arr) != napi_ok || !is_arr || napi_get_array_length(env, argv[1], &nbox) != napi_ok ||
        nbox > SB_MAX_PEERS) {
        napi_throw_type_error(env, NULL, "peerConnect(handle, mailboxes[], recvFloats, sendBegin, sendLen, dstBegin, theirSlot, timeoutMs)");
        return NULL;
    }
    void *boxes[SB_MAX_PEERS] = {0};
    for (uint32_t j = 0; j < nbox; j++) {
        napi_value v;
        if (napi_get_element(env, argv[1], j, &v) != napi_ok || !value_ptr(env, v, &boxes[j])) {
            napi_throw_type_error(env, NULL, "peerConnect: mailbox pointers must be numbers");
            return NULL;
        }
    }
    const uint32_t *a[5] = {0};
    uint32_t n[5] = {0};
    for (int k = 0; k < 5; k++)
        if (!get_u32(env, argv[k + 2], &a[k], &n[k]) || n[k] != nbox) {
            napi_throw_type_error(env, NULL, "peerConnect: five Uint32Arrays, one entry per mailbox");
            return NULL;
        }
    uint32_t timeout = 0;
    if (argc >= 8) (void)napi_get_value_uint32(env, argv[7], &timeout);
    sb_status st = sb.peer_connect(e, nbox, boxes, a[0], a[1], a[2], a[3], a[4], timeout);
    if (st != SB_OK) return throw_status(env, e, st, "sb_peer_connect");
    return NULL;
}

static void partition_finalize(napi_env env, void *data, void *hint)
{
    (void)env;
    (void)hint;
    sb_partition **box = (sb_partition **)data;
    if (*box && sb.partition_destroy) sb.partition_destroy(*box);
    free(box);
}
static sb_partition *get_partition(napi_env env, napi_value v, sb_partition ***boxp)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !*(sb_partition **)p) {
        napi_throw_type_error(env, NULL, "partition handle expected");
        return NULL;
    }
    if (boxp) *boxp = (sb_partition **)p;
    return *(sb_partition **)p;
}

static napi_value js_partition_create(napi_env env, napi_callback_info info)
{
    if (!need_lib(env)) return NULL;
    size_t argc = 10;
    napi_value argv[10];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    uint32_t u[5] = {0};
    double reach = 0;
    void *data[4];
    size_t len[4];
    if (argc < 10 || napi_get_value_uint32(env, argv[0], &u[0]) != napi_ok || napi_get_value_uint32(env, argv[1], &u[1]) != napi_ok ||
        napi_get_value_uint32(env, argv[2], &u[2]) != napi_ok || napi_get_value_uint32(env, argv[7], &u[3]) != napi_ok ||
        napi_get_value_uint32(env, argv[8], &u[4]) != napi_ok || napi_get_value_double(env, argv[9], &reach) != napi_ok) {
        napi_throw_type_error(env, NULL, "partitionCreate(layout, maxParticles, maxBeams, metadata, mapping, particles, beams, world, depth, contactReach)");
        return NULL;
    }
    for (int k = 0; k < 4; k++)
        if (!get_bytes(env, argv[k + 3], &data[k], &len[k])) {
            napi_throw_type_error(env, NULL, "partitionCreate: ArrayBuffer or TypedArray expected");
            return NULL;
        }
    const size_t bstride = u[0] == SB_LAYOUT_V1 ? SB_BEAM_STRIDE_V1 : SB_BEAM_STRIDE_V2, isz = u[0] == SB_LAYOUT_V1 ? 2 : 4;
    if (len[0] < SB_METADATA_BYTES || len[1] < ((size_t)u[1] + u[2]) * isz || len[2] < (size_t)u[1] * SB_PARTICLE_STRIDE ||
        len[3] < (size_t)u[2] * bstride) {
        napi_throw_range_error(env, NULL, "partitionCreate: a buffer is smaller than the capacities say");
        return NULL;
    }
    sb_partition *pt = NULL;
    sb_status st = sb.partition_create(u[0], u[1], u[2], data[0], data[1], data[2], data[3], u[3], u[4], (float)reach, &pt);
    if (st != SB_OK) return throw_status(env, NULL, st, "sb_partition_create");
    sb_partition **box = (sb_partition **)malloc(sizeof *box);
    napi_value ext;
    if (!box || (*box = pt, napi_create_external(env, box, partition_finalize, NULL, &ext) != napi_ok)) {
        sb.partition_destroy(pt);
        free(box);
        napi_throw_error(env, NULL, "out of memory");
        return NULL;
    }
    return ext;
}

static napi_value js_partition_destroy(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_partition **box = NULL;
    sb_partition *pt = argc >= 1 ? get_partition(env, argv[0], &box) : NULL;
    if (!pt) return NULL;
    *box = NULL;
    sb.partition_destroy(pt);
    return NULL;
}

static napi_value js_partition_rank_counts(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_partition *pt = argc >= 1 ? get_partition(env, argv[0], NULL) : NULL;
    if (!pt) return NULL;
    uint32_t rank = 0;
    void *out = NULL;
    if (argc < 2 || napi_get_value_uint32(env, argv[1], &rank) != napi_ok) {
        napi_throw_type_error(env, NULL, "partitionRankCounts(partition, rank)");
        return NULL;
    }
    napi_value ta = make_typed(env, napi_uint32_array, 8, 4, &out);
    if (!ta) return NULL;
    sb_status st = sb.partition_rank_counts(pt, rank, (uint32_t *)out);
    if (st != SB_OK) return throw_status(env, NULL, st, "sb_partition_rank_counts");
    return ta;
}

static napi_value js_partition_rank_scene(napi_env env, napi_callback_info info)
{
    size_t argc = 8;
    napi_value argv[8];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_partition *pt = argc >= 1 ? get_partition(env, argv[0], NULL) : NULL;
    if (!pt) return NULL;
    uint32_t rank = 0, maxp = 0, maxb = 0;
    void *data[4];
    size_t len[4];
    if (argc < 8 || napi_get_value_uint32(env, argv[1], &rank) != napi_ok || napi_get_value_uint32(env, argv[2], &maxp) != napi_ok ||
        napi_get_value_uint32(env, argv[3], &maxb) != napi_ok) {
        napi_throw_type_error(env, NULL, "partitionRankScene(partition, rank, maxParticles, maxBeams, metadata, mapping, particles, beams)");
        return NULL;
    }
    for (int k = 0; k < 4; k++)
        if (!get_bytes(env, argv[k + 4], &data[k], &len[k])) {
            napi_throw_type_error(env, NULL, "partitionRankScene: ArrayBuffer or TypedArray expected");
            return NULL;
        }
    /* sizes: what sb_partition_rank_scene writes follows the PARTITION's layout (2- or 4-byte mapping indices, 40- or 44-byte
     * beam records), not what the caller may have assumed: buffers sized for v1 handed in for a v2 partition are refused here
     * instead of being overrun */
    uint32_t layout = 0;
    if (sb.partition_layout(pt, &layout) != SB_OK) return throw_status(env, NULL, SB_ERR_INVALID, "sb_partition_layout");
    const size_t isz = layout == SB_LAYOUT_V1 ? 2 : 4, bstride = layout == SB_LAYOUT_V1 ? SB_BEAM_STRIDE_V1 : SB_BEAM_STRIDE_V2;
    if (len[0] < SB_METADATA_BYTES || len[1] < ((size_t)maxp + maxb) * isz || len[2] < (size_t)maxp * SB_PARTICLE_STRIDE ||
        len[3] < (size_t)maxb * bstride) {
        napi_throw_range_error(env, NULL, "partitionRankScene: a buffer is smaller than the capacities say");
        return NULL;
    }
    sb_status st = sb.partition_rank_scene(pt, rank, maxp, maxb, data[0], data[1], data[2], data[3]);
    if (st != SB_OK) return throw_status(env, NULL, st, "sb_partition_rank_scene");
    return NULL;
}

static napi_value js_partition_rank_ids(napi_env env, napi_callback_info info)
{
    size_t argc = 2;
    napi_value argv[2];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_partition *pt = argc >= 1 ? get_partition(env, argv[0], NULL) : NULL;
    if (!pt) return NULL;
    uint32_t rank = 0, c[8];
    if (argc < 2 || napi_get_value_uint32(env, argv[1], &rank) != napi_ok || sb.partition_rank_counts(pt, rank, c) != SB_OK) {
        napi_throw_type_error(env, NULL, "partitionRankIds(partition, rank)");
        return NULL;
    }
    void *pg, *po, *bg, *bo;
    napi_value obj, tpg = make_typed(env, napi_uint32_array, c[0], 4, &pg), tpo = make_typed(env, napi_uint8_array, c[0], 1, &po),
                    tbg = make_typed(env, napi_uint32_array, c[1], 4, &bg), tbo = make_typed(env, napi_uint8_array, c[1], 1, &bo);
    if (!tpg || !tpo || !tbg || !tbo) return NULL;
    sb_status st = sb.partition_rank_ids(pt, rank, (uint32_t *)pg, (uint8_t *)po, (uint32_t *)bg, (uint8_t *)bo);
    if (st != SB_OK) return throw_status(env, NULL, st, "sb_partition_rank_ids");
    CHECK_NAPI(napi_create_object(env, &obj));
    CHECK_NAPI(napi_set_named_property(env, obj, "particleGlobal", tpg));
    CHECK_NAPI(napi_set_named_property(env, obj, "particleOwned", tpo));
    CHECK_NAPI(napi_set_named_property(env, obj, "beamGlobal", tbg));
    CHECK_NAPI(napi_set_named_property(env, obj, "beamOwned", tbo));
    return obj;
}

static napi_value js_partition_peer(napi_env env, napi_callback_info info)
{
    size_t argc = 3;
    napi_value argv[3];
    CHECK_NAPI(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    sb_partition *pt = argc >= 1 ? get_partition(env, argv[0], NULL) : NULL;
    if (!pt) return NULL;
    uint32_t rank = 0, j = 0, peer = 0, c[4];
    if (argc < 3 || napi_get_value_uint32(env, argv[1], &rank) != napi_ok || napi_get_value_uint32(env, argv[2], &j) != napi_ok ||
        sb.partition_peer_counts(pt, rank, j, &peer, c) != SB_OK) {
        napi_throw_type_error(env, NULL, "partitionPeer(partition, rank, j): no such peer");
        return NULL;
    }
    static const char *names[4] = {"ghostP", "sendP", "ghostB", "sendB"};
    void *d[4];
    napi_value obj, vr, ta[4];
    for (int k = 0; k < 4; k++)
        if (!(ta[k] = make_typed(env, napi_uint32_array, c[k], 4, &d[k]))) return NULL;
    sb_status st = sb.partition_peer_lists(pt, rank, j, (uint32_t *)d[0], (uint32_t *)d[1], (uint32_t *)d[2], (uint32_t *)d[3]);
    if (st != SB_OK) return throw_status(env, NULL, st, "sb_partition_peer_lists");
    CHECK_NAPI(napi_create_object(env, &obj));
    CHECK_NAPI(napi_create_uint32(env, peer, &vr));
    CHECK_NAPI(napi_set_named_property(env, obj, "rank", vr));
    for (int k = 0; k < 4; k++) CHECK_NAPI(napi_set_named_property(env, obj, names[k], ta[k]));
    return obj;
}

static napi_value init(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"load", js_load}, {"create", js_create}, {"destroy", js_destroy},
        {"writeBuffers", js_write_buffers}, {"loadBuffers", js_load_buffers},
        {"writeUserInput", js_write_user_input}, {"setPhysicsConstants", js_set_constants},
        {"getPhysicsConstants", js_get_constants}, {"frame", js_frame}, {"step", js_step_plain},
        {"deletePass", js_delete_pass}, {"sync", js_sync}, {"stepTimed", js_step_timed},
        {"mark", js_mark}, {"markElapsed", js_mark_elapsed},
        {"getCounts", js_get_counts}, {"getInfo", js_get_info},
        {"haloConfigure", js_halo_configure}, {"haloSetLayout", js_halo_set_layout}, {"haloPack", js_halo_pack},
        {"haloUnpack", js_halo_unpack}, {"getStream", js_get_stream}, {"peerMailbox", js_peer_mailbox}, {"peerMap", js_peer_map},
        {"peerConnect", js_peer_connect}, {"peerExchange", js_peer_exchange}, {"haloDeleteGhosts", js_halo_delete_ghosts},
        {"partitionCreate", js_partition_create}, {"partitionDestroy", js_partition_destroy},
        {"partitionRankCounts", js_partition_rank_counts}, {"partitionRankScene", js_partition_rank_scene},
        {"partitionRankIds", js_partition_rank_ids}, {"partitionPeer", js_partition_peer},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
