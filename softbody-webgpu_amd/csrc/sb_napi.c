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
 *           collisionMode, path, tileParticles, device}) -> handle
 *   destroy(h)
 *   writeBuffers(h, metadata, mapping, particles, beams)     ArrayBuffers, engineWorker.ts:580-597
 *   loadBuffers(h, metadata, mapping, particles, beams)      ArrayBuffers, engineWorker.ts:548-579
 *   writeUserInput(h, bytes32)        ArrayBuffer/TypedArray, engineMapping.ts:323-325
 *   setPhysicsConstants(h, Float32Array(8)) / getPhysicsConstants(h) -> Float32Array(8)
 *   frame(h) / step(h, n) / deletePass(h) / sync(h) / stepTimed(h, n) -> ms
 *   getCounts(h) -> {particles, beams} / getInfo(h, key) -> number
 */
#define NAPI_VERSION 4
#include <dlfcn.h>
#include <node_api.h>
#include <stdio.h>
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
    sb_status (*sync)(sb_engine *);
    sb_status (*step_timed)(sb_engine *, uint32_t, float *);
    sb_status (*get_counts)(sb_engine *, uint32_t *, uint32_t *);
    sb_status (*get_info)(sb_engine *, const char *, uint64_t *);
    const char *(*last_error)(const sb_engine *);
    uint32_t (*abi_version)(void);
} sb;

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
    SYM(sync, "sb_sync");
    SYM(step_timed, "sb_step_timed");
    SYM(get_counts, "sb_get_counts");
    SYM(get_info, "sb_get_info");
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
        }
    }
    sb_engine *e = NULL;
    sb_status st = sb.create(&o, &e);
    if (st != SB_OK) return throw_status(env, NULL, st, "sb_create");
    napi_value ext;
    CHECK_NAPI(napi_create_external(env, e, NULL, NULL, &ext));
    return ext;
}

static sb_engine *get_engine(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "engine handle expected");
        return NULL;
    }
    return (sb_engine *)p;
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
    simple_fn f = which == 0 ? sb.frame : which == 1 ? sb.delete_pass : which == 2 ? sb.sync : sb.destroy;
    sb_status st = f(e);
    if (st != SB_OK) return throw_status(env, which == 3 ? NULL : e, st, name);
    return NULL;
}
static napi_value js_frame(napi_env env, napi_callback_info info) { return js_simple(env, info, 0, "sb_frame"); }
static napi_value js_delete_pass(napi_env env, napi_callback_info info) { return js_simple(env, info, 1, "sb_delete_pass"); }
static napi_value js_sync(napi_env env, napi_callback_info info) { return js_simple(env, info, 2, "sb_sync"); }
static napi_value js_destroy(napi_env env, napi_callback_info info) { return js_simple(env, info, 3, "sb_destroy"); }

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

static napi_value init(napi_env env, napi_value exports)
{
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"load", js_load}, {"create", js_create}, {"destroy", js_destroy},
        {"writeBuffers", js_write_buffers}, {"loadBuffers", js_load_buffers},
        {"writeUserInput", js_write_user_input}, {"setPhysicsConstants", js_set_constants},
        {"getPhysicsConstants", js_get_constants}, {"frame", js_frame}, {"step", js_step_plain},
        {"deletePass", js_delete_pass}, {"sync", js_sync}, {"stepTimed", js_step_timed},
        {"getCounts", js_get_counts}, {"getInfo", js_get_info},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
