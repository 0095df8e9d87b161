/*
 * sb_oracle.c -- CPU ORACLE (test infrastructure, see sb_oracle.h).
 *
 * Scalar restatement of /root/reference/src/shaders/compute.wgsl.  Every
 * block cites the WGSL lines it follows.  Compile with -ffp-contract=off:
 * the canonical arithmetic is IEEE binary32, round-to-nearest-even, one
 * rounding per WGSL operator, evaluated in the order the WGSL source writes
 * it.  WGSL built-ins are pinned to these definitions (the WGSL spec only
 * bounds them in ULP, so any of them is a valid reference outcome):
 *   length(v)    = sqrt(v.x*v.x + v.y*v.y)           (correctly rounded sqrt)
 *   normalize(v) = (v.x * r, v.y * r), r = 1 / length(v)  (one correctly rounded reciprocal; real
 *                  WebGPU back ends lower normalize to v * inverseSqrt(dot(v,v)), so the product
 *                  form is at least as faithful as two divisions and costs one divide less)
 *   strain       = (len - target) * (1 / length)   (compute.wgsl:112; WGSL division is only 2.5-ULP
 *                  accurate, the reciprocal of the constant rest length is one IEEE divide)
 *   distance(a,b)= length(a - b)
 *   dot(a,b)     = a.x*b.x + a.y*b.y
 *   min(a,b)     = b < a ? b : a ;  max(a,b) = a < b ? b : a
 *   clamp(x,l,h) = min(max(x,l),h)   (also when l > h)
 *   sign(x)      = (x > 0) - (x < 0)
 *   abs(x)       = clear sign bit
 *   pow(x,y)     = sbo_pow below
 *   i32(f)       = truncate toward zero, saturating, NaN -> 0
 */
#include "sb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- helpers */

typedef struct { float x, y; } v2;

static inline float f_min(float a, float b) { return (b < a) ? b : a; }
static inline float f_max(float a, float b) { return (a < b) ? b : a; }
static inline float f_clamp(float x, float lo, float hi) { return f_min(f_max(x, lo), hi); }
static inline float f_sign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
static inline float f_abs(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u &= 0x7fffffffu;
    memcpy(&x, &u, 4);
    return x;
}
static inline float v_length(v2 v) { return sqrtf(v.x * v.x + v.y * v.y); }
static inline float v_dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
/* SBO_VARIANT (default 0 = the canonical arithmetic above) builds the OTHER readings of the WGSL text that a
 * conformant WebGPU implementation may take, for tools/tolerance_study.py only (never used as the oracle):
 *   bit 0: normalize(v) = (v.x / length(v), v.y / length(v))   two IEEE divisions, the literal reading
 *   bit 1: strain = (len - target) / length                    one IEEE division, the literal reading of :112
 *   bit 2: normalize(v) = v * inverseSqrt(dot(v, v))           the usual back-end lowering, inverseSqrt correctly rounded
 * (plus -ffp-contract=fast -mfma at build time for the "contraction allowed" variant, see oracle/Makefile). */
#ifndef SBO_VARIANT
#define SBO_VARIANT 0
#endif
static inline v2 v_normalize(v2 v)
{
#if SBO_VARIANT & 1
    float len = v_length(v);
    v2 r = { v.x / len, v.y / len };
#elif SBO_VARIANT & 4
    float inv = (float)(1.0 / sqrt((double)(v.x * v.x + v.y * v.y)));
    v2 r = { v.x * inv, v.y * inv };
#else
    float inv = 1.0f / v_length(v);
    v2 r = { v.x * inv, v.y * inv };
#endif
    return r;
}

int32_t sbo_f32_to_i32(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}

/* Deterministic pow for the drag term (compute.wgsl:175).  x >= 0 (it is
 * abs(v)), y = drag_exp (UI range 1..4, src/main.ts:132).  Small integer
 * exponents are exact products; everything else is exp2(y*log2(x)) evaluated
 * in binary64 with +,-,*,/ only, so the same source gives the same bits on any
 * IEEE machine (CPU or GPU), then rounded once to binary32. */
static double sbo_log2_d(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    int e = (int)((u >> 52) & 0x7ff);
    if (e == 0) { /* binary64 subnormal: cannot come from a float input, but be total */
        x = x * 18014398509481984.0; /* 2^54 */
        memcpy(&u, &x, 8);
        e = (int)((u >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &u, 8); /* [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double t = (m - 1.0) / (m + 1.0);
    double t2 = t * t;
    double s = 1.0 / 23.0;
    s = s * t2 + 1.0 / 21.0;
    s = s * t2 + 1.0 / 19.0;
    s = s * t2 + 1.0 / 17.0;
    s = s * t2 + 1.0 / 15.0;
    s = s * t2 + 1.0 / 13.0;
    s = s * t2 + 1.0 / 11.0;
    s = s * t2 + 1.0 / 9.0;
    s = s * t2 + 1.0 / 7.0;
    s = s * t2 + 1.0 / 5.0;
    s = s * t2 + 1.0 / 3.0;
    s = s * t2 + 1.0;
    double ln_m = 2.0 * t * s;
    return (double)e + ln_m * 1.4426950408889634; /* 1/ln 2 */
}

static double sbo_exp2_d(double x)
{
    if (x >= 1024.0) return (double)INFINITY;
    if (x <= -1100.0) return 0.0;
    long long n = (long long)x;
    if ((double)n > x) n -= 1; /* floor */
    double f = x - (double)n;  /* [0,1) */
    if (f > 0.5) { f = f - 1.0; n += 1; }
    double z = f * 0.6931471805599453; /* |z| <= 0.3466 */
    double s = 1.0 / 6227020800.0;     /* 1/13! */
    s = s * z + 1.0 / 479001600.0;
    s = s * z + 1.0 / 39916800.0;
    s = s * z + 1.0 / 3628800.0;
    s = s * z + 1.0 / 362880.0;
    s = s * z + 1.0 / 40320.0;
    s = s * z + 1.0 / 5040.0;
    s = s * z + 1.0 / 720.0;
    s = s * z + 1.0 / 120.0;
    s = s * z + 1.0 / 24.0;
    s = s * z + 1.0 / 6.0;
    s = s * z + 0.5;
    s = s * z + 1.0;
    s = s * z + 1.0;
    /* scale by 2^n in two exact steps to stay in range */
    long long n1 = n / 2, n2 = n - n1;
    uint64_t b1 = (uint64_t)(n1 + 1023) << 52, b2 = (uint64_t)(n2 + 1023) << 52;
    double p1, p2;
    memcpy(&p1, &b1, 8);
    memcpy(&p2, &b2, 8);
    return s * p1 * p2;
}

float sbo_pow(float x, float y)
{
    if (x != x || y != y) return x + y;
    if (y == 1.0f) return x;
    if (y == 2.0f) return x * x;
    if (y == 3.0f) return x * x * x;
    if (y == 4.0f) return (x * x) * (x * x);
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : INFINITY);
    if (x == INFINITY) return (y > 0.0f) ? INFINITY : ((y == 0.0f) ? 1.0f : 0.0f);
    return (float)sbo_exp2_d((double)y * sbo_log2_d((double)x));
}

/* ---------------------------------------------------------------- layout */

/* Metadata word offsets (compute.wgsl:29-54, engineMapping.ts:254-262) */
enum {
    MD_PARTICLE_I_C = 1, MD_BEAM_I_C = 6, MD_MAX_PARTICLES = 10, MD_MAX_BEAMS = 11,
    MD_GRAVITY = 12, MD_BORDER_ELASTICITY = 14, MD_BORDER_FRICTION = 15, MD_ELASTICITY = 16,
    MD_FRICTION = 17, MD_DRAG_COEFF = 18, MD_DRAG_EXP = 19, MD_USER_STRENGTH = 20,
    MD_MOUSE_ACTIVE = 21, MD_MOUSE_POS = 22, MD_MOUSE_VEL = 24, MD_APPLIED_FORCE = 26
};

typedef struct {
    uint32_t particle_i_c, beam_i_c, max_particles, max_beams;
    v2 gravity;
    float border_elasticity, border_friction, elasticity, friction, drag_coeff, drag_exp, user_strength;
    uint32_t mouse_active;
    v2 mouse_pos, mouse_vel, applied_force;
} meta_t;

static inline uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline float rd_f32(const uint8_t *p) { float v; memcpy(&v, p, 4); return v; }
static inline void wr_u32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
static inline void wr_f32(uint8_t *p, float v) { memcpy(p, &v, 4); }

static meta_t read_meta(const uint8_t *md)
{
    meta_t m;
    m.particle_i_c = rd_u32(md + 4 * MD_PARTICLE_I_C);
    m.beam_i_c = rd_u32(md + 4 * MD_BEAM_I_C);
    m.max_particles = rd_u32(md + 4 * MD_MAX_PARTICLES);
    m.max_beams = rd_u32(md + 4 * MD_MAX_BEAMS);
    m.gravity.x = rd_f32(md + 4 * MD_GRAVITY);
    m.gravity.y = rd_f32(md + 4 * MD_GRAVITY + 4);
    m.border_elasticity = rd_f32(md + 4 * MD_BORDER_ELASTICITY);
    m.border_friction = rd_f32(md + 4 * MD_BORDER_FRICTION);
    m.elasticity = rd_f32(md + 4 * MD_ELASTICITY);
    m.friction = rd_f32(md + 4 * MD_FRICTION);
    m.drag_coeff = rd_f32(md + 4 * MD_DRAG_COEFF);
    m.drag_exp = rd_f32(md + 4 * MD_DRAG_EXP);
    m.user_strength = rd_f32(md + 4 * MD_USER_STRENGTH);
    m.mouse_active = rd_u32(md + 4 * MD_MOUSE_ACTIVE);
    m.mouse_pos.x = rd_f32(md + 4 * MD_MOUSE_POS);
    m.mouse_pos.y = rd_f32(md + 4 * MD_MOUSE_POS + 4);
    m.mouse_vel.x = rd_f32(md + 4 * MD_MOUSE_VEL);
    m.mouse_vel.y = rd_f32(md + 4 * MD_MOUSE_VEL + 4);
    m.applied_force.x = rd_f32(md + 4 * MD_APPLIED_FORCE);
    m.applied_force.y = rd_f32(md + 4 * MD_APPLIED_FORCE + 4);
    return m;
}

/* get_mapped_index, compute.wgsl:78-81 (v1: u16 halves of packed u32; v2: u32) */
static inline uint32_t mapped_index(const sbo_params *prm, const uint8_t *mapping, uint32_t id)
{
    if (prm->layout == SBO_LAYOUT_V1) {
        uint16_t v;
        memcpy(&v, mapping + 2 * (size_t)id, 2);
        return v;
    }
    return rd_u32(mapping + 4 * (size_t)id);
}
static inline void set_mapped_index(const sbo_params *prm, uint8_t *mapping, uint32_t id, uint32_t val)
{
    if (prm->layout == SBO_LAYOUT_V1) {
        uint16_t v = (uint16_t)val;
        memcpy(mapping + 2 * (size_t)id, &v, 2);
    } else {
        wr_u32(mapping + 4 * (size_t)id, val);
    }
}

typedef struct { v2 p, v, a; } particle_t; /* compute.wgsl:10-14 */

static inline particle_t rd_particle(const uint8_t *buf, uint32_t index)
{
    particle_t q;
    memcpy(&q, buf + (size_t)index * SBO_PARTICLE_STRIDE, SBO_PARTICLE_STRIDE);
    return q;
}
static inline void wr_particle(uint8_t *buf, uint32_t index, particle_t q)
{
    memcpy(buf + (size_t)index * SBO_PARTICLE_STRIDE, &q, SBO_PARTICLE_STRIDE);
}

/* ---------------------------------------------------------------- beam half */

/* compute.wgsl:96-131 for one beam mapping slot */
static void beam_update(const sbo_params *prm, const meta_t *md, const uint8_t *particles_read,
                        uint8_t *beams, const uint8_t *mapping, int32_t *particle_forces,
                        uint32_t *delete_mappings, uint32_t beam_mapping_index)
{
    const float particle_force_scale = 65536.0f;   /* compute.wgsl:70 */
    const float beam_stress_scale = 1.0f / 20.0f;  /* compute.wgsl:71 */
    const int v1 = prm->layout == SBO_LAYOUT_V1;
    /* :97 */
    uint32_t index = mapped_index(prm, mapping, md->max_particles + beam_mapping_index);
    uint8_t *b = beams + (size_t)index * (v1 ? SBO_BEAM_STRIDE_V1 : SBO_BEAM_STRIDE_V2);
    uint32_t index_a, index_b;
    uint8_t *f; /* first float field (length) */
    if (v1) {   /* :99-100 */
        uint32_t pair = rd_u32(b);
        index_a = pair & 0xffffu;
        index_b = pair >> 16;
        f = b + 4;
    } else {
        index_a = rd_u32(b);
        index_b = rd_u32(b + 4);
        f = b + 8;
    }
    float length = rd_f32(f + 0), target_length = rd_f32(f + 4), last_length = rd_f32(f + 8);
    float spring = rd_f32(f + 12), damp = rd_f32(f + 16), yield_strain = rd_f32(f + 20);
    float strain_break_limit = rd_f32(f + 24);
    /* :101-103 */
    particle_t pa = rd_particle(particles_read, index_a);
    particle_t pb = rd_particle(particles_read, index_b);
    v2 diff = { pb.p.x - pa.p.x, pb.p.y - pa.p.y };
    /* :104-107 */
    if (v_length(diff) == 0.0f) {
        diff.x = 0.0f;
        diff.y = -1.0e-10f;
    }
    /* :108-112 */
    float len = v_length(diff);
    float force_mag = (target_length - len) * spring + (last_length - len) * damp;
    v2 n = v_normalize(diff);
    v2 force = { force_mag * n.x, force_mag * n.y };
#if SBO_VARIANT & 2
    float strain = (len - target_length) / length;
#else
    float strain = (len - target_length) * (1.0f / length); /* x / y pinned as x * (1/y), see header */
#endif
    /* :113-116 */
    if (f_abs(strain) > yield_strain) {
        target_length = len - yield_strain * length * f_sign(strain);
    }
    /* :117-121, mark_beam_deleted :86-88 */
    if (f_abs(len - length) > length * strain_break_limit) {
        uint32_t bit = md->max_particles + beam_mapping_index;
#ifdef _OPENMP
#pragma omp atomic
#endif
        delete_mappings[bit / 32u] |= 1u << (bit % 32u);
    }
    /* :122-125 */
    float stress = force_mag * beam_stress_scale;
    float strain_out = f_abs(strain) / yield_strain;
    wr_f32(f + 4, target_length);
    wr_f32(f + 8, len); /* last_length */
    wr_f32(f + 28, strain_out);
    wr_f32(f + 32, stress);
    /* :127-130; i32 atomicAdd wraps, so add as u32 */
    uint32_t *pf = (uint32_t *)particle_forces;
    uint32_t ax = (uint32_t)sbo_f32_to_i32(-force.x * particle_force_scale);
    uint32_t ay = (uint32_t)sbo_f32_to_i32(-force.y * particle_force_scale);
    uint32_t bx = (uint32_t)sbo_f32_to_i32(force.x * particle_force_scale);
    uint32_t by = (uint32_t)sbo_f32_to_i32(force.y * particle_force_scale);
#ifdef _OPENMP
#pragma omp atomic
#endif
    pf[(size_t)index_a * 2] += ax;
#ifdef _OPENMP
#pragma omp atomic
#endif
    pf[(size_t)index_a * 2 + 1] += ay;
#ifdef _OPENMP
#pragma omp atomic
#endif
    pf[(size_t)index_b * 2] += bx;
#ifdef _OPENMP
#pragma omp atomic
#endif
    pf[(size_t)index_b * 2 + 1] += by;
}

/* ---------------------------------------------------------------- particle half */

/* body of the collision loop for one (particle, other) pair, compute.wgsl:148-169 */
static inline void collide_pair(const sbo_params *prm, const meta_t *md, float elasticity_coeff,
                                particle_t *particle, const particle_t *const_particle,
                                uint32_t index, uint32_t other_index, const particle_t *other)
{
    v2 d = { other->p.x - const_particle->p.x, other->p.y - const_particle->p.y };
    float dist = v_length(d); /* :150 */
    if (dist == 0.0f) {
        /* :151-154 */
        particle->p.y += f_sign((float)index - (float)other_index);
    } else if (dist < prm->particle_radius * 2.0f) {
        /* :155-168 */
        v2 normal = v_normalize(d);
        v2 tangent = { -normal.y, normal.x };
        v2 inv_rel_velocity = { const_particle->v.x - other->v.x, const_particle->v.y - other->v.y };
        float impulse_normal = elasticity_coeff * v_dot(inv_rel_velocity, normal);
        float max_friction = impulse_normal * md->friction;
        float impulse_tangent = f_clamp(v_dot(inv_rel_velocity, tangent), -max_friction, max_friction);
        particle->v.x -= impulse_normal * normal.x + impulse_tangent * tangent.x;
        particle->v.y -= impulse_normal * normal.y + impulse_tangent * tangent.y;
        float overlap = prm->particle_radius * 2.0f - dist;
        v2 clip_shift = { normal.x * overlap / 2.0f, normal.y * overlap / 2.0f };
        float dt2 = prm->time_step * prm->time_step;
        particle->a.x -= clip_shift.x / dt2;
        particle->a.y -= clip_shift.y / dt2;
    }
}

/* compute.wgsl:171-201, everything after the collision loop */
static inline void particle_finish(const sbo_params *prm, const meta_t *md, particle_t *particle,
                                   uint32_t index, uint8_t *particles_write, int32_t *particle_forces)
{
    const float particle_force_scale = 65536.0f;
    /* :172 gravity */
    particle->a.x += md->gravity.x;
    particle->a.y += md->gravity.y;
    /* :174-176 drag */
    if (v_length(particle->v) > 0.0f) {
        v2 n = v_normalize(particle->v);
        float px = sbo_pow(f_abs(particle->v.x), md->drag_exp);
        float py = sbo_pow(f_abs(particle->v.y), md->drag_exp);
        particle->a.x -= md->drag_coeff * px * n.x;
        particle->a.y -= md->drag_coeff * py * n.y;
    }
    /* :178 */
    particle->a.x += md->applied_force.x * md->user_strength;
    particle->a.y += md->applied_force.y * md->user_strength;
    /* :179-181 */
    if (md->mouse_active > 0u) {
        v2 dm = { md->mouse_pos.x - particle->p.x, md->mouse_pos.y - particle->p.y };
        if (v_length(dm) < prm->particle_radius * 10.0f) {
            particle->a.x += (md->mouse_vel.x - particle->v.x) * md->user_strength - md->gravity.x;
            particle->a.y += (md->mouse_vel.y - particle->v.y) * md->user_strength - md->gravity.y;
        }
    }
    /* :183-185 atomicExchange(...,0) */
    size_t bfi = (size_t)index * 2;
    int32_t fx = particle_forces[bfi], fy = particle_forces[bfi + 1];
    particle_forces[bfi] = 0;
    particle_forces[bfi + 1] = 0;
    particle->a.x += (float)fx / particle_force_scale;
    particle->a.y += (float)fy / particle_force_scale;
    /* :186-188 */
    particle->v.x += particle->a.x * prm->time_step;
    particle->v.y += particle->a.y * prm->time_step;
    particle->p.x += particle->v.x * prm->time_step;
    particle->p.y += particle->v.y * prm->time_step;
    particle->a.x = 0.0f;
    particle->a.y = 0.0f;
    /* :190 */
    float lo = prm->particle_radius, hi = prm->bounds_size - prm->particle_radius;
    v2 clamped = { f_clamp(particle->p.x, lo, hi), f_clamp(particle->p.y, lo, hi) };
    /* :191-194 */
    if (particle->p.x != clamped.x) {
        particle->a.y -= f_min(particle->a.y, f_sign(particle->v.y) * md->border_friction *
                                                  f_abs(particle->v.x) * (1.0f + md->border_elasticity));
        particle->v.x *= -md->border_elasticity;
    }
    /* :195-198 */
    if (particle->p.y != clamped.y) {
        particle->a.x -= f_min(particle->a.x, f_sign(particle->v.x) * md->border_friction *
                                                  f_abs(particle->v.y) * (1.0f + md->border_elasticity));
        particle->v.y *= -md->border_elasticity;
    }
    /* :199-201 */
    particle->p = clamped;
    wr_particle(particles_write, index, *particle);
}

/* uniform grid over the slot list; a broad phase that yields a SUPERSET of
 * the interacting pairs of compute.wgsl:144-170 and applies them in the same
 * ascending-slot order, hence bit-identical to ALLPAIRS. */
typedef struct {
    uint32_t n;          /* cells per side */
    float cell;          /* cell edge, >= 2r with margin for the float division below */
    uint32_t *cell_start;/* n*n+1 */
    uint32_t *cell_slots;/* slots sorted by cell (stable => ascending slot inside a cell) */
} grid_t;

static inline uint32_t grid_coord(const grid_t *g, float x)
{
    float q = x / g->cell;
    if (!(q > 0.0f)) return 0; /* negative, zero, NaN */
    if (q >= (float)g->n) return g->n - 1;
    return (uint32_t)q;
}

static int grid_build(grid_t *g, const sbo_params *prm, const meta_t *md,
                      const uint8_t *particles_read, const uint8_t *mapping)
{
    g->cell = prm->particle_radius * 2.0f * 1.015625f;
    float nf = prm->bounds_size / g->cell;
    uint32_t n = (nf >= 1.0f && nf < 32768.0f) ? (uint32_t)nf + 1u : (nf >= 32768.0f ? 32768u : 1u);
    g->n = n;
    size_t ncell = (size_t)n * n;
    g->cell_start = (uint32_t *)calloc(ncell + 1, sizeof(uint32_t));
    g->cell_slots = (uint32_t *)malloc(sizeof(uint32_t) * (md->particle_i_c ? md->particle_i_c : 1));
    uint32_t *cell_of = (uint32_t *)malloc(sizeof(uint32_t) * (md->particle_i_c ? md->particle_i_c : 1));
    if (!g->cell_start || !g->cell_slots || !cell_of) return -1;
    for (uint32_t s = 0; s < md->particle_i_c; s++) {
        particle_t q = rd_particle(particles_read, mapped_index(prm, mapping, s));
        uint32_t c = grid_coord(g, q.p.y) * n + grid_coord(g, q.p.x);
        cell_of[s] = c;
        g->cell_start[c + 1]++;
    }
    for (size_t c = 0; c < ncell; c++) g->cell_start[c + 1] += g->cell_start[c];
    uint32_t *cursor = (uint32_t *)malloc(sizeof(uint32_t) * ncell);
    if (!cursor) return -1;
    memcpy(cursor, g->cell_start, sizeof(uint32_t) * ncell);
    for (uint32_t s = 0; s < md->particle_i_c; s++) g->cell_slots[cursor[cell_of[s]]++] = s;
    free(cursor);
    free(cell_of);
    return 0;
}

static void grid_free(grid_t *g)
{
    free(g->cell_start);
    free(g->cell_slots);
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* compute.wgsl:134-202 for one particle mapping slot */
static void particle_update(const sbo_params *prm, const meta_t *md, const uint8_t *particles_read,
                            uint8_t *particles_write, const uint8_t *mapping, int32_t *particle_forces,
                            const grid_t *grid, uint32_t particle_mapping_index)
{
    uint32_t index = mapped_index(prm, mapping, particle_mapping_index); /* :136 */
    particle_t particle = rd_particle(particles_read, index);            /* :139 */
    const particle_t const_particle = particle;                          /* :141 */
    float elasticity_coeff = (md->elasticity + 1.0f) / 2.0f;             /* :143 */
    if (prm->collision_mode == SBO_COLLIDE_ALLPAIRS) {
        /* :144-170 */
        for (uint32_t o = 0; o < md->particle_i_c; o++) {
            if (o == particle_mapping_index) continue;
            uint32_t other_index = mapped_index(prm, mapping, o);
            particle_t other = rd_particle(particles_read, other_index);
            collide_pair(prm, md, elasticity_coeff, &particle, &const_particle, index, other_index, &other);
        }
    } else if (prm->collision_mode == SBO_COLLIDE_GRID) {
        uint32_t cx = grid_coord(grid, const_particle.p.x), cy = grid_coord(grid, const_particle.p.y);
        uint32_t stack_buf[64], *cand = stack_buf, ncand = 0, cap = 64;
        for (int dy = -1; dy <= 1; dy++) {
            int yy = (int)cy + dy;
            if (yy < 0 || yy >= (int)grid->n) continue;
            int x0 = (int)cx - 1 < 0 ? 0 : (int)cx - 1;
            int x1 = (int)cx + 1 >= (int)grid->n ? (int)grid->n - 1 : (int)cx + 1;
            uint32_t b = grid->cell_start[(size_t)yy * grid->n + x0];
            uint32_t e = grid->cell_start[(size_t)yy * grid->n + x1 + 1];
            for (uint32_t k = b; k < e; k++) {
                uint32_t o = grid->cell_slots[k];
                if (o == particle_mapping_index) continue;
                if (ncand == cap) {
                    uint32_t *nb = (uint32_t *)malloc(sizeof(uint32_t) * cap * 2);
                    memcpy(nb, cand, sizeof(uint32_t) * cap);
                    if (cand != stack_buf) free(cand);
                    cand = nb;
                    cap *= 2;
                }
                cand[ncand++] = o;
            }
        }
        qsort(cand, ncand, sizeof(uint32_t), cmp_u32);
        for (uint32_t k = 0; k < ncand; k++) {
            uint32_t other_index = mapped_index(prm, mapping, cand[k]);
            particle_t other = rd_particle(particles_read, other_index);
            collide_pair(prm, md, elasticity_coeff, &particle, &const_particle, index, other_index, &other);
        }
        if (cand != stack_buf) free(cand);
    }
    particle_finish(prm, md, &particle, index, particles_write, particle_forces);
}

/* ---------------------------------------------------------------- entry points */

void sbo_update(const sbo_params *prm, const uint8_t *metadata, const uint8_t *particles_read,
                uint8_t *particles_write, uint8_t *beams, const uint8_t *mapping,
                int32_t *particle_forces, uint32_t *delete_mappings)
{
    const meta_t md = read_meta(metadata);
    int nthreads = prm->threads > 1 ? prm->threads : 1;
    (void)nthreads;
    /* S0 phase 1: every beam (compute.wgsl:96-131) */
    long long nb = md.beam_i_c;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
#endif
    for (long long i = 0; i < nb; i++)
        beam_update(prm, &md, particles_read, beams, mapping, particle_forces, delete_mappings, (uint32_t)i);
    /* S0 phase 2: every particle (compute.wgsl:134-202) */
    grid_t grid;
    memset(&grid, 0, sizeof grid);
    if (prm->collision_mode == SBO_COLLIDE_GRID) {
        if (grid_build(&grid, prm, &md, particles_read, mapping) != 0) abort();
    }
    long long np = md.particle_i_c;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads) if (nthreads > 1)
#endif
    for (long long i = 0; i < np; i++)
        particle_update(prm, &md, particles_read, particles_write, mapping, particle_forces, &grid, (uint32_t)i);
    if (prm->collision_mode == SBO_COLLIDE_GRID) grid_free(&grid);
}

void sbo_delete(const sbo_params *prm, uint8_t *metadata, uint8_t *mapping,
                uint32_t *delete_mappings, size_t delete_words)
{
    /* compute.wgsl:205-246, INTENT (the author's comment at :220 says the
     * shader "doesnt work at all"; SURVEY A7 fixes the canonical semantics):
     * in-place forward stable compaction of the flagged beam slots.  Only
     * beams are ever flagged (mark_particle_deleted :83-85 has no caller). */
    meta_t md = read_meta(metadata);
    uint32_t w = 0;
    for (uint32_t s = 0; s < md.beam_i_c; s++) {
        uint32_t bit = md.max_particles + s;
        int flagged = (bit / 32u < delete_words) && ((delete_mappings[bit / 32u] >> (bit % 32u)) & 1u);
        if (!flagged) {
            if (w != s)
                set_mapped_index(prm, mapping, md.max_particles + w,
                                 mapped_index(prm, mapping, md.max_particles + s));
            w++;
        }
    }
    wr_u32(metadata + 4 * MD_BEAM_I_C, w); /* :238 */
    memset(delete_mappings, 0, delete_words * sizeof(uint32_t)); /* :241-244 */
}

int sbo_step(const sbo_params *prm, const uint8_t *metadata, uint8_t *particles_a,
             uint8_t *particles_b, uint8_t *beams, const uint8_t *mapping,
             int32_t *particle_forces, uint32_t *delete_mappings, uint32_t n_substeps)
{
    /* engineWorker.ts:655-661: even i reads A writes B, odd i reads B writes A */
    for (uint32_t i = 0; i < n_substeps; i++) {
        if (i % 2 == 0)
            sbo_update(prm, metadata, particles_a, particles_b, beams, mapping, particle_forces, delete_mappings);
        else
            sbo_update(prm, metadata, particles_b, particles_a, beams, mapping, particle_forces, delete_mappings);
    }
    return (int)(n_substeps % 2);
}

void sbo_frame(const sbo_params *prm, uint8_t *metadata, uint8_t *particles_a, uint8_t *particles_b,
               uint8_t *beams, uint8_t *mapping, int32_t *particle_forces,
               uint32_t *delete_mappings, size_t delete_words, uint32_t subticks)
{
    sbo_step(prm, metadata, particles_a, particles_b, beams, mapping, particle_forces, delete_mappings, subticks);
    sbo_delete(prm, metadata, mapping, delete_mappings, delete_words); /* engineWorker.ts:663-664 */
}
