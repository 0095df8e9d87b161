// sb_physics.h -- device-side arithmetic of one substep (gfx950).
//
// Follows /root/reference/src/shaders/compute.wgsl line by line (cited per block) in the
// canonical arithmetic DESIGN.md fixes: IEEE binary32, round-to-nearest-even, one rounding
// per WGSL operator (build with -ffp-contract=off), correctly rounded sqrt and divide,
// left-to-right evaluation as the WGSL source writes it.  Integer force accumulation
// (compute.wgsl:68-71,127-130) is exact and order-free, which is what lets the tiled path
// reduce forces in LDS without changing a bit of the result.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SB_DEV __device__ __forceinline__

// metadata bytes 48..111 (compute.wgsl:42-53, engineMapping.ts:260-262), uploaded verbatim
struct SbConsts {
    float gravity_x, gravity_y;
    float border_elasticity, border_friction, elasticity, friction, drag_coeff, drag_exp;
    float user_strength;
    uint32_t mouse_active;
    float mouse_pos_x, mouse_pos_y, mouse_vel_x, mouse_vel_y, applied_force_x, applied_force_y;
};
static_assert(sizeof(SbConsts) == 64, "metadata tail is 64 bytes");

// pipeline-overridable constants (compute.wgsl:1-3, engineWorker.ts:328-332)
struct SbParams {
    float bounds_size, particle_radius, time_step;
};

SB_DEV float sb_min(float a, float b) { return (b < a) ? b : a; }
SB_DEV float sb_max(float a, float b) { return (a < b) ? b : a; }
SB_DEV float sb_clamp(float x, float lo, float hi) { return sb_min(sb_max(x, lo), hi); }
SB_DEV float sb_sign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
SB_DEV float sb_abs(float x) { return __uint_as_float(__float_as_uint(x) & 0x7fffffffu); }
// Correctly rounded sqrt and divide.  On ROCm 7.2 `__fsqrt_rn` lowers to the bare 1-ulp
// v_sqrt_f32; `__builtin_sqrtf` / operator `/` under -fhip-fp32-correctly-rounded-divide-sqrt
// lower to the IEEE sequences (v_sqrt + residual fix-up; v_div_scale/fmas/fixup).  Checked in
// the .s and by the bit-exact parity tests.
#ifndef SB_ABLATE
#define SB_ABLATE 0
#endif
#if SB_ABLATE & 16 // diagnostic build: raw 1-ulp v_sqrt_f32
SB_DEV float sb_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
SB_DEV float sb_sqrt(float x) { return __builtin_sqrtf(x); }
#endif
#if SB_ABLATE & 8 // diagnostic build: divide = multiply by the raw v_rcp_f32
SB_DEV float sb_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
#else
SB_DEV float sb_div(float a, float b) { return a / b; }
#endif
SB_DEV float sb_length(float x, float y) { return sb_sqrt(x * x + y * y); }

// i32(f): truncate toward zero, saturating, NaN -> 0 (compute.wgsl:127-130).  That is exactly what
// one v_cvt_i32_f32 does on gfx950 (out-of-range clamps to INT_MIN/INT_MAX, NaN gives 0); spelled as
// inline asm so neither a chain of range checks nor a UB-exploiting fold of `(int)x` can appear.
SB_DEV int32_t sb_f32_to_i32(float x)
{
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// pow for the drag term (compute.wgsl:175).  Exact products for exponents 1..4, otherwise
// exp2(y*log2 x) in binary64 from +,-,*,/ only (bit-reproducible on any IEEE machine).
SB_DEV double sb_log2_d(double x)
{
    uint64_t u = (uint64_t)__double_as_longlong(x);
    int e = (int)((u >> 52) & 0x7ff);
    if (e == 0) {
        x = x * 18014398509481984.0;
        u = (uint64_t)__double_as_longlong(x);
        e = (int)((u >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = __longlong_as_double((long long)u);
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
    return (double)e + ln_m * 1.4426950408889634;
}

SB_DEV double sb_exp2_d(double x)
{
    if (x >= 1024.0) return __longlong_as_double(0x7ff0000000000000LL);
    if (x <= -1100.0) return 0.0;
    long long n = (long long)x;
    if ((double)n > x) n -= 1;
    double f = x - (double)n;
    if (f > 0.5) { f = f - 1.0; n += 1; }
    double z = f * 0.6931471805599453;
    double s = 1.0 / 6227020800.0;
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
    long long n1 = n / 2, n2 = n - n1;
    double p1 = __longlong_as_double((long long)((uint64_t)(n1 + 1023) << 52));
    double p2 = __longlong_as_double((long long)((uint64_t)(n2 + 1023) << 52));
    return s * p1 * p2;
}

SB_DEV float sb_pow(float x, float y)
{
    if (x != x || y != y) return x + y;
    if (y == 1.0f) return x;
    if (y == 2.0f) return x * x;
    if (y == 3.0f) return x * x * x;
    if (y == 4.0f) return (x * x) * (x * x);
    const float inf = __uint_as_float(0x7f800000u);
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : inf);
    if (x == inf) return (y > 0.0f) ? inf : ((y == 0.0f) ? 1.0f : 0.0f);
    return (float)sb_exp2_d((double)y * sb_log2_d((double)x));
}

// ---------------------------------------------------------------- beam (compute.wgsl:103-130)

struct SbBeamResult {
    float target_length, last_length, strain, stress;
    int32_t ax, ay, bx, by; // fixed-point contributions to endpoint A and endpoint B
    bool broken;            // mark_beam_deleted condition (:117)
};

// AUX = also produce strain/stress (compute.wgsl:122-123: outputs nobody reads before the caller
// gets control back).  Without AUX the strain division is only executed for beams close to or
// past their yield point: |len-target| <= 0.999*yield*length implies fl((len-target)/length) <= yield
// for every input (the filter is conservative; NaN/inf/zero parameters fall through to the exact
// path or agree with it), so the yield decision is bit-identical to evaluating :112-113 as written.
template <bool AUX>
SB_DEV SbBeamResult sb_beam_eval(float2 pa, float2 pb, float length, float target_length,
                                 float last_length, float spring, float damp, float yield_strain,
                                 float strain_break_limit)
{
    const float particle_force_scale = 65536.0f; // :70
    const float beam_stress_scale = 1.0f / 20.0f; // :71
    SbBeamResult r;
    float dx = pb.x - pa.x, dy = pb.y - pa.y; // :103
    float len = sb_length(dx, dy);            // :104 / :108 (same value unless the guard fires)
    if (len == 0.0f) {                        // :104-107
        dx = 0.0f;
        dy = -1.0e-10f;
        len = sb_length(0.0f, -1.0e-10f);     // folded at compile time, correctly rounded
    }
    float force_mag = (target_length - len) * spring + (last_length - len) * damp; // :110
    float nx = sb_div(dx, len), ny = sb_div(dy, len); // normalize(diff)
    float fx = force_mag * nx, fy = force_mag * ny;   // :111
    const float stretch = len - target_length;
    r.target_length = target_length;
    r.strain = 0.0f;
    r.stress = 0.0f;
    if (AUX || sb_abs(stretch) > yield_strain * length * 0.999f) {
        float strain = sb_div(stretch, length);  // :112
        if (sb_abs(strain) > yield_strain)       // :113-116
            r.target_length = len - yield_strain * length * sb_sign(strain);
        if (AUX) {
            r.stress = force_mag * beam_stress_scale;        // :122
            r.strain = sb_div(sb_abs(strain), yield_strain); // :123
        }
    }
    r.broken = sb_abs(len - length) > length * strain_break_limit; // :117
    r.last_length = len;                                           // :124
    const float sx = fx * particle_force_scale, sy = fy * particle_force_scale;
    r.bx = sb_f32_to_i32(sx);  // :129
    r.by = sb_f32_to_i32(sy);  // :130
    r.ax = sb_f32_to_i32(-sx); // :127  (-f*s == -(f*s) exactly)
    r.ay = sb_f32_to_i32(-sy); // :128
    return r;
}

// ---------------------------------------------------------------- particle (compute.wgsl:139-201)

struct SbParticle {
    float2 p, v, a;
};

// one iteration of the collision loop, compute.wgsl:148-169.  `self` is const_particle (:141).
SB_DEV void sb_collide_pair(const SbParams &prm, float friction, float elasticity_coeff,
                            SbParticle &particle, const SbParticle &self, uint32_t index,
                            uint32_t other_index, float2 op, float2 ov)
{
    float dx = op.x - self.p.x, dy = op.y - self.p.y;
    float dist = sb_length(dx, dy); // :150
    if (dist == 0.0f) {             // :151-154
        particle.p.y += sb_sign((float)index - (float)other_index);
    } else if (dist < prm.particle_radius * 2.0f) { // :155
        float nx = sb_div(dx, dist), ny = sb_div(dy, dist);   // :156
        float tx = -ny, ty = nx;                              // :157
        float ux = self.v.x - ov.x, uy = self.v.y - ov.y;     // :158
        float impulse_normal = elasticity_coeff * (ux * nx + uy * ny); // :159
        float max_friction = impulse_normal * friction;               // :160
        float impulse_tangent = sb_clamp(ux * tx + uy * ty, -max_friction, max_friction); // :161
        particle.v.x -= impulse_normal * nx + impulse_tangent * tx;   // :162
        particle.v.y -= impulse_normal * ny + impulse_tangent * ty;
        float overlap = prm.particle_radius * 2.0f - dist;            // :164
        float csx = sb_div(nx * overlap, 2.0f), csy = sb_div(ny * overlap, 2.0f);
        float dt2 = prm.time_step * prm.time_step;                    // :168
        particle.a.x -= sb_div(csx, dt2);
        particle.a.y -= sb_div(csy, dt2);
    }
}

// ---------------------------------------------------------------- spatial hash (SB_COLLIDE_GRID)

// Uniform grid over [x0, x0 + nx*cell) x [y0, y0 + ny*cell); coordinates outside are clamped into
// the edge cells.  The map x -> cell coordinate is monotone and cell >= 2r * (1 + 1/64), so two
// particles closer than 2r always land in the same or adjacent cells: the 3x3 neighbourhood is a
// SUPERSET of the interacting pairs of compute.wgsl:144-170 wherever the particles are (clamping
// only piles far-away particles into edge cells: slower, never wrong).
struct SbGrid {
    const uint32_t *cell_scan; // per cell: exclusive scan inside its 2048-cell block
    const uint32_t *block_off; // per 2048-cell block: offset of the block
    const uint2 *rec;          // sorted by cell: {slot, internal index}
    const uint32_t *cell_of;   // per particle: its cell at the last rebuild
    float x0, y0, cell;
    uint32_t nx, ny;
};
// The hash is rebuilt only when needed: cells are 2r*(1+1/64) + 2*skin wide, and a rebuild happens as
// soon as the sum of per-substep maximum displacements since the last build exceeds skin.  Until then
// every particle is within skin of where it was binned, so two particles closer than 2r NOW were closer
// than 2r + 2*skin THEN and sit in the same or adjacent cells of the (stale) binning: still a superset.
// Candidates are therefore looked up in the cells of the last build but tested at their CURRENT positions.
struct SbGridCtl {
    uint32_t rebuild;     // 1 while the build kernels of this substep must run
    uint32_t force;       // set by the host (upload, halo unpack): rebuild unconditionally
    uint32_t step_max;    // float bits: max displacement of any particle in the substep just run
    float accum;          // sum of step maxima since the last build
    float skin;
    uint32_t builds;      // statistics
};
#define SB_SCAN_BLOCK 2048u

// largest displacement of any particle in this substep -> ctl->step_max (positive float bits order
// like unsigned integers; anything not provably small, NaN included, reads as "huge")
SB_DEV void sb_track_displacement(SbGridCtl *ctl, float m)
{
    m = (m < 1.0e30f) ? m : 1.0e30f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63u) == 0u && m > 0.0f) atomicMax(&ctl->step_max, __float_as_uint(m));
}

SB_DEV uint32_t sb_grid_coord(float x, float x0, float cell, uint32_t n)
{
    float q = sb_div(x - x0, cell);
    if (!(q > 0.0f)) return 0u; // negative, zero, NaN
    if (q >= (float)n) return n - 1u;
    return (uint32_t)q;
}
SB_DEV uint32_t sb_grid_start(const SbGrid &g, uint32_t c) { return g.cell_scan[c] + g.block_off[c / SB_SCAN_BLOCK]; }

// The collision loop of compute.wgsl:144-170 restricted to the 3x3 cell neighbourhood, applying
// contacts in ASCENDING SLOT ORDER exactly like the all-pairs scan does: repeatedly pick the
// contact with the smallest slot above the last one applied.  Non-contacts are no-ops in the
// reference loop, so skipping them changes nothing; the result is bit-identical to all-pairs.
SB_DEV void sb_collide_grid(const SbGrid &g, const SbParams &prm, float friction, float elasticity_coeff,
                            SbParticle &particle, const SbParticle &self, uint32_t i,
                            const uint32_t *__restrict__ pidx, const float2 *__restrict__ pos_r,
                            const float2 *__restrict__ vel_r)
{
    const float two_r = prm.particle_radius * 2.0f;
    const float far2 = two_r * two_r * 1.001f;
    const uint32_t cell = g.cell_of[i]; // where this particle was binned at the last rebuild
    const uint32_t cx = cell % g.nx, cy = cell / g.nx;
    const uint32_t xa = cx > 0u ? cx - 1u : 0u, xb = cx + 1u < g.nx ? cx + 1u : g.nx - 1u;
    uint32_t rb[3], re[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int yy = (int)cy + r - 1;
        if (yy < 0 || yy >= (int)g.ny) {
            rb[r] = re[r] = 0u;
        } else {
            rb[r] = sb_grid_start(g, (uint32_t)yy * g.nx + xa);
            re[r] = sb_grid_start(g, (uint32_t)yy * g.nx + xb + 1u); // cell array has one spare entry
        }
    }
    bool have_last = false;
    uint32_t last = 0u;
    for (;;) {
        uint32_t best_slot = 0xFFFFFFFFu, best_id = 0u;
        float2 best_p = make_float2(0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            for (uint32_t k = rb[r]; k < re[r]; k++) {
                const uint2 rec = g.rec[k];
                const uint32_t slot = rec.x, id = rec.y;
                if (id == i || (have_last && slot <= last) || slot >= best_slot) continue;
                const float2 q = pos_r[id];
                const float ex = q.x - self.p.x, ey = q.y - self.p.y;
                const float d2 = ex * ex + ey * ey; // exactly the argument length() takes the root of
                // sqrt is monotone: d2 clearly above (2r)^2 cannot give d < 2r (and is not 0), so the
                // correctly rounded root is only evaluated for the few candidates near contact range
                if (d2 > far2) continue;
                const float d = sb_sqrt(d2);
                if (d == 0.0f || d < two_r) {
                    best_slot = slot;
                    best_id = id;
                    best_p = q;
                }
            }
        }
        if (best_slot == 0xFFFFFFFFu) break;
        sb_collide_pair(prm, friction, elasticity_coeff, particle, self, pidx[i], pidx[best_id], best_p, vel_r[best_id]);
        last = best_slot;
        have_last = true;
    }
}

// everything after the collision loop, compute.wgsl:171-199; fx,fy are the complete fixed-point
// beam force sums for this particle (the atomicExchange results of :184-185).
SB_DEV void sb_particle_finish(const SbParams &prm, const SbConsts &c, SbParticle &particle,
                               int32_t fx, int32_t fy)
{
    const float particle_force_scale = 65536.0f;
    particle.a.x += c.gravity_x; // :172
    particle.a.y += c.gravity_y;
    float vl = sb_length(particle.v.x, particle.v.y);
    if (vl > 0.0f) { // :174-176
        float nx = sb_div(particle.v.x, vl), ny = sb_div(particle.v.y, vl);
        float px = sb_pow(sb_abs(particle.v.x), c.drag_exp);
        float py = sb_pow(sb_abs(particle.v.y), c.drag_exp);
        particle.a.x -= c.drag_coeff * px * nx;
        particle.a.y -= c.drag_coeff * py * ny;
    }
    particle.a.x += c.applied_force_x * c.user_strength; // :178
    particle.a.y += c.applied_force_y * c.user_strength;
    if (c.mouse_active > 0u) { // :179-181
        float mx = c.mouse_pos_x - particle.p.x, my = c.mouse_pos_y - particle.p.y;
        if (sb_length(mx, my) < prm.particle_radius * 10.0f) {
            particle.a.x += (c.mouse_vel_x - particle.v.x) * c.user_strength - c.gravity_x;
            particle.a.y += (c.mouse_vel_y - particle.v.y) * c.user_strength - c.gravity_y;
        }
    }
    particle.a.x += sb_div((float)fx, particle_force_scale); // :184-185
    particle.a.y += sb_div((float)fy, particle_force_scale);
    particle.v.x += particle.a.x * prm.time_step; // :186
    particle.v.y += particle.a.y * prm.time_step;
    particle.p.x += particle.v.x * prm.time_step; // :187
    particle.p.y += particle.v.y * prm.time_step;
    particle.a.x = 0.0f; // :188
    particle.a.y = 0.0f;
    float lo = prm.particle_radius, hi = prm.bounds_size - prm.particle_radius; // :190
    float cx = sb_clamp(particle.p.x, lo, hi), cy = sb_clamp(particle.p.y, lo, hi);
    if (particle.p.x != cx) { // :191-194
        particle.a.y -= sb_min(particle.a.y, sb_sign(particle.v.y) * c.border_friction *
                                                 sb_abs(particle.v.x) * (1.0f + c.border_elasticity));
        particle.v.x *= -c.border_elasticity;
    }
    if (particle.p.y != cy) { // :195-198
        particle.a.x -= sb_min(particle.a.x, sb_sign(particle.v.x) * c.border_friction *
                                                 sb_abs(particle.v.y) * (1.0f + c.border_elasticity));
        particle.v.y *= -c.border_elasticity;
    }
    particle.p.x = cx; // :199
    particle.p.y = cy;
}
