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
    // time_step * time_step is a power of two whenever subticks is (the default 64 -> 2^-12): dividing by it (compute.wgsl:168)
    // is then the exact multiplication by inv_dt2, bit for bit the IEEE quotient.  inv_dt2 = 0 means "divide".
    float inv_dt2;
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

// ---------------------------------------------------------------- short correctly rounded sqrt / reciprocal
// The IEEE sequences the compiler emits cost ~15 (sqrt: range scaling, v_sqrt_f32, two neighbour residuals,
// class fix-up) and ~11 (1/x: v_div_scale x2, v_rcp_f32, five fmas, v_div_fmas, v_div_fixup) instructions, about a
// quarter of the arithmetic of a beam.  Away from the ends of the exponent range much shorter sequences return the
// same bits; "the same bits" is not an estimate: tools/exact_math_check.hip runs every one of the 2^32 binary32
// inputs through both on the GPU (gfx950: 1 509 949 441 inputs in the sqrt gate, 754 974 721 in the reciprocal
// gate, 0 mismatches each; the log is profiles/r02_exact_math_check.txt).
//   sqrt: y = v_rsq_f32(x); g = x*y; h = y/2; g' = fma(fma(-g, g, x), h, g)        2^-90 <= x <= 2^90
//   1/x : y = v_rcp_f32(x); y' = fma(y, fma(-x, y, 1), y)                          2^-45 <= x <= 2^45
// (the reciprocal gate holds every sqrt the sqrt gate can return).  Callers test the gate for the whole wave and take
// the IEEE sequences otherwise, so zero, subnormal, huge, infinite and NaN operands never reach the short forms.
// "Exhaustively checked" is a statement about ONE instruction set: v_rsq_f32 / v_rcp_f32 are only ULP-bounded by the ISA, and
// another target's tables may differ.  A device build for anything but gfx950 therefore stops here instead of silently
// resting every bit-exact guarantee on unverified sequences (run tools/exact_math_check.hip on the new target, commit its
// log under profiles/, then add the target to this list).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "sb_sqrt_gated / sb_rcp_gated are verified bit-exact on gfx950 only (profiles/r02_exact_math_check.txt)"
#endif
SB_DEV bool sb_in_sqrt_gate(float x) { return x >= 0x1p-90f && x <= 0x1p90f; } // false for NaN
SB_DEV float sb_sqrt_gated(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}
SB_DEV float sb_rcp_gated(float x)
{
    const float y = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(y, __builtin_fmaf(-x, y, 1.0f), y);
}
// true when EVERY active lane of the wave is inside the gate (one s_cmp on the ballot: the branch is wave-uniform)
SB_DEV bool sb_wave_all(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0ull; }

// Write-through stores for what a substep kernel streams out (particles, beam state).  A plain store leaves its line dirty in
// the XCD's L2, and the release at the end of the kernel writes all of them back at once, in front of the next launch: a launch
// boundary behind N dirty megabytes costs N / 6 TB/s more (MI355X guide, "boundary").  Agent-scope stores (sc1) leave during
// the kernel instead.  r04, 1 M particles: the single-substep kernel with the hash on 23.2 -> 22.2 us per substep.
SB_DEV void sb_store_wt(float2 *p, float2 v)
{
    __hip_atomic_store((unsigned long long *)p, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
SB_DEV void sb_store_wt(float *p, float v) { __hip_atomic_store((uint32_t *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// i32(f): truncate toward zero, saturating, NaN -> 0 (compute.wgsl:127-130).  That is exactly what
// one v_cvt_i32_f32 does on gfx950 (out-of-range clamps to INT_MIN/INT_MAX, NaN gives 0); spelled as
// inline asm so neither a chain of range checks nor a UB-exploiting fold of `(int)x` can appear.
SB_DEV int32_t sb_f32_to_i32(float x)
{
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// i32(-f) with the negation folded into the conversion's source modifier (one instruction instead of two)
SB_DEV int32_t sb_f32_to_i32_neg(float x)
{
    int32_t r;
    asm("v_cvt_i32_f32_e64 %0, -%1" : "=v"(r) : "v"(x));
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

// the general-exponent path is ~300 instructions of binary64 arithmetic that almost no scene runs
// (drag_exp defaults to 2): kept out of line so it is not replicated into every unrolled call site
__device__ __attribute__((noinline)) float sb_pow_general(float x, float y)
{
    const float inf = __uint_as_float(0x7f800000u);
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : inf);
    if (x == inf) return (y > 0.0f) ? inf : ((y == 0.0f) ? 1.0f : 0.0f);
    return (float)sb_exp2_d((double)y * sb_log2_d((double)x));
}

SB_DEV float sb_pow(float x, float y)
{
    if (x != x || y != y) return x + y;
    if (y == 1.0f) return x;
    if (y == 2.0f) return x * x;
    if (y == 3.0f) return x * x * x;
    if (y == 4.0f) return (x * x) * (x * x);
    return sb_pow_general(x, y);
}

// ---------------------------------------------------------------- beam (compute.wgsl:103-130)

struct SbBeamResult {
    float target_length, last_length, strain, stress;
    int32_t ax, ay, bx, by; // fixed-point contributions to endpoint A and endpoint B
    bool broken;            // mark_beam_deleted condition (:117)
};

// AUX = also produce strain/stress (compute.wgsl:122-123: outputs nobody reads before the caller
// gets control back).  Branch-free (selects only): the kernels evaluate a batch of beams per thread
// in straight-line code so the scheduler can interleave their dependency chains -- the IEEE sqrt and
// reciprocal sequences are long serial chains, and a lone chain issues one VALU instruction every
// ~4 cycles where interleaved chains issue one every 2.  inv_length = 1 / length (IEEE), computed
// once per material at upload (or per beam when rest lengths are not in the table).
template <bool AUX>
SB_DEV SbBeamResult sb_beam_eval(float2 pa, float2 pb, float length, float inv_length, float target_length,
                                 float last_length, float spring, float damp, float yield_strain,
                                 float strain_break_limit)
{
    const float particle_force_scale = 65536.0f; // :70
    const float beam_stress_scale = 1.0f / 20.0f; // :71
    SbBeamResult r;
    float dx = pb.x - pa.x, dy = pb.y - pa.y; // :103
    const float len2 = dx * dx + dy * dy;
    float len, inv_len;
    if (sb_wave_all(sb_in_sqrt_gate(len2))) { // every beam of the wave has an ordinary length: short exact forms
        len = sb_sqrt_gated(len2);            // :108 (the guard of :104 cannot fire: len2 >= 2^-90)
        inv_len = sb_rcp_gated(len);
    } else {
        const float len0 = sb_sqrt(len2);     // :104 / :108 (same value unless the guard fires)
        const bool degenerate = len0 == 0.0f; // :104-107
        dx = degenerate ? 0.0f : dx;
        dy = degenerate ? -1.0e-10f : dy;
        len = degenerate ? sb_length(0.0f, -1.0e-10f) : len0; // constant folded, correctly rounded
        inv_len = sb_div(1.0f, len);
    }
    const float force_mag = (target_length - len) * spring + (last_length - len) * damp; // :110
    const float nx = dx * inv_len, ny = dy * inv_len;     // normalize(diff) = diff * (1 / length(diff)), DESIGN.md 2
    const float fx = force_mag * nx, fy = force_mag * ny; // :111
    const float strain = (len - target_length) * inv_length; // :112, x / y pinned as x * (1 / y)
    const float yielded_target = len - yield_strain * length * sb_sign(strain); // :115
    r.target_length = (sb_abs(strain) > yield_strain) ? yielded_target : target_length; // :113-116
    r.broken = sb_abs(len - length) > length * strain_break_limit; // :117
    r.stress = AUX ? force_mag * beam_stress_scale : 0.0f;          // :122
    r.strain = AUX ? sb_div(sb_abs(strain), yield_strain) : 0.0f;   // :123
    r.last_length = len;                                            // :124
    const float sx = fx * particle_force_scale, sy = fy * particle_force_scale;
    r.bx = sb_f32_to_i32(sx);  // :129
    r.by = sb_f32_to_i32(sy);  // :130
    r.ax = sb_f32_to_i32_neg(sx); // :127  (-f*s == -(f*s) exactly)
    r.ay = sb_f32_to_i32_neg(sy); // :128
    return r;
}

// length of a beam exactly as sb_beam_eval / sb_beam_group compute it (:103-108): whichever branch is taken returns the
// same bits on the gate's domain, so a caller may take its own
SB_DEV float sb_beam_length(float2 pa, float2 pb)
{
    const float dx = pb.x - pa.x, dy = pb.y - pa.y;
    const float len2 = dx * dx + dy * dy;
    if (sb_wave_all(sb_in_sqrt_gate(len2))) return sb_sqrt_gated(len2);
    const float len0 = sb_sqrt(len2);
    return len0 == 0.0f ? sb_length(0.0f, -1.0e-10f) : len0; // the guard of :104-107
}

// G beams side by side, for the temporally blocked kernel (sb_blocked.hip): the same arithmetic as sb_beam_eval in
// the same order, written stage by stage over the group so that the G dependency chains interleave in one
// instruction stream and ONE wave-uniform gate decides between the short exact sqrt/reciprocal and the IEEE
// sequences for the whole group.  Per material the host precomputes yl = yield_strain * length (the first product of
// :115, evaluated left to right) and ll = length * strain_break_limit (:117): same operands, same single rounding.
// sign(strain) enters :115 only as a factor of +-1 (exact) when |strain| > yield_strain >= 0, so it is applied as a
// copied sign bit; the host routes scenes with a negative or NaN yield_strain to the single-substep kernel, where
// sign() is spelled out.
// Two floats in a register pair: + - * on them are ONE v_pk_add_f32 / v_pk_mul_f32 (each half rounded exactly as the scalar
// instruction rounds it; with -ffp-contract=off nothing is fused), and a scalar operand is broadcast by the instruction's
// operand select, not by a copy.  The blocked kernel's beam phase is bound by the vector ALU's issue rate, and half of a
// beam's arithmetic comes in such pairs: (x, y), and (target - len, last - len) against (spring, damp).
typedef float sb_v2 __attribute__((ext_vector_type(2)));
// `sd` = (spring, damp) TIMES the force scale 65536 (compute.wgsl:70): the reference scales each force component last,
// (force_mag * n) * 65536; a power of two commutes with every rounding on the way there -- (target - len) * spring, the sum,
// the product with n -- as long as no intermediate is subnormal or overflows.  The host only lets scenes in whose every spring
// and damp is zero or between 2^-50 and 2^100 in magnitude (SB_BK_SD_MIN / MAX: with lengths inside the sqrt gate no product
// can then be subnormal); an overflow shows up as a value the `mirrored` gate refuses, and that branch evaluates in the
// reference's order.  One packed multiplication less per beam.
#define SB_BK_SD_MIN 0x1p-50f
#define SB_BK_SD_MAX 0x1p100f
struct SbBeamMat {
    float length, inv_length;
    sb_v2 sd; // (spring, damp) * 65536
    float yield_strain, yl, ll;
};
// max(|a|, |b|, |c|) in one instruction (source modifiers; spelled as asm so that no canonicalising copies appear)
SB_DEV float sb_max3_abs(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// target_length and last_length of each beam go in and out.
// Forces: endpoint B gets i32(+f*s) (:129-130), endpoint A gets i32(-f*s) (:127-128).  The conversion truncates toward zero,
// so the two are exact negatives of each other -- unless it saturates (|f*s| >= 2^31: INT_MAX on one side, INT_MIN on the
// other) .  `mirrored` (wave-uniform) says that no lane of the wave is anywhere near that: the caller then SUBTRACTS fb from
// A's sums (ds_sub) and fa is not computed at all -- two conversions (each the price of two plain instructions) for three
// instructions of gate per group.  NaN converts to 0 on both sides and passes the gate as what it is: harmless.
// Plastic yield (:113-116) is rare: the comparison is made for every beam, the new target only computed when some lane of the
// wave needs it (the comparison's lane mask IS the ballot: the test costs scalar instructions only).
template <int G>
SB_DEV void sb_beam_group(const float2 (&pa)[G], const float2 (&pb)[G], const SbBeamMat (&m)[G], float (&target)[G], float (&last)[G],
                          int32_t (&fa)[G][2], int32_t (&fb)[G][2], bool &mirrored, bool (&broken)[G])
{
    sb_v2 tl[G]; // (target_length, last_length): the caller's two registers usually ARE a pair
#pragma unroll
    for (int u = 0; u < G; u++) tl[u] = sb_v2{target[u], last[u]};
    const float particle_force_scale = 65536.0f; // :70
    sb_v2 d[G], fs[G];
    float len2[G], len[G], inv_len[G], strain[G];
    // the gate as an OR of lane masks: each comparison lands in a scalar register pair and the branch tests their union
    // (a bool carried through && and then balloted went through a VGPR and back: two vector instructions per group)
    unsigned long long outside = 0ull;
#pragma unroll
    for (int u = 0; u < G; u++) {
        const sb_v2 a = {pa[u].x, pa[u].y}, b = {pb[u].x, pb[u].y};
        d[u] = b - a; // :103
        const sb_v2 sq = d[u] * d[u];
        len2[u] = sq.x + sq.y;
        outside |= __builtin_amdgcn_ballot_w64(!(len2[u] >= 0x1p-90f)) | __builtin_amdgcn_ballot_w64(!(len2[u] <= 0x1p90f));
    }
    if (outside == 0ull) {
#pragma unroll
        for (int u = 0; u < G; u++) len[u] = sb_sqrt_gated(len2[u]); // :108 (the guard of :104 cannot fire)
#pragma unroll
        for (int u = 0; u < G; u++) inv_len[u] = sb_rcp_gated(len[u]);
    } else {
#pragma unroll
        for (int u = 0; u < G; u++) {
            const float len0 = sb_sqrt(len2[u]);
            const bool degenerate = len0 == 0.0f; // :104-107
            d[u].x = degenerate ? 0.0f : d[u].x;
            d[u].y = degenerate ? -1.0e-10f : d[u].y;
            len[u] = degenerate ? sb_length(0.0f, -1.0e-10f) : len0;
            inv_len[u] = sb_div(1.0f, len[u]);
        }
    }
    unsigned long long yields = 0ull;
#pragma unroll
    for (int u = 0; u < G; u++) {
        const sb_v2 e = tl[u] - len[u];    // (target - len, last - len)
        const sb_v2 w = e * m[u].sd;       // ... * (spring, damp) * 65536
        const float force_scaled = w.x + w.y; // :110, times the force scale
        const sb_v2 n = d[u] * inv_len[u];
        fs[u] = n * force_scaled;                   // :111, :127-130: (force_mag * n) * scale, per component (see SbBeamMat::sd)
        strain[u] = -e.x * m[u].inv_length;         // :112 (len - target is -(target - len), exactly)
        yields |= __builtin_amdgcn_ballot_w64(sb_abs(strain[u]) > m[u].yield_strain);
        broken[u] = sb_abs(len[u] - m[u].length) > m[u].ll; // :117
        fb[u][0] = sb_f32_to_i32(fs[u].x);
        fb[u][1] = sb_f32_to_i32(fs[u].y);
    }
    float big = sb_max3_abs(fs[0].x, fs[0].y, fs[G - 1].x);
#pragma unroll
    for (int u = 1; u < G; u++) big = sb_max3_abs(big, fs[u].x, fs[u].y); // (the max of non-negative numbers: |big| = big)
    mirrored = __builtin_amdgcn_ballot_w64(!(big < 0x1p30f)) == 0ull;     // (NaN operands drop out of v_max3: see above)
    if (!mirrored) { // some lane is huge, infinite -- or only looks so because the scale was applied early: the reference's own order
#pragma unroll
        for (int u = 0; u < G; u++) {
            const sb_v2 e = tl[u] - len[u];
            const sb_v2 w = e * (m[u].sd * 0x1p-16f); // (spring, damp) themselves: the scaling was exact
            const float force_mag = w.x + w.y;
            const sb_v2 n = d[u] * inv_len[u];
            const sb_v2 f = n * force_mag * particle_force_scale;
            fb[u][0] = sb_f32_to_i32(f.x);
            fb[u][1] = sb_f32_to_i32(f.y);
            fa[u][0] = sb_f32_to_i32_neg(f.x);
            fa[u][1] = sb_f32_to_i32_neg(f.y);
        }
    }
    if (__builtin_expect(yields != 0ull, 0)) {
#pragma unroll
        for (int u = 0; u < G; u++) {
            const float signed_yl = m[u].yl * __uint_as_float((__float_as_uint(strain[u]) & 0x80000000u) | 0x3f800000u); // * (+-1)
            const float yielded_target = len[u] - signed_yl;                                                              // :115
            target[u] = (sb_abs(strain[u]) > m[u].yield_strain) ? yielded_target : target[u];                             // :113-116
        }
    }
#pragma unroll
    for (int u = 0; u < G; u++) last[u] = len[u]; // :124
}

// ---------------------------------------------------------------- particle (compute.wgsl:139-201)

struct SbParticle {
    float2 p, v, a;
};

// one iteration of the collision loop, compute.wgsl:148-169.  `self` is const_particle (:141); (dx, dy) = other - self,
// dist = length(dx, dy) as the caller already had to compute it for the contact test (:150).  The reciprocal of dist
// takes the short exact form when every lane of the wave that is in contact has an ordinary distance (gate of
// sb_rcp_gated; exhaustively checked, see above), x / 2 is x * 0.5 exactly, and x / dt^2 is a multiplication when dt^2
// is a power of two.
SB_DEV void sb_collide_pair_at(const SbParams &prm, float friction, float elasticity_coeff, SbParticle &particle,
                               const SbParticle &self, uint32_t index, uint32_t other_index, float dx, float dy, float dist,
                               float2 ov)
{
    if (dist == 0.0f) {             // :151-154
        particle.p.y += sb_sign((float)index - (float)other_index);
    } else if (dist < prm.particle_radius * 2.0f) { // :155
        const float inv_dist = sb_wave_all(dist >= 0x1p-45f && dist <= 0x1p45f) ? sb_rcp_gated(dist) : sb_div(1.0f, dist);
        float nx = dx * inv_dist, ny = dy * inv_dist;         // :156
        float tx = -ny, ty = nx;                              // :157
        float ux = self.v.x - ov.x, uy = self.v.y - ov.y;     // :158
        float impulse_normal = elasticity_coeff * (ux * nx + uy * ny); // :159
        float max_friction = impulse_normal * friction;               // :160
        float impulse_tangent = sb_clamp(ux * tx + uy * ty, -max_friction, max_friction); // :161
        particle.v.x -= impulse_normal * nx + impulse_tangent * tx;   // :162
        particle.v.y -= impulse_normal * ny + impulse_tangent * ty;
        float overlap = prm.particle_radius * 2.0f - dist;            // :164
        float csx = nx * overlap * 0.5f, csy = ny * overlap * 0.5f;   // (n * overlap) / 2
        if (prm.inv_dt2 != 0.0f) {                                    // :168
            particle.a.x -= csx * prm.inv_dt2;
            particle.a.y -= csy * prm.inv_dt2;
        } else {
            float dt2 = prm.time_step * prm.time_step;
            particle.a.x -= sb_div(csx, dt2);
            particle.a.y -= sb_div(csy, dt2);
        }
    }
}
SB_DEV void sb_collide_pair(const SbParams &prm, float friction, float elasticity_coeff,
                            SbParticle &particle, const SbParticle &self, uint32_t index,
                            uint32_t other_index, float2 op, float2 ov)
{
    float dx = op.x - self.p.x, dy = op.y - self.p.y;
    sb_collide_pair_at(prm, friction, elasticity_coeff, particle, self, index, other_index, dx, dy, sb_length(dx, dy), ov);
}

// ---------------------------------------------------------------- spatial hash (SB_COLLIDE_GRID)

// Uniform grid over [x0, x0 + nx*cell) x [y0, y0 + ny*cell); coordinates outside are clamped into
// the edge cells.  The origin and the extent are fixed at upload; the cell size (with it nx, ny) belongs to
// the CURRENT hash and lives in SbGridCtl, because the skin adapts to how fast the scene moves.  The map x -> cell coordinate is monotone and cell >= 2r * (1 + 1/64), so two
// particles closer than 2r always land in the same or adjacent cells: the 3x3 neighbourhood is a
// SUPERSET of the interacting pairs of compute.wgsl:144-170 wherever the particles are (clamping
// only piles far-away particles into edge cells: slower, never wrong).
struct SbGrid {
    const unsigned long long *head; // per cell: (number of the build that wrote it) << 32 | first record of its chain; a stale
                                    // build number reads as "empty", so no cell is ever cleared
    const float4 *rec;          // per particle (record index == internal index): {x, y AT BUILD TIME, bits(slot), bits(next record)}
    const uint32_t *cell_of;    // per particle: its cell at the last build
    // The hash is DOUBLE BUFFERED (r04): the three arrays above are buffer 0, these are buffer 1; SbGridCtl::cur says which one
    // the neighbour lists in use came from.  A substep that pushes the next hash (SbGridCtl::pushing) writes the other buffer
    // while particles with overflowed lists still scan the cells of the current one.
    const unsigned long long *head1;
    const float4 *rec1;
    const uint32_t *cell_of1;
    float x0, y0, width, height; // the TIGHT frame: the uploaded bounding box plus a margin
    float two_r, cell_min;      // cells are never smaller than cell_min (what the arrays were sized for)
    uint32_t nx_cap, ny_cap;
    float bounds;               // the WIDE frame is the whole domain [0, bounds]^2, wide_side cells per side at most
    uint32_t wide_side;         // (what fits the same arrays); used once the scene has left the tight frame
    // neighbour lists (sb_neighbour_list_build): per particle the internal indices of everybody within
    // 2r + 2*skin at build time, in ascending slot order; entry k of particle i at nl[k * nl_stride + i].
    // Written by the particle kernel of the substep that follows a hash build (*fresh != 0: every particle
    // makes its own list from the fresh hash, with the whole chip's parallelism, and uses it at once), read by
    // the same thread on the substeps after it.
    uint32_t *nl_count;         // entries, or SB_NL_OVERFLOW: more than SB_NL_CAP, scan the cells instead
    uint32_t *nl;
    uint32_t nl_stride;
    // filled in by the kernel itself once it has decided (sb_grid_view): the geometry of the hash the lists in use came from
    // (a generic pointer into LDS) and the drift C accumulated since that hash was built
    const struct SbGridGeom *geo;
    float Cx, Cy;
};
#define SB_NL_CAP 16u
#ifndef SB_NL_SEL
#define SB_NL_SEL 6 // candidates one sweep of the list builder can take
#endif
#define SB_NL_OVERFLOW 0xFFFFFFFFu
// The hash is rebuilt only when needed.  Cells are 2r*(1+1/64) + 2*skin wide; the engine keeps a bound
// D on how far any particle can have moved since the last build RELATIVE TO THE COMMON DRIFT C of the scene,
// and rebuilds before a substep whose READ state has D > skin.  Per substep every particle's displacement
// d_i is measured against one vector c chosen beforehand (the mean displacement of the substep before: a
// falling or sliding body moves almost rigidly, and a translation changes no distance); D accumulates
// max_i |d_i - c| and C accumulates c.  Any choice of c keeps the bound valid: for a pair (i, j) the
// relative motion since the build is sum (d_i - c) - sum (d_j - c), at most 2D.  So while D <= skin, two
// particles closer than 2r NOW were closer than 2r + 2*skin AT BUILD TIME and sit in the same or
// adjacent cells of the stale binning (still a superset of the contacts), and a candidate whose
// build-time position, carried along by C, is farther than 2r + skin from the querying particle's
// current position cannot be in contact: candidates are found in the stale cells, pre-filtered on
// their drifted stale positions and tested exactly at their CURRENT positions.
struct SbGridGeom {
    float skin, cell, reach2; // cell width 2r*(1+1/64) + 2*skin (>= cell_min); (2r + 2*skin)^2 with a rounding margin
    uint32_t nx, ny;
    float x0, y0;       // origin of the frame in use
    uint32_t wide;      // 0: tight frame; 1: whole domain (more than 1/64 of the particles fell outside the tight one)
    uint32_t gen;       // number of the build this hash came from (what its head words carry)
};
// The decision state, one block per substep parity.  r04: the block a substep kernel reads was published by the LAST WORKGROUP TO
// FINISH of the substep before it (sb_grid_tail: per-workgroup displacement slots, sharded arrival tickets, the last arriver
// reduces and decides) -- until r03 a helper launch per substep (k_grid_maintain) did that, 18 % of config 3's GPU time for a
// decision that is "nothing to do" on nine substeps in ten.  A launch reads ctl[par] with plain loads (written by an earlier
// launch) and its tail writes ctl[par ^ 1] whole.  Two schedules for the hash itself (the `mode` of a launch, the host's choice):
//   SB_GRID_LAGGED   no helper launch at all.  When the bound is about to run out (D + 2 x the last step > skin) the tail ORDERS
//                    a push: during the next substep every workgroup pushes its own particles (READ state) into the OTHER hash
//                    buffer while the lists in use -- still valid -- serve that substep; the substep after that makes the new
//                    lists (fresh), which are born with the push substep's displacement already on their bound.  Kernel
//                    boundaries are the only synchronisation: no device-wide barrier, no residency requirement, any number of
//                    tiles.  If a single substep moves somebody farther than predicted (lists invalid and no hash in the
//                    making, or new lists invalid at birth) the tail raises `abort`: that launch has still produced a correct
//                    substep -- its lists were valid -- but every launch queued behind it returns at once, and the host, which
//                    looks at the end of every call, builds a hash with the helper and runs a stretch in the classic schedule.
//   SB_GRID_CLASSIC  the r03 schedule: `fresh` + `need_build` ask the helper launch in front of the next substep
//                    (k_grid_build, which otherwise returns at once) for a hash of that substep's READ state.  Violent scenes
//                    (a hash per substep or two) and the atomic path run this way.
// A rebuild the HOST knows about (upload, ghost refresh, the hybrid's fresh start, recovery from an abort) is a forced helper
// launch, not a flag.
#define SB_GRID_LAGGED 0u
#define SB_GRID_CLASSIC 1u
struct SbGridCtl {
    uint32_t fresh;      // the coming substep makes the neighbour lists from hash `cur` ...
    uint32_t need_build; // ... which the helper launch in front of it has yet to build (classic schedule only)
    uint32_t pushing;    // the coming substep pushes every particle into hash cur ^ 1, geometry `pgeo` (lagged schedule only)
    uint32_t abort;      // sticky, in BOTH blocks: the lists are not known to be valid for the coming substep; launches return at once
    uint32_t cur;        // the hash buffer the lists in use came from
    uint32_t executed;   // substeps run since the upload (the host's roll-back after an abort counts on it)
    uint32_t builds;     // statistics: hashes built (or on order)
    uint32_t since;      // age of the hash in use in substeps, the coming one included (what its bound `accum` has been collected over)
    float accum;         // D for the READ state of the coming substep
    float cx, cy;        // c for the coming substep
    float Cx, Cy;        // C for the READ state of the coming substep
    float skin_min, skin_max; // the skin ADAPTS at every build: when the last hash lasted 3 substeps or less it doubles (up to
                         // skin_max) provided that promises two substeps at the rate the bound has been growing -- if not even
                         // that, the scene is too violent for any hash to last and the skin drops back to skin_min; when a hash
                         // lasted 64 substeps or more the skin halves (down to skin_min)
    uint32_t wide_next;  // 1: some build counted more than 1/64 of the particles outside its frame: the next one frames the whole domain
    uint32_t settled;    // 1: the displacement slots of the last substep are already in `accum` (or the lists were replaced since): the
                         // decision of the coming substep does not add them again (forced helper launch, k_grid_settle, the hybrid's write-back)
    uint32_t transient;  // 1: hash `cur` came from a forced helper launch (upload, ghost refresh, recovery): how long it lasts says
                         // nothing about the scene (a lattice relaxes its upload jitter in one big step) and does not move the skin
    uint32_t short_lived; // 1: the last hash was replaced at an age of 6 substeps or less: the lagged schedule, which orders a hash one substep
                         // early and spends another making lists, is the wrong one for this scene (a hint to the host: grid_substeps)
    uint32_t pad_[3];
    SbGridGeom geo;      // hash `cur`
    SbGridGeom pgeo;     // hash cur ^ 1 while it is being pushed
};
static_assert(sizeof(SbGridCtl) % 16 == 0, "SbGridCtl is moved in 16-byte chunks");
#define SB_CHAIN_END 0xFFFFFFFFu
#define SB_MAX_WAVES 16
#define SB_AGENT_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// SbGridCtl words are written by ONE launch (k_grid_maintain) and read by LATER launches only: the readers use plain loads
// (a uniform address: scalar loads through the constant cache, invalidated at every kernel start) -- an agent-scope atomic
// load per word and per particle was a trip past the L2 each
#define SB_CTL_LOAD(p) (*(p))
#define SB_AGENT_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

SB_DEV SbGridGeom sb_grid_geom_load(const SbGridGeom *c)
{
    SbGridGeom m;
    m.skin = SB_CTL_LOAD(&c->skin);
    m.cell = SB_CTL_LOAD(&c->cell);
    m.reach2 = SB_CTL_LOAD(&c->reach2);
    m.nx = SB_CTL_LOAD(&c->nx);
    m.ny = SB_CTL_LOAD(&c->ny);
    m.x0 = SB_CTL_LOAD(&c->x0);
    m.y0 = SB_CTL_LOAD(&c->y0);
    m.wide = SB_CTL_LOAD(&c->wide);
    m.gen = SB_CTL_LOAD(&c->gen);
    return m;
}
// the hash the lists in use came from, as a view: buffer `cur` in the first three members, its geometry, the drift since its build
SB_DEV SbGrid sb_grid_view(const SbGrid &g, uint32_t cur, const SbGridGeom *geo, float Cx, float Cy)
{
    SbGrid v = g;
    v.geo = geo;
    v.Cx = Cx;
    v.Cy = Cy;
    if (cur != 0u) {
        v.head = g.head1;
        v.rec = g.rec1;
        v.cell_of = g.cell_of1;
    }
    return v;
}
// the geometry that goes with a skin (every workgroup of k_grid_maintain computes the same values)
SB_DEV SbGridGeom sb_grid_geom_for(const SbGrid &g, float skin, uint32_t wide)
{
    SbGridGeom m;
    m.skin = skin;
    m.wide = wide;
    m.gen = 0u; // (the builder fills it in)
    const float reach = g.two_r + 2.0f * skin;
    m.reach2 = reach * reach * 1.001f;
    const float want = g.two_r * 1.015625f + 2.0f * skin;
    if (wide) { // the whole domain, with cells as coarse as the arrays demand (coarser is slower, never wrong)
        m.x0 = 0.0f;
        m.y0 = 0.0f;
        m.cell = fmaxf(fmaxf(want, g.cell_min), sb_div(g.bounds, (float)g.wide_side));
        const float f = ceilf(sb_div(g.bounds, m.cell));
        m.nx = m.ny = f >= 1.0f ? (f < (float)g.wide_side ? (uint32_t)f : g.wide_side) : 1u;
    } else {
        m.x0 = g.x0;
        m.y0 = g.y0;
        m.cell = fmaxf(want, g.cell_min);
        const float fx = ceilf(sb_div(g.width, m.cell)), fy = ceilf(sb_div(g.height, m.cell));
        m.nx = fx >= 1.0f ? (fx < (float)g.nx_cap ? (uint32_t)fx : g.nx_cap) : 1u; // never more cells than allocated
        m.ny = fy >= 1.0f ? (fy < (float)g.ny_cap ? (uint32_t)fy : g.ny_cap) : 1u;
    }
    return m;
}

// ---- the start of every particle kernel under SB_COLLIDE_GRID: last substep's displacement slots, the decision (SbGridCtl)
// Every workgroup takes the SAME decision from the same words, in the shadow of its own first loads -- what the helper launch
// of rounds 1-3 did, as the prologue of the kernel that needs the answer.  (r04 first put the decision at the END of the
// substep kernel -- arrival tickets, the last workgroup reduces and publishes: five dependent trips to memory at ~1 us each
// behind the last workgroup's last store made every launch 9 us longer; profiles/r04_grid_schedules.txt.  Nothing may hang off
// the end of a launch that fills the chip exactly once.)
//   slots: float4[3][SB_GRID_SLOTS] = {largest drift-relative displacement (float bits, by atomic max: order-free), sample dx,
//   sample dy, -} of the workgroups b with b % SB_GRID_SLOTS == slot (samples: the first SB_GRID_SLOTS workgroups only).
//   Triple buffered by the number of the substep: substep k reads buffer (k - 1) % 3, fills k % 3, and its workgroup 0 zeroes
//   (k + 1) % 3 -- all three indices come from the HOST's count of executed substeps, which an abort corrects.
#define SB_GRID_SLOTS 64u // (one wave reduces them all; a thousand workgroups: fifteen atomic maxima per slot, spread over the launch)
struct SbGridStep {
    SbGridCtl *ctl;           // both parity blocks
    uint32_t par;             // the block the substep before this one published; workgroup 0 of this launch publishes the other
    uint32_t mode;            // SB_GRID_LAGGED / SB_GRID_CLASSIC
    const float4 *slots_in;   // what the substep before this one measured
    float4 *slots_out;        // what this one measures
    float4 *slots_zero;       // the buffer the next one fills
    float inv_samples;        // 1 / min(workgroups of the particle kernel, SB_GRID_SLOTS)
    uint32_t *outside;        // [4] particles each build found outside its frame, by build number & 3 (sb_grid_decide)
    uint32_t P;
#ifdef SB_STAMPS // diagnostic build: where one workgroup's time goes (10 ns ticks since its start; tools/grid_schedule_probe.py)
    uint32_t *stamps;
#endif
};
#ifdef SB_STAMPS
#define SB_STAMP(t, k)                                                                                                  \
    do {                                                                                                                \
        if (blockIdx.x == gridDim.x / 2u && threadIdx.x == 0u) (t).stamps[k] = (uint32_t)(wall_clock64() - sb_t_start); \
    } while (0)
#else
#define SB_STAMP(t, k) do { } while (0)
#endif
// The words of the decision that gate the kernel itself, and the numbers every thread needs of it.  ONE lane per workgroup
// computes them (sb_grid_stage: a few dozen instructions in front of the first barrier, while the other waves are still
// waiting for their own loads); everybody reads them behind that barrier.  r04 measured what the alternatives cost on a kernel
// that is bound by instruction issue: every wave computing them for itself (a dozen LDS reads, two divisions: 130 instructions
// x 32 waves per CU) +4 us per launch; thread 0 computing the WHOLE block with the workgroup waiting at a second barrier +4 us
// as well (profiles/r04_grid_schedules.txt).
struct SbGridFlags {
    uint32_t abort, fresh, pushing, need_build;
};
struct SbGridHot {
    SbGridFlags f;
    float step;       // the largest drift-relative displacement of the substep just done
    float cx, cy;     // the common displacement this substep is measured against
    float Cx, Cy;     // the drift accumulated since the hash in use was built -- meaningful on a substep that neither makes lists nor
                      // aborts (the cell scans of particles whose lists overflowed use it); on the others: SbGridShared::now
    uint32_t pad_[3];
};
struct __attribute__((aligned(16))) SbGridShared { // LDS of a kernel that decides
    uint32_t outside[4]; // the builders' counts (SbGridStep::outside)
    SbGridHot hot;
    SbGridCtl prev;    // the block the launch read
    SbGridCtl now;     // what it runs under -- in workgroup 0, which publishes it, and wherever a substep makes lists or pushes
};

// The scene's common drift is estimated from a SAMPLE: the first particle of each of the first SB_GRID_SLOTS workgroups reports
// its displacement; any estimate keeps the bound valid.  Called by thread 0.
SB_DEV void sb_store_sample_displacement(float4 *slots_out, float dx, float dy)
{
    if (blockIdx.x < SB_GRID_SLOTS) {
        slots_out[blockIdx.x].y = sb_abs(dx) < 1.0e30f ? dx : 0.0f;
        slots_out[blockIdx.x].z = sb_abs(dy) < 1.0e30f ? dy : 0.0f;
    }
}

// End of the particle kernel, called by EVERY thread: this workgroup's largest drift-relative displacement joins its slot
// (anything not provably small reads as huge; non-negative floats order like their bits).  Fire and forget: nothing waits for it.
SB_DEV void sb_store_block_displacement(float4 *slots_out, float m)
{
    __shared__ float s_wave_max[SB_MAX_WAVES];
    m = (m < 1.0e30f) ? m : 1.0e30f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63u) == 0u) s_wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = 0.0f;
        for (uint32_t w = 0; w < (blockDim.x >> 6); w++) b = fmaxf(b, s_wave_max[w]);
        (void)__hip_atomic_fetch_max((uint32_t *)&slots_out[blockIdx.x % SB_GRID_SLOTS].x, __float_as_uint(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the skin of the hash about to be built follows how long the last one lasted (SbGridCtl)
SB_DEV float sb_grid_next_skin(float skin, float skin_min, float skin_max, float accum, uint32_t since)
{
    if (since <= 3u) {
        // short-lived hash: a wider skin must promise at least two substeps at the rate the bound has been growing -- the
        // smallest doubling that does (r03 tried one doubling only: a scene that speeds up while its skin is lean, 4 units
        // against 4.4 per substep, then never gets off the minimum although 16 would last four substeps) -- else the scene is
        // simply too violent for any hash to last and lean cells are the cheapest
        const float rate = accum / (float)since;
        for (float wider = skin * 2.0f; wider <= skin_max; wider *= 2.0f)
            if (wider >= 2.0f * rate) return wider;
        if (skin_max >= 2.0f * rate && skin_max > skin) return skin_max;
        if (!(skin >= 2.0f * rate)) return skin_min;
    } else if (since >= 64u) {
        return fmaxf(skin * 0.5f, skin_min);
    }
    return skin;
}

// Requested by every thread together with its own particles (cold lines, a full trip to memory like them; the wait counter
// retires in order, so asked for any earlier they would stand in front of requests that come out of the L2).  Unconditional
// (an exec-masked load is a branch with a full drain of the wait counter behind it); only wave 0's copies are used: lane k its
// slot, lanes 0 .. 8 the control block in 16-byte chunks, lane 9 the four `outside` counters.  Everything the decision reads
// comes in with these two requests: a load of its own inside the decision would be a memory latency with the whole workgroup
// waiting behind it.
struct SbGridEarly {
    float4 slot, ctl;
};
SB_DEV SbGridEarly sb_grid_begin(const SbGridStep &t)
{
    SbGridEarly e;
    e.slot = t.slots_in[threadIdx.x % SB_GRID_SLOTS];
    const uint32_t chunks = sizeof(SbGridCtl) / 16u, lane = threadIdx.x & 63u, k = lane < chunks ? lane : chunks;
    const float4 *ctl = (const float4 *)(t.ctl + t.par), *cnt = (const float4 *)t.outside;
    e.ctl = (k < chunks ? ctl : cnt - chunks)[k];
    return e;
}

// the flags of this substep from those of the block it read and the displacement of the substep just done
SB_DEV SbGridFlags sb_grid_flags(uint32_t e_abort, uint32_t e_settled, uint32_t e_fresh, uint32_t e_pushing, uint32_t e_need_build,
                                 float e_accum, float skin, float pskin, float step, uint32_t mode, uint32_t advance)
{
    SbGridFlags f{e_abort, e_fresh, e_pushing, e_need_build}; // (adopted as they stand: abort is sticky, a settled block was decided ahead of time)
    if (e_abort == 0u && e_settled == 0u) {
        const float accum1 = e_accum + step; // the bound of the lists in use, for the READ state of this substep
        const bool over = !(accum1 <= skin); // NaN-safe
        f.fresh = f.pushing = f.need_build = 0u;
        if (e_pushing != 0u) { // the substep just done pushed ITS read state into hash cur ^ 1: this one makes the lists, born
            f.fresh = 1u;      // with that substep's displacement on their bound
            if (!(step <= pskin)) f.abort = 1u;
        } else if (mode == SB_GRID_CLASSIC) {
            if (over) f.fresh = f.need_build = 1u; // the helper launch in front of this substep builds from its READ state (it takes this same decision)
        } else if (over) {
            f.abort = 1u; // somebody moved farther than predicted: the lists in use are not known to be valid any more
        } else if (!(accum1 + 2.0f * step <= skin)) {
            f.pushing = 1u; // one more substep like the last could use the skin up: this substep pushes the next hash as it goes
        }
    }
    // a lagged launch cannot serve a build the classic schedule ordered from a helper it does not have (the host changed
    // schedules without serving it: never, by construction)
    if (mode == SB_GRID_LAGGED && advance != 0u && f.need_build != 0u) f.abort = 1u;
    return f;
}

// In front of the kernel's first workgroup barrier, by wave 0 (the caller's other waves pass through): the slots reduced, the
// block and the counters into LDS, and -- lane 0 -- what everybody needs of the decision (SbGridHot).
SB_DEV void sb_grid_stage(const SbGridStep &t, const SbGridEarly &e, SbGridShared &sh, uint32_t advance)
{
    if (threadIdx.x >= 64u) return;
    const uint32_t lane = threadIdx.x, chunks = sizeof(SbGridCtl) / 16u;
    float mx = e.slot.x, sx = e.slot.y, sy = e.slot.z; // (non-negative floats order like their bits: the atomic maximum was taken on those)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        sx += __shfl_xor(sx, off, 64);
        sy += __shfl_xor(sy, off, 64);
    }
    if (lane < chunks) {
        ((float4 *)&sh.prev)[lane] = e.ctl;
        ((float4 *)&sh.now)[lane] = e.ctl; // (the decision starts from a copy and patches it in place)
    } else if (lane == chunks) {
        *(float4 *)sh.outside = e.ctl;
    }
    // the words of the block the flags depend on, out of the lanes that hold them (a word of chunk c sits in lane c)
#define SB_CTL_WORD(field) __builtin_amdgcn_readlane(__float_as_uint(((offsetof(SbGridCtl, field) / 4u) & 3u) == 0u   ? e.ctl.x \
                                                                      : ((offsetof(SbGridCtl, field) / 4u) & 3u) == 1u ? e.ctl.y \
                                                                      : ((offsetof(SbGridCtl, field) / 4u) & 3u) == 2u ? e.ctl.z \
                                                                                                                        : e.ctl.w), \
                                                     (int)(offsetof(SbGridCtl, field) / 16u))
    const uint32_t e_abort = SB_CTL_WORD(abort), e_settled = SB_CTL_WORD(settled), e_fresh = SB_CTL_WORD(fresh);
    const uint32_t e_pushing = SB_CTL_WORD(pushing), e_need = SB_CTL_WORD(need_build);
    const float e_accum = __uint_as_float(SB_CTL_WORD(accum)), skin = __uint_as_float(SB_CTL_WORD(geo.skin));
    const float pskin = __uint_as_float(SB_CTL_WORD(pgeo.skin));
    const float e_cx = __uint_as_float(SB_CTL_WORD(cx)), e_cy = __uint_as_float(SB_CTL_WORD(cy));
    const float e_Cx = __uint_as_float(SB_CTL_WORD(Cx)), e_Cy = __uint_as_float(SB_CTL_WORD(Cy));
#undef SB_CTL_WORD
    if (lane == 0u) {
        SbGridHot h;
        h.step = mx;
        h.f = sb_grid_flags(e_abort, e_settled, e_fresh, e_pushing, e_need, e_accum, skin, pskin, mx, t.mode, advance);
        if (e_abort != 0u || e_settled != 0u) { // adopted as it stands
            h.cx = e_cx;
            h.cy = e_cy;
            h.Cx = e_Cx;
            h.Cy = e_Cy;
        } else { // the mean of the sample displacements of the substep just done
            float mean_x = sx * t.inv_samples, mean_y = sy * t.inv_samples;
            if (!(sb_abs(mean_x) < 1.0e30f) || !(sb_abs(mean_y) < 1.0e30f)) mean_x = mean_y = 0.0f;
            h.cx = mean_x;
            h.cy = mean_y;
            h.Cx = e_Cx + e_cx;
            h.Cy = e_Cy + e_cy;
        }
        h.pad_[0] = h.pad_[1] = h.pad_[2] = 0u;
        sh.hot = h;
    }
}

// The state this substep runs under (see SbGridCtl for the two schedules), by ONE thread, into sh.now (a copy of sh.prev so
// far: sb_grid_stage), behind the kernel's first barrier; whoever reads sh.now has another barrier in front of that.  Every
// workgroup of a launch computes the same block: everything read here was written by earlier launches.
//   advance   1: a substep (it counts itself); 0: the host's settling launch (k_grid_settle: the decision ahead of time, for a
//             look) -- the block it publishes is `settled`, and the substep that finds it adopts it as it stands
//   publish   workgroup 0 stores the block for the next launch (the unforced helper decides too, but only for itself)
// outside[4]: particles each build found outside its frame, by build number & 3: the builders of hash G add to slot G & 3, the
// decision reads the slot of the hash in use (complete long ago) and whoever orders hash G zeroes slot (G + 1) & 3.
SB_DEV void sb_grid_decide(const SbGridStep &t, const SbGrid &g, SbGridShared &sh, uint32_t advance, bool publish)
{
    const SbGridCtl &E = sh.prev;
    SbGridCtl &N = sh.now; // (patched in LDS, field by field, so that this once-per-launch code does not size the kernel's register file)
    const float step = sh.hot.step; // (lane 0 of wave 0 computed them in front of the barrier: sb_grid_stage)
    const SbGridFlags f = sh.hot.f;
    if (E.abort != 0u) {
        // sticky: nothing moves until the host has been here
    } else if (E.settled != 0u) {
        N.settled = advance ? 0u : 1u;
        N.executed = E.executed + advance;
    } else {
        // the frame: tight until some build found more than 1/64 of the particles outside it
        const uint32_t wide_next = (E.wide_next != 0u || sh.outside[E.geo.gen & 3u] > t.P / 64u) ? 1u : 0u;
        const float accum1 = E.accum + step;
        N.wide_next = wide_next;
        N.fresh = f.fresh;
        N.need_build = f.need_build;
        N.pushing = f.pushing;
        N.settled = advance ? 0u : 1u;
        N.executed = E.executed + advance;
        N.cx = sh.hot.cx;
        N.cy = sh.hot.cy;
        // unless something below says otherwise: the lists in use serve this substep too
        N.since = E.since + 1u;
        N.accum = accum1;
        N.Cx = E.Cx + E.cx;
        N.Cy = E.Cy + E.cy;
        auto order = [&](SbGridGeom &out) { // the geometry of a hash to come (the skin adapts to how long this one lasted)
            const float skin_new = E.transient != 0u ? E.geo.skin : sb_grid_next_skin(E.geo.skin, E.skin_min, E.skin_max, accum1, E.since);
            if (E.transient == 0u) N.short_lived = E.since <= 6u ? 1u : 0u;
            out = sb_grid_geom_for(g, skin_new, (E.geo.wide != 0u || wide_next != 0u) ? 1u : 0u);
            out.gen = E.builds + 1u;
            N.builds = E.builds + 1u;
            if (publish && blockIdx.x == 0u) SB_AGENT_STORE(&t.outside[(E.builds + 2u) & 3u], 0u);
        };
        if (E.pushing != 0u) { // lists from the hash the last substep pushed
            N.transient = 0u;
            N.cur = E.cur ^ 1u;
            N.geo = E.pgeo;
            N.since = 2u; // (the age of the hash: the substep that pushed it is on its bound)
            N.accum = step;
            N.Cx = E.cx;
            N.Cy = E.cy;
        } else if (f.need_build != 0u) { // (classic: the helper builds hash cur ^ 1 with this geometry)
            order(N.geo);
            N.transient = 0u;
            N.cur = E.cur ^ 1u;
            N.since = 1u;
            N.accum = N.Cx = N.Cy = 0.0f;
        } else if (f.pushing != 0u) {
            order(N.pgeo);
        }
    }
    if (f.abort != 0u && E.abort == 0u) {
        N.abort = 1u;
        N.executed = E.executed; // (this launch returns at once)
    }
    if (publish && blockIdx.x == 0u) {
        const uint32_t *src = (const uint32_t *)&sh.now;
        uint32_t *dst = (uint32_t *)(t.ctl + (t.par ^ 1u));
        for (uint32_t k = 0; k < sizeof(SbGridCtl) / 4u; k++) SB_AGENT_STORE(&dst[k], src[k]);
        if (f.abort != 0u) SB_AGENT_STORE(&t.ctl[t.par].abort, 1u); // both blocks: later launches alternate between them
    }
}

// the arrays a build writes (both hash buffers)
struct SbGridBuild {
    unsigned long long *head[2];
    uint32_t *cell_of[2];
    float4 *rec[2];
    uint32_t *outside; // [4] particles each build found outside its frame, by build number & 3 (sb_grid_decide)
};

// cell coordinate with the clamp made visible: *outside is set when a FINITE coordinate had to be clamped
SB_DEV uint32_t sb_grid_coord_flag(float x, float x0, float cell, uint32_t n, bool *outside)
{
    float q = sb_div(x - x0, cell);
    if (!(q > 0.0f)) {
        if (q < 0.0f) *outside = true; // (zero and NaN are not "outside")
        return 0u;
    }
    if (q >= (float)n) {
        if (q < 1.0e30f) *outside = true;
        return n - 1u;
    }
    return (uint32_t)q;
}

SB_DEV uint32_t sb_grid_coord(float x, float x0, float cell, uint32_t n)
{
    float q = sb_div(x - x0, cell);
    if (!(q > 0.0f)) return 0u; // negative, zero, NaN
    if (q >= (float)n) return n - 1u;
    return (uint32_t)q;
}

// One particle into hash buffer `buf`, in two halves so that a caller can keep several returning atomics in flight per lane:
// its cell and itself pushed on the front of that cell's list with one returning 64-bit exchange (the head word carries the
// number of the build, so the cells of older builds read as empty and nothing is ever cleared; the order inside a list is
// arbitrary, which is fine: contacts are re-ordered by slot) ...
SB_DEV unsigned long long sb_grid_push_begin(const SbGridBuild &w, uint32_t buf, const SbGridGeom &geo, uint32_t i, float2 p,
                                             uint32_t *cell, bool *outside)
{
    *cell = sb_grid_coord_flag(p.y, geo.y0, geo.cell, geo.ny, outside) * geo.nx + sb_grid_coord_flag(p.x, geo.x0, geo.cell, geo.nx, outside);
    return atomicExch(&w.head[buf][*cell], ((unsigned long long)geo.gen << 32) | i);
}
// ... then its record, whose `next` is what the exchange returned if that came from the same build
SB_DEV void sb_grid_push_end(const SbGridBuild &w, uint32_t buf, const SbGridGeom &geo, uint32_t i, float2 p, uint32_t slot,
                             uint32_t cell, unsigned long long old)
{
    const uint32_t next = (uint32_t)(old >> 32) == geo.gen ? (uint32_t)old : SB_CHAIN_END;
    w.cell_of[buf][i] = cell;
    w.rec[buf][i] = make_float4(p.x, p.y, __uint_as_float(slot), __uint_as_float(next));
}

// the builders' count of particles outside the frame (slot: build number & 3): one add per wave, and only when somebody left
// it; read by the decisions of later launches only
SB_DEV void sb_grid_count_outside(uint32_t *outside_slot, uint32_t n_out)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n_out += __shfl_xor(n_out, off, 64);
    if (n_out && (threadIdx.x & 63u) == 0u) (void)__hip_atomic_fetch_add(outside_slot, n_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// what a particle kernel under SB_COLLIDE_GRID gets besides the hash itself
struct SbGridJob {
    SbGridStep step;
    SbGridBuild build;     // lagged schedule: the arrays a push writes
    const uint32_t *pslot; // ... and every particle's slot (its record carries it)
    // cooperative list build (sb_lists_cooperative): internal index per slot, and the workgroup's LDS area for it
    const uint32_t *islot;
    uint32_t coop_off, coop_bytes; // byte offset into the dynamic LDS, size (0: none -- every particle walks the hash for itself)
};

// The hash is a linked list per cell (r02; round 1 counted, scanned and scattered into cell-sorted records: three
// device-wide barriers and a scan over every cell, 70-90 us per build against ~20 for one pass of exchanges): the first
// records of the 3x3 cells around a (stale) cell, SB_CHAIN_END where a cell is empty or off the grid.
struct SbGridHood {
    uint32_t head[9];
};
SB_DEV SbGridHood sb_grid_hood(const SbGrid &g, const SbGridGeom &m, uint32_t cell)
{
    SbGridHood h;
    const int cx = (int)(cell % m.nx), cy = (int)(cell / m.nx);
    unsigned long long v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int yy = cy + k / 3 - 1, xx = cx + k % 3 - 1;
        const bool in = yy >= 0 && yy < (int)m.ny && xx >= 0 && xx < (int)m.nx;
        v[k] = in ? g.head[(uint32_t)yy * m.nx + (uint32_t)xx] : 0ull; // nine independent loads
    }
#pragma unroll
    for (int k = 0; k < 9; k++) h.head[k] = (uint32_t)(v[k] >> 32) == m.gen ? (uint32_t)v[k] : SB_CHAIN_END; // (gen >= 1: 0 is never current)
    return h;
}
// f(x, y, slot, id) for every record of the nine chains; three chains are walked side by side so that their loads overlap
// f(x, y, slot, id) for every record of the nine chains; three chains are walked side by side so that their loads overlap.
// Branch-free inside a round: a chain that has ended re-reads `self` (the caller's own record, which every f ignores anyway:
// a particle is not its own neighbour) instead of masking its load off -- exec-mask bookkeeping around every load and every
// early return made the list builder instruction-bound (r02 counters: 1300 VALU + 650 SALU per particle and sweep).
template <typename F>
SB_DEV void sb_grid_walk(const SbGrid &g, const SbGridHood &h, uint32_t self, F f)
{
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
        uint32_t cur[3] = {h.head[3 * r], h.head[3 * r + 1], h.head[3 * r + 2]};
        while ((cur[0] & cur[1] & cur[2]) != SB_CHAIN_END) { // all ones only when all three have ended
            uint32_t id[3];
            float4 rc[3];
#pragma unroll
            for (int c = 0; c < 3; c++) id[c] = cur[c] != SB_CHAIN_END ? cur[c] : self;
#pragma unroll
            for (int c = 0; c < 3; c++) rc[c] = g.rec[id[c]];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                cur[c] = cur[c] != SB_CHAIN_END ? __float_as_uint(rc[c].w) : SB_CHAIN_END;
                f(rc[c].x, rc[c].y, __float_as_uint(rc[c].z), id[c]);
            }
        }
    }
}

// The collision loop of compute.wgsl:144-170 restricted to the 3x3 cell neighbourhood, applying
// contacts in ASCENDING SLOT ORDER exactly like the all-pairs scan does: repeatedly pick the
// contact with the smallest slot above the last one applied.  Non-contacts are no-ops in the
// reference loop, so skipping them changes nothing; the result is bit-identical to all-pairs.
SB_DEV void sb_collide_grid(const SbGrid &g, const SbGridGeom &m, const SbGridHood &hood, const SbParams &prm, float friction,
                            float elasticity_coeff, SbParticle &particle, const SbParticle &self, uint32_t i,
                            const uint32_t *__restrict__ pidx, const float2 *__restrict__ pos_r,
                            const float2 *__restrict__ vel_r)
{
    const float two_r = prm.particle_radius * 2.0f;
    const float far2 = two_r * two_r * 1.001f;
    const float reach = two_r + m.skin, stale_far2 = reach * reach * 1.001f;
    // where this particle would be in the frame of the build: current position minus the common drift
    const float qx = self.p.x - g.Cx, qy = self.p.y - g.Cy;
    bool have_last = false;
    uint32_t last = 0u;
    for (;;) {
        // one sweep collects the FOUR contacts with the smallest slots above `last` (sorted insert into four
        // registers); they are then applied in that order.  A pile of K contacts costs K/4 sweeps, not K.
        uint32_t bs[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, bi[4] = {0u, 0u, 0u, 0u};
        sb_grid_walk(g, hood, i, [&](float rx, float ry, uint32_t slot, uint32_t id) {
            if (id == i || (have_last && slot <= last) || slot >= bs[3]) return;
            const float sx = rx - qx, sy = ry - qy;
            if (sx * sx + sy * sy > stale_far2) return; // cannot have come within 2r (see SbGridCtl)
            const float2 q = pos_r[id];
            const float ex = q.x - self.p.x, ey = q.y - self.p.y;
            const float d2 = ex * ex + ey * ey; // exactly the argument length() takes the root of
            // sqrt is monotone: d2 clearly above (2r)^2 cannot give d < 2r (and is not 0), so the
            // correctly rounded root is only evaluated for the few candidates near contact range
            if (d2 > far2) return;
            const float d = sb_sqrt(d2);
            if (!(d == 0.0f || d < two_r)) return;
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) { // carry the larger one down the four registers
                const bool lt = slot < bs[q4];
                const uint32_t ts = lt ? bs[q4] : slot, ti = lt ? bi[q4] : id;
                bs[q4] = lt ? slot : bs[q4];
                bi[q4] = lt ? id : bi[q4];
                slot = ts;
                id = ti;
            }
        });
        bool full = true;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            if (bs[q4] == 0xFFFFFFFFu) {
                full = false;
            } else {
                sb_collide_pair(prm, friction, elasticity_coeff, particle, self, pidx[i], pidx[bi[q4]], pos_r[bi[q4]],
                                vel_r[bi[q4]]);
                last = bs[q4];
                have_last = true;
            }
        }
        if (!full) break; // fewer than four contacts left above `last`
    }
}

// On the substep of a hash build (positions in the records are the READ state): everybody within 2r + 2*skin of particle i,
// in ascending slot order (repeated selection of the smallest slot above the last one taken, as above).
// While the displacement bound D <= skin, a pair closer than 2r NOW was closer than 2r + 2D at build time,
// so the list is a superset of i's contacts until the next build.  NaN distances are kept (conservative).
SB_DEV uint32_t sb_neighbour_list_build(const SbGrid &g, const SbGridGeom &m, uint32_t i)
{
    const float reach2 = m.reach2;
    // distances AT BUILD TIME, record against record: in the lagged schedule (SbGridCtl) the hash holds the READ state of the
    // substep before this one, so the particle's own current position is not the one its neighbours were binned against
    const float4 own = g.rec[i];
    const float2 p = make_float2(own.x, own.y);
    const SbGridHood hood = sb_grid_hood(g, m, g.cell_of[i]);
    uint32_t n = 0u, last = 0u;
    bool have_last = false;
    for (;;) {
        // one sweep over the candidates collects the SB_NL_SEL smallest slots above `last` (sorted insert into that many
        // registers), so a typical list of 4-7 entries costs one sweep, not one per entry
        uint32_t bs[SB_NL_SEL], bi[SB_NL_SEL];
#pragma unroll
        for (int q = 0; q < SB_NL_SEL; q++) bs[q] = 0xFFFFFFFFu, bi[q] = 0u;
        sb_grid_walk(g, hood, i, [&](float rx, float ry, uint32_t slot, uint32_t id) {
            const float dx = rx - p.x, dy = ry - p.y;
            // (one test, no short-circuit branches)
            if ((id == i) | (have_last & (slot <= last)) | (slot >= bs[SB_NL_SEL - 1]) | (dx * dx + dy * dy > reach2)) return;
#pragma unroll
            for (int q = 0; q < SB_NL_SEL; q++) { // carry the larger one down the four registers
                const bool lt = slot < bs[q];
                const uint32_t ts = lt ? bs[q] : slot, ti = lt ? bi[q] : id;
                bs[q] = lt ? slot : bs[q];
                bi[q] = lt ? id : bi[q];
                slot = ts;
                id = ti;
            }
        });
        bool full = true;
#pragma unroll
        for (int q = 0; q < SB_NL_SEL; q++) {
            if (bs[q] == 0xFFFFFFFFu) {
                full = false;
            } else if (n != SB_NL_OVERFLOW) {
                if (n == SB_NL_CAP) {
                    n = SB_NL_OVERFLOW;
                } else {
                    g.nl[n * g.nl_stride + i] = bi[q];
                    n++;
                    last = bs[q];
                    have_last = true;
                }
            }
        }
        if (!full || n == SB_NL_OVERFLOW) break; // fewer than SB_NL_SEL left above `last`: that was everybody
    }
    g.nl_count[i] = n;
    return n;
}

// ---- the neighbour lists of a whole tile, made by its workgroup TOGETHER through LDS (r04; `north_star`: "LDS-staged spatial-hash
// cell buckets").  sb_neighbour_list_build has every particle load the nine list heads around its cell and chase nine chains of
// 16-byte records through global memory: three dependent trips per particle, each record fetched by every particle near it
// (r02 counters: the walk is bound by those per-lane loads).  A tile is spatially compact, so the cells its particles need are the
// bounding rectangle of their own cells plus a rim of one: the workgroup loads every head of that rectangle once (coalesced along
// rows), walks every chain once and leaves each record in LDS as 8 bytes -- its position quantised to 16 bits per axis RELATIVE TO
// THE RECTANGLE (a list only has to be a superset of "everybody within 2r + 2 skin at build time": the reach is widened by three
// quantisation steps) and its slot -- a contiguous block of records per cell.  Every particle then selects its neighbours out of LDS,
// in ascending slot order as before (the SB_NL_SEL smallest slots above the last one taken, per sweep), and turns slots into
// internal indices through `islot` only for the entries it stores.  Returns false (uniform, nothing written) when the rectangle
// or its records do not fit the area -- a tile that has scattered (free particles after a long flight): the caller falls back on
// sb_neighbour_list_build.  Same lists as that function up to extra entries at the rim of the reach; the contacts taken from them
// are decided on current positions at walk time, so the physics is the same bit for bit.
struct SbCoopMeta {
    int minx, maxx, miny, maxy;
    uint32_t nrec, fail, pad_[2];
};
#define SB_COOP_ANYWHERE 0xFFFFFFFFu // quantised position of a record that cannot be placed (clamped into an edge cell from outside the frame, NaN): always a candidate
#define SB_COOP_CPT 3                // cells a thread loads side by side (512 threads x 3 cover the ~1400 cells an area holds in one round)
SB_DEV uint32_t sb_coop_quantise(float x, float y, float ox, float oy, float to_q)
{
    const float ux = (x - ox) * to_q, uy = (y - oy) * to_q;
    return (ux >= 0.0f && ux < 65535.0f && uy >= 0.0f && uy < 65535.0f) ? ((uint32_t)ux | ((uint32_t)uy << 16)) : SB_COOP_ANYWHERE; // (false for NaN)
}
SB_DEV bool sb_lists_cooperative(const SbGrid &g, const SbGridGeom &m, uint32_t p0, uint32_t n_own, const uint32_t *__restrict__ pslot,
                                 const uint32_t *__restrict__ islot, unsigned char *area, uint32_t area_bytes
#ifdef SB_STAMPS
                                 , const SbGridStep &dbg, uint64_t sb_t_start
#endif
)
{
    SbCoopMeta *meta = (SbCoopMeta *)area;
    const uint32_t cap = (area_bytes - (uint32_t)sizeof(SbCoopMeta)) / 12u; // per record 8 bytes, per cell 4 (first record | count << 16)
    uint2 *s_rec = (uint2 *)(area + sizeof(SbCoopMeta));
    uint32_t *s_cell = (uint32_t *)(s_rec + cap);
    const uint32_t tid = threadIdx.x, nthreads = blockDim.x, lane = tid & 63u;
    if (tid == 0u) {
        meta->minx = meta->miny = 0x7fffffff;
        meta->maxx = meta->maxy = -1;
        meta->nrec = meta->fail = 0u;
    }
    __syncthreads();
    { // the rectangle of the tile's own cells (reduced per wave first: 512 lanes on four LDS words would queue up behind each other)
        int lox = 0x7fffffff, hix = -1, loy = 0x7fffffff, hiy = -1;
        for (uint32_t i = tid; i < n_own; i += nthreads) {
            const uint32_t c = g.cell_of[p0 + i];
            const int cx = (int)(c % m.nx), cy = (int)(c / m.nx);
            lox = min(lox, cx);
            hix = max(hix, cx);
            loy = min(loy, cy);
            hiy = max(hiy, cy);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lox = min(lox, __shfl_xor(lox, off, 64));
            hix = max(hix, __shfl_xor(hix, off, 64));
            loy = min(loy, __shfl_xor(loy, off, 64));
            hiy = max(hiy, __shfl_xor(hiy, off, 64));
        }
        if (lane == 0u) {
            atomicMin(&meta->minx, lox);
            atomicMax(&meta->maxx, hix);
            atomicMin(&meta->miny, loy);
            atomicMax(&meta->maxy, hiy);
        }
    }
    __syncthreads();
    SB_STAMP(dbg, 10);
    const int x_lo = meta->minx - 1, y_lo = meta->miny - 1; // (with the rim)
    const uint32_t w = (uint32_t)(meta->maxx - meta->minx + 3), h = (uint32_t)(meta->maxy - meta->miny + 3);
    if (n_own == 0u || cap < 64u || cap > 0xFFF0u || (unsigned long long)w * h > cap) return false; // (uniform: every thread reads the same words)
    const uint32_t ncell = w * h;
    // the rectangle in world units and the quantisation step
    const float ox = m.x0 + (float)x_lo * m.cell, oy = m.y0 + (float)y_lo * m.cell;
    const float extent = fmaxf((float)w, (float)h) * m.cell, to_q = 65536.0f / extent, res = extent * (1.0f / 65536.0f);
    // ---- every chain of the rectangle once: SB_COOP_CPT cells per thread side by side (heads together, first records together;
    // a chain longer than one record -- rare: cells are about one particle wide -- is walked to its end for the count and again
    // to fill), a contiguous block of records per cell, allocated per wave (one LDS add per wave, not per record)
    for (uint32_t base = 0; base < ncell; base += nthreads * SB_COOP_CPT) {
        uint32_t first[SB_COOP_CPT], count[SB_COOP_CPT];
        float4 r0[SB_COOP_CPT];
        unsigned long long hd[SB_COOP_CPT];
#pragma unroll
        for (int u = 0; u < SB_COOP_CPT; u++) {
            const uint32_t t = base + tid + (uint32_t)u * nthreads;
            const int gx = x_lo + (int)(t % w), gy = y_lo + (int)(t / w);
            const bool in = t < ncell && gx >= 0 && gy >= 0 && gx < (int)m.nx && gy < (int)m.ny;
            hd[u] = g.head[in ? (uint32_t)gy * m.nx + (uint32_t)gx : 0u];
            if (!in) hd[u] = 0ull;
        }
#pragma unroll
        for (int u = 0; u < SB_COOP_CPT; u++) {
            first[u] = (uint32_t)(hd[u] >> 32) == m.gen ? (uint32_t)hd[u] : SB_CHAIN_END; // (gen >= 1: 0 is never current)
            r0[u] = g.rec[first[u] != SB_CHAIN_END ? first[u] : 0u];
        }
        uint32_t mine = 0u;
#pragma unroll
        for (int u = 0; u < SB_COOP_CPT; u++) {
            count[u] = first[u] != SB_CHAIN_END ? 1u : 0u;
            if (count[u]) {
                uint32_t cur = __float_as_uint(r0[u].w);
                while (cur != SB_CHAIN_END && count[u] < 0xFFFFu) { // (rare)
                    count[u]++;
                    cur = __float_as_uint(g.rec[cur].w);
                }
            }
            mine += count[u];
        }
        // this wave's block of records: an inclusive scan of the lanes' totals, one add by the last lane
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += v;
        }
        uint32_t wave_base = 0u;
        if (lane == 63u) wave_base = atomicAdd(&meta->nrec, incl);
        wave_base = __shfl(wave_base, 63, 64);
        uint32_t at = wave_base + incl - mine;
        if (at + mine > cap) {
            meta->fail = 1u;
            mine = 0u; // (nothing is written; the caller falls back)
        }
#pragma unroll
        for (int u = 0; u < SB_COOP_CPT; u++) {
            const uint32_t t = base + tid + (uint32_t)u * nthreads;
            if (t < ncell) s_cell[t] = mine ? (at | (count[u] << 16)) : 0u;
            if (mine && count[u]) {
                s_rec[at] = make_uint2(sb_coop_quantise(r0[u].x, r0[u].y, ox, oy, to_q), __float_as_uint(r0[u].z));
                uint32_t cur = __float_as_uint(r0[u].w);
                for (uint32_t j = 1; j < count[u]; j++) { // (rare)
                    const float4 r = g.rec[cur];
                    s_rec[at + j] = make_uint2(sb_coop_quantise(r.x, r.y, ox, oy, to_q), __float_as_uint(r.z));
                    cur = __float_as_uint(r.w);
                }
                at += count[u];
            }
        }
    }
    __syncthreads();
    SB_STAMP(dbg, 11);
    if (meta->fail != 0u) return false;
    // ---- every own particle: its neighbours out of LDS, in ascending slot order
    const float reach = sb_sqrt(m.reach2) + 3.0f * res, reach_q2 = reach * reach * to_q * to_q; // (m.reach2 already carries a rounding margin)
    for (uint32_t i = tid; i < n_own; i += nthreads) {
        const uint32_t gi = p0 + i, c = g.cell_of[gi], own_slot = pslot[gi];
        const float4 own = g.rec[gi];
        const float px = (own.x - ox) * to_q, py = (own.y - oy) * to_q; // (in quantisation steps, like the records; NaN: every test below fails -> kept)
        const uint32_t lx = (uint32_t)((int)(c % m.nx) - x_lo), ly = (uint32_t)((int)(c / m.nx) - y_lo);
        uint32_t sc[9]; // the nine cells' blocks, requested together
#pragma unroll
        for (int nb = 0; nb < 9; nb++) sc[nb] = s_cell[(ly + (uint32_t)(nb / 3) - 1u) * w + (lx + (uint32_t)(nb % 3) - 1u)];
        uint32_t n = 0u, last = 0u;
        bool have_last = false;
        for (;;) {
            uint32_t bs[SB_NL_SEL];
#pragma unroll
            for (int q = 0; q < SB_NL_SEL; q++) bs[q] = 0xFFFFFFFFu;
#pragma unroll
            for (int nb = 0; nb < 9; nb++) {
                const uint32_t k0 = sc[nb] & 0xFFFFu, kn = sc[nb] >> 16;
                for (uint32_t j = 0; j < kn; j++) {
                    const uint2 r = s_rec[k0 + j];
                    uint32_t slot = r.y;
                    const float dx = ((float)(r.x & 0xFFFFu) + 0.5f) - px, dy = ((float)(r.x >> 16) + 0.5f) - py;
                    const bool far = r.x != SB_COOP_ANYWHERE && dx * dx + dy * dy > reach_q2; // (NaN compares false: kept)
                    if ((slot == own_slot) | (have_last & (slot <= last)) | (slot >= bs[SB_NL_SEL - 1]) | far) continue;
#pragma unroll
                    for (int q = 0; q < SB_NL_SEL; q++) { // carry the larger one down the registers
                        const bool lt = slot < bs[q];
                        const uint32_t ts = lt ? bs[q] : slot;
                        bs[q] = lt ? slot : bs[q];
                        slot = ts;
                    }
                }
            }
            bool full = true;
#pragma unroll
            for (int q = 0; q < SB_NL_SEL; q++) {
                if (bs[q] == 0xFFFFFFFFu) {
                    full = false;
                } else if (n != SB_NL_OVERFLOW) {
                    if (n == SB_NL_CAP) {
                        n = SB_NL_OVERFLOW;
                    } else {
                        g.nl[n * g.nl_stride + gi] = islot[bs[q]];
                        n++;
                        last = bs[q];
                        have_last = true;
                    }
                }
            }
            if (!full || n == SB_NL_OVERFLOW) break;
        }
        g.nl_count[gi] = n;
    }
    return true;
}

// The collision loop of compute.wgsl:144-170 over particle i's neighbour list (`count` = nl_count[i], fetched
// by the caller ahead of time).  The list is slot-sorted and every test uses the frozen copy `self`, so
// walking it and applying the contacts found is the ascending-slot order of the all-pairs scan.
#ifndef SB_NL_BATCH
#define SB_NL_BATCH 2
#endif
SB_DEV void sb_collide_list(const SbGrid &g, uint32_t count, const SbParams &prm, float friction,
                            float elasticity_coeff, SbParticle &particle, const SbParticle &self, uint32_t i,
                            const uint32_t *__restrict__ pidx, const float2 *__restrict__ pos_r,
                            const float2 *__restrict__ vel_r)
{
    const float two_r = prm.particle_radius * 2.0f;
    const float far2 = two_r * two_r * 1.001f;
    // Entries are taken SB_NL_BATCH at a time (2 measured best: 54.0 / 48.2 / 49.2 us per substep of the config-3 pile for 1 / 2 / 4):
    // all their indices are requested together, then all their positions, then
    // the velocities of those near enough to matter -- three dependent rounds of loads per batch instead of two per
    // entry (a list of four used to cost eight dependent L2 latencies; the walk is latency-bound).  The contacts of
    // a batch are still applied in list order, i.e. ascending slot order.
    for (uint32_t k0 = 0; k0 < count; k0 += SB_NL_BATCH) { // count is a real length here, never SB_NL_OVERFLOW
        uint32_t id[SB_NL_BATCH];
        float2 q[SB_NL_BATCH], ov[SB_NL_BATCH];
        float d2[SB_NL_BATCH];
#pragma unroll
        for (int j = 0; j < SB_NL_BATCH; j++) id[j] = k0 + (uint32_t)j < count ? g.nl[(k0 + (uint32_t)j) * g.nl_stride + i] : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < SB_NL_BATCH; j++) q[j] = id[j] != 0xFFFFFFFFu ? pos_r[id[j]] : make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < SB_NL_BATCH; j++) {
            const float ex = q[j].x - self.p.x, ey = q[j].y - self.p.y;
            d2[j] = ex * ex + ey * ey; // exactly the argument length() takes the root of
            // sqrt is monotone: d2 clearly above (2r)^2 cannot give d < 2r, and is not 0
            if (id[j] == 0xFFFFFFFFu || d2[j] > far2) id[j] = 0xFFFFFFFFu;
        }
#pragma unroll
        for (int j = 0; j < SB_NL_BATCH; j++) ov[j] = id[j] != 0xFFFFFFFFu ? vel_r[id[j]] : make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < SB_NL_BATCH; j++) {
            if (id[j] == 0xFFFFFFFFu) continue;
            const float ex = q[j].x - self.p.x, ey = q[j].y - self.p.y;
            const float d = sb_wave_all(sb_in_sqrt_gate(d2[j])) ? sb_sqrt_gated(d2[j]) : sb_sqrt(d2[j]);
            if (d == 0.0f || d < two_r)
                sb_collide_pair_at(prm, friction, elasticity_coeff, particle, self, pidx[i], pidx[id[j]], ex, ey, d, ov[j]);
        }
    }
}

// The uncommon cases in one place (the callers keep them out of their main loop, so that its registers are
// not sized by code that almost never runs): on the substep after a hash build the list is made first; a
// pile denser than the list holds is served by the cell scan.
SB_DEV void sb_collide_slow(const SbGrid &g, bool fresh, const SbParams &prm, float friction, float elasticity_coeff,
                            SbParticle &particle, const SbParticle &self, uint32_t i,
                            const uint32_t *__restrict__ pidx, const float2 *__restrict__ pos_r,
                            const float2 *__restrict__ vel_r)
{
    const SbGridGeom m = sb_grid_geom_load(g.geo); // only these paths need the geometry of the current hash
    const uint32_t count = fresh ? sb_neighbour_list_build(g, m, i) : g.nl_count[i];
    if (count == SB_NL_OVERFLOW) {
        const SbGridHood hood = sb_grid_hood(g, m, g.cell_of[i]);
        sb_collide_grid(g, m, hood, prm, friction, elasticity_coeff, particle, self, i, pidx, pos_r, vel_r);
    } else {
        sb_collide_list(g, count, prm, friction, elasticity_coeff, particle, self, i, pidx, pos_r, vel_r);
    }
}

// everything after the collision loop, compute.wgsl:171-199; fx,fy are the complete fixed-point
// beam force sums for this particle (the atomicExchange results of :184-185).
// PLAIN (compile time; the host launches that variant when the constants of the call say so): drag_exp == 2 and no mouse
// grab -- the reference's defaults (engineMapping.ts:271; mouse_active is 0 unless a button is down).  The general form
// dispatches pow() over five cases and tests the mouse per particle: all uniform, all scalar branches, about a dozen of them
// per particle in a kernel whose scalar instructions were a quarter of its vector ones (r02 counters).  Same bits: the case
// pow() takes for y == 2 is x * x.
template <bool PLAIN = false>
SB_DEV void sb_particle_finish(const SbParams &prm, const SbConsts &c, SbParticle &particle,
                               int32_t fx, int32_t fy)
{
    const float particle_force_scale = 65536.0f;
    particle.a.x += c.gravity_x; // :172
    particle.a.y += c.gravity_y;
    const float v2 = particle.v.x * particle.v.x + particle.v.y * particle.v.y;
    float vl, inv_vl = 0.0f;
    if (sb_wave_all(sb_in_sqrt_gate(v2) || v2 == 0.0f)) { // (a particle at rest has no drag: nothing to take a root of)
        vl = v2 == 0.0f ? 0.0f : sb_sqrt_gated(v2);
        inv_vl = sb_rcp_gated(vl); // unused (infinite) at rest
    } else {
        vl = sb_sqrt(v2);
        if (vl > 0.0f) inv_vl = sb_div(1.0f, vl);
    }
    if (vl > 0.0f) { // :174-176
        float nx = particle.v.x * inv_vl, ny = particle.v.y * inv_vl;
        float px = PLAIN ? particle.v.x * particle.v.x : sb_pow(sb_abs(particle.v.x), c.drag_exp); // (|x| * |x| == x * x, bit for bit)
        float py = PLAIN ? particle.v.y * particle.v.y : sb_pow(sb_abs(particle.v.y), c.drag_exp);
        particle.a.x -= c.drag_coeff * px * nx;
        particle.a.y -= c.drag_coeff * py * ny;
    }
    particle.a.x += c.applied_force_x * c.user_strength; // :178
    particle.a.y += c.applied_force_y * c.user_strength;
    if (!PLAIN && c.mouse_active > 0u) { // :179-181
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
