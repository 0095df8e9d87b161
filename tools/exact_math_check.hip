// exact_math_check.hip -- EXHAUSTIVE check (all 2^32 binary32 bit patterns) of the short correctly
// rounded sqrt / reciprocal sequences of sb_physics.h against the compiler's IEEE sequences
// (__builtin_sqrtf, 1.0f / x under -fhip-fp32-correctly-rounded-divide-sqrt) on the GPU that will run
// them.  The canonical arithmetic pins length() to the correctly rounded sqrt and
// normalize() to one correctly rounded reciprocal; any instruction sequence that returns the same bits
// for every input is the same function.  A unary binary32 function has only 2^32 inputs, so "every
// input" is checked literally, on the hardware's own v_rsq_f32 / v_rcp_f32 / v_sqrt_f32.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         -o exact_math_check tools/exact_math_check.hip && ./exact_math_check
// Prints one line per candidate: inputs in its domain, mismatches, first few failing inputs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r)); exit(1); } } while (0)

#define NCAND 12
#define NEX 8

struct Result {
    unsigned long long domain[NCAND], bad[NCAND];
    uint32_t ex[NCAND][NEX];
};

// the fast-path gate of sb_physics.h: 2^-90 <= x <= 2^90 (positive, normal, far from both ends)
__device__ __forceinline__ bool in_gate(float x) { return x >= 0x1p-90f && x <= 0x1p90f; }

__device__ __forceinline__ float sqrt_ieee(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float rcp_ieee(float x) { return 1.0f / x; }

// S1: rsq seed, one coupled step
__device__ __forceinline__ float sqrt_s1(float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    const float g0 = x * y0, h0 = 0.5f * y0;
    const float d = __builtin_fmaf(-g0, g0, x);
    return __builtin_fmaf(d, h0, g0);
}
// S2: rsq seed, Goldschmidt refinement, then the final step
__device__ __forceinline__ float sqrt_s2(float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    const float g0 = x * y0, h0 = 0.5f * y0;
    const float r0 = __builtin_fmaf(-h0, g0, 0.5f);
    const float g1 = __builtin_fmaf(g0, r0, g0), h1 = __builtin_fmaf(h0, r0, h0);
    const float d = __builtin_fmaf(-g1, g1, x);
    return __builtin_fmaf(d, h1, g1);
}
// S3: v_sqrt seed, neighbour residuals (the compiler's own fix-up without the range scaling)
__device__ __forceinline__ float sqrt_s3(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    float r = (rd <= 0.0f) ? sd : s;
    r = (ru > 0.0f) ? su : r;
    return r;
}
// S4: v_sqrt seed + one Newton step with h = 0.5 * rsq
__device__ __forceinline__ float sqrt_s4(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(d, h, s);
}
// S5: S1 followed by a second residual step (same h0)
__device__ __forceinline__ float sqrt_s5(float x)
{
    const float y0 = __builtin_amdgcn_rsqf(x);
    const float g0 = x * y0, h0 = 0.5f * y0;
    const float d0 = __builtin_fmaf(-g0, g0, x);
    const float g1 = __builtin_fmaf(d0, h0, g0);
    const float d1 = __builtin_fmaf(-g1, g1, x);
    return __builtin_fmaf(d1, h0, g1);
}

// reciprocal of len with seed y0 (~1/len): one or two Newton steps
__device__ __forceinline__ float rcp_n1(float len, float y0)
{
    const float e = __builtin_fmaf(-len, y0, 1.0f);
    return __builtin_fmaf(y0, e, y0);
}
__device__ __forceinline__ float rcp_n2(float len, float y0)
{
    const float y1 = rcp_n1(len, y0);
    const float e = __builtin_fmaf(-len, y1, 1.0f);
    return __builtin_fmaf(y1, e, y1);
}
// Markstein-style final step: q = y1 + y1 * (1 - len * y1) is n2; variant with the residual taken against y0
__device__ __forceinline__ float rcp_n1r(float len, float y0)
{
    // one step, then a correction that re-uses the seed as the multiplier (cheaper dependency chain)
    const float e0 = __builtin_fmaf(-len, y0, 1.0f);
    const float y1 = __builtin_fmaf(y0, e0, y0);
    const float e1 = __builtin_fmaf(-len, y1, 1.0f);
    return __builtin_fmaf(y0, e1, y1);
}

__global__ void check(uint32_t base, Result *res)
{
    const uint32_t bits = base + blockIdx.x * blockDim.x + threadIdx.x;
    const float x = __uint_as_float(bits);
    auto report = [&](int c, bool in_domain, float got, float want) {
        if (!in_domain) return;
        atomicAdd(&res->domain[c], 1ull);
        if (__float_as_uint(got) != __float_as_uint(want)) {
            unsigned long long k = atomicAdd(&res->bad[c], 1ull);
            if (k < NEX) res->ex[c][k] = bits;
        }
    };
    const bool g = in_gate(x);
    const float s = sqrt_ieee(x);
    // sqrt candidates over the gate
    report(0, g, sqrt_s1(x), s);
    report(1, g, sqrt_s2(x), s);
    report(2, g, sqrt_s3(x), s);
    report(3, g, sqrt_s4(x), s);
    report(4, g, sqrt_s5(x), s);
    // reciprocal of len = sqrt(x), seeded with rsq(x) (shares the transcendental with the sqrt)
    const float want = rcp_ieee(s);
    const float y0 = __builtin_amdgcn_rsqf(x);
    report(5, g, rcp_n1(s, y0), want);
    report(6, g, rcp_n2(s, y0), want);
    report(7, g, rcp_n1r(s, y0), want);
    // reciprocal of x itself, seeded with v_rcp_f32(x), over 2^-45 <= x <= 2^45 (every length a gated sqrt can return)
    const bool gr = x >= 0x1p-45f && x <= 0x1p45f;
    const float r0 = __builtin_amdgcn_rcpf(x), wr = rcp_ieee(x);
    report(8, gr, rcp_n1(x, r0), wr);
    report(9, gr, rcp_n2(x, r0), wr);
    report(10, gr, r0, wr);                          // how often the bare instruction is already right
    report(11, g, __builtin_amdgcn_sqrtf(x), s);     // likewise for v_sqrt_f32
}

int main()
{
    Result *d;
    CK(hipMalloc(&d, sizeof(Result)));
    CK(hipMemset(d, 0, sizeof(Result)));
    const uint32_t chunk = 1u << 24;
    for (uint64_t base = 0; base < (1ull << 32); base += chunk) {
        check<<<chunk / 256, 256>>>((uint32_t)base, d);
        CK(hipGetLastError());
    }
    CK(hipDeviceSynchronize());
    Result h;
    CK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
    const char *names[NCAND] = {
        "sqrt S1 rsq seed + 1 step            ", "sqrt S2 rsq seed + goldschmidt + step ", "sqrt S3 v_sqrt + neighbour residuals  ",
        "sqrt S4 v_sqrt + step (h = rsq/2)     ", "sqrt S5 rsq seed + 2 steps            ", "rcp(len) rsq seed, 1 newton step      ",
        "rcp(len) rsq seed, 2 newton steps     ", "rcp(len) rsq seed, step + seed-step   ", "rcp(x) v_rcp seed, 1 newton step      ",
        "rcp(x) v_rcp seed, 2 newton steps     ", "rcp(x) bare v_rcp_f32                 ", "sqrt bare v_sqrt_f32                  "};
    int rc = 0;
    for (int c = 0; c < NCAND; c++) {
        printf("%s domain %llu mismatches %llu", names[c], h.domain[c], h.bad[c]);
        for (unsigned long long k = 0; k < h.bad[c] && k < NEX; k++) printf(" %08x", h.ex[c][k]);
        printf("\n");
    }
    CK(hipFree(d));
    return rc;
}
