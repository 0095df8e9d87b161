// valu_rate.hip -- how many cycles one SIMD of gfx950 spends per wave64 VALU instruction, by waves per SIMD.
// (MI355X_MICROARCH.md: v_fma_f32 2 cycles with several waves, 4 for one wave alone; this prints the same table on the
// box at hand, plus the packed and the transcendental forms the beam arithmetic uses.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void spin(float *out, int iters, float seed)
{
    float a[16];
    v2f p[8];
    for (int k = 0; k < 16; k++) a[k] = seed + k + threadIdx.x;
    for (int k = 0; k < 8; k++) p[k] = v2f{a[2 * k], a[2 * k + 1]};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (KIND == 0) a[k] = __builtin_fmaf(a[k], 1.0000001f, 0.5f);
            if (KIND == 1 && k < 8) p[k] = __builtin_elementwise_fma(p[k], v2f{1.0000001f, 1.0000001f}, v2f{0.5f, 0.5f});
            if (KIND == 2) a[k] = __builtin_amdgcn_rsqf(a[k]);
            if (KIND == 3) a[k] = a[k] * 1.0000001f;
            if (KIND == 4) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(a[k]) : "v"(a[k]));
        }
    }
    float s = 0;
    for (int k = 0; k < 16; k++) s += a[k];
    for (int k = 0; k < 8; k++) s += p[k].x + p[k].y;
    if (s == 12345.678f) out[0] = s;
}
template <int KIND> void run(const char *name, int per_iter)
{
    float *d;
    CK(hipMalloc(&d, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-28s", name);
    for (int wps = 1; wps <= 8; wps *= 2) { // waves per SIMD: one block per CU, 4 * wps waves
        const int iters = 20000;
        spin<KIND><<<256, 256 * wps>>>(d, 100, 1.0f);
        CK(hipEventRecord(e0));
        spin<KIND><<<256, 256 * wps>>>(d, iters, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = (double)iters * per_iter * wps;
        printf("  %dw/SIMD %5.2f cyc/instr", wps, ms * 1e-3 * 2.4e9 / instr_per_simd);
    }
    printf("   (at 2.4 GHz)\n");
}
int main()
{
    run<0>("v_fma_f32", 16);
    run<1>("v_pk_fma_f32", 8);
    run<2>("v_rsq_f32", 16);
    run<3>("v_mul_f32", 16);
    run<4>("v_cvt_i32_f32", 16);
    return 0;
}
