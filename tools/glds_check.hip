// LDS-DMA on gfx950 as k_substep_blocked uses it (known-answer check, run on the GPU box):
//   global_load_lds_dwordx4 voff, s[base]  -- a coalesced stream: lane l's 16 bytes land at M0 + 16 l
//   global_load_lds_dword   voff, s[base]  -- a gather: lane l's dword, from ITS OWN address, lands at M0 + 4 l
//   ... offset:4                           -- the instruction offset is added to the global AND to the LDS address
// M0 is written in the same asm statement that uses it and restored after the issue (it is read at issue).
//   hipcc -O3 --offload-arch=gfx950 -o scratch/glds_check tools/glds_check.hip && gpurun -- scratch/glds_check
#include <hip/hip_runtime.h>
__device__ __forceinline__ void glds16(const void *g, uint32_t voff, uint32_t lds)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void glds4(const void *g, uint32_t voff, uint32_t lds)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(g), "s"(lds) : "memory");
}
__device__ __forceinline__ void glds4_4(const void *g, uint32_t voff, uint32_t lds)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(g), "s"(lds) : "memory");
}
extern __shared__ float dyn[];
__global__ void k(const float *a, const uint32_t *idx, float *out, uint32_t n)
{
    const uint32_t tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const uint32_t base = (uint32_t)(uintptr_t)dyn;
    for (uint32_t c = wave * 1024u; c < n * 4u; c += (blockDim.x >> 6) * 1024u) {
        uint32_t off = c + lane * 16u;
        if (off < n * 4u) glds16(a, off, base + c);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint32_t c = wave * 64u; c < n; c += (blockDim.x >> 6) * 64u) {
        uint32_t h = c + lane;
        if (h < n) { uint32_t i = idx[h]; glds4(a, i * 8u, base + 16384u + c * 4u); glds4_4(a, i * 8u, base + 32768u + c * 4u - 4u); /* the instruction offset moves BOTH addresses: global +4 and LDS +4 */ }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint32_t i = tid; i < n; i += blockDim.x) out[i] = dyn[i] + dyn[4096 + i] + dyn[8192 + i];
}
#include <cstdio>
#include <vector>
int main()
{
    const uint32_t n = 3001, N = 20000;
    std::vector<float> a(2 * N);
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < 2 * N; i++) a[i] = (float)(i % 9973) * 0.5f;
    for (uint32_t i = 0; i < n; i++) idx[i] = (i * 7919u + 13u) % N;
    float *da, *dout; uint32_t *didx;
    hipMalloc(&da, a.size() * 4); hipMalloc(&dout, n * 4); hipMalloc(&didx, n * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(didx, idx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4), dim3(512), 49152, 0, da, didx, dout, n);
    std::vector<float> out(n);
    if (hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    uint32_t bad = 0;
    for (uint32_t i = 0; i < n; i++) { float e = a[i] + a[2 * idx[i]] + a[2 * idx[i] + 1]; if (out[i] != e) { if (bad < 5) printf("i=%u got %g exp %g\n", i, out[i], e); bad++; } }
    printf("glds test: %u bad of %u\n", bad, n);
    return bad != 0;
}
