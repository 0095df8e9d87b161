// hbm_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// widths the softbody kernels use (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of the bytes
// of a 16 B/lane stream; "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern").  Each kernel moves a KNOWN number of bytes over a 1 GiB
// buffer (4x the 256 MiB Infinity Cache), so counter / known = the correction factor.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_calib tools/hbm_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./hbm_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t r = (x); if (r != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r)); exit(1); } } while (0)

template <typename T> __global__ void calib_read(const T *__restrict__ src, size_t n, T *sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    T acc{};
    for (; i < n; i += stride) {
        T v = src[i];
        const unsigned *p = (const unsigned *)&v;
        unsigned *q = (unsigned *)&acc;
        for (unsigned k = 0; k < sizeof(T) / 4; k++) q[k] ^= p[k];
    }
    const unsigned *q = (const unsigned *)&acc;
    unsigned x = 0;
    for (unsigned k = 0; k < sizeof(T) / 4; k++) x ^= q[k];
    if (x == 0x12345678u) *sink = acc; // never true for the fill pattern; keeps the loads live
}

template <typename T> __global__ void calib_write(T *dst, size_t n, unsigned seed)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        T v;
        unsigned *q = (unsigned *)&v;
        for (unsigned k = 0; k < sizeof(T) / 4; k++) q[k] = seed + (unsigned)i;
        dst[i] = v;
    }
}

int main()
{
    const size_t bytes = 1ull << 30;
    void *a, *b, *sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    dim3 g(2048), t(256);
    for (int rep = 0; rep < 3; rep++) {
        calib_read<unsigned><<<g, t>>>((const unsigned *)a, bytes / 4, (unsigned *)sink);
        calib_read<uint2><<<g, t>>>((const uint2 *)b, bytes / 8, (uint2 *)sink);
        calib_read<uint4><<<g, t>>>((const uint4 *)a, bytes / 16, (uint4 *)sink);
        calib_write<unsigned><<<g, t>>>((unsigned *)b, bytes / 4, rep);
        calib_write<uint2><<<g, t>>>((uint2 *)a, bytes / 8, rep);
        calib_write<uint4><<<g, t>>>((uint4 *)b, bytes / 16, rep);
    }
    CK(hipDeviceSynchronize());
    printf("each kernel moved %zu bytes\n", bytes);
    return 0;
}
