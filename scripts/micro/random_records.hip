// Micro-benchmark: rate of dependent random record reads from a large array, as the traversal of a tree that
// does not fit any cache issues them.  Every lane reads `LOADS` x 16 bytes of one record at a pseudo-random
// index, folds the data into its next index (so the next read depends on this one), and repeats.
//   ./random_records <array MiB> <record stride bytes> <16-byte loads per record> <threads per block> <iterations>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LOADS>
__global__ __launch_bounds__(1024) void k_chase(const uint4 *__restrict__ base, unsigned long long n_rec, unsigned stride16, int iters,
                                                unsigned *__restrict__ out)
{
    unsigned long long x = (blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const unsigned long long r = x % n_rec;
        const uint4 *p = base + r * stride16;
        uint4 v[LOADS];
#pragma unroll
        for (int k = 0; k < LOADS; k++) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < LOADS; k++) acc += v[k].x ^ v[k].w;
        x += acc;                       // the next address depends on the data
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? atoll(argv[1]) : 1280;
    const unsigned stride = argc > 2 ? atoi(argv[2]) : 64;
    const int loads = argc > 3 ? atoi(argv[3]) : 3;
    const int threads = argc > 4 ? atoi(argv[4]) : 1024;
    const int iters = argc > 5 ? atoi(argv[5]) : 2000;
    const size_t bytes = mib << 20;
    uint4 *d = nullptr; unsigned *o = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 256 * 1024 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    const unsigned long long n_rec = bytes / stride;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        switch (loads) {
        case 1: hipLaunchKernelGGL(k_chase<1>, dim3(256), dim3(threads), 0, 0, d, n_rec, stride / 16, iters, o); break;
        case 3: hipLaunchKernelGGL(k_chase<3>, dim3(256), dim3(threads), 0, 0, d, n_rec, stride / 16, iters, o); break;
        case 4: hipLaunchKernelGGL(k_chase<4>, dim3(256), dim3(threads), 0, 0, d, n_rec, stride / 16, iters, o); break;
        case 5: hipLaunchKernelGGL(k_chase<5>, dim3(256), dim3(threads), 0, 0, d, n_rec, stride / 16, iters, o); break;
        case 6: hipLaunchKernelGGL(k_chase<6>, dim3(256), dim3(threads), 0, 0, d, n_rec, stride / 16, iters, o); break;
        case 8: hipLaunchKernelGGL(k_chase<8>, dim3(256), dim3(threads), 0, 0, d, n_rec, stride / 16, iters, o); break;
        default: printf("loads must be 1,3,4,5,6,8\n"); return 1;
        }
        hipEventRecord(e1);
        if (hipEventSynchronize(e1) != hipSuccess) { printf("kernel failed\n"); return 1; }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        const double recs = 256.0 * threads * iters;
        if (rep == 2)
            printf("array %zu MiB stride %u B loads %d x16B threads/CU %d : %.1f G records/s, %.2f TB/s of requested bytes, %.1f ms\n",
                   mib, stride, loads, threads, recs / ms / 1e6, recs * loads * 16 / ms / 1e9, ms);
    }
    return 0;
}
