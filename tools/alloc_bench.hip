// alloc_bench.hip -- how long does it take to get 120 GB of device memory on this pool?  (engine_create_s of the signed-9 table
// varies between 0.4 and 5 s; the table build is 0.285 s of that.)  Build: hipcc --offload-arch=gfx950 -O2 -o tools/alloc_bench tools/alloc_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(unsigned long long *p, size_t stride_words, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i * stride_words] = i; }
int main()
{
    const size_t bytes = (size_t)120 << 30;
    hipFree(0);
    for (int rep = 0; rep < 3; rep++) {
        void *p = nullptr;
        double t = now();
        hipError_t e = hipMalloc(&p, bytes);
        double t1 = now();
        // one write per 2 MiB page: first-touch cost, if any
        size_t n = bytes / (2 << 20);
        hipLaunchKernelGGL(touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned long long *)p, (size_t)(2 << 20) / 8, n);
        hipDeviceSynchronize();
        double t2 = now();
        hipFree(p);
        double t3 = now();
        printf("hipMalloc(120 GiB) %s: alloc %.3f s, first touch of every 2 MiB %.3f s, free %.3f s\n", hipGetErrorString(e), t1 - t, t2 - t1, t3 - t2);
    }
    {
        hipStream_t s; hipStreamCreate(&s);
        void *p = nullptr;
        double t = now();
        hipError_t e = hipMallocAsync(&p, bytes, s);
        hipStreamSynchronize(s);
        double t1 = now();
        printf("hipMallocAsync(120 GiB) %s: %.3f s\n", hipGetErrorString(e), t1 - t);
        if (e == hipSuccess) { hipFreeAsync(p, s); hipStreamSynchronize(s); }
        t = now();
        e = hipMallocAsync(&p, bytes, s);
        hipStreamSynchronize(s);
        printf("hipMallocAsync again (pool reuse) %s: %.3f s\n", hipGetErrorString(e), now() - t);
    }
    return 0;
}
