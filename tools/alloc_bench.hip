// alloc_bench.hip -- how long does it take to get 120 GB of device memory on this pool?  (engine_create_s of the signed-9 table
// varies between 0.4 and 5 s; the table build is 0.285 s of that.)  Build: hipcc --offload-arch=gfx950 -O2 -o tools/alloc_bench tools/alloc_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
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
    // The same 120 GiB as several hipMallocs issued from parallel host threads (a table need not be contiguous: one allocation
    // per window): does the driver clear / map them concurrently?
    for (int parts : {9, 30, 120}) {
        std::vector<void *> ps(parts, nullptr);
        std::vector<double> done(parts, 0.0);
        const size_t each = bytes / parts;
        const double t = now();
        std::vector<std::thread> th;
        for (int i = 0; i < parts; i++)
            th.emplace_back([&, i]() {
                hipSetDevice(0);
                hipMalloc(&ps[i], each);
                done[i] = now() - t;
            });
        for (auto &x : th) x.join();
        const double t1 = now();
        double first = 1e9;
        for (double d : done) first = d < first ? d : first;
        for (void *p : ps) hipFree(p);
        printf("%d parallel hipMalloc of %.1f GiB: all done after %.3f s (first after %.3f s), free %.3f s\n", parts, each / 1073741824.0, t1 - t, first, now() - t1);
    }
    // sequential pieces: when does the FIRST piece become usable?
    {
        const int parts = 9;
        std::vector<void *> ps(parts, nullptr);
        const size_t each = bytes / parts;
        const double t = now();
        double first = 0;
        for (int i = 0; i < parts; i++) {
            hipMalloc(&ps[i], each);
            if (i == 0) first = now() - t;
        }
        const double t1 = now();
        for (void *p : ps) hipFree(p);
        printf("%d sequential hipMalloc of %.1f GiB: all done after %.3f s (first after %.3f s), free %.3f s\n", parts, each / 1073741824.0, t1 - t, first, now() - t1);
    }
    // The virtual-memory path: one reserved range, physical memory created and mapped in 1 GiB chunks.  If
    // this returns chunk by chunk, a table window could be built while later windows are still being mapped.
    for (int rep = 0; rep < 2; rep++) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        const size_t chunk = (size_t)1 << 30;
        const int nchunks = (int)(bytes / chunk);
        void *base = nullptr;
        double t = now();
        e = hipMemAddressReserve(&base, bytes, gran, nullptr, 0);
        if (e != hipSuccess) { printf("hipMemAddressReserve: %s\n", hipGetErrorString(e)); break; }
        double t_res = now();
        static hipMemGenericAllocationHandle_t h[256];
        double t_first = 0, worst = 0;
        int done = 0;
        for (int i = 0; i < nchunks; i++) {
            const double c0 = now();
            e = hipMemCreate(&h[i], chunk, &prop, 0);
            if (e != hipSuccess) { printf("hipMemCreate chunk %d: %s\n", i, hipGetErrorString(e)); break; }
            e = hipMemMap((char *)base + (size_t)i * chunk, chunk, 0, h[i], 0);
            if (e != hipSuccess) { printf("hipMemMap chunk %d: %s\n", i, hipGetErrorString(e)); break; }
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess((char *)base + (size_t)i * chunk, chunk, &acc, 1);
            if (e != hipSuccess) { printf("hipMemSetAccess chunk %d: %s\n", i, hipGetErrorString(e)); break; }
            const double c1 = now();
            if (i == 0) t_first = c1 - c0;
            if (c1 - c0 > worst) worst = c1 - c0;
            done++;
        }
        double t_map = now();
        if (done == nchunks) {
            size_t n = bytes / (2 << 20);
            hipLaunchKernelGGL(touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned long long *)base, (size_t)(2 << 20) / 8, n);
            e = hipDeviceSynchronize();
        }
        double t_touch = now();
        for (int i = 0; i < done; i++) {
            hipMemUnmap((char *)base + (size_t)i * chunk, chunk);
            hipMemRelease(h[i]);
        }
        hipMemAddressFree(base, bytes);
        double t_free = now();
        printf("VMM 120 x 1 GiB (granularity %zu): reserve %.4f s, create+map+access %.3f s (first chunk %.4f s, slowest %.4f s), touch %.3f s (%s), release %.3f s\n", gran,
               t_res - t, t_map - t_res, t_first, worst, t_touch - t_map, hipGetErrorString(e), t_free - t_touch);
    }
    return 0;
}
