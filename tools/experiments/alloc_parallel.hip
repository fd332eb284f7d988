// alloc_parallel.hip -- how long does this box take to hand out 192 GiB of HBM, and does it go faster in pieces taken by several host threads at once?
//   (a) one hipMalloc                                  (what bsgs_lines_malloc does for lines above 0.6 of the HBM: 3.9 s in profiles/r08j_startup_stages_36g.log)
//   (b) K threads, one hipMalloc of 1/K each           (K separate address ranges: not usable for one table, the reference point for (c))
//   (c) one reserved address range, K threads each hipMemCreate + hipMemMap 1/K of it, one hipMemSetAccess
// build: hipcc -O2 --offload-arch=gfx950 -o alloc_parallel alloc_parallel.hip -pthread ; run: ./alloc_parallel [GiB=192]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void touch(unsigned long long *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i; }
int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoll(argv[1]) : 192, bytes = gib << 30;
    CK(hipSetDevice(0));
    CK(hipFree(0));
    for (int rep = 0; rep < 2; rep++) {
        {   // (a)
            void *p = nullptr;
            double t = now();
            CK(hipMalloc(&p, bytes));
            const double ta = now() - t;
            t = now();
            hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (unsigned long long *)p, bytes / 8);
            CK(hipDeviceSynchronize());
            const double tt = now() - t;
            t = now();
            CK(hipFree(p));
            printf("{\"how\": \"one hipMalloc\", \"GiB\": %zu, \"alloc_s\": %.3f, \"first_touch_s\": %.3f, \"free_s\": %.3f}\n", gib, ta, tt, now() - t);
        }
        for (int K : {4, 8}) {   // (b)
            std::vector<void *> p(K, nullptr);
            std::vector<std::thread> th;
            double t = now();
            for (int k = 0; k < K; k++) th.emplace_back([&, k] { CK(hipSetDevice(0)); CK(hipMalloc(&p[k], bytes / K)); });
            for (auto &x : th) x.join();
            const double ta = now() - t;
            t = now();
            for (int k = 0; k < K; k++) CK(hipFree(p[k]));
            printf("{\"how\": \"%d threads, one hipMalloc each\", \"GiB\": %zu, \"alloc_s\": %.3f, \"free_s\": %.3f}\n", K, gib, ta, now() - t);
        }
        for (int K : {1, 4, 8, 16}) {   // (c)
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            const size_t piece = bytes / K;
            if (piece % gran) { printf("piece not a multiple of the granularity %zu\n", gran); continue; }
            void *va = nullptr;
            double t = now();
            CK(hipMemAddressReserve(&va, bytes, 2ull << 20, nullptr, 0));
            std::vector<hipMemGenericAllocationHandle_t> h(K);
            std::vector<std::thread> th;
            for (int k = 0; k < K; k++) th.emplace_back([&, k] { CK(hipSetDevice(0)); CK(hipMemCreate(&h[k], piece, &prop, 0)); CK(hipMemMap((char *)va + k * piece, piece, 0, h[k], 0)); });
            for (auto &x : th) x.join();
            const double tc = now() - t;
            hipMemAccessDesc acc = {};
            acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, bytes, &acc, 1));
            const double ta = now() - t;
            t = now();
            hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (unsigned long long *)va, bytes / 8);
            CK(hipDeviceSynchronize());
            const double tt = now() - t;
            t = now();
            CK(hipMemUnmap(va, bytes));
            for (int k = 0; k < K; k++) CK(hipMemRelease(h[k]));
            CK(hipMemAddressFree(va, bytes));
            printf("{\"how\": \"one address range, %d threads hipMemCreate + hipMemMap\", \"GiB\": %zu, \"create_map_s\": %.3f, \"alloc_s\": %.3f, \"first_touch_s\": %.3f, \"free_s\": %.3f}\n", K, gib, tc, ta, tt, now() - t);
        }
    }
    return 0;
}
