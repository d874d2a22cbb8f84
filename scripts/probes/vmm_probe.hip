// Does the virtual-memory API work on this box?  Reserve a range of `fields` fields, map the first one, run a kernel over it,
// map the rest later, run a kernel over all of it; time the mapping of one 8 GiB field and the page-granular zero fill.
//   hipcc --offload-arch=gfx950 -O2 -o vmm_probe vmm_probe.hip && ./vmm_probe [field_MiB] [fields]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("FAILED %s: %s\n", #x, hipGetErrorString(e_));                      \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

__global__ void k_fill(double* p, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_sum(const double* p, size_t n, double* out) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    atomicAdd(out, s);
}

int main(int argc, char** argv) {
    const size_t field = (size_t)(argc > 1 ? atol(argv[1]) : 1024) << 20;
    const int fields = argc > 2 ? atoi(argv[2]) : 6;
    int dev = 0;
    CK(hipSetDevice(dev));
    int vmm = 0;
    CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("virtual memory management supported: %d\n", vmm);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    size_t gran_rec = 0;
    CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu\n", gran, gran_rec);
    size_t free0, total;
    CK(hipMemGetInfo(&free0, &total));
    void* base = nullptr;
    CK(hipMemAddressReserve(&base, field * fields, gran_rec, nullptr, 0));
    printf("reserved %zu MiB at %p\n", (field * fields) >> 20, base);
    std::vector<hipMemGenericAllocationHandle_t> h(fields);
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    double* sum;
    CK(hipMalloc(&sum, 8));
    auto map_one = [&](int f) -> int {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipMemCreate(&h[f], field, &prop, 0));
        CK(hipMemMap((char*)base + (size_t)f * field, field, 0, h[f], 0));
        CK(hipMemSetAccess((char*)base + (size_t)f * field, field, &acc, 1));
        auto t1 = std::chrono::steady_clock::now();
        printf("  field %d mapped in %.2f ms\n", f, std::chrono::duration<double, std::milli>(t1 - t0).count());
        return 0;
    };
    if (map_one(0)) return 1;
    size_t free1;
    CK(hipMemGetInfo(&free1, &total));
    printf("free memory went down by %zu MiB after one field\n", (free0 - free1) >> 20);
    const size_t n1 = field / 8;
    k_fill<<<1024, 256>>>((double*)base, n1, 1.0);
    CK(hipDeviceSynchronize());
    for (int f = 1; f < fields; ++f)
        if (map_one(f)) return 1;
    k_fill<<<4096, 256>>>((double*)base + n1, n1 * (fields - 1), 2.0);
    CK(hipMemset(sum, 0, 8));
    k_sum<<<4096, 256>>>((double*)base, n1 * fields, sum);
    double hs = 0;
    CK(hipMemcpy(&hs, sum, 8, hipMemcpyDeviceToHost));
    printf("sum %.1f expected %.1f\n", hs, (double)n1 * (1.0 + 2.0 * (fields - 1)));
    // a device-to-device copy and a peer-style memcpy through the range
    double* plain;
    CK(hipMalloc(&plain, field));
    CK(hipMemcpy(plain, (char*)base + field, field, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(&hs, plain + 5, 8, hipMemcpyDeviceToHost));
    printf("copied value %.1f (expected 2.0)\n", hs);
    // timing of a streaming kernel over mapped memory against hipMalloc'ed memory
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        k_fill<<<4096, 256>>>((double*)base, n1, 3.0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fill mapped : %.3f ms (%.0f GB/s)\n", ms, field / ms / 1e6);
        CK(hipEventRecord(e0));
        k_fill<<<4096, 256>>>(plain, n1, 3.0);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("fill malloc : %.3f ms (%.0f GB/s)\n", ms, field / ms / 1e6);
    }
    for (int f = 0; f < fields; ++f) {
        CK(hipMemUnmap((char*)base + (size_t)f * field, field));
        CK(hipMemRelease(h[f]));
    }
    CK(hipMemAddressFree(base, field * fields));
    size_t free2;
    CK(hipMemGetInfo(&free2, &total));
    printf("after release: free memory back to within %zd MiB (one plain field still held)\n", (ssize_t)(free0 - free2) >> 20);
    printf("OK\n");
    return 0;
}
