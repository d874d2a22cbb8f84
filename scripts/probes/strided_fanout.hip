// Probe: what would a y pass cost that READS one spectrum and WRITES five (the pointwise stage of a sweep that recomputes the
// iterate, fused into the strided pass instead of the contiguous one)?  Same tiles as k_ffty (rows n*16 bytes apart, W complex
// columns), one source plane per kx, five destination planes; nothing computed.  Against it: the in-place pass (strided_rw.hip).
//   hipcc --offload-arch=gfx950 -O3 -o strided_fanout strided_fanout.hip && ./strided_fanout [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cd;

template <int WCOLS, int WL, int NF, bool NT>
__global__ __launch_bounds__(512, 2) void k_fan(const cd* __restrict__ src, cd* __restrict__ dst, int n, size_t fstride) {
    const int lx = threadIdx.x % WL, ly = threadIdx.x / WL;
    constexpr int RS = 512 / WL, VEC = WCOLS / WL;
    const size_t off = (size_t)blockIdx.y * n * n + (size_t)blockIdx.x * WCOLS;
    for (int r = ly; r < n; r += RS) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const size_t i = off + (size_t)r * n + lx * VEC + v;
            const cd x = src[i];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                cd* p = dst + f * fstride + i;
                if (NT) {
                    __builtin_nontemporal_store(x.x + f, &p->x);
                    __builtin_nontemporal_store(x.y, &p->y);
                } else {
                    *p = cd{x.x + f, x.y};
                }
            }
        }
    }
}

template <class F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-52s %8.3f ms  %7.1f GB/s\n", name, ms / 3, bytes / (ms / 3) / 1e6);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const int planes = n / 2 + 1;
    const size_t fstride = (size_t)planes * n * n;
    cd *src, *dst;
    if (hipMalloc(&src, sizeof(cd) * fstride) != hipSuccess || hipMalloc(&dst, sizeof(cd) * fstride * 5) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    (void)hipMemset(src, 0, sizeof(cd) * fstride);
    const double bytes = (double)sizeof(cd) * fstride * 6;
#define FAN(WC, WL, NT_, TAG) \
    timeit("1 in / 5 out, W=" #WC " lanes/row=" #WL TAG, bytes, \
           [&] { hipLaunchKernelGGL((k_fan<WC, WL, 5, NT_>), dim3(n / WC, planes), dim3(512), 0, 0, src, dst, n, fstride); });
    FAN(8, 8, false, "") FAN(8, 8, true, " nontemporal") FAN(16, 16, false, "") FAN(16, 16, true, " nontemporal")
    FAN(32, 16, true, " nontemporal") FAN(64, 32, true, " nontemporal")
    return 0;
}
