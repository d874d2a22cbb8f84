// Probe: the y pass's access pattern by itself - tiles of ROWS rows x W complex columns, rows n*16 bytes apart (one kx
// plane of n x n complex values per (field, kx)), read and written back in place - for W = 8 (128-byte row segments, what
// k_ffty uses) and wider.  build: hipcc --offload-arch=gfx950 -O3 -o strided_rw strided_rw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cd;

template <int WCOLS, int WL, bool WRITE>
__global__ __launch_bounds__(512, 2) void k_tile(cd* __restrict__ a, int n, double* out) {
    // blockIdx.x: column tile, blockIdx.y: plane (kx and field folded)
    const int lx = threadIdx.x % WL, ly = threadIdx.x / WL;
    constexpr int RS = 512 / WL, VEC = WCOLS / WL;
    cd* __restrict__ base = a + (size_t)blockIdx.y * n * n + (size_t)blockIdx.x * WCOLS;
    double m = 0;
    for (int r = ly; r < n; r += RS) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            cd x = base[(size_t)r * n + lx * VEC + v];
            if (WRITE) {
                x.x += 1.0;
                base[(size_t)r * n + lx * VEC + v] = x;
            } else {
                m = fmax(m, fmax(x.x, x.y));
            }
        }
    }
    if (!WRITE && m > 1e300) out[0] = m;
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
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms / 3, bytes / (ms / 3) / 1e6);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const int planes = (n / 2 + 1) * 5;
    cd* a; double* out;
    const size_t total = (size_t)planes * n * n;
    if (hipMalloc(&a, sizeof(cd) * total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&out, 8);
    (void)hipMemset(a, 0, sizeof(cd) * total);
    const double bytes = (double)sizeof(cd) * total;
#define TILE(WC, WL, WR) \
    timeit(WR ? "tile W=" #WC " lanes/row=" #WL " read+write" : "tile W=" #WC " lanes/row=" #WL " read only", \
           bytes * (WR ? 2 : 1), [&] { hipLaunchKernelGGL((k_tile<WC, WL, WR>), dim3(n / WC, planes), dim3(512), 0, 0, a, n, out); });
    TILE(8, 8, false) TILE(16, 16, false) TILE(16, 8, false) TILE(32, 16, false) TILE(64, 32, false)
    TILE(8, 8, true) TILE(16, 16, true) TILE(16, 8, true) TILE(32, 16, true) TILE(64, 32, true)
    return 0;
}
