// Probe: what does the access pattern of the x pass (rows 16 MB apart, 128..1024-byte row segments) cost by itself?
// Read-only kernels over a [ROWS][REST] complex<double> array (ROWS = 513, REST = 1024*1024, 5 fields), max-reduced so
// that nothing is optimised away.  Variants:
//   stream      : plain coalesced streaming read of the same bytes (reference rate)
//   tile<W,R>   : a block of 512 threads reads a tile of ROWS rows x W*16 bytes; each wave-instruction covers
//                 R rows x (64/R)*16 contiguous bytes; rows per thread as in k_fftx_inv (9 loads of 2 x 16 B for W=16)
// build: hipcc --offload-arch=gfx950 -O3 -o strided_read strided_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double2 cd;

__device__ __forceinline__ double wmax(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void k_stream(const cd* __restrict__ a, size_t n, double* out) {
    double m = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        cd v = a[i];
        m = fmax(m, fmax(v.x, v.y));
    }
    m = wmax(m);
    if ((threadIdx.x & 63) == 0 && m > 1e300) out[0] = m;
}

// tile of ROWS x WCOLS complex columns; 512 threads; lane layout: WL consecutive lanes along a row (16 B each), then rows
template <int WCOLS, int WL, int VEC>
__global__ __launch_bounds__(512, 4) void k_tile(const cd* __restrict__ a, size_t fstride, int rows, size_t rest, double* out) {
    const int lx = threadIdx.x % WL, ly = threadIdx.x / WL;      // WL lanes per row segment, 512/WL row slots
    constexpr int RS = 512 / WL;
    const size_t c0 = (size_t)blockIdx.x * WCOLS;
    const cd* __restrict__ f = a + blockIdx.y * fstride;
    double m = 0;
    for (int r = ly; r < rows; r += RS) {
#pragma unroll
        for (int cc = 0; cc < WCOLS / (WL * VEC); ++cc) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                cd x = f[(size_t)r * rest + c0 + (size_t)(cc * WL + lx) * VEC + v];
                m = fmax(m, fmax(x.x, x.y));
            }
        }
    }
    m = wmax(m);
    if ((threadIdx.x & 63) == 0 && m > 1e300) out[0] = m;
}

template <class F>
static void timeit(const char* name, double bytes, F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.3f ms  %7.1f GB/s\n", name, ms / 3, bytes / (ms / 3) / 1e6);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024;
    const int rows = n / 2 + 1, nf = 5;
    const size_t rest = (size_t)n * n, fstride = rows * rest;
    cd* a; double* out;
    if (hipMalloc(&a, sizeof(cd) * fstride * nf) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 8);
    hipMemset(a, 0, sizeof(cd) * fstride * nf);
    const double bytes = (double)sizeof(cd) * fstride * nf;
    timeit("stream", bytes, [&] { hipLaunchKernelGGL(k_stream, dim3(65536), dim3(256), 0, 0, a, fstride * nf, out); });
#define TILE(WC, WL, V) \
    timeit("tile W=" #WC " cols, lanes/row=" #WL ", vec=" #V, bytes, [&] { \
        hipLaunchKernelGGL((k_tile<WC, WL, V>), dim3((unsigned)(rest / WC), nf), dim3(512), 0, 0, a, fstride, rows, rest, out); });
    TILE(16, 8, 2)    // the x pass: 256-B row segments, 8 lanes x 32 B, a wave-instruction touches 8 rows
    TILE(16, 16, 1)   // same tile, 16 lanes x 16 B per row
    TILE(8, 8, 1)     // 128-B segments
    TILE(32, 16, 2)   // 512-B segments, 4 rows per wave-instruction
    TILE(64, 32, 2)   // 1 KB segments, 2 rows per wave-instruction
    TILE(64, 64, 1)   // 1 KB segments, 1 row per wave-instruction
    TILE(128, 64, 2)  // 2 KB segments
    TILE(256, 64, 2)  // 4 KB segments
    return 0;
}
