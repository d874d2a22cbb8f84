// Streaming kernels and the 8 XCDs: does it matter which XCD touches which part of an array?  Workgroup b runs on XCD b % 8.
// copy / read-modify-write / read-only over 2 GiB with a grid-stride loop of 16-byte accesses;
//   order 0: workgroup b takes chunk b of every pass (the 8 XCDs share every 32 KiB window - what all pointwise kernels do)
//   order 1: within a pass every XCD takes a contiguous eighth of the pass
//   order 2: every XCD owns a contiguous eighth of the whole array
// hipcc --offload-arch=gfx950 -O3 -o xcd_stream_probe xcd_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int ORDER, int OP>
__global__ __launch_bounds__(256) void k(const double2* __restrict__ in, double2* __restrict__ out, size_t n, double* sink) {
    const unsigned G = gridDim.x, b = blockIdx.x;
    const size_t per_pass = (size_t)G * 256, passes = n / per_pass;
    double acc = 0.0;
    for (size_t p = 0; p < passes; ++p) {
        size_t i;
        if (ORDER == 0) i = p * per_pass + (size_t)b * 256 + threadIdx.x;
        else if (ORDER == 1) i = p * per_pass + ((size_t)(b & 7u) * (G >> 3) + (b >> 3)) * 256 + threadIdx.x;
        else i = (size_t)(b & 7u) * (n >> 3) + (p * (G >> 3) + (b >> 3)) * 256 + threadIdx.x;
        double2 v = in[i];
        if (OP == 0) out[i] = v;                       // copy
        else if (OP == 1) { v.x += 1.0; out[i] = v; }  // in place when out == in
        else acc += v.x + v.y;                         // read only
    }
    if (OP == 2 && acc == 1.2345) sink[0] = acc;
}

template <int ORDER, int OP>
float run(const double2* in, double2* out, size_t n, double* sink, hipEvent_t e0, hipEvent_t e1) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<ORDER, OP>), dim3(4096), dim3(256), 0, 0, in, out, n, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        if (rep && t < best) best = t;
    }
    return best;
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    double2 *a, *b;
    double* sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double gb = bytes / 1e9;
    printf("copy (read + write bytes):   order0 %6.0f GB/s  order1 %6.0f  order2 %6.0f\n", 2 * gb / (run<0, 0>(a, b, n, sink, e0, e1) * 1e-3),
           2 * gb / (run<1, 0>(a, b, n, sink, e0, e1) * 1e-3), 2 * gb / (run<2, 0>(a, b, n, sink, e0, e1) * 1e-3));
    printf("in place (read + write):     order0 %6.0f GB/s  order1 %6.0f  order2 %6.0f\n", 2 * gb / (run<0, 1>(a, a, n, sink, e0, e1) * 1e-3),
           2 * gb / (run<1, 1>(a, a, n, sink, e0, e1) * 1e-3), 2 * gb / (run<2, 1>(a, a, n, sink, e0, e1) * 1e-3));
    printf("read only:                   order0 %6.0f GB/s  order1 %6.0f  order2 %6.0f\n", gb / (run<0, 2>(a, b, n, sink, e0, e1) * 1e-3),
           gb / (run<1, 2>(a, b, n, sink, e0, e1) * 1e-3), gb / (run<2, 2>(a, b, n, sink, e0, e1) * 1e-3));
    return 0;
}
