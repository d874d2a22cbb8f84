// Does the 256 MB memory-side cache (Infinity Cache / MALL) of the MI355X serve a kernel that re-reads what the kernel
// before it wrote (or read)?  For buffer sizes from 32 MB to 1 GB: (a) write then read, (b) read then read, (c) read alone
// after flushing with a 2 GB stream.  Prints GB/s of the second kernel.  hipcc --offload-arch=gfx950 -O3 -o mall_probe mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_write(double2* p, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = double2{v, v + 1.0};
}
__global__ __launch_bounds__(256) void k_read(const double2* p, size_t n, double* out) {
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = p[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_rw(double2* p, size_t n) {  // in place: read, modify, write
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double2 v = p[i];
        v.x += 1.0;
        p[i] = v;
    }
}

int main() {
    const size_t big = (size_t)2 << 30;
    double2 *buf, *flush;
    double* out;
    CK(hipMalloc(&buf, (size_t)1 << 30));
    CK(hipMalloc(&flush, big));
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 4096;
    auto flush_all = [&]() { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, flush, big / 16, 3.0); };
    for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024}) {
        const size_t n = mb * ((size_t)1 << 20) / 16;
        float ms[5] = {0, 0, 0, 0, 0};
        for (int rep = 0; rep < 3; ++rep) {
            // (a) write then read
            flush_all();
            hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, buf, n, 1.0);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[0] += t / 2;
            // (b) read then read
            flush_all();
            hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n, out);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[1] += t / 2;
            // (c) read cold
            flush_all();
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n, out);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[2] += t / 2;
            // (d) write then in-place read-modify-write (the y pass after the z pass)
            flush_all();
            hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, buf, n, 1.0);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_rw, dim3(grid), dim3(256), 0, 0, buf, n);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[3] += t / 2;
            // (e) in-place read-modify-write cold
            flush_all();
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_rw, dim3(grid), dim3(256), 0, 0, buf, n);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1)); if (rep) ms[4] += t / 2;
        }
        const double gb = mb / 1024.0;
        printf("%5zu MB: read after write %7.0f GB/s | read after read %7.0f | read cold %7.0f | rmw after write %7.0f (bytes r+w) | rmw cold %7.0f\n",
               mb, gb / (ms[0] * 1e-3), gb / (ms[1] * 1e-3), gb / (ms[2] * 1e-3), 2 * gb / (ms[3] * 1e-3), 2 * gb / (ms[4] * 1e-3));
    }
    return 0;
}
