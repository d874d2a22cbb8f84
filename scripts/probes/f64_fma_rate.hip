// What the f64 vector pipes of this GPU deliver: independent fused multiply-add chains, nothing else.
//   hipcc --offload-arch=gfx950 -O3 -o f64_fma_rate f64_fma_rate.hip && ./f64_fma_rate
// Prints TFLOP/s and the clock that rate corresponds to (256 CUs x 4 SIMDs x 16 lanes x 2 flop per clock), for
// 1, 2, 4 and 8 waves per SIMD - the number to hold "f64 instructions per wave" counts of the sweep kernels against.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a, double b) {
    double x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    if (s == 1.2345e300) out[0] = s;
}

int main() {
    double* d;
    (void)hipMalloc((void**)&d, 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 1 << 14;
    constexpr int CH = 8;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;  // 256 threads = 4 waves = one per SIMD of a CU
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL((k_fma<CH>), dim3(blocks), dim3(256), 0, 0, d, iters, 0.999999, 1e-7);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double flops = 2.0 * CH * (double)iters * 256.0 * blocks;
            const double tf = flops / (ms * 1e-3) / 1e12;
            if (rep == 2)
                printf("%d wave(s)/SIMD: %.2f ms  %.1f TFLOP/s  = %.2f GHz x 256 CU x 4 SIMD x 16 lanes x 2\n", wps, ms, tf,
                       tf * 1e12 / (256.0 * 4 * 16 * 2) / 1e9);
        }
    }
    return 0;
}
