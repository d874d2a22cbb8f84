// what the runtime says about resident workgroups per CU for the launches of a 1024^3 sweep
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o build_variants/occ_probe scripts/probes/occupancy_probe.hip
#include "../../pysdc_amd/csrc/kernels_pointwise.hpp"
#include "../../pysdc_amd/csrc/kernels_fft.hpp"
#include <cstdio>
template <class K>
static void show(const char* name, K k, int threads, size_t lds) {
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, threads, lds);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k));
    printf("%-34s threads %4d dyn LDS %6zu static LDS %6zu regs %3d -> %d workgroups / CU (%s)\n", name, threads, lds,
           (size_t)fa.sharedSizeBytes, fa.numRegs, nb, hipGetErrorString(e));
}
int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, LDS per block max %zu, per CU %zu, regs per block %d, max threads / CU %d\n", p.gcnArchName, p.multiProcessorCount,
           (size_t)p.sharedMemPerBlock, (size_t)p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.maxThreadsPerMultiProcessor);
    constexpr int N = 1024, NF = 5;
    const size_t ldsz = (size_t)LayContig<N>::doubles(NF) * sizeof(double);
    show("k_spec_z<1024,5,4,0>", k_spec_z<N, NF, 4, 0>, 320, ldsz);
    show("k_spec_z<1024,5,1,0>", k_spec_z<N, NF, 1, 0>, 320, ldsz);
    constexpr int T = 8, P = N / 16;
    show("k_ffty<1024,8,+1>", k_ffty<N, T, +1>, P * T, (size_t)LayStrided<N, T>::doubles(T) * sizeof(double));
    show("k_fftx_inv<1024,8,norm>", k_fftx_inv<N, T, true, false>, P * T, (size_t)LayCols<N>::doubles(T) * sizeof(double));
    for (size_t l : {32768ul, 40960ul, 43520ul, 49152ul, 53248ul, 65536ul}) show("k_spec_z<1024,5,4,0> other LDS", k_spec_z<N, NF, 4, 0>, 320, l);
    return 0;
}
