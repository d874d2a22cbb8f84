// Probe: operand / result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (one f64 per lane for A, B, C/D; 4 blocks).
// For every lane la: A one-hot at la, B all ones  -> the D lanes that read 1 share la's (block, row i).
// For every lane lb: A all ones, B one-hot at lb  -> the D lanes that read 1 share lb's (block, col j).
// A one-hot at la and B one-hot at lb give a non-zero D iff block and k agree.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_layout mfma_f64_4x4x4_layout.hip ; prints the maps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

int main() {
    double *da, *db, *dd;
    hipMalloc(&da, 64 * 8); hipMalloc(&db, 64 * 8); hipMalloc(&dd, 64 * 8);
    std::vector<double> a(64), b(64), d(64);
    auto run = [&]() {
        hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost);
    };
    printf("A one-hot at lane la (B = 1): D lanes equal to 1\n");
    for (int la = 0; la < 64; ++la) {
        for (int l = 0; l < 64; ++l) { a[l] = l == la; b[l] = 1.0; }
        run();
        printf("la=%2d:", la);
        for (int l = 0; l < 64; ++l) if (d[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    printf("B one-hot at lane lb (A = 1): D lanes equal to 1\n");
    for (int lb = 0; lb < 64; ++lb) {
        for (int l = 0; l < 64; ++l) { b[l] = l == lb; a[l] = 1.0; }
        run();
        printf("lb=%2d:", lb);
        for (int l = 0; l < 64; ++l) if (d[l] != 0.0) printf(" %d", l);
        printf("\n");
    }
    printf("A one-hot at la, B one-hot at lb: pairs (la, lb) with a non-zero D (same block, same k), la < 16\n");
    for (int la = 0; la < 16; ++la) {
        printf("la=%2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            for (int l = 0; l < 64; ++l) { a[l] = l == la; b[l] = l == lb; }
            run();
            bool nz = false;
            for (int l = 0; l < 64; ++l) nz |= d[l] != 0.0;
            if (nz) printf(" %d", lb);
        }
        printf("\n");
    }
    return 0;
}
