// Stand-alone timing of the one-launch nested transfers (kernels_transfer.hpp) at config 5's sizes: 256^3 <-> 128^3, three
// fields.  hipcc --offload-arch=gfx950 -O3 -std=c++17 [-D...] -o build_variants/xfer_probe scripts/probes/xfer_nested_probe.hip
#include "../../pysdc_amd/csrc/kernels_transfer.hpp"
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#ifndef PROBE_TI
#define PROBE_TI 8
#endif
int main() {
    const int nc = 128, nf = 256, K = 3, W = 6;
    const size_t Nc = (size_t)nc * nc * nc, Nf = (size_t)nf * nf * nf;
    // tables: P (nf rows, width 6), R (nc rows, width 3), entry-major, columns ascending like np.nonzero
    std::vector<int> pi(W * nf), ri(3 * nc);
    std::vector<double> pw(W * nf), rw(3 * nc);
    const double w6[6] = {3.0 / 256, -25.0 / 256, 150.0 / 256, 150.0 / 256, -25.0 / 256, 3.0 / 256};
    for (int r = 0; r < nf; ++r) {
        std::vector<std::pair<int, double>> e;
        if (r % 2 == 0) e.push_back({r / 2, 1.0});
        else for (int t = 0; t < 6; ++t) e.push_back({((r / 2 - 2 + t) % nc + nc) % nc, w6[t]});
        std::sort(e.begin(), e.end());
        for (int c = 0; c < W; ++c) {
            pi[c * nf + r] = c < (int)e.size() ? e[c].first : e[0].first;
            pw[c * nf + r] = c < (int)e.size() ? e[c].second : 0.0;
        }
    }
    for (int i = 0; i < nc; ++i) {
        std::vector<std::pair<int, double>> e = {{(2 * i - 1 + nf) % nf, 0.25}, {2 * i, 0.5}, {2 * i + 1, 0.25}};
        std::sort(e.begin(), e.end());
        for (int c = 0; c < 3; ++c) { ri[c * nc + i] = e[c].first; rw[c * nc + i] = e[c].second; }
    }
    int *dpi, *dri; double *dpw, *drw, *fine, *coarse, *cold, *cout_;
    CK(hipMalloc(&dpi, pi.size() * 4)); CK(hipMalloc(&dri, ri.size() * 4));
    CK(hipMalloc(&dpw, pw.size() * 8)); CK(hipMalloc(&drw, rw.size() * 8));
    CK(hipMalloc(&fine, K * Nf * 8)); CK(hipMalloc(&coarse, K * Nc * 8)); CK(hipMalloc(&cold, K * Nc * 8)); CK(hipMalloc(&cout_, K * Nc * 8));
    CK(hipMemcpy(dpi, pi.data(), pi.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dri, ri.data(), ri.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpw, pw.data(), pw.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(drw, rw.data(), rw.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(fine, 0, K * Nf * 8)); CK(hipMemset(coarse, 0, K * Nc * 8)); CK(hipMemset(cold, 0, K * Nc * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    NestedArgs a; memset(&a, 0, sizeof a);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (which == 0) {
                a.in = fine; a.in_minus = nullptr; a.out = cout_; a.out_minus = cold; a.idx = dri; a.w = drw; a.n_out = nc; a.n_in = nf; a.W = 3; a.accumulate = 0;
#ifndef PROBE_JY
#define PROBE_JY 2
#endif
                const unsigned kx = 128, jy = PROBE_JY;
                hipLaunchKernelGGL((k_restrict3_nested<PROBE_TI>), dim3(nc / jy, nc / PROBE_TI, K), dim3(kx, jy), 0, 0, a, 1u);
            } else {
                a.in = coarse; a.in_minus = cold; a.out = fine; a.out_minus = nullptr; a.idx = dpi; a.w = dpw; a.n_out = nf; a.n_in = nc; a.W = W; a.accumulate = 1;
                hipLaunchKernelGGL((k_prolong3_nested<6>), dim3(nc / 8, nc / 8, (nc / 8) * K), dim3(256), 0, 0, a);
            }
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 1 && ms < best) best = ms;
        }
        printf("%s: %.1f us\n", which == 0 ? "restrict 3 x 256^3 -> 128^3" : "prolong  3 x 128^3 -> 256^3 (+=)", best * 1e3);
    }
    return 0;
}
