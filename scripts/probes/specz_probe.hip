// Round 6 experiment ("port k_trail_z's persistent / next-line-prefetch structure into k_spec_z MODE 4").
// Result on MI355X (profiles/r06/specz_ablation.txt): the persistent launch is bit-identical and SLOWER (1.94 - 2.00 ms against
// 1.66 ms for 65536 lines), with two or three workgroups per CU alike; ablations (-DSDC_ABL=bits: 1 no line transform, 2 no
// multiplier arithmetic, 8 no start-value loads) show why: loads + stores + the LDS hand-over alone (ABL 3) take 1.28 ms =
// 5.9 TB/s for one spectrum read and six written - the rate a plain fill reaches on this part (5.3 - 5.6 TB/s,
// scripts/hbm_peak.py).  The z launch of the serial sweep (51.6 GB: 11.4 - 11.7 ms) is within 25 % of that floor (8.9 ms).
// k_spec_z<1024,5,4,0> (one line per workgroup and launch) against k_spec_z_pers<1024,5> (persistent, next line prefetched):
// same lines bit for bit, time of both.  A slab of `nlines` z lines (default 65536 = 1 GB of start-value spectrum, 5.4 GB written).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude [-D...] -o build_variants/specz_probe scripts/probes/specz_probe.hip
#include "../../pysdc_amd/csrc/kernels_pointwise.hpp"
#include "../../pysdc_amd/csrc/kernels_fft.hpp"

// ---- the experiment (round 6): NOT part of the library - measured slower than the launch it was to replace, see below ----
// MODE 4 of k_spec_z (iterate recomputed from S0 through real node multipliers, mode pairs, residual lines handed to the
// line transforms) as a PERSISTENT launch: a few workgroups per CU (as many as its LDS holds: three at 1024 x 5) walk over the
// lines, and the start-value modes of a workgroup's NEXT line are on their way while it makes the multipliers of this one,
// hands the residual lines over and transforms them; the stores of a line drain under the next one's arithmetic.  With one
// line per workgroup and launch neither the vector ALUs (0.48 busy) nor the memory pipes (4.5 of ~7 TB/s) were kept busy: every
// workgroup began with a full memory latency in which its five waves held their registers and their LDS for nothing
// (profiles/r05/sq_counters_n1024.json).  Same arithmetic in the same order: the lines it writes are bit-identical.
// All loads are unconditional and of a fixed number (lanes without an item fetch a valid mode and ignore it; past the last line
// the last line is fetched again), so that every s_waitcnt the compiler places waits for exactly what is needed.
template <int N, int NF>
__global__ __launch_bounds__((specz_threads<N, NF, true>()), (specz_min_waves<N, true>()))
void k_spec_z_pers(SpecArgs a, unsigned nlines) {
    constexpr int E = specz_elems<N, true>(), P = N / E, NT = P * NF;
    static_assert(specz_lines<N, true>() == 1, "one line per field and workgroup");
    constexpr int CH = N > SDC_SPECZ_CH ? SDC_SPECZ_CH : N, NCH = N / CH;
    constexpr int NI = N / 2 + 1, IT2 = (NI + NT - 1) / NT;
    using LAY = LayContig<N>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    cd* rbuf = reinterpret_cast<cd*>(lds);  // [NF][CH]; the exchange planes of the transforms alias it
    // the twiddle factors the line transform looks up, in LDS: a table look-up in global memory inside the transform would
    // have to wait for the start-value loads of the NEXT line issued before it (loads return in order)
    constexpr int PLANES = LAY::doubles(NF) > 2 * NF * CH ? LAY::doubles(NF) : 2 * NF * CH;
    cd* twl = reinterpret_cast<cd*>(lds + ((PLANES + 1) & ~1));
    for (int i = threadIdx.x; i < N / 2; i += NT) twl[i] = a.tw[i];
    // ... and the sweep's coefficients (65 doubles at five nodes: more than the scalar registers hold beside everything else -
    // as kernel arguments they are fetched from the argument segment again inside the loop over the replayed sweeps, a wait
    // of the whole wave each time), read from LDS with one address for the whole wave
    double* cf = reinterpret_cast<double*>(twl + N / 2);   // gI[NF][NF], cI[NF][NF], rQ[NF][NF], alpha[NF]
    if (threadIdx.x < NF * NF) {
        cf[threadIdx.x] = a.gI[threadIdx.x / NF][threadIdx.x % NF];
        cf[NF * NF + threadIdx.x] = a.cI[threadIdx.x / NF][threadIdx.x % NF];
        cf[2 * NF * NF + threadIdx.x] = a.rQ[threadIdx.x / NF][threadIdx.x % NF];
    }
    if (threadIdx.x < NF) cf[3 * NF * NF + threadIdx.x] = a.alpha[threadIdx.x];
    __syncthreads();
    const bool coupled = a.coupled != 0;
    const int nsw = a.replay + 1;
    const int f = threadIdx.x / P, j = threadIdx.x % P;
    // the items of this thread: mode pairs (p, N - p), p = threadIdx.x + it * NT; what does not depend on the line
    int pl[IT2], ph[IT2];
    bool has[IT2], pair[IT2];
    double lz[IT2];
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
        const int p_ = threadIdx.x + it * NT;
        has[it] = p_ < NI;
        pair[it] = p_ >= 1 && p_ < N / 2;
        pl[it] = has[it] ? p_ : NI - 1;
        ph[it] = pair[it] ? N - p_ : pl[it];
        lz[it] = a.lamI[pl[it]].x;
    }
    const unsigned last = nlines - 1, stride = gridDim.x;
    const bool d3 = a.ndim == 3, d2 = a.ndim == 2;
    auto lxy_of = [&](unsigned ln) {
        double v = 0.0;
        if (d3) v = a.lamI[ln / N].x + a.lamI[ln % N].x;
        else if (d2) v = a.lamI[ln].x;
        return v;
    };
    unsigned line = blockIdx.x;
    cd lo[IT2], hi[IT2];
    double lxy;
    {
        const unsigned lc = line < last ? line : last;
        const size_t base = (size_t)lc * N;
        lxy = lxy_of(lc);
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            lo[it] = a.S0[base + pl[it]];
            hi[it] = a.S0[base + ph[it]];
        }
    }
    while (line < nlines) {
        const size_t base = (size_t)line * N;
        const unsigned nl = line + stride, nc = nl < last ? nl : last;
        double hm[IT2][NF];
#pragma unroll
        for (int it = 0; it < IT2; ++it) {
            const double lam = lz[it] + lxy;
            // (virt_multipliers_real / virt_step_real with the coefficients from LDS: same operations, same order)
            double g[NF], inv[NF];
#pragma unroll
            for (int m = 0; m < NF; ++m) {
                inv[m] = fast_rcp(1.0 - cf[3 * NF * NF + m] * lam);
                g[m] = 1.0;
            }
#ifndef SDC_ABL
#define SDC_ABL 0
#endif
            for (int sw = 0; sw < ((SDC_ABL & 2) ? 0 : nsw); ++sw) {
                double o[NF];
#pragma unroll
                for (int q = 0; q < NF; ++q) o[q] = g[q];
#pragma unroll
                for (int m = 0; m < NF; ++m) {
                    double t = 0.0;
#pragma unroll
                    for (int q = 0; q < NF; ++q) t = fma(cf[m * NF + q], o[q], t);
                    if (coupled) {
#pragma unroll
                        for (int q = 0; q < m; ++q) t = fma(cf[NF * NF + m * NF + q], g[q], t);
                    }
                    g[m] = fma(lam, t, 1.0) * inv[m];
                }
            }
            if (a.store_last && has[it]) {
                a.SL[base + pl[it]] = cd{g[NF - 1] * lo[it].x, g[NF - 1] * lo[it].y};
                if (pair[it]) a.SL[base + ph[it]] = cd{g[NF - 1] * hi[it].x, g[NF - 1] * hi[it].y};
            }
            if (a.virt == 2) {  // the iterate itself is wanted in real space (node values stored by every sweep)
#pragma unroll
                for (int m = 0; m < NF; ++m) hm[it][m] = g[m] * a.invN;
            } else {
#pragma unroll
                for (int m = 0; m < NF; ++m) {
                    double t = 0.0;
#pragma unroll
                    for (int q = 0; q < NF; ++q) t = fma(cf[2 * NF * NF + m * NF + q], g[q], t);
                    hm[it][m] = fma(lam, t, 1.0 - g[m]) * a.invN;
                }
            }
        }
        cd r[E];
        __syncthreads();   // the transforms of the previous line are done with their exchange planes
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
            for (int it = 0; it < IT2; ++it) {
                if (has[it]) {
                    if (pl[it] / CH == ch) {
#pragma unroll
                        for (int m = 0; m < NF; ++m) rbuf[m * CH + (pl[it] % CH)] = cscale(lo[it], hm[it][m]);
                    }
                    if (pair[it] && ph[it] / CH == ch) {
#pragma unroll
                        for (int m = 0; m < NF; ++m) rbuf[m * CH + (ph[it] % CH)] = cscale(hi[it], hm[it][m]);
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int o = j + i * P;
                if (o / CH == ch) r[i] = rbuf[f * CH + (o % CH)];
            }
            __syncthreads();
        }
        {   // the next line's start-value modes, into the registers this line's are done with: in flight while the line is
            // transformed and stored
            const size_t nb = (size_t)nc * N;
            lxy = lxy_of(nc);
#pragma unroll
            for (int it = 0; it < IT2; ++it) {
                if (!(SDC_ABL & 8)) {
                    lo[it] = a.S0[nb + pl[it]];
                    hi[it] = a.S0[nb + ph[it]];
                }
            }
        }
        if (!(SDC_ABL & 1)) fft_line<N, +1, LAY, P <= 64, E>(r, j, f, lds, twl);
        cd* __restrict__ dst = a.W + f * a.fstride + ((SDC_ABL & 4) ? (size_t)blockIdx.x * N : base);
#pragma unroll
        for (int i = 0; i < E; ++i) {
#if SDC_SPECZ_NT & 2
            __builtin_nontemporal_store(r[i].x, &dst[j + i * P].x);
            __builtin_nontemporal_store(r[i].y, &dst[j + i * P].y);
#else
            dst[j + i * P] = r[i];
#endif
        }
        line = nl;
    }
}


#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#ifndef PROBE_WGS_PER_CU
#define PROBE_WGS_PER_CU 3
#endif
int main(int argc, char** argv) {
    constexpr int N = 1024, NF = 5;
    const unsigned nlines = argc > 1 ? (unsigned)atoi(argv[1]) : 65536u;
    const int replay = argc > 2 ? atoi(argv[2]) : 3;
    const size_t Nc = (size_t)nlines * N;
    cd *S0, *W0, *W1, *SL0, *SL1, *tw, *lam;
    CK(hipMalloc(&S0, Nc * 16)); CK(hipMalloc(&W0, NF * Nc * 16)); CK(hipMalloc(&W1, NF * Nc * 16));
    CK(hipMalloc(&SL0, Nc * 16)); CK(hipMalloc(&SL1, Nc * 16));
    std::vector<cd> htw(N), hl(N);
    for (int m = 0; m < N; ++m) {
        const long double ang = 2.0L * 3.14159265358979323846264338327950288L * m / N;
        htw[m] = cd{(double)cosl(ang), (double)(-sinl(ang))};
        hl[m] = cd{0.1 * (2.0 * cos((double)ang) - 2.0) * N * N, 0.0};   // heat symbol nu (2 cos - 2) / dx^2
    }
    CK(hipMalloc(&tw, N * 16)); CK(hipMalloc(&lam, N * 16));
    CK(hipMemcpy(tw, htw.data(), N * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(lam, hl.data(), N * 16, hipMemcpyHostToDevice));
    {   // pseudo-random start-value spectrum
        std::vector<cd> h((size_t)1 << 20);
        unsigned long long s = 88172645463325252ull;
        for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v.x = (double)(s >> 11) / 9007199254740992.0 - 0.5; s ^= s << 13; s ^= s >> 7; s ^= s << 17; v.y = (double)(s >> 11) / 9007199254740992.0 - 0.5; }
        for (size_t o = 0; o < Nc; o += h.size()) CK(hipMemcpy(S0 + o, h.data(), std::min(h.size(), Nc - o) * 16, hipMemcpyHostToDevice));
    }
    SpecArgs a; memset(&a, 0, sizeof a);
    a.S0 = S0; a.fstride = Nc; a.tw = tw; a.lamI = lam; a.nf = NF; a.ndim = 3; a.coupled = 1; a.real_sym = 1; a.replay = replay;
    a.invN = 1.0 / ((double)N * N * N); a.store_last = 1;
    const double dt = 2.5e-4;
    for (int m = 0; m < NF; ++m) {
        a.alpha[m] = dt * (0.05 + 0.17 * m);
        for (int q = 0; q < NF; ++q) {
            a.gI[m][q] = dt * 0.07 * ((m + 2 * q) % 5 - 2);
            a.cI[m][q] = q < m ? dt * 0.11 * (1 + (m + q) % 3) : 0.0;
            a.rQ[m][q] = dt * 0.09 * (1 + (2 * m + q) % 4);
        }
    }
    const size_t ldsz = (size_t)LayContig<N>::doubles(NF) * sizeof(double);
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 6; ++rep)
        for (int which = 0; which < 2; ++which) {
            a.W = which ? W1 : W0; a.SL = which ? SL1 : SL0;
            CK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL((k_spec_z<N, NF, 4, 0>), dim3(nlines), dim3(320), ldsz, 0, a, nlines);
            else hipLaunchKernelGGL((k_spec_z_pers<N, NF>), dim3(std::min<unsigned>(nlines, cus * PROBE_WGS_PER_CU)), dim3(320), ldsz + 16 + (N / 2) * 16 + (3 * NF * NF + NF + 1) * 8, 0, a, nlines);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best[which]) best[which] = ms;
        }
    CK(hipGetLastError());
    // bit-identical?
    std::vector<cd> x((size_t)1 << 22), y(x.size());
    size_t bad = 0;
    for (size_t o = 0; o < NF * Nc; o += x.size() * 37) {
        const size_t len = std::min(x.size(), NF * Nc - o);
        CK(hipMemcpy(x.data(), W0 + o, len * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), W1 + o, len * 16, hipMemcpyDeviceToHost));
        bad += memcmp(x.data(), y.data(), len * 16) != 0;
    }
    CK(hipMemcpy(x.data(), SL0, std::min(x.size(), Nc) * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), SL1, std::min(x.size(), Nc) * 16, hipMemcpyDeviceToHost));
    bad += memcmp(x.data(), y.data(), std::min(x.size(), Nc) * 16) != 0;
    const double gb = (double)(1 + NF + 1) * Nc * 16 / 1e9;
    printf("lines %u replay %d: per-line launch %.3f ms (%.0f GB/s), persistent x%d %.3f ms (%.0f GB/s), scaled to 1024^3: %.2f -> %.2f ms; mismatching blocks %zu\n",
           nlines, replay, best[0], gb / best[0] * 1e3, PROBE_WGS_PER_CU, best[1], gb / best[1] * 1e3, best[0] * 525312.0 / nlines, best[1] * 525312.0 / nlines, bad);
    return bad != 0;
}
