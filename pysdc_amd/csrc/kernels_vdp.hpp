// libsdcmi kernels: van der Pol ensemble.
#pragma once
#include "context.hpp"

// ------------------------------------------------------------------------------------------------------
// van der Pol ensemble: one trajectory per lane (SoA state [2][T])
// ------------------------------------------------------------------------------------------------------
struct VdpSweepArgs {
    double* U;   // node index base: U + m N = U[m] ([2][T] each), m = 1..M
    const double* u0;   // u[0] (a block of its own)
    double* F;
    const double* tau;  // or null
    size_t T;
    double mu, dt, tol;
    int maxiter;
    double Q[MAXM][MAXM], QI[MAXM][MAXM];  // dt * Qmat, dt * QI (inner MxM blocks): the reference forms dt * Q[m][j]
                                           // first and multiplies by f afterwards, so the product is made once on the host
    unsigned long long* counters;
    unsigned long long* norms;  // node-wise max of the collocation residual after the sweep, or null
    int spread;  // the old iterate is the spread start value (U[1..M] = U[0] was never written out): only U[0] is read
};

// Newton for u - h f(u) = r with the closed-form 2x2 inverse (Van_der_Pol_implicit.py:131-201)
__device__ __forceinline__ bool vdp_newton(double& x1, double& x2, double r0, double r1, double h, double mu, double tol,
                                           int maxiter, unsigned long long& newton) {
#pragma clang fp contract(off)
    int it = 0;
    double res = 99.0;
    while (it < maxiter) {
        const double e0 = x1 - h * x2 - r0;
        const double e1 = x2 - h * (mu * (1 - x1 * x1) * x2 - x1) - r1;
        res = fmax(fabs(e0), fabs(e1));
        if (e0 != e0 || e1 != e1) res = e0 + e1;  // NaN
        if (res < tol || res != res) break;
        const double c = 1.0 / (-2 * h * h * mu * x1 * x2 - h * h - 1 + h * mu * (1 - x1 * x1));
        const double d00 = c * (h * mu * (1 - x1 * x1) - 1), d01 = c * (-h);
        const double d10 = c * (2 * h * mu * x1 * x2 + h), d11 = c * (-1.0);
        const double nx1 = x1 - (d00 * e0 + d01 * e1);
        const double nx2 = x2 - (d10 * e0 + d11 * e1);
        x1 = nx1;
        x2 = nx2;
        ++it;
        ++newton;
    }
    return !(res != res || it == maxiter);
}

// Work counters of a launch: summed over the workgroup first, then ONE atomic per counter and workgroup (and none for a
// zero) - 16 384 waves adding to the same three words one after the other were a measurable part of a 0.2 ms launch.
// Every thread of the workgroup must call this (it contains a barrier).
__device__ __forceinline__ void vdp_add_counters(unsigned long long* counters, unsigned long long newton, unsigned long long rhs,
                                                 unsigned long long failed) {
    __shared__ unsigned long long part[3][16];
    for (int o = 32; o > 0; o >>= 1) {
        newton += __shfl_xor(newton, o, 64);
        rhs += __shfl_xor(rhs, o, 64);
        failed += __shfl_xor(failed, o, 64);
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        part[0][w] = newton;
        part[1][w] = rhs;
        part[2][w] = failed;
    }
    __syncthreads();
    if (threadIdx.x < 3 && counters) {
        unsigned long long t = 0;
        for (int k = 0; k < nw; ++k) t += part[threadIdx.x][k];
        if (t) atomicAdd(counters + threadIdx.x, t);
    }
}

// one generic_implicit sweep (generic_implicit.py:51-103) for every trajectory, node values on the slabs
// LAZYF: the right-hand sides of the old iterate are recomputed from its node values (read anyway as Newton
// guesses; f costs five flops, the same formula gives the same bits) and the new ones are not stored - the F slab
// is brought up to date on demand (sdc_materialize).  Halves the bytes of the sweep.
#ifndef SDC_VDP_LDS_COEFFS
#define SDC_VDP_LDS_COEFFS 0  // 1: quadrature coefficients from LDS - the compiler then holds all 50 in VGPRs (249): worse
#endif
#ifndef SDC_VDP_KEEP_NODES
#define SDC_VDP_KEEP_NODES 1  // 0: node values re-read instead of held (150 instead of 160 VGPRs; measured 0.757 vs 0.736 ms)
#endif
#ifndef SDC_VDP_WAVES
#define SDC_VDP_WAVES 0  // > 0: waves per SIMD the sweep kernel is compiled for (register budget 512 / waves) whatever M
#endif
// Default: 4 waves per SIMD up to M = 5 (128 registers, a few spilled: 0.480 -> 0.468 ms per sweep of 1e7 trajectories once the
// counter atomics were out of the way - with them in it, round 3 had measured this setting as a loss), the compiler's choice
// above (160+ registers at M = 5 already).
#define VDP_BOUNDS __launch_bounds__(256, SDC_VDP_WAVES > 0 ? SDC_VDP_WAVES : (M <= 5 ? 4 : 1))
template <int M, bool LAZYF>
__global__ VDP_BOUNDS void k_vdp_sweep(VdpSweepArgs a) {
#pragma clang fp contract(off)
    unsigned long long newton = 0, rhs = 0, failed = 0;
    const size_t T = a.T, N = 2 * a.T;
#if SDC_VDP_LDS_COEFFS
    // the 2 M^2 quadrature coefficients from LDS (same-address reads: broadcasts) instead of ~100 scalar registers that
    // would be spilled into vector registers
    __shared__ double sQ[M][M], sQI[M][M];
    if (threadIdx.x < M * M) {
        sQ[threadIdx.x / M][threadIdx.x % M] = a.Q[threadIdx.x / M][threadIdx.x % M];
        sQI[threadIdx.x / M][threadIdx.x % M] = a.QI[threadIdx.x / M][threadIdx.x % M];
    }
    __syncthreads();
#define VQ(m, j) sQ[m][j]
#define VQI(m, j) sQI[m][j]
#else
#define VQ(m, j) a.Q[m][j]
#define VQI(m, j) a.QI[m][j]
#endif
    double nmax[M];
#pragma unroll
    for (int m = 0; m < M; ++m) nmax[m] = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < T; i += (size_t)gridDim.x * blockDim.x) {
        const double mu = a.mu;
        const double u00 = a.u0[i], u01 = a.u0[T + i];
        // Live state per trajectory: the right-hand sides of all nodes (old ones, replaced node by node by the new ones)
        // and the gathered sums - 4 M doubles.  Node values are NOT kept: the old one of node m is read again when its
        // Newton iteration starts (as the guess), the new ones are read back for the residual (this thread wrote them:
        // L2 hits) - 2 M doubles fewer in registers than holding them.  Measured round 3: no faster (0.757 against 0.736 ms
        // per sweep of 1e7 trajectories), and compiled for 4 waves per SIMD it still spills (0.95 ms): occupancy is not
        // what this kernel waits for.  SDC_VDP_KEEP_NODES=1 (default) holds them.
        double f0[M], f1[M], g0[M], g1[M];
#if SDC_VDP_KEEP_NODES
        double un0[M], un1[M];
#endif
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (LAZYF) {
                const double o0 = a.spread ? u00 : a.U[(size_t)(m + 1) * N + i];  // old node value: f here, guess below
                const double o1 = a.spread ? u01 : a.U[(size_t)(m + 1) * N + T + i];
#if SDC_VDP_KEEP_NODES
                un0[m] = o0;
                un1[m] = o1;
#endif
                f0[m] = o1;
                f1[m] = mu * (1 - o0 * o0) * o1 - o0;
            } else {
                f0[m] = a.F[(size_t)(m + 1) * N + i];
                f1[m] = a.F[(size_t)(m + 1) * N + T + i];
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                s0 += VQ(m, j) * f0[j];
                s1 += VQ(m, j) * f1[j];
            }
#pragma unroll
            for (int j = 0; j < M; ++j) {
                s0 -= VQI(m, j) * f0[j];
                s1 -= VQI(m, j) * f1[j];
            }
            g0[m] = s0 + u00;
            g1[m] = s1 + u01;
            if (a.tau) {
                g0[m] += a.tau[(size_t)m * N + i];
                g1[m] += a.tau[(size_t)m * N + T + i];
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double r0 = g0[m], r1 = g1[m];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                if (j < m) {
                    r0 += VQI(m, j) * f0[j];
                    r1 += VQI(m, j) * f1[j];
                }
            }
            const double h = VQI(m, m);
#if SDC_VDP_KEEP_NODES
            double x1 = LAZYF ? un0[m] : a.U[(size_t)(m + 1) * N + i];
            double x2 = LAZYF ? un1[m] : a.U[(size_t)(m + 1) * N + T + i];
#else
            double x1 = a.U[(size_t)(m + 1) * N + i];
            double x2 = a.U[(size_t)(m + 1) * N + T + i];
#endif
            if (h == 0.0) {
                x1 = r0;
                x2 = r1;
            } else if (!vdp_newton(x1, x2, r0, r1, h, mu, a.tol, a.maxiter, newton)) {
                failed += 1;
            }
            a.U[(size_t)(m + 1) * N + i] = x1;
            a.U[(size_t)(m + 1) * N + T + i] = x2;
#if SDC_VDP_KEEP_NODES
            un0[m] = x1;
            un1[m] = x2;
#endif
            f0[m] = x2;
            f1[m] = mu * (1 - x1 * x1) * x2 - x1;
            if (!LAZYF) {
                a.F[(size_t)(m + 1) * N + i] = f0[m];
                a.F[(size_t)(m + 1) * N + T + i] = f1[m];
            }
            rhs += 1;
        }
        if (a.norms) {
            // collocation residual of the new iterate (core/sweeper.py:186-199): the right-hand sides are in registers,
            // the node values are read back
#pragma unroll
            for (int m = 0; m < M; ++m) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    s0 += VQ(m, j) * f0[j];
                    s1 += VQ(m, j) * f1[j];
                }
#if SDC_VDP_KEEP_NODES
                s0 += u00 - un0[m];
                s1 += u01 - un1[m];
#else
                s0 += u00 - a.U[(size_t)(m + 1) * N + i];
                s1 += u01 - a.U[(size_t)(m + 1) * N + T + i];
#endif
                if (a.tau) {
                    s0 += a.tau[(size_t)m * N + i];
                    s1 += a.tau[(size_t)m * N + T + i];
                }
                const double r0 = fabs(s0), r1 = fabs(s1);
                const double r = (r0 > r1 || r0 != r0) ? r0 : r1;
                nmax[m] = (nmax[m] > r || nmax[m] != nmax[m]) ? nmax[m] : r;
            }
        }
    }
    vdp_add_counters(a.counters, newton, rhs, failed);
    if (a.norms) block_max_to_slots<M>(a.norms, nmax);
#undef VQ
#undef VQI
}

// ------------------------------------------------------------------------------------------------------
// MFMA variant of the Newton step (BASELINE.json north_star: "MFMA used only for the small dense (I - dt Qdiag J)
// block solves").  The 2x2 inverse is still the reference's closed form (Van_der_Pol_implicit.py:190-201); its
// application  delta = (dg/du)^{-1} g  runs on the matrix cores: v_mfma_f64_4x4x4_4b_f64 multiplies four independent
// 4x4 blocks per instruction, each block carrying TWO trajectories as diag(D_a, D_b) times the column (g_a, g_b).
// Operand layout on gfx950 (scripts/probes/mfma_f64_4x4x4_layout.hip): with k = lane / 16, blk = (lane % 16) / 4,
//   A[blk][i][k] sits in lane 16 k + 4 blk + i,  B[blk][k][j] in lane 16 k + 4 blk + j,  D[blk][i][j] in lane 16 i + 4 blk + j.
// A wave holds 64 trajectories (one per lane), so one Newton step of the wave is 8 instructions of 8 trajectories,
// with the operands redistributed through LDS (6 values out, 2 back per trajectory).  Only one column of B and the
// two diagonal 2x2 blocks of A carry data: 1/8 of the instruction's multiply-adds are useful.  The loop is made
// wave-uniform (lanes that have converged keep their values and contribute zeros).
// MEASURED (profiles/r02): slower than the closed-form VALU kernel - see DESIGN.md; selected by sdc_set_vdp_block_solver.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wl: this wave's LDS area, 64 x 8 doubles in (d00 d01 d10 d11 e0 e1 . .), then 64 x 2 doubles out
__device__ __forceinline__ bool vdp_newton_mfma(double& x1, double& x2, double r0, double r1, double h, double mu, double tol,
                                                int maxiter, unsigned long long& newton, double* wl, bool valid) {
#pragma clang fp contract(off)
    const int lane = threadIdx.x & 63;
    const int k = lane >> 4, blk = (lane & 15) >> 2, ij = lane & 3;
    double* out = wl + 64 * 8;
    int it = 0;
    double res = 99.0;
    bool running = valid;
    while (__builtin_amdgcn_ballot_w64(running) != 0) {
        double d00 = 0.0, d01 = 0.0, d10 = 0.0, d11 = 0.0, e0 = 0.0, e1 = 0.0;
        if (running && !(it < maxiter)) running = false;
        if (running) {
            e0 = x1 - h * x2 - r0;
            e1 = x2 - h * (mu * (1 - x1 * x1) * x2 - x1) - r1;
            res = fmax(fabs(e0), fabs(e1));
            if (e0 != e0 || e1 != e1) res = e0 + e1;  // NaN
            if (res < tol || res != res) running = false;
        }
        if (running) {
            const double c = 1.0 / (-2 * h * h * mu * x1 * x2 - h * h - 1 + h * mu * (1 - x1 * x1));
            d00 = c * (h * mu * (1 - x1 * x1) - 1);
            d01 = c * (-h);
            d10 = c * (2 * h * mu * x1 * x2 + h);
            d11 = c * (-1.0);
        } else {
            e0 = e1 = 0.0;
        }
        double* mine = wl + lane * 8;
        mine[0] = d00;
        mine[1] = d01;
        mine[2] = d10;
        mine[3] = d11;
        mine[4] = e0;
        mine[5] = e1;
        wave_lds_sync();
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            // A[blk][i][k]: rows 0,1 <- trajectory a = 8n + 2 blk, rows 2,3 <- trajectory a + 1 (block diagonal)
            const int ha = ij >> 1;
            const double av = (k >> 1) == ha ? wl[(8 * n + 2 * blk + ha) * 8 + 2 * (ij & 1) + (k & 1)] : 0.0;
            // B[blk][k][j]: column 0 = (g_a, g_b)
            const double bv = ij == 0 ? wl[(8 * n + 2 * blk + (k >> 1)) * 8 + 4 + (k & 1)] : 0.0;
            const double dv = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0.0, 0, 0, 0);
            // D[blk][i][0] (i = lane / 16): component i & 1 of trajectory 8n + 2 blk + i / 2
            if (ij == 0) out[(8 * n + 2 * blk + (k >> 1)) * 2 + (k & 1)] = dv;
        }
        wave_lds_sync();
        if (running) {
            const double nx1 = x1 - out[lane * 2], nx2 = x2 - out[lane * 2 + 1];
            x1 = nx1;
            x2 = nx2;
            ++it;
            ++newton;
        }
        wave_lds_sync();
    }
    return !valid || !(res != res || it == maxiter);
}

template <int M>
__global__ __launch_bounds__(256) void k_vdp_sweep_mfma(VdpSweepArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) double vdp_lds[];
    double* wl = vdp_lds + (threadIdx.x >> 6) * (64 * 10);
    unsigned long long newton = 0, rhs = 0, failed = 0;
    const size_t T = a.T, N = 2 * a.T;
    double nmax[M];
#pragma unroll
    for (int m = 0; m < M; ++m) nmax[m] = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t rounds = (T + stride - 1) / stride;  // every lane takes part in every round (wave-uniform MFMA)
    for (size_t rd = 0; rd < rounds; ++rd) {
        const size_t i = rd * stride + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
        const bool valid = i < T;
        const size_t ii = valid ? i : 0;
        const double mu = a.mu;
        const double u00 = a.u0[ii], u01 = a.u0[T + ii];
        double f0[M], f1[M], g0[M], g1[M], un0[M], un1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            un0[m] = a.U[(size_t)(m + 1) * N + ii];
            un1[m] = a.U[(size_t)(m + 1) * N + T + ii];
            f0[m] = un1[m];
            f1[m] = mu * (1 - un0[m] * un0[m]) * un1[m] - un0[m];
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                s0 += a.Q[m][j] * f0[j];
                s1 += a.Q[m][j] * f1[j];
            }
#pragma unroll
            for (int j = 0; j < M; ++j) {
                s0 -= a.QI[m][j] * f0[j];
                s1 -= a.QI[m][j] * f1[j];
            }
            g0[m] = s0 + u00;
            g1[m] = s1 + u01;
            if (a.tau) {
                g0[m] += a.tau[(size_t)m * N + ii];
                g1[m] += a.tau[(size_t)m * N + T + ii];
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double r0 = g0[m], r1 = g1[m];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                if (j < m) {
                    r0 += a.QI[m][j] * f0[j];
                    r1 += a.QI[m][j] * f1[j];
                }
            }
            const double h = a.QI[m][m];
            double x1 = un0[m], x2 = un1[m];
            if (h == 0.0) {
                x1 = r0;
                x2 = r1;
            } else if (!vdp_newton_mfma(x1, x2, r0, r1, h, mu, a.tol, a.maxiter, newton, wl, valid)) {
                failed += 1;
            }
            if (valid) {
                a.U[(size_t)(m + 1) * N + i] = x1;
                a.U[(size_t)(m + 1) * N + T + i] = x2;
                rhs += 1;
            }
            un0[m] = x1;
            un1[m] = x2;
            f0[m] = x2;
            f1[m] = mu * (1 - x1 * x1) * x2 - x1;
        }
        if (a.norms && valid) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    s0 += a.Q[m][j] * f0[j];
                    s1 += a.Q[m][j] * f1[j];
                }
                s0 += u00 - un0[m];
                s1 += u01 - un1[m];
                if (a.tau) {
                    s0 += a.tau[(size_t)m * N + i];
                    s1 += a.tau[(size_t)m * N + T + i];
                }
                const double r0 = fabs(s0), r1 = fabs(s1);
                const double r = (r0 > r1 || r0 != r0) ? r0 : r1;
                nmax[m] = (nmax[m] > r || nmax[m] != nmax[m]) ? nmax[m] : r;
            }
        }
    }
    vdp_add_counters(a.counters, newton, rhs, failed);
    if (a.norms) block_max_to_slots<M>(a.norms, nmax);
}

// f(u) of the ensemble (Van_der_Pol_implicit.py:76-98); fmax (or null): max |f| over both components on the way (the
// residual of a spread predictor).  Two trajectories per thread and iteration when T is even (16-byte accesses: this
// launch moves little per thread and waits for memory latency).
__global__ __launch_bounds__(256) void k_vdp_eval(const double* __restrict__ u, double* __restrict__ f, size_t T, double mu,
                                                  unsigned long long* counters, unsigned long long* fmax = nullptr) {
#pragma clang fp contract(off)
    unsigned long long rhs = 0;
    double mx = 0.0;
    auto upd = [&](double v) {
        v = fabs(v);
        mx = (mx > v || mx != mx) ? mx : v;
    };
    if ((T & 1) == 0 && ((reinterpret_cast<size_t>(u) | reinterpret_cast<size_t>(f)) & 15) == 0) {
        const size_t H = T / 2;
        const double2* __restrict__ u1 = reinterpret_cast<const double2*>(u);
        const double2* __restrict__ u2 = reinterpret_cast<const double2*>(u + T);
        double2* __restrict__ f1 = reinterpret_cast<double2*>(f);
        double2* __restrict__ f2 = reinterpret_cast<double2*>(f + T);
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < H; i += (size_t)gridDim.x * blockDim.x) {
            const double2 x1 = u1[i], x2 = u2[i];
            double2 g;
            g.x = mu * (1 - x1.x * x1.x) * x2.x - x1.x;
            g.y = mu * (1 - x1.y * x1.y) * x2.y - x1.y;
            f1[i] = x2;
            f2[i] = g;
            if (fmax) {
                upd(x2.x);
                upd(x2.y);
                upd(g.x);
                upd(g.y);
            }
            rhs += 2;
        }
    } else {
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < T; i += (size_t)gridDim.x * blockDim.x) {
            const double x1 = u[i], x2 = u[T + i];
            const double g = mu * (1 - x1 * x1) * x2 - x1;
            f[i] = x2;
            f[T + i] = g;
            if (fmax) {
                upd(x2);
                upd(g);
            }
            rhs += 1;
        }
    }
    if (fmax) {
        mx = wave_max(mx);
        if ((threadIdx.x & 63) == 0) atomic_max_abs(fmax, mx);
    }
    // (counters == null: storing deferred values - those evaluations were counted by the sweep)
    vdp_add_counters(counters, 0, rhs, 0);
}

__global__ void k_vdp_solve(const double* __restrict__ rhsv, const double* __restrict__ guess, double* __restrict__ out,
                            size_t T, double h, double mu, double tol, int maxiter, unsigned long long* counters) {
    unsigned long long newton = 0, failed = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < T; i += (size_t)gridDim.x * blockDim.x) {
        double x1 = guess[i], x2 = guess[T + i];
        if (!vdp_newton(x1, x2, rhsv[i], rhsv[T + i], h, mu, tol, maxiter, newton)) failed += 1;
        out[i] = x1;
        out[T + i] = x2;
    }
    vdp_add_counters(counters, newton, 0, failed);
}

// (dg/du)^{-1} rhs for g(u) = u - dt f(u): the closed-form 2x2 inverse of Van_der_Pol_implicit.py:190-201, one
// trajectory per lane
__global__ void k_vdp_jac_solve(const double* __restrict__ rhs, const double* __restrict__ u, double* __restrict__ out,
                                size_t T, double dt, double mu) {
#pragma clang fp contract(off)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < T; i += (size_t)gridDim.x * blockDim.x) {
        const double u1 = u[i], u2 = u[T + i], r0 = rhs[i], r1 = rhs[T + i];
        const double c = 1.0 / (-2 * dt * dt * mu * u1 * u2 - dt * dt - 1 + dt * mu * (1 - u1 * u1));
        const double d00 = c * (dt * mu * (1 - u1 * u1) - 1), d01 = c * (-dt);
        const double d10 = c * (2 * dt * mu * u1 * u2 + dt), d11 = c * (-1.0);
        out[i] = d00 * r0 + d01 * r1;
        out[T + i] = d10 * r0 + d11 * r1;
    }
}

