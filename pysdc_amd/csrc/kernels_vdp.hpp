// libsdcmi kernels: van der Pol ensemble.
#pragma once
#include "context.hpp"

// ------------------------------------------------------------------------------------------------------
// van der Pol ensemble: one trajectory per lane (SoA state [2][T])
// ------------------------------------------------------------------------------------------------------
struct VdpSweepArgs {
    double* U;   // slab [(M+1)][2][T]
    double* F;
    const double* tau;  // or null
    size_t T;
    double mu, dt, tol;
    int maxiter;
    double Q[MAXM][MAXM], QI[MAXM][MAXM];  // dt * Qmat, dt * QI (inner MxM blocks): the reference forms dt * Q[m][j]
                                           // first and multiplies by f afterwards, so the product is made once on the host
    unsigned long long* counters;
    unsigned long long* norms;  // node-wise max of the collocation residual after the sweep, or null
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

// one generic_implicit sweep (generic_implicit.py:51-103) for every trajectory, node values on the slabs
// LAZYF: the right-hand sides of the old iterate are recomputed from its node values (read anyway as Newton
// guesses; f costs five flops, the same formula gives the same bits) and the new ones are not stored - the F slab
// is brought up to date on demand (sdc_materialize).  Halves the bytes of the sweep.
template <int M, bool LAZYF>
__global__ __launch_bounds__(256) void k_vdp_sweep(VdpSweepArgs a) {
#pragma clang fp contract(off)
    unsigned long long newton = 0, rhs = 0, failed = 0;
    const size_t T = a.T, N = 2 * a.T;
    double nmax[M];
#pragma unroll
    for (int m = 0; m < M; ++m) nmax[m] = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < T; i += (size_t)gridDim.x * blockDim.x) {
        const double mu = a.mu;
        const double u00 = a.U[i], u01 = a.U[T + i];
        double f0[M], f1[M], g0[M], g1[M], un0[M], un1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if (LAZYF) {
                un0[m] = a.U[(size_t)(m + 1) * N + i];  // old node value: guess below, f here
                un1[m] = a.U[(size_t)(m + 1) * N + T + i];
                f0[m] = un1[m];
                f1[m] = mu * (1 - un0[m] * un0[m]) * un1[m] - un0[m];
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
                    r0 += a.QI[m][j] * f0[j];
                    r1 += a.QI[m][j] * f1[j];
                }
            }
            const double h = a.QI[m][m];
            double x1 = LAZYF ? un0[m] : a.U[(size_t)(m + 1) * N + i];
            double x2 = LAZYF ? un1[m] : a.U[(size_t)(m + 1) * N + T + i];
            if (h == 0.0) {
                x1 = r0;
                x2 = r1;
            } else if (!vdp_newton(x1, x2, r0, r1, h, mu, a.tol, a.maxiter, newton)) {
                failed += 1;
            }
            a.U[(size_t)(m + 1) * N + i] = x1;
            a.U[(size_t)(m + 1) * N + T + i] = x2;
            un0[m] = x1;
            un1[m] = x2;
            f0[m] = x2;
            f1[m] = mu * (1 - x1 * x1) * x2 - x1;
            if (!LAZYF) {
                a.F[(size_t)(m + 1) * N + i] = f0[m];
                a.F[(size_t)(m + 1) * N + T + i] = f1[m];
            }
            rhs += 1;
        }
        if (a.norms) {
            // collocation residual of the new iterate (core/sweeper.py:186-199), all values still in registers
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
    for (int o = 32; o > 0; o >>= 1) {
        newton += __shfl_xor(newton, o, 64);
        rhs += __shfl_xor(rhs, o, 64);
        failed += __shfl_xor(failed, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(a.counters + 0, newton);
        atomicAdd(a.counters + 1, rhs);
        atomicAdd(a.counters + 2, failed);
    }
    if (a.norms) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const double v = wave_max(nmax[m]);
            if ((threadIdx.x & 63) == 0) atomic_max_abs(a.norms + m, v);
        }
    }
}

__global__ void k_vdp_eval(const double* __restrict__ u, double* __restrict__ f, size_t T, double mu,
                           unsigned long long* counters) {
#pragma clang fp contract(off)
    unsigned long long rhs = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < T; i += (size_t)gridDim.x * blockDim.x) {
        const double x1 = u[i], x2 = u[T + i];
        f[i] = x2;
        f[T + i] = mu * (1 - x1 * x1) * x2 - x1;
        rhs += 1;
    }
    if (!counters) return;  // storing deferred values: those evaluations were counted by the sweep
    for (int o = 32; o > 0; o >>= 1) rhs += __shfl_xor(rhs, o, 64);
    if ((threadIdx.x & 63) == 0 && rhs) atomicAdd(counters + 1, rhs);
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
    for (int o = 32; o > 0; o >>= 1) {
        newton += __shfl_xor(newton, o, 64);
        failed += __shfl_xor(failed, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counters + 0, newton);
        atomicAdd(counters + 2, failed);
    }
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

