// libsdcmi kernels: quadrature gathers / residual norms / predictor fill / vector operations.
#pragma once
#include "context.hpp"

// ------------------------------------------------------------------------------------------------------
// elementwise kernels
// ------------------------------------------------------------------------------------------------------
struct QuadArgs {
    const double* u0;    // may be null
    const double* F;     // F slab base; field (j, comp) at F + (j*ncomp + comp)*N
    const double* tau;   // TAU base or null
    const double* Usub;  // U slab base for the residual (subtract U[mo+1]) or null
    double* out[MAXM];   // MODE 0 outputs
    double cI[MAXM][MAXM];  // [mo][j-1]
    double cE[MAXM][MAXM];
    unsigned long long* norms;  // MODE 1: per-node max |.| as ordered bit patterns
    size_t N;
    int nout;  // number of output rows (M, or 1 for the end point)
    int tau_row0;  // tau row used for output 0 (end point uses the last row)
};

__device__ inline void atomic_max_abs(unsigned long long* slot, double v) {
    // |v| >= 0: IEEE order == unsigned order of the bit pattern; NaN (0x7ff8...) wins, like np.max
    // The slot only grows: a relaxed read first keeps the millions of waves that cannot raise it away from
    // the read-modify-write queue of that one address.
    const unsigned long long bits = (unsigned long long)__double_as_longlong(fabs(v));
    if (bits > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bits);
}

__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double w = __shfl_xor(v, o, 64);
        v = (v > w || v != v) ? v : w;  // propagate NaN
    }
    return v;
}

// NV per-thread maxima -> slots[0..NV): reduced over the wave, then over the workgroup through LDS, then ONE guarded
// atomic per value and workgroup.  Used by the van der Pol sweep (0.507 -> 0.487 ms per sweep of 1e7 trajectories); the
// norm-only transform pass and the marching stencil keep one guarded atomic per wave (measured: the barrier costs the x
// pass 3.5 %, the stencil nothing either way).  Every thread of the workgroup must call it (barrier inside);
// SDC_BLOCK_NORMS 0: one per wave.
#ifndef SDC_BLOCK_NORMS
#define SDC_BLOCK_NORMS 1
#endif
template <int NV>
__device__ inline void block_max_to_slots(unsigned long long* slots, const double (&v)[NV], int nvalid = NV) {
#if SDC_BLOCK_NORMS
    __shared__ double red[NV][16];
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double m = wave_max(v[k]);
        if ((threadIdx.x & 63) == 0) red[k][w] = m;
    }
    __syncthreads();
    if ((int)threadIdx.x < NV && (int)threadIdx.x < nvalid) {
        double m = red[threadIdx.x][0];
        for (int q = 1; q < nw; ++q) {
            const double o = red[threadIdx.x][q];
            m = (m > o || m != m) ? m : o;
        }
        atomic_max_abs(slots + threadIdx.x, m);
    }
#else
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const double m = wave_max(v[k]);
        if ((threadIdx.x & 63) == 0 && k < nvalid) atomic_max_abs(slots + k, m);
    }
#endif
}

// out[mo] = u0 + sum_j cI[mo][j] F_impl[j] (+ cE[mo][j] F_expl[j]) (+ tau[mo]) (- U[mo+1], max-norm)
// MODE 0: the sums are stored; 1: node norms of the residual only; 2: the norms AND the bare quadrature sums (out[mo]): the
// residual of a fine level and the integrals its restriction asks for next (core/base_transfer.py:120-127) in one pass over F
#ifndef SDC_QUAD_NT
#define SDC_QUAD_NT 3   // bit 0: nontemporal loads of F, bit 1: nontemporal stores of the sums (round 6, 256^3 x 3 nodes: residual +
                        // integrals 351 -> 310 us, gather 265 -> 245, integrate 226 -> 197; 1024^3: 16.4 -> 16.2 ms)
#endif
__device__ __forceinline__ double2 quad_ld(const double2* p) {
#if SDC_QUAD_NT & 1
    return double2{__builtin_nontemporal_load(&p->x), __builtin_nontemporal_load(&p->y)};
#else
    return *p;
#endif
}
__device__ __forceinline__ void quad_st(double2* p, double2 v) {
#if SDC_QUAD_NT & 2
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
#else
    *p = v;
#endif
}
template <int M, int NCOMP, int MODE>
__global__ __launch_bounds__(256) void k_quad(QuadArgs a) {
    const size_t n2 = a.N >> 1;
    double nmax[M];
#pragma unroll
    for (int m = 0; m < M; ++m) nmax[m] = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 fi[M], fe[M];
#pragma unroll
        for (int j = 0; j < M; ++j) {
            fi[j] = quad_ld(reinterpret_cast<const double2*>(a.F + ((size_t)(j + 1) * NCOMP) * a.N) + i);
            if (NCOMP == 2) fe[j] = quad_ld(reinterpret_cast<const double2*>(a.F + ((size_t)(j + 1) * NCOMP + 1) * a.N) + i);
        }
        double2 u0 = a.u0 ? reinterpret_cast<const double2*>(a.u0)[i] : double2{0.0, 0.0};
#pragma unroll
        for (int mo = 0; mo < M; ++mo) {
            if (mo < a.nout) {
                double2 acc = double2{0.0, 0.0};
#pragma unroll
                for (int j = 0; j < M; ++j) {
                    if (NCOMP == 2) {
                        // same grouping as imex_1st_order.py:52: Q * (impl + expl) when both weights agree
                        acc.x += a.cI[mo][j] * fi[j].x + a.cE[mo][j] * fe[j].x;
                        acc.y += a.cI[mo][j] * fi[j].y + a.cE[mo][j] * fe[j].y;
                    } else {
                        acc.x += a.cI[mo][j] * fi[j].x;
                        acc.y += a.cI[mo][j] * fi[j].y;
                    }
                }
                if (MODE == 2) quad_st(reinterpret_cast<double2*>(a.out[mo]) + i, acc);   // the quadrature alone: what integrate() returns
                acc.x += u0.x;
                acc.y += u0.y;
                if (a.tau) {
                    double2 t = reinterpret_cast<const double2*>(a.tau + (size_t)(a.tau_row0 + mo) * a.N)[i];
                    acc.x += t.x;
                    acc.y += t.y;
                }
                if (MODE == 0) {
                    quad_st(reinterpret_cast<double2*>(a.out[mo]) + i, acc);
                } else {
                    double2 us = reinterpret_cast<const double2*>(a.Usub + (size_t)(mo + 1) * a.N)[i];
                    double r0 = fabs(acc.x - us.x), r1 = fabs(acc.y - us.y);
                    double r = (r0 > r1 || r0 != r0) ? r0 : r1;
                    nmax[mo] = (nmax[mo] > r || nmax[mo] != nmax[mo]) ? nmax[mo] : r;
                }
            }
        }
    }
    if (MODE >= 1) {
        // one guarded atomic per value and WORKGROUP: on a small field all waves of the launch are resident at once, every one
        // of them finds the slot at zero and queues its atomic on the same address (128^3, M = 3: 16 384 waves x 3 atomics made a
        // 35 us pass take 200 us - config 5's coarse level)
        block_max_to_slots<M>(a.norms, nmax, a.nout);
    }
}

struct LinArgs {
    double* out;
    const double* base;  // may alias out
    const double* x[2 * MAXM];
    double c[2 * MAXM];
    int nterms;
    size_t n;
};

// out = base + sum_k c[k] * x[k]   (right-hand side of one node: generic_implicit.py:87-89 / imex_1st_order.py:92-94)
__global__ __launch_bounds__(256) void k_lincomb(LinArgs a) {
    const size_t n2 = a.n >> 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 acc = reinterpret_cast<const double2*>(a.base)[i];
        for (int k = 0; k < a.nterms; ++k) {
            const double2 v = reinterpret_cast<const double2*>(a.x[k])[i];
            acc.x += a.c[k] * v.x;
            acc.y += a.c[k] * v.y;
        }
        reinterpret_cast<double2*>(a.out)[i] = acc;
    }
}

// odd (Dirichlet-zero) extension of a 1-D field stored as [0, u_0..u_{n-1}, 0, -u_{n-1}..-u_0] (length 2(n+1)):
// rebuild the zero end points and the mirrored half from the interior
__global__ void k_odd_mirror(double* __restrict__ f, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[2 * n + 1 - i] = -f[1 + i];
    if (i == 0) {
        f[0] = 0.0;
        f[n + 1] = 0.0;
    }
}

// odd extension in every axis of an n^ndim interior field into a (2(n+1))^ndim array (dirichlet-zero in 2-D / 3-D: the
// Dirichlet operator on the interior is the periodic one on the extension), and the way back.  One thread per element of
// the extension: coordinate e of an axis maps to interior index e-1 (1 <= e <= n), to 2n+1-e with a sign flip
// (n+2 <= e <= 2n+1), or to the boundary value zero (e = 0, n+1).
__global__ void k_odd_extend_nd(const double* __restrict__ interior, double* __restrict__ ext, int n, int ndim) {
    const int m = 2 * (n + 1);
    size_t total = 1;
    for (int d = 0; d < ndim; ++d) total *= (size_t)m;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        size_t rem = g, src = 0, stride = 1;
        double sign = 1.0;
        bool zero = false;
        for (int d = 0; d < ndim; ++d) {  // fastest axis first
            const int e = (int)(rem % m);
            rem /= m;
            int i;
            if (e == 0 || e == n + 1) { zero = true; i = 0; }
            else if (e <= n) i = e - 1;
            else { i = 2 * n + 1 - e; sign = -sign; }
            src += (size_t)i * stride;
            stride *= (size_t)n;
        }
        ext[g] = zero ? 0.0 : sign * interior[src];
    }
}
__global__ void k_odd_extract_nd(const double* __restrict__ ext, double* __restrict__ interior, int n, int ndim) {
    const int m = 2 * (n + 1);
    size_t total = 1;
    for (int d = 0; d < ndim; ++d) total *= (size_t)n;
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        size_t rem = g, src = 0, stride = 1;
        for (int d = 0; d < ndim; ++d) {
            const int i = (int)(rem % n);
            rem /= n;
            src += (size_t)(i + 1) * stride;
            stride *= (size_t)m;
        }
        interior[g] = ext[src];
    }
}

__global__ void k_amax(const double* __restrict__ x, size_t n, unsigned long long* slot) {
    double m = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = fabs(x[i]);
        m = (m > v || m != m) ? m : v;
    }
    const double mm[1] = {m};
    block_max_to_slots<1>(slot, mm);   // (one guarded atomic per workgroup: small fields have all their waves in flight at once)
}

// max |x + y| (y may be null): max |f_impl(u0) + f_expl(u0)| for the residual of a deferred spread predictor
__global__ void k_amax_sum(const double* __restrict__ x, const double* __restrict__ y, size_t n,
                           unsigned long long* slot) {
    double m = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = fabs(y ? x[i] + y[i] : x[i]);
        m = (m > v || m != m) ? m : v;
    }
    const double mm[1] = {m};
    block_max_to_slots<1>(slot, mm);
}

__global__ void k_axpby(size_t n, double a, const double* __restrict__ x, double b, const double* __restrict__ y,
                        double* __restrict__ z) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (x) v = a * x[i];
        if (y) v += b * y[i];
        z[i] = v;
    }
}

// A box of a field (basic indexing of the reference's ndarray datatype, datatype_classes/mesh.py:12-60: integers and slices
// per axis): element (i0, i1, i2, i3) of the box lies at field[off + i0 s0 + i1 s1 + i2 s2 + i3 s3] (strides in elements, may be
// negative) and at compact[((i0 c1 + i1) c2 + i2) c3 + i3].  dir 0: compact <- box, 1: box <- compact, 2: box <- value.
struct BoxArgs {
    long long off, s[4];
    long long c[4];
    double* field;
    double* compact;
    double value;
    int dir;
};
__global__ void k_box(BoxArgs a) {
    const size_t total = (size_t)(a.c[0] * a.c[1] * a.c[2] * a.c[3]);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const long long i3 = (long long)(r % (size_t)a.c[3]);
        r /= (size_t)a.c[3];
        const long long i2 = (long long)(r % (size_t)a.c[2]);
        r /= (size_t)a.c[2];
        const long long i1 = (long long)(r % (size_t)a.c[1]);
        const long long i0 = (long long)(r / (size_t)a.c[1]);
        double* at = a.field + (a.off + i0 * a.s[0] + i1 * a.s[1] + i2 * a.s[2] + i3 * a.s[3]);
        if (a.dir == 0) a.compact[i] = *at;
        else if (a.dir == 1) *at = a.compact[i];
        else *at = a.value;
    }
}

__global__ void k_fill(size_t n, double a, double* __restrict__ y) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = a;
}

struct SpreadArgs {
    const double* u0;       // U[0]
    const double* f0;       // F[0] base (ncomp fields)
    const double* profile;  // forcing profile or null
    double* U;              // slab
    double* F;              // slab
    double g[MAXM + 1];     // forcing scalars at t and the node times
    size_t N;
    int M, ncomp, guess, forcing;
    double fill_u, fill_f;
    unsigned long long* f0max;  // max |F[0]| (implicit + explicit) for the residual of the spread state, or null
};

// predictor fill of the node values; core/sweeper.py:140-158
__global__ __launch_bounds__(256) void k_spread(SpreadArgs a) {
    const size_t n2 = a.N >> 1;
    double fmaxv = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        double2 u = reinterpret_cast<const double2*>(a.u0)[i];
        double2 fi = reinterpret_cast<const double2*>(a.f0)[i];
        double2 fe = double2{0.0, 0.0}, pr = double2{0.0, 0.0};
        if (a.ncomp == 2) fe = reinterpret_cast<const double2*>(a.f0 + a.N)[i];
        if (a.f0max) {
            const double s0 = fabs(fi.x + fe.x), s1 = fabs(fi.y + fe.y);
            const double sm = (s0 > s1 || s0 != s0) ? s0 : s1;
            fmaxv = (fmaxv > sm || fmaxv != fmaxv) ? fmaxv : sm;
        }
        if (a.forcing) pr = reinterpret_cast<const double2*>(a.profile)[i];
        for (int m = 1; m <= a.M; ++m) {
            double2 um = u, fim = fi, fem = fe;
            if (a.guess == SDC_GUESS_SPREAD) {
                if (a.forcing) fem = double2{pr.x * a.g[m], pr.y * a.g[m]};
            } else if (a.guess == SDC_GUESS_ZERO) {
                um = fim = fem = double2{0.0, 0.0};
            } else if (a.guess == SDC_GUESS_CONST) {
                um = double2{a.fill_u, a.fill_u};
                fim = fem = double2{a.fill_f, a.fill_f};
            }
            reinterpret_cast<double2*>(a.U + (size_t)m * a.N)[i] = um;
            reinterpret_cast<double2*>(a.F + ((size_t)m * a.ncomp) * a.N)[i] = fim;
            if (a.ncomp == 2) reinterpret_cast<double2*>(a.F + ((size_t)m * a.ncomp + 1) * a.N)[i] = fem;
        }
    }
    if (a.f0max) {
        fmaxv = wave_max(fmaxv);
        if ((threadIdx.x & 63) == 0) atomic_max_abs(a.f0max, fmaxv);
    }
}

// u[0] is replaced while the residual FIELDS r_m = u0 + dt sum_j Q[m][j] f_j - u_m of the current iterate are
// at hand (kept by the sweep in the U[1..M] slab): r_m changes by the same difference for every node, so the
// node norms against the new u[0] follow in one pass:  d = new - old, u0 <- new, norms[m] = max |r_m + d|.
template <int M>
__global__ __launch_bounds__(256) void k_replace_u0(const double* __restrict__ src, double* __restrict__ U0,
                                                    const double* __restrict__ U, size_t N,
                                                    unsigned long long* __restrict__ norms) {
    double nm[M];
#pragma unroll
    for (int m = 0; m < M; ++m) nm[m] = 0.0;
    const size_t n2 = N >> 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        const double2 nw = reinterpret_cast<const double2*>(src)[i];
        const double2 od = reinterpret_cast<const double2*>(U0)[i];
        const double d0 = nw.x - od.x, d1 = nw.y - od.y;
        reinterpret_cast<double2*>(U0)[i] = nw;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const double2 r = reinterpret_cast<const double2*>(U + (size_t)(m + 1) * N)[i];
            const double v0 = fabs(r.x + d0), v1 = fabs(r.y + d1);
            const double v = (v0 > v1 || v0 != v0) ? v0 : v1;
            nm[m] = (nm[m] > v || nm[m] != nm[m]) ? nm[m] : v;
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double v = wave_max(nm[m]);
        if ((threadIdx.x & 63) == 0) atomic_max_abs(norms + m, v);
    }
}

// ------------------------------------------------------------------------------------------------------
// conjugate gradients for (I - factor A) x = b, the reference's solver_type='CG' (generic_ND_FD.py:252-260 ->
// scipy.sparse.linalg.cg: no preconditioner, stop when ||r|| < rtol ||b||).  Fused vector updates with
// DETERMINISTIC dot products: every workgroup writes one partial sum, a single workgroup adds them in a fixed
// order - iteration counts are observable (work_counters['CG']) and must not depend on scheduling.
//   mode 0: o0 = a0 - a1 + s a2,            sum o0^2      (r = b - x + factor A x)
//   mode 1: o0 = a0 + s o0                                (p = r + beta p)
//   mode 2: o0 = a0 - s a1,                 sum a0 o0     (q = p - factor A p,  p.q)
//   mode 3: o0 += s a0,  o1 -= s a1,        sum o1^2      (x += alpha p, r -= alpha q,  r.r)
//   mode 4:                                 sum a0^2
// ------------------------------------------------------------------------------------------------------
struct CgArgs {
    const double *a0, *a1, *a2;
    double *o0, *o1;
    double s;
    size_t n;
    double* part;
    int mode;
    // iterations enqueued without a host round trip (cg_solve): the scalar comes from device memory (sdev[0], written by
    // k_cg_scalars) and the launch does nothing once the iteration has stopped (*done != 0)
    const double* sdev;
    const double* done;
};

// scalars of the conjugate-gradient iteration kept on the device (cg_solve): slots of `scal`
enum { CGS_RHO = 1, CGS_ALPHA = 3, CGS_BETA = 4, CGS_ATOL = 5, CGS_DONE = 6, CGS_ITERS = 7, CGS_MAXITER = 8 };

__global__ __launch_bounds__(256) void k_cg(CgArgs a) {
#pragma clang fp contract(off)
    if (a.done && *a.done != 0.0) return;
    if (a.sdev) a.s = *a.sdev;
    double acc = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < a.n; i += (size_t)gridDim.x * blockDim.x) {
        if (a.mode == 0) {
            const double v = (a.a0[i] - a.a1[i]) + a.s * a.a2[i];
            a.o0[i] = v;
            acc += v * v;
        } else if (a.mode == 1) {
            a.o0[i] = a.a0[i] + a.s * a.o0[i];
        } else if (a.mode == 2) {
            const double p = a.a0[i];
            const double v = p - a.s * a.a1[i];
            a.o0[i] = v;
            acc += p * v;
        } else if (a.mode == 3) {
            a.o0[i] += a.s * a.a0[i];
            const double v = a.o1[i] - a.s * a.a1[i];
            a.o1[i] = v;
            acc += v * v;
        } else if (a.mode == 4) {
            const double v = a.a0[i];
            acc += v * v;
        } else if (a.mode == 6) {          // GMRES: w = v - s * (A v), |w|^2
            const double v = a.a0[i] - a.s * a.a1[i];
            a.o0[i] = v;
            acc += v * v;
        } else if (a.mode == 7) {          // <a0, a1>
            acc += a.a0[i] * a.a1[i];
        } else if (a.mode == 8) {          // modified Gram-Schmidt: w -= s * v_k, then <next, w>  (next may be w itself)
            const double v = a.o0[i] - a.s * a.a0[i];
            a.o0[i] = v;
            acc += (a.a1 == a.o0 ? v : a.a1[i]) * v;
        } else if (a.mode == 9) {          // v_new = s * w
            a.o0[i] = a.s * a.a0[i];
        } else if (a.mode == 10) {         // t = s * v (a1 null) or t += s * v
            a.o0[i] = a.a1 ? a.o0[i] + a.s * a.a0[i] : a.s * a.a0[i];
        } else {                           // 11: x += t
            a.o0[i] += a.a0[i];
        }
    }
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && a.part) a.part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// sum of the partial sums of the launch before (same order as k_sum_parts), then the scalar step of the iteration
// (scipy.sparse.linalg.cg as the reference calls it, generic_ND_FD.py:252-260; the arithmetic the host loop of round 2 did):
//   step 0 (after p.Ap):  alpha = rho / pq
//   step 1 (after r.r):   iterations += 1;  stop if sqrt(rr) < atol or the iteration limit is reached, else
//                         beta = rr / rho,  rho = rr
__global__ __launch_bounds__(256) void k_cg_scalars(const double* __restrict__ part, int nb, double* __restrict__ scal, int step) {
#pragma clang fp contract(off)
    if (scal[CGS_DONE] != 0.0) return;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += part[i];
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    if (step == 0) {
        scal[CGS_ALPHA] = scal[CGS_RHO] / v;
    } else {
        const double its = scal[CGS_ITERS] + 1.0;
        scal[CGS_ITERS] = its;
        if (sqrt(v) < scal[CGS_ATOL]) scal[CGS_DONE] = 1.0;
        else if (its >= scal[CGS_MAXITER]) scal[CGS_DONE] = 2.0;
        else scal[CGS_BETA] = v / scal[CGS_RHO];
        scal[CGS_RHO] = v;
    }
}

__global__ __launch_bounds__(256) void k_sum_parts(const double* __restrict__ part, int nb, double* __restrict__ out) {
#pragma clang fp contract(off)
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += part[i];
    __shared__ double sh[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

