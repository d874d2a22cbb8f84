// libsdcmi: SDC sweep engine for MI355X (gfx950).  Kernels + the C-ABI declared in include/sdcmi.h.
//
// Data layout (all f64, spatial index fastest):
//   U[(M+1)][N]   F[(M+1)][ncomp][N]   TAU[M][N]   UEND[N]   W[M][Nc] (complex half spectrum, work)
// Sweep pipeline for the periodic FD problems (DESIGN.md "kernels"):
//   gather (Q-weighted sums, all nodes, one pass)            -> R[m] stored in U[1+m]
//   r2c FFT along axis 0 (strided tiles, LDS)                -> W
//   c2c FFT along axis 1 (3-D only, in place)
//   c2c FFT along the contiguous axis + node-coupled solve + inverse, all M nodes of one line per workgroup
//   inverse axis 1, inverse axis 0 (c2r)                     -> U[1..M]
//   stencil A*U[m] (+ explicit stencil)                      -> F[1..M]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/sdcmi.h"
#include "fft.hpp"

#define MAXM 8
#define MAXSTEN 12

// ------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------
struct Stencil {
    int npts = 0;
    int off[MAXSTEN];
    double w[MAXSTEN];
};

struct ProfEntry {
    double ms = 0;
    int calls = 0;
};

struct sdc_ctx {
    int device = 0, ndim = 0, n = 0, M = 0, ncomp = 1;
    size_t N = 0;       // n^ndim
    size_t Nc = 0;      // complex entries of one spectrum field
    hipStream_t stream = nullptr;
    double *U = nullptr, *F = nullptr, *TAU = nullptr, *UEND = nullptr, *profile = nullptr;
    cd* W = nullptr;
    cd *S = nullptr, *S0 = nullptr;  // spectral cache: transforms of U[1..M] and of U[0] (lazy)
    bool spec_valid = false, spec0_valid = false, spec_spread = false, reuse = true, force_gather = false;
    cd *tw = nullptr, *lamI = nullptr, *lamE = nullptr;
    unsigned long long* red = nullptr;  // reduction slots (device)
    unsigned long long* red_host = nullptr;
    bool tau_active = false, have_coeffs = false, have_stencil[2] = {false, false}, unlocked = false;
    int expl_kind = SDC_EXPL_NONE;
    bool res_spread = false;  // state = spread predictor of an autonomous f: residual_m = dt |sum_j Q[m][j]| max|f(u0)|
    bool fuse_residual = true;
    bool res_valid = false;   // node norms of the residual were produced by the fused stencil kernel
    double res_dt = 0.0;
    unsigned long long* res_dev = nullptr;  // device slots of those norms
    bool spectral_op = false;  // implicit operator given by its Fourier symbol only (no stencil): eval_f by FFT
    int react_kind = 0, react_nu = 2;
    double react_p0 = 0, react_p1 = 0;
    int kind = 0;  // 0: periodic finite differences, 1: van der Pol ensemble (N = 2 * ntraj, SoA)
    double vdp_mu = 0, vdp_tol = 1e-9;
    int vdp_maxiter = 100;
    unsigned long long* counters = nullptr;  // device: [0] newton, [1] rhs, [2] failed solves
    unsigned long long rhs_host = 0;         // evaluations the reference would have made where the engine copies
    double Q[MAXM + 1][MAXM + 1], QI[MAXM + 1][MAXM + 1], QE[MAXM + 1][MAXM + 1], nodes[MAXM], weights[MAXM];
    double gvals[MAXM + 1];
    Stencil st[2];
    size_t bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, pev0 = nullptr, pev1 = nullptr;
    bool profiling = false;
    std::vector<hipEvent_t> pool;            // event pairs recorded around launches while profiling
    std::vector<const char*> pool_names;
    size_t pool_used = 0;                    // pairs in flight
    std::map<std::string, ProfEntry> prof;
    std::vector<std::string> prof_names;
    std::string err;
};

static thread_local std::string g_create_err;

static int fail(sdc_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_err = buf;
    return code;
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(c, e_ == hipErrorOutOfMemory ? SDC_ERR_NOMEM : SDC_ERR_HIP, "%s: %s", #call, \
                        hipGetErrorString(e_));                                                      \
    } while (0)

// Per-kernel device time: a pair of events from a pool is recorded around every launch on the context's
// stream; nothing synchronises until the pool is full or the profile is read, so the timed region of bench.py
// is not perturbed.
static void prof_flush(sdc_ctx* c) {
    if (c->pool_used == 0) return;
    (void)hipEventSynchronize(c->pool[2 * c->pool_used - 1]);
    for (size_t i = 0; i < c->pool_used; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->pool[2 * i], c->pool[2 * i + 1]) == hipSuccess) {
            ProfEntry& e = c->prof[c->pool_names[i]];
            e.ms += ms;
            e.calls += 1;
        }
    }
    c->pool_used = 0;
}

struct LaunchTimer {
    sdc_ctx* c;
    size_t slot = 0;
    bool on;
    LaunchTimer(sdc_ctx* c_, const char* n) : c(c_), on(c_->profiling) {
        if (!on) return;
        constexpr size_t kPairs = 2048;
        if (c->pool.empty()) {
            c->pool.resize(2 * kPairs);
            c->pool_names.resize(kPairs);
            for (auto& e : c->pool) (void)hipEventCreate(&e);
        }
        if (c->pool_used == kPairs) prof_flush(c);
        slot = c->pool_used++;
        c->pool_names[slot] = n;
        (void)hipEventRecord(c->pool[2 * slot], c->stream);
    }
    ~LaunchTimer() {
        if (on) (void)hipEventRecord(c->pool[2 * slot + 1], c->stream);
    }
};

static inline int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// profile names that carry the number of fields of the launch, e.g. "fft_x_fwd[5]" (interned, static lifetime)
static const char* pname(const char* base, int nf) {
    static std::map<std::string, std::string> table;
    std::string key = std::string(base) + "[" + std::to_string(nf) + "]";
    auto it = table.find(key);
    if (it == table.end()) it = table.emplace(key, key).first;
    return it->second.c_str();
}

// 1/(1 - alpha*lambda): |denominator|^2 is finite and away from zero for the dissipative / skew operators
// handled here, so the reciprocal is v_rcp_f64 refined by two Newton steps (~1 ulp) instead of the IEEE
// division sequence (v_div_scale / v_div_fmas / v_div_fixup).
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ cd cinv_fast(cd d) {
    const double m = fast_rcp(d.x * d.x + d.y * d.y);
    return cd{d.x * m, -d.y * m};
}

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
    atomicMax(slot, (unsigned long long)__double_as_longlong(fabs(v)));
}

__device__ inline double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double w = __shfl_xor(v, o, 64);
        v = (v > w || v != v) ? v : w;  // propagate NaN
    }
    return v;
}

// out[mo] = u0 + sum_j cI[mo][j] F_impl[j] (+ cE[mo][j] F_expl[j]) (+ tau[mo]) (- U[mo+1], max-norm)
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
            fi[j] = reinterpret_cast<const double2*>(a.F + ((size_t)(j + 1) * NCOMP) * a.N)[i];
            if (NCOMP == 2) fe[j] = reinterpret_cast<const double2*>(a.F + ((size_t)(j + 1) * NCOMP + 1) * a.N)[i];
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
                acc.x += u0.x;
                acc.y += u0.y;
                if (a.tau) {
                    double2 t = reinterpret_cast<const double2*>(a.tau + (size_t)(a.tau_row0 + mo) * a.N)[i];
                    acc.x += t.x;
                    acc.y += t.y;
                }
                if (MODE == 0) {
                    reinterpret_cast<double2*>(a.out[mo])[i] = acc;
                } else {
                    double2 us = reinterpret_cast<const double2*>(a.Usub + (size_t)(mo + 1) * a.N)[i];
                    double r0 = fabs(acc.x - us.x), r1 = fabs(acc.y - us.y);
                    double r = (r0 > r1 || r0 != r0) ? r0 : r1;
                    nmax[mo] = (nmax[mo] > r || nmax[mo] != nmax[mo]) ? nmax[mo] : r;
                }
            }
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int mo = 0; mo < M; ++mo) {
            double v = wave_max(nmax[mo]);
            if ((threadIdx.x & 63) == 0 && mo < a.nout) atomic_max_abs(a.norms + mo, v);
        }
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

__global__ void k_amax(const double* __restrict__ x, size_t n, unsigned long long* slot) {
    double m = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double v = fabs(x[i]);
        m = (m > v || m != m) ? m : v;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomic_max_abs(slot, m);
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

// ------------------------------------------------------------------------------------------------------
// periodic finite-difference operator (eval_f), any stencil width, 1-3 dimensions
// ------------------------------------------------------------------------------------------------------
struct StencilArgs {
    const double* in[MAXM];
    double* outI[MAXM];  // implicit-operator result or null
    double* outE[MAXM];  // explicit-stencil result or null
    const double* profile;  // forcing profile (outE = profile * g[f]) or null
    double g[MAXM];
    Stencil sI, sE;
    int nf, ndim, n;
    int useE;  // 0 none, 1 stencil, 2 forcing
};

__device__ inline int wrapi(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

__global__ __launch_bounds__(256) void k_stencil(StencilArgs a) {
    const int n = a.n;
    const size_t N = a.ndim == 1 ? (size_t)n : (a.ndim == 2 ? (size_t)n * n : (size_t)n * n * n);
    const size_t n2 = N >> 1;
    const int f = blockIdx.y;
    const double* __restrict__ u = a.in[f];
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n2; p += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = p * 2;
        const int z = (int)(i0 % n);
        const size_t rest = i0 / n;
        const int y = a.ndim >= 2 ? (int)(rest % n) : 0;
        const int x = a.ndim == 3 ? (int)(rest / n) : 0;
        const size_t line = i0 - z;  // start of the contiguous line
        for (int which = 0; which < 2; ++which) {
            double* out = which == 0 ? a.outI[f] : a.outE[f];
            if (!out) continue;
            if (which == 1 && a.useE == 2) {
                double2 pr = reinterpret_cast<const double2*>(a.profile)[p];
                reinterpret_cast<double2*>(out)[p] = double2{pr.x * a.g[f], pr.y * a.g[f]};
                continue;
            }
            const Stencil& s = which == 0 ? a.sI : a.sE;
            double r0 = 0.0, r1 = 0.0;
            // axis order follows the Kronecker sum of problem_helper.py:226-235: slowest axis first
            if (a.ndim == 3) {
                for (int k = 0; k < s.npts; ++k) {
                    const size_t q = ((size_t)wrapi(x + s.off[k], n) * n + y) * n + z;
                    double2 v = *reinterpret_cast<const double2*>(u + q);
                    r0 += s.w[k] * v.x;
                    r1 += s.w[k] * v.y;
                }
            }
            if (a.ndim >= 2) {
                for (int k = 0; k < s.npts; ++k) {
                    const size_t q = ((size_t)x * n + wrapi(y + s.off[k], n)) * n + z;
                    double2 v = *reinterpret_cast<const double2*>(u + q);
                    r0 += s.w[k] * v.x;
                    r1 += s.w[k] * v.y;
                }
            }
            for (int k = 0; k < s.npts; ++k) {
                r0 += s.w[k] * u[line + wrapi(z + s.off[k], n)];
                r1 += s.w[k] * u[line + wrapi(z + 1 + s.off[k], n)];
            }
            reinterpret_cast<double2*>(out)[p] = double2{r0, r1};
        }
    }
}

// 3-D fast path for 3-point stencils (offsets -1, 0, 1 per axis): 2.5-D blocking.  A workgroup owns a
// (TY x TZ) tile of the y-z plane and marches along x; the x neighbours stay in registers, the y/z neighbours
// of the current plane come from a double-buffered LDS tile with halo, so every input word is read from
// global memory once per tile (+ halo) instead of seven times.
struct Stencil3Args {
    const double* in[MAXM];
    double* outI[MAXM];
    double* outE[MAXM];
    double wI[3], wE[3];  // weights for offsets -1, 0, +1
    int n, xchunk, ntiles, nchunks;
};

// Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only): give every XCD a contiguous
// range of the logical grid so that tiles sharing halo lines meet in the same L2.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned total) {
    return (total & 7u) ? b : (b & 7u) * (total >> 3) + (b >> 3);
}

template <int RPT>
__global__ __launch_bounds__(256, 4) void k_stencil3d(Stencil3Args a) {
    constexpr int TZ = 64, TYB = 8, TY = TYB * RPT, LW = TZ + 4;  // LDS row: [halo | 64 | halo | pad]
    __shared__ double tile[2][TY + 2][LW];
    const int n = a.n;
    const int tz = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ntz = n / TZ;
    // logical order: z-tile fastest, then y-tile (halo partners stay close), then x-chunk, then field
    unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);
    const int zt = lb % ntz;
    lb /= ntz;
    const int nty = n / TY;
    const int yt = lb % nty;
    lb /= nty;
    const int chunk = lb % a.nchunks;
    const int f = lb / a.nchunks;
    const int z0 = zt * TZ, y0 = yt * TY;
    const int x0 = chunk * a.xchunk;
    const double* __restrict__ u = a.in[f];
    double* __restrict__ oI = a.outI[f];
    double* __restrict__ oE = a.outE[f];
    const size_t sx = (size_t)n * n;
    // halo duty of this thread: 0..63 -> y halo rows (below / above), 64..64+2*TY-1 -> z halo columns
    const int t = threadIdx.x;
    const bool hy = t < 64, hz = t >= 64 && t < 64 + 2 * TY;
    size_t hoff = 0;   // offset of the halo element(s) within a plane
    int hrow = 0, hcol = 0;
    if (hy) {
        const int side = t >> 5, pz = t & 31;
        const int yy = side == 0 ? (y0 == 0 ? n - 1 : y0 - 1) : (y0 + TY == n ? 0 : y0 + TY);
        hoff = (size_t)yy * n + z0 + 2 * pz;
        hrow = side == 0 ? 0 : TY + 1;
        hcol = 1 + 2 * pz;
    } else if (hz) {
        const int q = t - 64, side = q / TY, r = q % TY;
        const int zz = side == 0 ? (z0 == 0 ? n - 1 : z0 - 1) : (z0 + TZ == n ? 0 : z0 + TZ);
        hoff = (size_t)(y0 + r) * n + zz;
        hrow = r + 1;
        hcol = side == 0 ? 0 : TZ + 1;
    }
    size_t off[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) off[r] = (size_t)(y0 + ty + r * TYB) * n + z0 + 2 * tz;

    auto plane = [&](int x) { return u + (size_t)(x < 0 ? x + n : (x >= n ? x - n : x)) * sx; };
    double2 prev[RPT], cur[RPT], nxt[RPT], nx2[RPT];
    double2 hcur = double2{0.0, 0.0}, hnxt = double2{0.0, 0.0}, hnx2 = double2{0.0, 0.0};
    {
        const double* pm = plane(x0 - 1);
        const double* p0 = plane(x0);
        const double* p1 = plane(x0 + 1);
        const double* p2 = plane(x0 + 2);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            prev[r] = *reinterpret_cast<const double2*>(pm + off[r]);
            cur[r] = *reinterpret_cast<const double2*>(p0 + off[r]);
            nxt[r] = *reinterpret_cast<const double2*>(p1 + off[r]);
            nx2[r] = *reinterpret_cast<const double2*>(p2 + off[r]);
        }
        if (hy) {
            hcur = *reinterpret_cast<const double2*>(p0 + hoff);
            hnxt = *reinterpret_cast<const double2*>(p1 + hoff);
            hnx2 = *reinterpret_cast<const double2*>(p2 + hoff);
        } else if (hz) {
            hcur.x = p0[hoff];
            hnxt.x = p1[hoff];
            hnx2.x = p2[hoff];
        }
    }
    auto put = [&](int b, const double2 (&v)[RPT], double2 h) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            tile[b][ty + r * TYB + 1][1 + 2 * tz] = v[r].x;
            tile[b][ty + r * TYB + 1][2 + 2 * tz] = v[r].y;
        }
        if (hy) {
            tile[b][hrow][hcol] = h.x;
            tile[b][hrow][hcol + 1] = h.y;
        } else if (hz) {
            tile[b][hrow][hcol] = h.x;
        }
    };
    put(0, cur, hcur);
    const double cI = 3.0 * a.wI[1], cE = 3.0 * a.wE[1];
    for (int p = 0; p < a.xchunk; ++p) {
        const int b = p & 1;
        const int x = x0 + p;
        __syncthreads();
        // prefetch plane x+3 (interior + halo): two planes are always in flight behind the one in use
        double2 nn[RPT];
        double2 hnn = double2{0.0, 0.0};
        const bool more = p + 1 < a.xchunk;
        if (p + 2 < a.xchunk) {
            const double* p3 = plane(x + 3);
#pragma unroll
            for (int r = 0; r < RPT; ++r) nn[r] = *reinterpret_cast<const double2*>(p3 + off[r]);
            if (hy) hnn = *reinterpret_cast<const double2*>(p3 + hoff);
            else if (hz) hnn.x = p3[hoff];
        }
        const size_t po = (size_t)x * sx;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int row = ty + r * TYB + 1, col = 1 + 2 * tz;
            const double ym0 = tile[b][row - 1][col], ym1 = tile[b][row - 1][col + 1];
            const double yp0 = tile[b][row + 1][col], yp1 = tile[b][row + 1][col + 1];
            const double zm = tile[b][row][col - 1], zp = tile[b][row][col + 2];
            const double c0 = cur[r].x, c1 = cur[r].y;
            // same association as the row sums of the Kronecker-sum matrix: per axis (w-,w0,w+), axes added
            double2 res;
            res.x = (a.wI[0] * prev[r].x + a.wI[2] * nxt[r].x) + (a.wI[0] * ym0 + a.wI[2] * yp0) +
                    (a.wI[0] * zm + a.wI[2] * c1) + cI * c0;
            res.y = (a.wI[0] * prev[r].y + a.wI[2] * nxt[r].y) + (a.wI[0] * ym1 + a.wI[2] * yp1) +
                    (a.wI[0] * c0 + a.wI[2] * zp) + cI * c1;
            if (oI) *reinterpret_cast<double2*>(oI + po + off[r]) = res;
            if (oE) {
                double2 re;
                re.x = (a.wE[0] * prev[r].x + a.wE[2] * nxt[r].x) + (a.wE[0] * ym0 + a.wE[2] * yp0) +
                       (a.wE[0] * zm + a.wE[2] * c1) + cE * c0;
                re.y = (a.wE[0] * prev[r].y + a.wE[2] * nxt[r].y) + (a.wE[0] * ym1 + a.wE[2] * yp1) +
                       (a.wE[0] * c0 + a.wE[2] * zp) + cE * c1;
                *reinterpret_cast<double2*>(oE + po + off[r]) = re;
            }
        }
        if (more) {
            put(b ^ 1, nxt, hnxt);
#pragma unroll
            for (int r = 0; r < RPT; ++r) {
                prev[r] = cur[r];
                cur[r] = nxt[r];
                nxt[r] = nx2[r];
                nx2[r] = nn[r];
            }
            hnxt = hnx2;
            hnx2 = hnn;
        }
    }
}

// eval_f for ALL nodes fused with the collocation residual: a workgroup marches the (y,z) tile through x for
// the M fields U[1..M] at once, so at every point all f_j = A u_j are in registers when the residual
// u0 + dt sum_j Q[m][j] f_j - u_m (core/sweeper.py:186-199) is formed.  Replaces stencil (10 field passes) +
// residual (11) by one kernel with 6 reads + 5 writes.
struct StencilResArgs {
    const double* U;  // slab: U[0] = u0, U[1..M]
    double* F;        // slab (ncomp == 1)
    double wI[3];
    double cQ[MAXM][MAXM];  // dt * Q[m+1][j+1]
    unsigned long long* norms;
    int n, xchunk, nchunks;
    size_t N;
};

template <int M>
__global__ __launch_bounds__(256, 3) void k_stencil3d_res(StencilResArgs a) {
    // LDS: 2 buffers x M fields x (8+2) rows x 66 doubles = 52.8 KB at M = 5 -> three workgroups per CU
    constexpr int TZ = 64, TY = 8, LW = TZ + 2;
    __shared__ double tile[2][M][TY + 2][LW];
    const int n = a.n;
    const int tz = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int ntz = n / TZ, nty = n / TY;
    unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);
    const int zt = lb % ntz;
    lb /= ntz;
    const int yt = lb % nty;
    const int chunk = lb / nty;
    const int z0 = zt * TZ, y0 = yt * TY, x0 = chunk * a.xchunk;
    const size_t sx = (size_t)n * n;
    const int t = threadIdx.x;
    const bool hy = t < 64, hz = t >= 64 && t < 64 + 2 * TY;
    size_t hoff = 0;
    int hrow = 0, hcol = 0;
    if (hy) {
        const int side = t >> 5, pz = t & 31;
        const int yy = side == 0 ? (y0 == 0 ? n - 1 : y0 - 1) : (y0 + TY == n ? 0 : y0 + TY);
        hoff = (size_t)yy * n + z0 + 2 * pz;
        hrow = side == 0 ? 0 : TY + 1;
        hcol = 1 + 2 * pz;
    } else if (hz) {
        const int q = t - 64, side = q / TY, r = q % TY;
        const int zz = side == 0 ? (z0 == 0 ? n - 1 : z0 - 1) : (z0 + TZ == n ? 0 : z0 + TZ);
        hoff = (size_t)(y0 + r) * n + zz;
        hrow = r + 1;
        hcol = side == 0 ? 0 : TZ + 1;
    }
    const size_t off = (size_t)(y0 + ty) * n + z0 + 2 * tz;
    auto wrapx = [&](int x) { return (size_t)(x < 0 ? x + n : (x >= n ? x - n : x)) * sx; };
    auto halo_load = [&](const double* plane) {
        double2 h = double2{0.0, 0.0};
        if (hy) h = *reinterpret_cast<const double2*>(plane + hoff);
        else if (hz) h.x = plane[hoff];
        return h;
    };
    auto put = [&](int b, int j, double2 v, double2 h) {
        tile[b][j][ty + 1][1 + 2 * tz] = v.x;
        tile[b][j][ty + 1][2 + 2 * tz] = v.y;
        if (hy) {
            tile[b][j][hrow][hcol] = h.x;
            tile[b][j][hrow][hcol + 1] = h.y;
        } else if (hz) {
            tile[b][j][hrow][hcol] = h.x;
        }
    };
    double2 prev[M], cur[M], nxt[M];
    double2 u0c, u0n = double2{0.0, 0.0};
    double nmax[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const double* uj = a.U + (size_t)(j + 1) * a.N;
        prev[j] = *reinterpret_cast<const double2*>(uj + wrapx(x0 - 1) + off);
        cur[j] = *reinterpret_cast<const double2*>(uj + wrapx(x0) + off);
        nxt[j] = *reinterpret_cast<const double2*>(uj + wrapx(x0 + 1) + off);
        put(0, j, cur[j], halo_load(uj + wrapx(x0)));
        nmax[j] = 0.0;
    }
    u0c = *reinterpret_cast<const double2*>(a.U + wrapx(x0) + off);
    const double cI = 3.0 * a.wI[1];
    for (int p = 0; p < a.xchunk; ++p) {
        const int b = p & 1;
        const int x = x0 + p;
        __syncthreads();
        const bool more = p + 1 < a.xchunk;
        // in flight while this plane is computed: the interior of plane x+2 and the halo of plane x+1
        double2 nn[M], hn[M];
        if (more) {
            const size_t px1 = wrapx(x + 1), px2 = wrapx(x + 2);
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const double* uj = a.U + (size_t)(j + 1) * a.N;
                nn[j] = *reinterpret_cast<const double2*>(uj + px2 + off);
                hn[j] = halo_load(uj + px1);
            }
            u0n = *reinterpret_cast<const double2*>(a.U + px1 + off);
        }
        const size_t po = (size_t)x * sx + off;
        double2 fv[M];
        const int row = ty + 1, col = 1 + 2 * tz;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const double ym0 = tile[b][j][row - 1][col], ym1 = tile[b][j][row - 1][col + 1];
            const double yp0 = tile[b][j][row + 1][col], yp1 = tile[b][j][row + 1][col + 1];
            const double zm = tile[b][j][row][col - 1], zp = tile[b][j][row][col + 2];
            const double c0 = cur[j].x, c1 = cur[j].y;
            fv[j].x = (a.wI[0] * prev[j].x + a.wI[2] * nxt[j].x) + (a.wI[0] * ym0 + a.wI[2] * yp0) +
                      (a.wI[0] * zm + a.wI[2] * c1) + cI * c0;
            fv[j].y = (a.wI[0] * prev[j].y + a.wI[2] * nxt[j].y) + (a.wI[0] * ym1 + a.wI[2] * yp1) +
                      (a.wI[0] * c0 + a.wI[2] * zp) + cI * c1;
            *reinterpret_cast<double2*>(a.F + (size_t)(j + 1) * a.N + po) = fv[j];
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double r0 = 0.0, r1 = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                r0 += a.cQ[m][j] * fv[j].x;
                r1 += a.cQ[m][j] * fv[j].y;
            }
            r0 = fabs((r0 + u0c.x) - cur[m].x);
            r1 = fabs((r1 + u0c.y) - cur[m].y);
            const double r = (r0 > r1 || r0 != r0) ? r0 : r1;
            nmax[m] = (nmax[m] > r || nmax[m] != nmax[m]) ? nmax[m] : r;
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < M; ++j) {
                put(b ^ 1, j, nxt[j], hn[j]);
                prev[j] = cur[j];
                cur[j] = nxt[j];
                nxt[j] = nn[j];
            }
            u0c = u0n;
        }
    }
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const double v = wave_max(nmax[m]);
        if ((threadIdx.x & 63) == 0) atomic_max_abs(a.norms + m, v);
    }
}

// pointwise explicit (reaction) terms of the Allen-Cahn problems
//   kind 1: c * u * (1 - u^nu),  c = 1/eps^2          (AllenCahn_2D_FFT.py:140-141)
//   kind 2: -2/eps^2 u (1-u)(1-2u) - 6 dw u (1-u)     (AllenCahn_MPIFFT.py:83-85)
__global__ void k_reaction(const double* __restrict__ u, double* __restrict__ out, size_t n, int kind, double p0,
                           double p1, int nu) {
#pragma clang fp contract(off)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = u[i];
        double r;
        if (kind == 1) {
            double pw = 1.0;
            for (int q = 0; q < nu; ++q) pw *= v;
            r = p0 * v * (1.0 - pw);
        } else {
            r = p0 * v * (1.0 - v) * (1.0 - 2.0 * v) - p1 * v * (1.0 - v);
        }
        out[i] = r;
    }
}

// ------------------------------------------------------------------------------------------------------
// FFT kernels
// ------------------------------------------------------------------------------------------------------
struct FieldPtrs {
    const double* in[MAXM];
    double* out[MAXM];
};

// 1-D problems: promote the real line to complex / take the real part back
__global__ void k_promote(FieldPtrs p, cd* W, size_t N) {
    const int f = blockIdx.y;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x)
        W[(size_t)f * N + i] = cd{p.in[f][i], 0.0};
}
__global__ void k_realpart(FieldPtrs p, const cd* W, size_t N) {
    const int f = blockIdx.y;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x)
        p.out[f][i] = W[(size_t)f * N + i].x;
}

// r2c along axis 0: real field [N][rest] seen as complex pairs [N][rest/2]; two real columns per complex
// column ("two for one"), unpacked to the half spectra W[k][rest], k = 0..N/2.
template <int N, int T>
__global__ __launch_bounds__((N / fft_elems(N)) * T, 4) void k_fftx_fwd(FieldPtrs p, cd* __restrict__ W, size_t fstride,
                                                                      int rest, const cd* __restrict__ tw) {
    constexpr int E = fft_elems(N), P = N / E;
    using LAY = LayStrided<N, T>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int col = threadIdx.x % T, j = threadIdx.x / T;
    const int ncol = rest >> 1;  // complex columns
    const int c = blockIdx.x * T + col;
    const bool ok = c < ncol;
    const double* __restrict__ in = p.in[blockIdx.y];
    cd r[E];
#pragma unroll
    for (int i = 0; i < E; ++i)
        r[i] = ok ? *reinterpret_cast<const cd*>(in + (size_t)(j + i * P) * rest + 2 * (size_t)c) : cd{0.0, 0.0};
    fft_line<N, -1, LAY>(r, j, col, lds, tw);
    // unpack: A[k] = (C[k] + conj C[N-k]) / 2, B[k] = (C[k] - conj C[N-k]) / (2i)
    cd A[E], B[E];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int i = 0; i < E; ++i) lds[LAY::idx(col, j + i * P)] = part == 0 ? r[i].x : r[i].y;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int k = j + i * P;
            const double v = lds[LAY::idx(col, (N - k) & (N - 1))];
            if (part == 0) {
                A[i].x = 0.5 * (r[i].x + v);
                B[i].y = -0.5 * (r[i].x - v);
            } else {
                A[i].y = 0.5 * (r[i].y - v);
                B[i].x = 0.5 * (r[i].y + v);
            }
        }
        __syncthreads();
    }
    cd* __restrict__ Wf = W + blockIdx.y * fstride;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int k = j + i * P;
        if (ok && k <= N / 2) {
            cd* dst = Wf + (size_t)k * rest + 2 * (size_t)c;
            dst[0] = A[i];
            dst[1] = B[i];
        }
    }
}

// c2r along axis 0 (inverse of the above, unnormalised)
template <int N, int T>
__global__ __launch_bounds__((N / fft_elems(N)) * T, 4) void k_fftx_inv(FieldPtrs p, const cd* __restrict__ W,
                                                                      size_t fstride, int rest,
                                                                      const cd* __restrict__ tw) {
    constexpr int E = fft_elems(N), P = N / E;
    using LAY = LayStrided<N, T>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int col = threadIdx.x % T, j = threadIdx.x / T;
    const int ncol = rest >> 1;
    const int c = blockIdx.x * T + col;
    const bool ok = c < ncol;
    const cd* __restrict__ Wf = W + blockIdx.y * fstride;
    cd A[E], B[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int k = j + i * P;
        if (ok && k <= N / 2) {
            const cd* src = Wf + (size_t)k * rest + 2 * (size_t)c;
            A[i] = src[0];
            B[i] = src[1];
        } else {
            A[i] = B[i] = cd{0.0, 0.0};
        }
    }
    cd r[E];
    // C[k] = A[k] + i B[k] (k <= N/2), C[N-k] = conj A[k] + i conj B[k]
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int k = j + i * P;
            if (k <= N / 2) {
                const bool edge = (k == 0) || (k == N / 2);
                double own, mir;
                if (part == 0) {
                    own = edge ? A[i].x : A[i].x - B[i].y;
                    mir = A[i].x + B[i].y;
                } else {
                    own = edge ? B[i].x : A[i].y + B[i].x;
                    mir = -A[i].y + B[i].x;
                }
                lds[LAY::idx(col, k)] = own;
                if (!edge) lds[LAY::idx(col, N - k)] = mir;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const double v = lds[LAY::idx(col, j + i * P)];
            if (part == 0) r[i].x = v;
            else r[i].y = v;
        }
        __syncthreads();
    }
    fft_line<N, +1, LAY>(r, j, col, lds, tw);
    double* __restrict__ out = p.out[blockIdx.y];
    if (ok) {
#pragma unroll
        for (int i = 0; i < E; ++i) *reinterpret_cast<cd*>(out + (size_t)(j + i * P) * rest + 2 * (size_t)c) = r[i];
    }
}

// c2c in place along the middle axis of W[f][kx][y][z] (3-D only): tile = all y x T z-columns
template <int N, int T, int DIR>
__global__ __launch_bounds__((N / fft_elems(N)) * T, 4) void k_ffty(cd* __restrict__ W, size_t fstride,
                                                                  const cd* __restrict__ tw) {
    constexpr int E = fft_elems(N), P = N / E;
    using LAY = LayStrided<N, T>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int col = threadIdx.x % T, j = threadIdx.x / T;
    const int c = blockIdx.x * T + col;
    const bool ok = c < N;
    cd* __restrict__ base = W + blockIdx.z * fstride + (size_t)blockIdx.y * N * N + c;
    cd r[E];
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = ok ? base[(size_t)(j + i * P) * N] : cd{0.0, 0.0};
    fft_line<N, DIR, LAY>(r, j, col, lds, tw);
    if (ok) {
#pragma unroll
        for (int i = 0; i < E; ++i) base[(size_t)(j + i * P) * N] = r[i];
    }
}

struct ZArgs {
    cd* W;
    size_t fstride;
    const cd *tw, *lamI, *lamE;  // lamE may be null
    double cI[MAXM][MAXM];       // strictly lower: dt*QI[m+1][j+1], j < m
    double cE[MAXM][MAXM];       // strictly lower: dt*QE[m+1][j+1]
    double alpha[MAXM];          // dt*QI[m+1][m+1]
    double invN;
    int nf, ndim, coupled;
    int apply;  // 1: multiply by the symbol (operator application) instead of dividing by 1 - alpha*symbol
};

// forward FFT along the contiguous axis, node-coupled implicit solve in Fourier space, inverse FFT.
// One workgroup = LPB lines x all nf fields; column c = f*LPB + l occupies threads [c*P, (c+1)*P).
// After the forward transform the spectra go through LDS once more so that one thread holds ALL nf node
// values of a mode: the lower-triangular node coupling is then a register recurrence with wave-uniform
// coefficient indices (scalar kernarg loads, no per-lane table look-ups).
template <int N>
constexpr int z_lines_per_block() {
    constexpr int P = N / fft_elems(N);
    return P >= 64 ? 1 : 64 / P;
}

template <int N>
__global__ __launch_bounds__(z_lines_per_block<N>() * (N / fft_elems(N)) * MAXM, 3) void k_fftz_solve(ZArgs a, unsigned nlines) {
    constexpr int E = fft_elems(N), P = N / E, LPB = z_lines_per_block<N>();
    constexpr int NCH = E == 16 ? 2 : 1;  // the solve buffer holds N/NCH modes per column at a time
    constexpr int CH = N / NCH, ECH = E / NCH;
    using LAY = LayContig<N>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = threadIdx.x / P, j = threadIdx.x % P;
    const int f = c / LPB, l = c % LPB;
    const size_t line = (size_t)blockIdx.x * LPB + l;
    const bool ok = line < nlines;
    cd* __restrict__ Wl = a.W + f * a.fstride + line * N;
    cd r[E];
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = ok ? Wl[j + i * P] : cd{0.0, 0.0};
    fft_line<N, -1, LAY, (N / fft_elems(N)) <= 64>(r, j, c, lds, a.tw);
    __syncthreads();  // the solve buffer aliases other waves' exchange planes

    cd* buf = reinterpret_cast<cd*>(lds);  // [column][CH]
    const int nthreads = a.nf * LPB * P;
#pragma unroll
    for (int ph = 0; ph < NCH; ++ph) {
#pragma unroll
        for (int i = 0; i < ECH; ++i) buf[c * CH + j + i * P] = r[ph * ECH + i];
        __syncthreads();
        for (int item = threadIdx.x; item < LPB * CH; item += nthreads) {
            const int ll = item / CH, kk = item % CH;
            const size_t ln = (size_t)blockIdx.x * LPB + ll;
            const int kz = ph * CH + kk;
            cd lam = a.lamI[kz], mu = cd{0.0, 0.0};
            if (a.lamE) mu = a.lamE[kz];
            if (a.ndim == 3) {
                const int kx = (int)(ln / N) % (N / 2 + 1), ky = (int)(ln % N);
                lam = cadd(lam, cadd(a.lamI[kx], a.lamI[ky]));
                if (a.lamE) mu = cadd(mu, cadd(a.lamE[kx], a.lamE[ky]));
            } else if (a.ndim == 2) {
                const int kx = (int)(ln % (N / 2 + 1));
                lam = cadd(lam, a.lamI[kx]);
                if (a.lamE) mu = cadd(mu, a.lamE[kx]);
            }
            cd u[MAXM];
#pragma unroll
            for (int m = 0; m < MAXM; ++m) {
                if (m < a.nf) {
                    cd acc = buf[(m * LPB + ll) * CH + kk];
                    if (a.coupled) {
#pragma unroll
                        for (int q = 0; q < m; ++q) {
                            const double ci = a.cI[m][q], ce = a.cE[m][q];
                            const cd coef = cd{ci * lam.x + ce * mu.x, ci * lam.y + ce * mu.y};
                            acc = cfma(coef, u[q], acc);
                        }
                    }
                    const double al = a.alpha[m];
                    u[m] = a.apply ? cmul(acc, lam) : cmul(acc, cinv_fast(cd{1.0 - al * lam.x, -al * lam.y}));
                    buf[(m * LPB + ll) * CH + kk] = cscale(u[m], a.invN);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < ECH; ++i) r[ph * ECH + i] = buf[c * CH + j + i * P];
        __syncthreads();
    }
    // opaque copy of the lane index: without it the forward transform's twiddles stay live (~100 VGPRs)
    // through the whole kernel for reuse in the inverse
    int j2 = j;
    asm volatile("" : "+v"(j2));
    fft_line<N, +1, LAY, (N / fft_elems(N)) <= 64>(r, j2, c, lds, a.tw);
    if (ok) {
#pragma unroll
        for (int i = 0; i < E; ++i) Wl[j2 + i * P] = r[i];
    }
}

// plain transform along the contiguous axis, src -> dst (may alias), optionally scaled: forward to bring u0 /
// node values into the fully transformed domain of the spectral cache, inverse after the spectral sweep
template <int N, int DIR>
__global__ __launch_bounds__(z_lines_per_block<N>() * (N / fft_elems(N)) * MAXM, 4) void k_fftz_plain(
    const cd* __restrict__ src, cd* __restrict__ dst, size_t fstride, const cd* __restrict__ tw, unsigned nlines,
    double scale) {
    constexpr int E = fft_elems(N), P = N / E, LPB = z_lines_per_block<N>();
    using LAY = LayContig<N>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int c = threadIdx.x / P, j = threadIdx.x % P;
    const int f = c / LPB, l = c % LPB;
    const size_t line = (size_t)blockIdx.x * LPB + l;
    const bool ok = line < nlines;
    const size_t base = f * fstride + line * N;
    cd r[E];
#pragma unroll
    for (int i = 0; i < E; ++i) r[i] = ok ? src[base + j + i * P] : cd{0.0, 0.0};
    if (scale != 1.0) {
#pragma unroll
        for (int i = 0; i < E; ++i) r[i] = cscale(r[i], scale);
    }
    fft_line<N, DIR, LAY, (N / fft_elems(N)) <= 64>(r, j, c, lds, tw);
    if (ok) {
#pragma unroll
        for (int i = 0; i < E; ++i) dst[base + j + i * P] = r[i];
    }
}

// Sweep in the transformed domain (DESIGN.md "spectral reuse").  For linear f(u) = A u (+ B u) the gathered
// right-hand side of node m is  u0 + dt sum_j (Q-QI)[m][j] A u_j^k (+ explicit part): its transform follows
// from the transforms of u0 and of the previous iterate, which the previous sweep left in S.  One launch
// reads S0 and S[0..nf), applies gather + node-coupled solve per mode, writes the new spectra back to S and
// their inverse transform along the contiguous axis to W (input of the inverse y / x passes).
struct SpecArgs {
    cd* S;
    size_t fstride;
    const cd* S0;
    cd* W;
    const cd *tw, *lamI, *lamE;
    double gI[MAXM][MAXM], gE[MAXM][MAXM];  // dt (Q - QI), dt (Q - QE), inner MxM blocks
    double cI[MAXM][MAXM], cE[MAXM][MAXM], alpha[MAXM];
    double invN;
    int nf, ndim, coupled, spread;
};

// one thread per Fourier mode: gather on the cached transforms + node-coupled solve, S updated in place
template <int NF>
__global__ __launch_bounds__(256) void k_spec_point(SpecArgs a, int n, size_t nmodes) {
    for (size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x; g < nmodes; g += (size_t)gridDim.x * blockDim.x) {
        const int kz = (int)(g % n);
        const size_t ln = g / n;
        cd lam = a.lamI[kz], mu = cd{0.0, 0.0};
        if (a.lamE) mu = a.lamE[kz];
        if (a.ndim == 3) {
            const int kx = (int)(ln / n), ky = (int)(ln % n);
            lam = cadd(lam, cadd(a.lamI[kx], a.lamI[ky]));
            if (a.lamE) mu = cadd(mu, cadd(a.lamE[kx], a.lamE[ky]));
        } else if (a.ndim == 2) {
            lam = cadd(lam, a.lamI[ln]);
            if (a.lamE) mu = cadd(mu, a.lamE[ln]);
        }
        const cd u0h = a.S0[g];
        cd old[NF], u[NF];
#pragma unroll
        for (int q = 0; q < NF; ++q) old[q] = a.spread ? u0h : a.S[q * a.fstride + g];
#pragma unroll
        for (int m = 0; m < NF; ++m) {
            cd acc = u0h;
#pragma unroll
            for (int q = 0; q < NF; ++q) {
                const double gi = a.gI[m][q], ge = a.gE[m][q];
                acc = cfma(cd{gi * lam.x + ge * mu.x, gi * lam.y + ge * mu.y}, old[q], acc);
            }
            if (a.coupled) {
#pragma unroll
                for (int q = 0; q < m; ++q) {
                    const double ci = a.cI[m][q], ce = a.cE[m][q];
                    acc = cfma(cd{ci * lam.x + ce * mu.x, ci * lam.y + ce * mu.y}, u[q], acc);
                }
            }
            const double al = a.alpha[m];
            u[m] = cmul(acc, cinv_fast(cd{1.0 - al * lam.x, -al * lam.y}));
            a.S[m * a.fstride + g] = u[m];
        }
    }
}

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
    double Q[MAXM][MAXM], QI[MAXM][MAXM];
    unsigned long long* counters;
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
template <int M>
__global__ __launch_bounds__(256) void k_vdp_sweep(VdpSweepArgs a) {
#pragma clang fp contract(off)
    unsigned long long newton = 0, rhs = 0, failed = 0;
    const size_t T = a.T, N = 2 * a.T;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < T; i += (size_t)gridDim.x * blockDim.x) {
        const double mu = a.mu, dt = a.dt;
        const double u00 = a.U[i], u01 = a.U[T + i];
        double f0[M], f1[M], g0[M], g1[M];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            f0[m] = a.F[(size_t)(m + 1) * N + i];
            f1[m] = a.F[(size_t)(m + 1) * N + T + i];
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                s0 += dt * a.Q[m][j] * f0[j];
                s1 += dt * a.Q[m][j] * f1[j];
            }
#pragma unroll
            for (int j = 0; j < M; ++j) {
                s0 -= dt * a.QI[m][j] * f0[j];
                s1 -= dt * a.QI[m][j] * f1[j];
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
                    r0 += dt * a.QI[m][j] * f0[j];
                    r1 += dt * a.QI[m][j] * f1[j];
                }
            }
            const double h = dt * a.QI[m][m];
            double x1 = a.U[(size_t)(m + 1) * N + i], x2 = a.U[(size_t)(m + 1) * N + T + i];
            if (h == 0.0) {
                x1 = r0;
                x2 = r1;
            } else if (!vdp_newton(x1, x2, r0, r1, h, mu, a.tol, a.maxiter, newton)) {
                failed += 1;
            }
            a.U[(size_t)(m + 1) * N + i] = x1;
            a.U[(size_t)(m + 1) * N + T + i] = x2;
            f0[m] = x2;
            f1[m] = mu * (1 - x1 * x1) * x2 - x1;
            a.F[(size_t)(m + 1) * N + i] = f0[m];
            a.F[(size_t)(m + 1) * N + T + i] = f1[m];
            rhs += 1;
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

// ------------------------------------------------------------------------------------------------------
// space transfer between nested periodic grids (coarsening by 2 per axis): tensor products of the 1-D
// interpolation  fine[2i] = coarse[i], fine[2i+1] = sum_j w[j] coarse[i - k/2 + 1 + j]  and of its scaled
// transpose (TransferMesh.py:49-146 with helpers/transfer_helper.py:153-186, periodic / equidist_nested)
// ------------------------------------------------------------------------------------------------------
struct XferArgs {
    const double* in;
    double* out;
    const int* idx;     // [n_out][W] source indices along the axis (device)
    const double* w;    // [n_out][W] weights (zero-padded)
    size_t outer, inner;
    int n_out, n_in, W;
};

// one axis of the tensor product: out[o][i][q] = sum_j w[i][j] * in[o][idx[i][j]][q]
__global__ void k_xfer_axis(XferArgs a) {
    const size_t total = a.outer * a.n_out * a.inner;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t q = p % a.inner;
        const size_t r = p / a.inner;
        const int i = (int)(r % a.n_out);
        const size_t o = r / a.n_out;
        const double* __restrict__ src = a.in + o * a.n_in * a.inner + q;
        double acc = 0.0;
        for (int j = 0; j < a.W; ++j) {
            const double wj = a.w[i * a.W + j];
            if (wj != 0.0) acc += wj * src[(size_t)a.idx[i * a.W + j] * a.inner];
        }
        a.out[p] = acc;
    }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static int ensure_work(sdc_ctx* c);
static inline int grid_for(size_t work, int block) {
    size_t g = (work + block - 1) / block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

static double* slot_ptr(sdc_ctx* c, int slot, int m, int comp) {
    switch (slot) {
        case SDC_SLOT_U: return (m >= 0 && m <= c->M) ? c->U + (size_t)m * c->N : nullptr;
        case SDC_SLOT_F:
            return (m >= 0 && m <= c->M && comp >= 0 && comp < c->ncomp) ? c->F + ((size_t)m * c->ncomp + comp) * c->N
                                                                         : nullptr;
        case SDC_SLOT_TAU: return (m >= 0 && m < c->M) ? c->TAU + (size_t)m * c->N : nullptr;
        case SDC_SLOT_UEND: return c->UEND;
        default: return nullptr;
    }
}

template <int MODE>
static int launch_quad(sdc_ctx* c, const QuadArgs& a, const char* name) {
    LaunchTimer lt(c, name);
    const int grid = grid_for(c->N / 2, 256);
#define QCASE(MM)                                                                                   \
    case MM:                                                                                        \
        if (c->ncomp == 2) hipLaunchKernelGGL((k_quad<MM, 2, MODE>), dim3(grid), dim3(256), 0, c->stream, a); \
        else hipLaunchKernelGGL((k_quad<MM, 1, MODE>), dim3(grid), dim3(256), 0, c->stream, a);     \
        break;
    switch (c->M) {
        QCASE(1) QCASE(2) QCASE(3) QCASE(4) QCASE(5) QCASE(6) QCASE(7) QCASE(8)
        default: return fail(c, SDC_ERR_PARAM, "num_nodes %d > %d", c->M, MAXM);
    }
#undef QCASE
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

static void quad_base(sdc_ctx* c, QuadArgs& a) {
    memset(&a, 0, sizeof a);
    a.F = c->F;
    a.N = c->N;
    a.nout = c->M;
}

static int run_stencil(sdc_ctx* c, int nf, const double* const* in, double* const* outI, double* const* outE,
                       const double* g) {
    StencilArgs a;
    memset(&a, 0, sizeof a);
    a.nf = nf;
    a.ndim = c->ndim;
    a.n = c->n;
    a.sI = c->st[0];
    a.sE = c->st[1];
    a.useE = c->expl_kind;
    a.profile = c->profile;
    for (int f = 0; f < nf; ++f) {
        a.in[f] = in[f];
        a.outI[f] = outI ? outI[f] : nullptr;
        a.outE[f] = (outE && c->expl_kind != SDC_EXPL_NONE) ? outE[f] : nullptr;
        a.g[f] = g ? g[f] : 0.0;
    }
    LaunchTimer lt(c, pname("stencil", nf));
    // 3-D fast path: both operators (when present) are 3-point stencils with offsets -1, 0, +1
    auto three = [](const Stencil& s) { return s.npts == 3 && s.off[0] == -1 && s.off[1] == 0 && s.off[2] == 1; };
    const bool needE = c->expl_kind == SDC_EXPL_STENCIL && outE != nullptr;
    constexpr int RPT = 4;
    if (c->ndim == 3 && c->n % 64 == 0 && c->n % (8 * RPT) == 0 && outI && three(c->st[0]) &&
        (!needE || three(c->st[1])) && c->expl_kind != SDC_EXPL_FORCING) {
        Stencil3Args s3;
        memset(&s3, 0, sizeof s3);
        for (int f = 0; f < nf; ++f) {
            s3.in[f] = in[f];
            s3.outI[f] = outI[f];
            s3.outE[f] = needE ? outE[f] : nullptr;
        }
        for (int k = 0; k < 3; ++k) {
            s3.wI[k] = c->st[0].w[k];
            s3.wE[k] = needE ? c->st[1].w[k] : 0.0;
        }
        s3.n = c->n;
        s3.xchunk = c->n >= 64 ? 64 : c->n;
        s3.nchunks = c->n / s3.xchunk;
        s3.ntiles = (c->n / 64) * (c->n / (8 * RPT));
        hipLaunchKernelGGL((k_stencil3d<RPT>), dim3(s3.ntiles * s3.nchunks * nf), dim3(256), 0, c->stream, s3);
        HIPCHK(c, hipGetLastError());
        return SDC_OK;
    }
    hipLaunchKernelGGL(k_stencil, dim3(grid_for(c->N / 2, 256), nf), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// F[1..M] = A U[1..M] for the new iterate; fused with the residual when the fast 3-D kernel applies
static int eval_nodes(sdc_ctx* c, double dt) {
    const int M = c->M;
    auto three = [](const Stencil& s) { return s.npts == 3 && s.off[0] == -1 && s.off[1] == 0 && s.off[2] == 1; };
    if (c->fuse_residual && c->ndim == 3 && c->ncomp == 1 && !c->tau_active && c->n % 64 == 0 && M <= 6 &&
        three(c->st[0])) {
        StencilResArgs a;
        memset(&a, 0, sizeof a);
        a.U = c->U;
        a.F = c->F;
        for (int k = 0; k < 3; ++k) a.wI[k] = c->st[0].w[k];
        for (int m = 0; m < M; ++m)
            for (int j = 0; j < M; ++j) a.cQ[m][j] = dt * c->Q[m + 1][j + 1];
        a.norms = c->res_dev;
        a.n = c->n;
        a.N = c->N;
        a.xchunk = c->n >= 64 ? 64 : c->n;
        a.nchunks = c->n / a.xchunk;
        HIPCHK(c, hipMemsetAsync(c->res_dev, 0, sizeof(unsigned long long) * 8, c->stream));
        const unsigned grid = (unsigned)((c->n / 64) * (c->n / 8) * a.nchunks);
        {
            LaunchTimer lt(c, pname("stencil_res", M));
#define RCASE(MM) \
    case MM: hipLaunchKernelGGL((k_stencil3d_res<MM>), dim3(grid), dim3(256), 0, c->stream, a); break;
            switch (M) { RCASE(1) RCASE(2) RCASE(3) RCASE(4) RCASE(5) RCASE(6) }
#undef RCASE
        }
        HIPCHK(c, hipGetLastError());
        c->res_valid = true;
        c->res_dt = dt;
        return SDC_OK;
    }
    const double* in[MAXM];
    double* oi[MAXM];
    double* oe[MAXM];
    double g[MAXM];
    for (int m = 0; m < M; ++m) {
        in[m] = c->U + (size_t)(m + 1) * c->N;
        oi[m] = c->F + ((size_t)(m + 1) * c->ncomp) * c->N;
        oe[m] = (c->ncomp == 2 && c->expl_kind == SDC_EXPL_STENCIL) ? oi[m] + c->N : nullptr;
        g[m] = c->gvals[m + 1];
    }
    return run_stencil(c, M, in, oi, c->expl_kind == SDC_EXPL_STENCIL ? oe : nullptr, g);
}

template <int N>
static int fft_pipeline_n(sdc_ctx* c, int nf, const FieldPtrs& p, ZArgs& z) {
    constexpr int E = fft_elems(N), P = N / E;
    constexpr int T = N >= 2048 ? 4 : 8;  // complex columns per strided tile (128-byte row segments up to N = 1024)
    const size_t lds_str = (size_t)LayStrided<N, T>::doubles(T) * sizeof(double);
    const int n = c->n;
    z.W = c->W;
    z.fstride = c->Nc;
    z.tw = c->tw;
    z.lamI = c->lamI;
    z.nf = nf;
    z.ndim = c->ndim;
    z.invN = 1.0 / (double)c->N;
    size_t lines;
    if (c->ndim == 1) {
        LaunchTimer lt(c, pname("promote", nf));
        hipLaunchKernelGGL(k_promote, dim3(grid_for(c->N, 256), nf), dim3(256), 0, c->stream, p, c->W, c->N);
        lines = 1;
    } else {
        const int rest = (int)(c->N / n);
        const int tiles = (rest / 2 + T - 1) / T;
        {
            LaunchTimer lt(c, pname("fft_x_fwd", nf));
            hipLaunchKernelGGL((k_fftx_fwd<N, T>), dim3(tiles, nf), dim3(P * T), lds_str, c->stream, p, c->W, c->Nc,
                               rest, c->tw);
        }
        if (c->ndim == 3) {
            LaunchTimer lt(c, pname("fft_y_fwd", nf));
            hipLaunchKernelGGL((k_ffty<N, T, -1>), dim3((n + T - 1) / T, n / 2 + 1, nf), dim3(P * T), lds_str,
                               c->stream, c->W, c->Nc, c->tw);
        }
        lines = (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    }
    {
        LaunchTimer lt(c, pname("fft_z_solve", nf));
        constexpr int LPB = z_lines_per_block<N>();
        constexpr int NCH = fft_elems(N) == 16 ? 2 : 1;
        size_t ldsz = (size_t)LayContig<N>::doubles(nf * LPB) * sizeof(double);  // FFT exchange planes
        const size_t solve_sz = (size_t)nf * LPB * (N / NCH) * sizeof(cd);         // node-coupling buffer
        if (solve_sz > ldsz) ldsz = solve_sz;
        hipLaunchKernelGGL((k_fftz_solve<N>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(P * LPB * nf), ldsz,
                           c->stream, z, (unsigned)lines);
    }
    if (c->ndim == 1) {
        LaunchTimer lt(c, pname("realpart", nf));
        hipLaunchKernelGGL(k_realpart, dim3(grid_for(c->N, 256), nf), dim3(256), 0, c->stream, p, c->W, c->N);
    } else {
        const int rest = (int)(c->N / n);
        const int tiles = (rest / 2 + T - 1) / T;
        if (c->ndim == 3) {
            LaunchTimer lt(c, pname("fft_y_inv", nf));
            hipLaunchKernelGGL((k_ffty<N, T, +1>), dim3((n + T - 1) / T, n / 2 + 1, nf), dim3(P * T), lds_str,
                               c->stream, c->W, c->Nc, c->tw);
        }
        LaunchTimer lt(c, pname("fft_x_inv", nf));
        hipLaunchKernelGGL((k_fftx_inv<N, T>), dim3(tiles, nf), dim3(P * T), lds_str, c->stream, p, c->W, c->Nc, rest,
                           c->tw);
    }
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// forward transform of nf real fields into fully transformed spectra dst[f] (dst + f*fstride)
template <int N>
static int fwd_transform_n(sdc_ctx* c, int nf, const FieldPtrs& p, cd* dst, size_t fstride) {
    constexpr int E = fft_elems(N), P = N / E, T = N >= 2048 ? 4 : 8, LPB = z_lines_per_block<N>();
    const size_t lds_str = (size_t)LayStrided<N, T>::doubles(T) * sizeof(double);
    const int n = c->n;
    size_t lines;
    if (c->ndim == 1) {
        LaunchTimer lt(c, pname("promote", nf));
        hipLaunchKernelGGL(k_promote, dim3(grid_for(c->N, 256), nf), dim3(256), 0, c->stream, p, dst, c->N);
        lines = 1;
    } else {
        const int rest = (int)(c->N / n);
        const int tiles = (rest / 2 + T - 1) / T;
        {
            LaunchTimer lt(c, pname("fft_x_fwd", nf));
            hipLaunchKernelGGL((k_fftx_fwd<N, T>), dim3(tiles, nf), dim3(P * T), lds_str, c->stream, p, dst, fstride, rest,
                               c->tw);
        }
        if (c->ndim == 3) {
            LaunchTimer lt(c, pname("fft_y_fwd", nf));
            hipLaunchKernelGGL((k_ffty<N, T, -1>), dim3((n + T - 1) / T, n / 2 + 1, nf), dim3(P * T), lds_str, c->stream,
                               dst, fstride, c->tw);
        }
        lines = (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    }
    LaunchTimer lt(c, pname("fft_z_fwd", nf));
    const size_t ldsz = (size_t)LayContig<N>::doubles(nf * LPB) * sizeof(double);
    hipLaunchKernelGGL((k_fftz_plain<N, -1>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(P * LPB * nf), ldsz,
                       c->stream, dst, dst, fstride, c->tw, (unsigned)lines, 1.0);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// spectral sweep + inverse passes into out[f]
template <int N>
static int spec_sweep_n(sdc_ctx* c, int nf, SpecArgs& a, const FieldPtrs& p) {
    constexpr int E = fft_elems(N), P = N / E, T = N >= 2048 ? 4 : 8, LPB = z_lines_per_block<N>();
    constexpr int NCH = E == 16 ? 2 : 1;
    const size_t lds_str = (size_t)LayStrided<N, T>::doubles(T) * sizeof(double);
    const int n = c->n;
    const size_t lines = c->ndim == 1 ? 1 : (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    {
        LaunchTimer lt(c, pname("spec_point", nf));
        const size_t nmodes = lines * N;
        const dim3 grid(grid_for(nmodes, 256));
#define SCASE(MM) \
    case MM: hipLaunchKernelGGL((k_spec_point<MM>), grid, dim3(256), 0, c->stream, a, n, nmodes); break;
        switch (nf) { SCASE(1) SCASE(2) SCASE(3) SCASE(4) SCASE(5) SCASE(6) SCASE(7) SCASE(8) }
#undef SCASE
    }
    {
        LaunchTimer lt(c, pname("fft_z_inv", nf));
        const size_t ldsz = (size_t)LayContig<N>::doubles(nf * LPB) * sizeof(double);
        hipLaunchKernelGGL((k_fftz_plain<N, +1>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(P * LPB * nf), ldsz,
                           c->stream, c->S, c->W, c->Nc, c->tw, (unsigned)lines, a.invN);
    }
    (void)NCH;
    if (c->ndim == 1) {
        LaunchTimer lt(c, pname("realpart", nf));
        hipLaunchKernelGGL(k_realpart, dim3(grid_for(c->N, 256), nf), dim3(256), 0, c->stream, p, c->W, c->N);
    } else {
        const int rest = (int)(c->N / n);
        const int tiles = (rest / 2 + T - 1) / T;
        if (c->ndim == 3) {
            LaunchTimer lt(c, pname("fft_y_inv", nf));
            hipLaunchKernelGGL((k_ffty<N, T, +1>), dim3((n + T - 1) / T, n / 2 + 1, nf), dim3(P * T), lds_str, c->stream,
                               c->W, c->Nc, c->tw);
        }
        LaunchTimer lt(c, pname("fft_x_inv", nf));
        hipLaunchKernelGGL((k_fftx_inv<N, T>), dim3(tiles, nf), dim3(P * T), lds_str, c->stream, p, c->W, c->Nc, rest,
                           c->tw);
    }
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

#define N_DISPATCH(c, CALL)                                                                                 \
    switch ((c)->n) {                                                                                       \
        case 2: return CALL(2);                                                                             \
        case 4: return CALL(4);                                                                             \
        case 8: return CALL(8);                                                                             \
        case 16: return CALL(16);                                                                           \
        case 32: return CALL(32);                                                                           \
        case 64: return CALL(64);                                                                           \
        case 128: return CALL(128);                                                                         \
        case 256: return CALL(256);                                                                         \
        case 512: return CALL(512);                                                                         \
        case 1024: return CALL(1024);                                                                       \
        case 2048: return CALL(2048);                                                                       \
        default: return fail(c, SDC_ERR_UNSUPPORTED, "spectral solve needs n = 2^p <= 2048, got %d", (c)->n); \
    }

static int fwd_transform(sdc_ctx* c, int nf, const FieldPtrs& p, cd* dst, size_t fstride) {
#define CALL(NN) fwd_transform_n<NN>(c, nf, p, dst, fstride)
    N_DISPATCH(c, CALL)
#undef CALL
}
static int spec_sweep(sdc_ctx* c, int nf, SpecArgs& a, const FieldPtrs& p) {
    {
        int rw = ensure_work(c);
        if (rw != SDC_OK) return rw;
        a.W = c->W;
    }
#define CALL(NN) spec_sweep_n<NN>(c, nf, a, p)
    N_DISPATCH(c, CALL)
#undef CALL
}

// (I - alpha_f A) out_f = in_f + sum_{j<f} (cI[f][j] A + cE[f][j] B) out_j for f = 0..nf-1
static int fft_pipeline(sdc_ctx* c, int nf, const FieldPtrs& p, ZArgs& z) {
    if (!c->have_stencil[0]) return fail(c, SDC_ERR_STATE, "implicit operator not set (sdc_set_stencil)");
    if (!is_pow2(c->n) || c->n > 2048 || (c->n > 1024 && c->ndim > 1))
        return fail(c, SDC_ERR_UNSUPPORTED,
                    "spectral solve needs n = 2^p <= 1024 per dimension (<= 2048 in 1-D), got %d", c->n);
    {
        int rw = ensure_work(c);
        if (rw != SDC_OK) return rw;
    }
    switch (c->n) {
        case 2: return fft_pipeline_n<2>(c, nf, p, z);
        case 4: return fft_pipeline_n<4>(c, nf, p, z);
        case 8: return fft_pipeline_n<8>(c, nf, p, z);
        case 16: return fft_pipeline_n<16>(c, nf, p, z);
        case 32: return fft_pipeline_n<32>(c, nf, p, z);
        case 64: return fft_pipeline_n<64>(c, nf, p, z);
        case 128: return fft_pipeline_n<128>(c, nf, p, z);
        case 256: return fft_pipeline_n<256>(c, nf, p, z);
        case 512: return fft_pipeline_n<512>(c, nf, p, z);
        case 1024: return fft_pipeline_n<1024>(c, nf, p, z);
        case 2048: return fft_pipeline_n<2048>(c, nf, p, z);
    }
    return fail(c, SDC_ERR_UNSUPPORTED, "n = %d", c->n);
}

static int build_symbol(sdc_ctx* c, int which) {
    const int n = c->n;
    std::vector<cd> lam(n);
    const Stencil& s = c->st[which];
    for (int k = 0; k < n; ++k) {
        long double re = 0, im = 0;
        for (int q = 0; q < s.npts; ++q) {
            // exact argument reduction: (k*off) mod n
            long long kk = ((long long)k * s.off[q]) % n;
            if (kk < 0) kk += n;
            const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)kk / (long double)n;
            re += (long double)s.w[q] * cosl(ang);
            im += (long double)s.w[q] * sinl(ang);
        }
        lam[k] = cd{(double)re, (double)im};
    }
    cd** dst = which == 0 ? &c->lamI : &c->lamE;
    if (!*dst) {
        HIPCHK(c, hipMalloc((void**)dst, sizeof(cd) * n));
        c->bytes += sizeof(cd) * n;
    }
    HIPCHK(c, hipMemcpyAsync(*dst, lam.data(), sizeof(cd) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SDC_OK;
}

static inline int grid_for(size_t work, int block);
// context-less datatype operations (ctx == NULL) run on the null stream of the current device
static sdc_ctx* default_ctx() {
    static thread_local sdc_ctx* d = nullptr;
    if (d) return d;
    sdc_ctx* c = new sdc_ctx();
    if (hipMalloc((void**)&c->red, sizeof(unsigned long long) * 16) != hipSuccess ||
        hipHostMalloc((void**)&c->red_host, sizeof(unsigned long long) * 16) != hipSuccess ||
        hipEventCreate(&c->pev0) != hipSuccess || hipEventCreate(&c->pev1) != hipSuccess) {
        g_create_err = "cannot set up the default context (no GPU?)";
        delete c;
        return nullptr;
    }
    d = c;
    return d;
}
#define CTX_OR_DEFAULT(c)                 \
    if (!(c)) {                           \
        (c) = default_ctx();              \
        if (!(c)) return SDC_ERR_HIP;     \
    }

// deterministic synthetic field: prod_d sin(pi*freq_d*x_d) + amp * N(0,1) from a counter-based hash
__device__ inline unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void k_init_field(double* __restrict__ out, int ndim, int n, int f0, int f1, int f2, double amp,
                             unsigned long long seed) {
    const size_t N = ndim == 1 ? (size_t)n : (ndim == 2 ? (size_t)n * n : (size_t)n * n * n);
    const double dx = 1.0 / n, pi = 3.14159265358979323846;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        const int i2 = (int)(i % n);
        const size_t rest = i / n;
        const int i1 = ndim >= 2 ? (int)(rest % n) : 0;
        const int i0 = ndim == 3 ? (int)(rest / n) : 0;
        double v;
        // grid orientation of generic_ND_FD.py:171-180: 2-D x on axis 1, y on axis 0; 3-D x on axis 1, y on
        // axis 0, z on axis 2
        if (ndim == 1) v = sin(pi * f0 * (i2 * dx));
        else if (ndim == 2) v = sin(pi * f0 * (i2 * dx)) * sin(pi * f1 * (i1 * dx));
        else v = sin(pi * f0 * (i1 * dx)) * sin(pi * f1 * (i0 * dx)) * sin(pi * f2 * (i2 * dx));
        if (amp != 0.0) {
            const unsigned long long h1 = splitmix64(seed * 0x100000001B3ull + 2 * i);
            const unsigned long long h2 = splitmix64(seed * 0x100000001B3ull + 2 * i + 1);
            const double u1 = ((double)(h1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            const double u2 = ((double)(h2 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            v += amp * sqrt(-2.0 * log(u1)) * cos(2.0 * pi * u2);
        }
        out[i] = v;
    }
}

extern "C" {

int sdc_version(void) { return 100; }

int sdc_init_field(sdc_ctx* c, double* dst, const int* freq, double amp, unsigned long long seed) {
    if (!c || !dst || !freq) return fail(c, SDC_ERR_PARAM, "null pointer");
    LaunchTimer lt(c, "init_field");
    hipLaunchKernelGGL(k_init_field, dim3(grid_for(c->N, 256)), dim3(256), 0, c->stream, dst, c->ndim, c->n, freq[0],
                       c->ndim > 1 ? freq[1] : 0, c->ndim > 2 ? freq[2] : 0, amp, seed);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

const char* sdc_last_error(const sdc_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

size_t sdc_ctx_bytes(const sdc_ctx* ctx) { return ctx ? ctx->bytes : 0; }

int sdc_ctx_create(sdc_ctx** out, int device, int ndim, int n, int num_nodes, int ncomp, void* stream) {
    if (!out) return fail(nullptr, SDC_ERR_PARAM, "out is null");
    *out = nullptr;
    if (ndim < 1 || ndim > 3) return fail(nullptr, SDC_ERR_PARAM, "can work with up to three dimensions, got %d", ndim);
    if (n < 2 || (n & 1)) return fail(nullptr, SDC_ERR_PARAM, "need an even number of points per dimension, got %d", n);
    if (num_nodes < 1 || num_nodes > MAXM) return fail(nullptr, SDC_ERR_PARAM, "num_nodes must be in 1..%d", MAXM);
    if (ncomp != 1 && ncomp != 2) return fail(nullptr, SDC_ERR_PARAM, "ncomp must be 1 or 2");
    sdc_ctx* c = new sdc_ctx();
    c->device = device;
    c->ndim = ndim;
    c->n = n;
    c->M = num_nodes;
    c->ncomp = ncomp;
    c->stream = (hipStream_t)stream;
    c->N = 1;
    for (int d = 0; d < ndim; ++d) c->N *= (size_t)n;
    c->Nc = ndim == 1 ? c->N : (size_t)(n / 2 + 1) * (c->N / n);
    memset(c->Q, 0, sizeof c->Q);
    memset(c->QI, 0, sizeof c->QI);
    memset(c->QE, 0, sizeof c->QE);
    memset(c->gvals, 0, sizeof c->gvals);
    int rc = [&]() -> int {
        HIPCHK(nullptr, hipSetDevice(device));
        const size_t fb = c->N * sizeof(double);
        HIPCHK(nullptr, hipMalloc((void**)&c->U, fb * (c->M + 1)));
        HIPCHK(nullptr, hipMalloc((void**)&c->F, fb * (c->M + 1) * ncomp));
        HIPCHK(nullptr, hipMalloc((void**)&c->UEND, fb));
        HIPCHK(nullptr, hipMalloc((void**)&c->red, sizeof(unsigned long long) * 16));
        HIPCHK(nullptr, hipHostMalloc((void**)&c->red_host, sizeof(unsigned long long) * 16));
        c->bytes = fb * (c->M + 1) * (1 + ncomp) + fb;
        HIPCHK(nullptr, hipMalloc((void**)&c->counters, sizeof(unsigned long long) * 4));
        HIPCHK(nullptr, hipMalloc((void**)&c->res_dev, sizeof(unsigned long long) * 8));
        HIPCHK(nullptr, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long) * 4, c->stream));
        HIPCHK(nullptr, hipMemsetAsync(c->U, 0, fb * (c->M + 1), c->stream));
        HIPCHK(nullptr, hipMemsetAsync(c->F, 0, fb * (c->M + 1) * ncomp, c->stream));
        HIPCHK(nullptr, hipMemsetAsync(c->UEND, 0, fb, c->stream));
        HIPCHK(nullptr, hipEventCreate(&c->ev0));
        HIPCHK(nullptr, hipEventCreate(&c->ev1));
        HIPCHK(nullptr, hipEventCreate(&c->pev0));
        HIPCHK(nullptr, hipEventCreate(&c->pev1));
        if (is_pow2(n)) {
            std::vector<cd> tw(n);
            for (int m = 0; m < n; ++m) {
                const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
                tw[m] = cd{(double)cosl(ang), (double)(-sinl(ang))};
            }
            HIPCHK(nullptr, hipMalloc((void**)&c->tw, sizeof(cd) * n));
            HIPCHK(nullptr, hipMemcpyAsync(c->tw, tw.data(), sizeof(cd) * n, hipMemcpyHostToDevice, c->stream));
        }
        HIPCHK(nullptr, hipStreamSynchronize(c->stream));
        return SDC_OK;
    }();
    if (rc != SDC_OK) {
        sdc_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return SDC_OK;
}

int sdc_ctx_destroy(sdc_ctx* c) {
    if (!c) return SDC_OK;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->U);
    (void)hipFree(c->F);
    (void)hipFree(c->TAU);
    (void)hipFree(c->UEND);
    (void)hipFree(c->W);
    (void)hipFree(c->S);
    (void)hipFree(c->S0);
    (void)hipFree(c->tw);
    (void)hipFree(c->lamI);
    (void)hipFree(c->lamE);
    (void)hipFree(c->profile);
    (void)hipFree(c->red);
    (void)hipFree(c->counters);
    (void)hipFree(c->res_dev);
    if (c->red_host) (void)hipHostFree(c->red_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->pev0) (void)hipEventDestroy(c->pev0);
    if (c->pev1) (void)hipEventDestroy(c->pev1);
    for (auto& e : c->pool) (void)hipEventDestroy(e);
    delete c;
    return SDC_OK;
}

int sdc_set_coeffs(sdc_ctx* c, const double* Qmat, const double* QI, const double* QE, const double* nodes,
                   const double* weights) {
    if (!c || !Qmat || !QI || !nodes || !weights) return fail(c, SDC_ERR_PARAM, "null coefficient pointer");
    if (c->ncomp == 2 && !QE) return fail(c, SDC_ERR_PARAM, "QE required for the IMEX sweeper");
    const int M1 = c->M + 1;
    for (int i = 0; i < M1; ++i)
        for (int j = 0; j < M1; ++j) {
            c->Q[i][j] = Qmat[i * M1 + j];
            c->QI[i][j] = QI[i * M1 + j];
            c->QE[i][j] = QE ? QE[i * M1 + j] : 0.0;
            if (j > i && c->QI[i][j] != 0.0) return fail(c, SDC_ERR_PARAM, "Lower triangular matrix expected!");
            if (j >= i && j > 0 && c->QE[i][j] != 0.0)
                return fail(c, SDC_ERR_PARAM, "Strictly lower triangular matrix expected!");
        }
    for (int m = 0; m < c->M; ++m) {
        c->nodes[m] = nodes[m];
        c->weights[m] = weights[m];
    }
    c->have_coeffs = true;
    c->res_valid = false;
    return SDC_OK;
}

int sdc_set_stencil(sdc_ctx* c, int which, int npts, const int* offsets, const double* weights) {
    if (!c || which < 0 || which > 1 || npts < 1 || npts > MAXSTEN || !offsets || !weights)
        return fail(c, SDC_ERR_PARAM, "bad stencil (npts must be 1..%d)", MAXSTEN);
    Stencil& s = c->st[which];
    s.npts = npts;
    for (int k = 0; k < npts; ++k) {
        if (offsets[k] <= -c->n || offsets[k] >= c->n) return fail(c, SDC_ERR_PARAM, "stencil offset exceeds grid");
        s.off[k] = offsets[k];
        s.w[k] = weights[k];
    }
    c->have_stencil[which] = true;
    if (which == 1) c->expl_kind = SDC_EXPL_STENCIL;
    return build_symbol(c, which);
}

int sdc_set_problem_vdp(sdc_ctx* c, double mu, double newton_tol, int newton_maxiter) {
    if (!c) return SDC_ERR_PARAM;
    if (c->ndim != 1 || c->ncomp != 1)
        return fail(c, SDC_ERR_PARAM, "van der Pol ensemble needs a 1-D context with n = 2 * ntraj and ncomp = 1");
    c->kind = 1;
    c->vdp_mu = mu;
    c->vdp_tol = newton_tol;
    c->vdp_maxiter = newton_maxiter;
    return SDC_OK;
}

int sdc_work_counters(sdc_ctx* c, unsigned long long* out) {
    if (!c || !out) return SDC_ERR_PARAM;
    HIPCHK(c, hipMemcpyAsync(c->red_host + 12, c->counters, sizeof(unsigned long long) * 3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) out[k] = c->red_host[12 + k];
    out[1] += c->rhs_host;
    return SDC_OK;
}

static int vdp_check_failures(sdc_ctx* c) {
    unsigned long long v[3];
    int rc = sdc_work_counters(c, v);
    if (rc != SDC_OK) return rc;
    if (v[2] != 0) {
        HIPCHK(c, hipMemsetAsync(c->counters + 2, 0, sizeof(unsigned long long), c->stream));
        return fail(c, SDC_ERR_NEWTON, "Newton did not converge after %d iterations (or got nan) for %llu solves",
                    c->vdp_maxiter, v[2]);
    }
    return SDC_OK;
}

int sdc_set_symbol(sdc_ctx* c, int which, const double* table) {
    if (!c || which < 0 || which > 1 || !table) return fail(c, SDC_ERR_PARAM, "bad symbol table");
    cd** dst = which == 0 ? &c->lamI : &c->lamE;
    if (!*dst) {
        HIPCHK(c, hipMalloc((void**)dst, sizeof(cd) * c->n));
        c->bytes += sizeof(cd) * c->n;
    }
    HIPCHK(c, hipMemcpyAsync(*dst, table, sizeof(cd) * c->n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_stencil[which] = true;
    if (which == 0) c->spectral_op = true;
    c->spec_valid = c->spec0_valid = false;
    return SDC_OK;
}

int sdc_set_reaction(sdc_ctx* c, int kind, double p0, double p1, int nu) {
    if (!c || kind < 1 || kind > 2 || nu < 1 || nu > 16) return fail(c, SDC_ERR_PARAM, "bad reaction term");
    if (c->ncomp != 2) return fail(c, SDC_ERR_PARAM, "explicit part needs ncomp == 2");
    c->react_kind = kind;
    c->react_p0 = p0;
    c->react_p1 = p1;
    c->react_nu = nu;
    c->expl_kind = SDC_EXPL_REACTION;
    return SDC_OK;
}

int sdc_set_expl_kind(sdc_ctx* c, int kind) {
    if (!c || kind < 0 || kind > 3) return fail(c, SDC_ERR_PARAM, "bad explicit kind");
    if (kind != SDC_EXPL_NONE && c->ncomp != 2) return fail(c, SDC_ERR_PARAM, "explicit part needs ncomp == 2");
    c->expl_kind = kind;
    return SDC_OK;
}

int sdc_set_forcing_profile(sdc_ctx* c, const double* host_profile) {
    if (!c || !host_profile) return fail(c, SDC_ERR_PARAM, "null profile");
    if (c->ncomp != 2) return fail(c, SDC_ERR_PARAM, "forcing needs ncomp == 2");
    if (!c->profile) {
        HIPCHK(c, hipMalloc((void**)&c->profile, c->N * sizeof(double)));
        c->bytes += c->N * sizeof(double);
    }
    HIPCHK(c, hipMemcpyAsync(c->profile, host_profile, c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->expl_kind = SDC_EXPL_FORCING;
    return SDC_OK;
}

int sdc_set_forcing_values(sdc_ctx* c, const double* g) {
    if (!c || !g) return fail(c, SDC_ERR_PARAM, "null forcing values");
    for (int m = 0; m <= c->M; ++m) c->gvals[m] = g[m];
    return SDC_OK;
}

extern "C" int sdc_invalidate_spectra(sdc_ctx* c, int which);
extern "C" int sdc_solve(sdc_ctx* c, const double* rhs, double factor, const double* guess, double* out);
extern "C" int sdc_eval_f(sdc_ctx* c, const double* u, double g_t, double* f_impl, double* f_expl);

static int ensure_work(sdc_ctx* c) {
    if (!c->W) {
        HIPCHK(c, hipMalloc((void**)&c->W, sizeof(cd) * c->Nc * c->M));
        c->bytes += sizeof(cd) * c->Nc * c->M;
    }
    return SDC_OK;
}

static int ensure_tau(sdc_ctx* c) {
    if (!c->TAU) {
        HIPCHK(c, hipMalloc((void**)&c->TAU, c->N * sizeof(double) * c->M));
        HIPCHK(c, hipMemsetAsync(c->TAU, 0, c->N * sizeof(double) * c->M, c->stream));
        c->bytes += c->N * sizeof(double) * c->M;
    }
    return SDC_OK;
}

void* sdc_slot_ptr(sdc_ctx* c, int slot, int m, int comp) {
    if (!c) return nullptr;
    if (slot == SDC_SLOT_TAU && ensure_tau(c) != SDC_OK) return nullptr;
    if (slot == SDC_SLOT_WORK) return c->W;
    return slot_ptr(c, slot, m, comp);
}

int sdc_invalidate_spectra(sdc_ctx* c, int which) {
    if (!c) return SDC_ERR_PARAM;
    c->res_valid = false;
    c->res_spread = false;
    if (which & 1) {
        c->spec0_valid = false;
        c->spec_spread = false;  // "all nodes equal U[0]" no longer holds for the new U[0]
    }
    if (which & 2) {
        c->spec_valid = false;
        c->spec_spread = false;
    }
    if (which & 4) c->force_gather = true;  // some F[m >= 1] no longer equals f(U[m]): gather on F itself
    return SDC_OK;
}

int sdc_set_fused_residual(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    c->fuse_residual = on != 0;
    c->res_valid = false;
    return SDC_OK;
}

int sdc_set_unlocked(sdc_ctx* c, int unlocked) {
    if (!c) return SDC_ERR_PARAM;
    c->unlocked = unlocked != 0;
    return SDC_OK;
}

int sdc_set_spectral_reuse(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    c->reuse = on != 0;
    c->spec_valid = c->spec0_valid = c->spec_spread = false;
    return SDC_OK;
}

int sdc_set_tau_active(sdc_ctx* c, int active) {
    if (!c) return SDC_ERR_PARAM;
    if (active) {
        int rc = ensure_tau(c);
        if (rc != SDC_OK) return rc;
    }
    c->tau_active = active != 0;
    c->res_valid = false;
    c->res_spread = false;
    return SDC_OK;
}

int sdc_upload(sdc_ctx* c, int slot, int m, int comp, const double* host) {
    if (!c || !host) return fail(c, SDC_ERR_PARAM, "null pointer");
    double* d = (double*)sdc_slot_ptr(c, slot, m, comp);
    if (!d) return fail(c, SDC_ERR_PARAM, "bad slot (%d, %d, %d)", slot, m, comp);
    HIPCHK(c, hipMemcpyAsync(d, host, c->N * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->res_valid = false;
    c->res_spread = false;
    if (slot == SDC_SLOT_U) sdc_invalidate_spectra(c, m == 0 ? 1 : 2);
    return SDC_OK;
}

int sdc_download(sdc_ctx* c, int slot, int m, int comp, double* host) {
    if (!c || !host) return fail(c, SDC_ERR_PARAM, "null pointer");
    double* d = (double*)sdc_slot_ptr(c, slot, m, comp);
    if (!d) return fail(c, SDC_ERR_PARAM, "bad slot (%d, %d, %d)", slot, m, comp);
    HIPCHK(c, hipMemcpyAsync(host, d, c->N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SDC_OK;
}

int sdc_eval_f(sdc_ctx* c, const double* u, double g_t, double* f_impl, double* f_expl) {
    if (!c || !u || !f_impl) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (c->kind == 1) {
        LaunchTimer lt(c, "vdp_eval");
        hipLaunchKernelGGL(k_vdp_eval, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, u, f_impl, c->N / 2,
                           c->vdp_mu, c->counters);
        HIPCHK(c, hipGetLastError());
        return SDC_OK;
    }
    if (!c->have_stencil[0]) return fail(c, SDC_ERR_STATE, "implicit operator not set (sdc_set_stencil)");
    if (c->expl_kind == SDC_EXPL_FORCING && !c->profile) return fail(c, SDC_ERR_STATE, "forcing profile not set");
    if (c->spectral_op || c->expl_kind == SDC_EXPL_REACTION) {
        if (c->spectral_op) {
            FieldPtrs p;
            memset(&p, 0, sizeof p);
            ZArgs z;
            memset(&z, 0, sizeof z);
            p.in[0] = u;
            p.out[0] = f_impl;
            z.apply = 1;
            int rc0 = fft_pipeline(c, 1, p, z);
            if (rc0 != SDC_OK) return rc0;
        } else {
            const double* in1[1] = {u};
            double* oi1[1] = {f_impl};
            int rc0 = run_stencil(c, 1, in1, oi1, nullptr, nullptr);
            if (rc0 != SDC_OK) return rc0;
        }
        if (f_expl && c->expl_kind == SDC_EXPL_REACTION) {
            LaunchTimer lt(c, "reaction");
            hipLaunchKernelGGL(k_reaction, dim3(grid_for(c->N, 256)), dim3(256), 0, c->stream, u, f_expl, c->N,
                               c->react_kind, c->react_p0, c->react_p1, c->react_nu);
            HIPCHK(c, hipGetLastError());
        } else if (f_expl && c->expl_kind == SDC_EXPL_STENCIL) {
            return fail(c, SDC_ERR_UNSUPPORTED, "explicit stencil together with a spectral implicit operator");
        }
        return SDC_OK;
    }
    const double* in[1] = {u};
    double* oi[1] = {f_impl};
    double* oe[1] = {f_expl};
    double g[1] = {g_t};
    return run_stencil(c, 1, in, oi, f_expl ? oe : nullptr, g);
}

int sdc_predict(sdc_ctx* c, double t, double dt, int guess, double fill_u, double fill_f) {
    (void)t;
    (void)dt;
    if (!c) return SDC_ERR_PARAM;
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    if (guess < 0 || guess > 3) return fail(c, SDC_ERR_PARAM, "initial_guess option %d not implemented", guess);
    int rc = sdc_eval_f(c, c->U, c->gvals[0], c->F, c->ncomp == 2 ? c->F + c->N : nullptr);
    if (rc != SDC_OK) return rc;
    SpreadArgs a;
    memset(&a, 0, sizeof a);
    a.u0 = c->U;
    a.f0 = c->F;
    a.profile = c->profile;
    a.U = c->U;
    a.F = c->F;
    a.N = c->N;
    a.M = c->M;
    a.ncomp = c->ncomp;
    a.guess = guess;
    a.forcing = c->expl_kind == SDC_EXPL_FORCING;
    a.fill_u = fill_u;
    a.fill_f = fill_f;
    for (int m = 0; m <= c->M; ++m) a.g[m] = c->gvals[m];
    // all nodes equal u0 and f does not depend on t: every f_j equals f(u0), so the node residuals are
    // dt * |sum_j Q[m][j]| * max|f(u0)| and the fill kernel can reduce max|f(u0)| on the way
    const bool spread_res = guess == SDC_GUESS_SPREAD && c->expl_kind != SDC_EXPL_FORCING && !c->tau_active;
    if (spread_res) {
        HIPCHK(c, hipMemsetAsync(c->res_dev, 0, sizeof(unsigned long long) * 8, c->stream));
        a.f0max = c->res_dev + 7;
    }
    {
        LaunchTimer lt(c, "spread");
        hipLaunchKernelGGL(k_spread, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    // 'spread' evaluates f at every node in the reference (core/sweeper.py:142-143); the engine copies F[0]
    if (c->kind == 1 && guess == SDC_GUESS_SPREAD) c->rhs_host += (unsigned long long)c->M * (c->N / 2);
    c->unlocked = true;
    c->res_valid = false;
    c->res_spread = spread_res;
    c->spec_valid = false;
    c->spec_spread = (guess == SDC_GUESS_SPREAD || guess == SDC_GUESS_COPY);  // all nodes equal U[0]
    return SDC_OK;
}

// Node-by-node sweep on the device for right-hand sides that are not linear in u (pointwise reaction terms) or
// whose implicit operator is given by its symbol only: the reference's loop (imex_1st_order.py:57-108 /
// generic_implicit.py:51-103) with every step a kernel on the stream - gather for all nodes, then per node
// right-hand side, solve (FFT pipeline), f evaluation (operator by FFT or stencil + explicit part).
static int sweep_nodewise(sdc_ctx* c, double dt) {
    const int M = c->M;
    const bool imex = c->ncomp == 2;
    QuadArgs q;
    quad_base(c, q);
    q.u0 = c->U;
    q.tau = c->tau_active ? c->TAU : nullptr;
    for (int m = 0; m < M; ++m) {
        q.out[m] = c->U + (size_t)(m + 1) * c->N;  // the old iterate is only a solver guess: reuse its storage
        for (int j = 0; j < M; ++j) {
            q.cI[m][j] = dt * (c->Q[m + 1][j + 1] - c->QI[m + 1][j + 1]);
            q.cE[m][j] = dt * (c->Q[m + 1][j + 1] - c->QE[m + 1][j + 1]);
        }
    }
    int rc = launch_quad<0>(c, q, "gather");
    if (rc != SDC_OK) return rc;
    c->spec_valid = c->spec_spread = false;
    for (int m = 0; m < M; ++m) {
        double* um = c->U + (size_t)(m + 1) * c->N;
        if (m > 0) {
            LinArgs la;
            memset(&la, 0, sizeof la);
            la.out = um;
            la.base = um;
            la.n = c->N;
            for (int j = 0; j < m; ++j) {
                const double ci = dt * c->QI[m + 1][j + 1], ce = dt * c->QE[m + 1][j + 1];
                if (ci != 0.0) {
                    la.x[la.nterms] = c->F + ((size_t)(j + 1) * c->ncomp) * c->N;
                    la.c[la.nterms++] = ci;
                }
                if (imex && ce != 0.0) {
                    la.x[la.nterms] = c->F + ((size_t)(j + 1) * c->ncomp + 1) * c->N;
                    la.c[la.nterms++] = ce;
                }
            }
            if (la.nterms > 0) {
                LaunchTimer lt(c, "node_rhs");
                hipLaunchKernelGGL(k_lincomb, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, la);
                HIPCHK(c, hipGetLastError());
            }
        }
        const double alpha = dt * c->QI[m + 1][m + 1];
        if (alpha != 0.0) {
            rc = sdc_solve(c, um, alpha, um, um);
            if (rc != SDC_OK) return rc;
        }
        rc = sdc_eval_f(c, um, c->gvals[m + 1], c->F + ((size_t)(m + 1) * c->ncomp) * c->N,
                        imex ? c->F + ((size_t)(m + 1) * c->ncomp + 1) * c->N : nullptr);
        if (rc != SDC_OK) return rc;
    }
    return SDC_OK;
}

int sdc_sweep(sdc_ctx* c, double t, double dt) {
    (void)t;
    if (!c) return SDC_ERR_PARAM;
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    if (!c->unlocked) return fail(c, SDC_ERR_STATE, "level is locked: predict first (assert L.status.unlocked)");
    const int M = c->M;
    c->res_valid = false;
    c->res_spread = false;
    if (c->kind == 1) {
        VdpSweepArgs a;
        memset(&a, 0, sizeof a);
        a.U = c->U;
        a.F = c->F;
        a.tau = c->tau_active ? c->TAU : nullptr;
        a.T = c->N / 2;
        a.mu = c->vdp_mu;
        a.dt = dt;
        a.tol = c->vdp_tol;
        a.maxiter = c->vdp_maxiter;
        a.counters = c->counters;
        for (int m = 0; m < M; ++m)
            for (int j = 0; j < M; ++j) {
                a.Q[m][j] = c->Q[m + 1][j + 1];
                a.QI[m][j] = c->QI[m + 1][j + 1];
            }
        const int grid = grid_for(a.T, 256);
        {
            LaunchTimer lt(c, "vdp_sweep");
#define VCASE(MM) \
    case MM: hipLaunchKernelGGL((k_vdp_sweep<MM>), dim3(grid), dim3(256), 0, c->stream, a); break;
            switch (M) {
                VCASE(1) VCASE(2) VCASE(3) VCASE(4) VCASE(5) VCASE(6) VCASE(7) VCASE(8)
            }
#undef VCASE
        }
        HIPCHK(c, hipGetLastError());
        return vdp_check_failures(c);
    }
    if (c->expl_kind == SDC_EXPL_STENCIL && !c->have_stencil[1])
        return fail(c, SDC_ERR_STATE, "explicit operator not set (sdc_set_stencil which=1)");
    if (c->expl_kind == SDC_EXPL_REACTION || c->spectral_op) return sweep_nodewise(c, dt);
    const bool gather_once = c->force_gather;
    c->force_gather = false;
    if (c->reuse && !gather_once && !c->tau_active && c->expl_kind != SDC_EXPL_FORCING && c->have_stencil[0] &&
        is_pow2(c->n) && (c->n <= 1024 || (c->n == 2048 && c->ndim == 1))) {
        // ---- spectral reuse: f is linear in u, so the gather happens on the cached transforms ----
        if (!c->S) {
            HIPCHK(c, hipMalloc((void**)&c->S, sizeof(cd) * c->Nc * M));
            HIPCHK(c, hipMalloc((void**)&c->S0, sizeof(cd) * c->Nc));
            c->bytes += sizeof(cd) * c->Nc * (M + 1);
            c->spec_valid = c->spec0_valid = false;
        }
        FieldPtrs p;
        memset(&p, 0, sizeof p);
        if (!c->spec0_valid) {
            p.in[0] = c->U;
            int rc0 = fwd_transform(c, 1, p, c->S0, 0);
            if (rc0 != SDC_OK) return rc0;
            c->spec0_valid = true;
        }
        if (!c->spec_valid && !c->spec_spread) {
            for (int m = 0; m < M; ++m) p.in[m] = c->U + (size_t)(m + 1) * c->N;
            int rc0 = fwd_transform(c, M, p, c->S, c->Nc);
            if (rc0 != SDC_OK) return rc0;
            c->spec_valid = true;
        }
        SpecArgs a;
        memset(&a, 0, sizeof a);
        a.S = c->S;
        a.fstride = c->Nc;
        a.S0 = c->S0;
        a.W = c->W;
        a.tw = c->tw;
        a.lamI = c->lamI;
        a.lamE = c->expl_kind == SDC_EXPL_STENCIL ? c->lamE : nullptr;
        a.invN = 1.0 / (double)c->N;
        a.nf = M;
        a.ndim = c->ndim;
        a.spread = (!c->spec_valid && c->spec_spread) ? 1 : 0;
        bool coupled = false;
        for (int m = 0; m < M; ++m) {
            p.out[m] = c->U + (size_t)(m + 1) * c->N;
            a.alpha[m] = dt * c->QI[m + 1][m + 1];
            for (int j = 0; j < M; ++j) {
                a.gI[m][j] = dt * (c->Q[m + 1][j + 1] - c->QI[m + 1][j + 1]);
                a.gE[m][j] = a.lamE ? dt * (c->Q[m + 1][j + 1] - c->QE[m + 1][j + 1]) : 0.0;
                if (j < m) {
                    a.cI[m][j] = dt * c->QI[m + 1][j + 1];
                    a.cE[m][j] = a.lamE ? dt * c->QE[m + 1][j + 1] : 0.0;
                    if (a.cI[m][j] != 0.0 || a.cE[m][j] != 0.0) coupled = true;
                }
            }
        }
        a.coupled = coupled;
        int rc0 = spec_sweep(c, M, a, p);
        if (rc0 != SDC_OK) return rc0;
        c->spec_valid = true;
        c->spec_spread = false;
        return eval_nodes(c, dt);
    }
    c->spec_valid = false;
    c->spec_spread = false;
    // 1. gather u0 + dt (Q - QI) F_impl + dt (Q - QE) F_expl (+ tau) for all nodes into U[1..M]
    QuadArgs q;
    quad_base(c, q);
    q.u0 = c->U;
    q.tau = c->tau_active ? c->TAU : nullptr;
    const bool forcing = c->expl_kind == SDC_EXPL_FORCING;
    for (int m = 0; m < M; ++m) {
        q.out[m] = c->U + (size_t)(m + 1) * c->N;
        for (int j = 0; j < M; ++j) {
            q.cI[m][j] = dt * (c->Q[m + 1][j + 1] - c->QI[m + 1][j + 1]);
            // u-independent forcing: the new explicit values equal the old ones, so the strictly lower
            // QE add-back of imex_1st_order.py:94 folds into the gather
            q.cE[m][j] = forcing ? dt * c->Q[m + 1][j + 1] : dt * (c->Q[m + 1][j + 1] - c->QE[m + 1][j + 1]);
        }
    }
    int rc = launch_quad<0>(c, q, "gather");
    if (rc != SDC_OK) return rc;
    // 2. node-coupled spectral solve
    FieldPtrs p;
    memset(&p, 0, sizeof p);
    ZArgs z;
    memset(&z, 0, sizeof z);
    bool coupled = false;
    for (int m = 0; m < M; ++m) {
        p.in[m] = c->U + (size_t)(m + 1) * c->N;
        p.out[m] = c->U + (size_t)(m + 1) * c->N;
        z.alpha[m] = dt * c->QI[m + 1][m + 1];
        for (int j = 0; j < m; ++j) {
            z.cI[m][j] = dt * c->QI[m + 1][j + 1];
            z.cE[m][j] = (c->expl_kind == SDC_EXPL_STENCIL) ? dt * c->QE[m + 1][j + 1] : 0.0;
            if (z.cI[m][j] != 0.0 || z.cE[m][j] != 0.0) coupled = true;
        }
    }
    z.coupled = coupled;
    z.lamE = c->expl_kind == SDC_EXPL_STENCIL ? c->lamE : nullptr;
    rc = fft_pipeline(c, M, p, z);
    if (rc != SDC_OK) return rc;
    // 3. F[m] = f(U[m]) for the new values (+ the residual when it fuses)
    return eval_nodes(c, dt);
}

int sdc_solve(sdc_ctx* c, const double* rhs, double factor, const double* guess, double* out) {
    if (!c || !rhs || !out) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (c->kind == 1) {
        if (!guess) return fail(c, SDC_ERR_PARAM, "the Newton solver needs an initial guess");
        {
            LaunchTimer lt(c, "vdp_solve");
            hipLaunchKernelGGL(k_vdp_solve, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, rhs, guess, out,
                               c->N / 2, factor, c->vdp_mu, c->vdp_tol, c->vdp_maxiter, c->counters);
        }
        HIPCHK(c, hipGetLastError());
        return vdp_check_failures(c);
    }
    FieldPtrs p;
    memset(&p, 0, sizeof p);
    ZArgs z;
    memset(&z, 0, sizeof z);
    p.in[0] = rhs;
    p.out[0] = out;
    z.alpha[0] = factor;
    z.coupled = 0;
    z.lamE = nullptr;
    return fft_pipeline(c, 1, p, z);
}

int sdc_residual(sdc_ctx* c, double dt, int type, double* node_norms, double* residual) {
    if (!c || !residual) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    if (type < 0 || type > 3)
        return fail(c, SDC_ERR_PARAM,
                    "residual_type = %d not implemented, choose full_abs, last_abs, full_rel or last_rel instead", type);
    const int M = c->M;
    HIPCHK(c, hipMemsetAsync(c->red, 0, sizeof(unsigned long long) * 16, c->stream));
    bool from_spread = false;
    if (c->res_valid && c->res_dt == dt) {
        // the sweep's fused eval_f kernel already reduced the node norms of this very state
        HIPCHK(c, hipMemcpyAsync(c->red, c->res_dev, sizeof(unsigned long long) * 8, hipMemcpyDeviceToDevice, c->stream));
    } else if (c->res_spread) {
        HIPCHK(c, hipMemcpyAsync(c->red + 7, c->res_dev + 7, sizeof(unsigned long long), hipMemcpyDeviceToDevice, c->stream));
        from_spread = true;
    } else {
        QuadArgs q;
        quad_base(c, q);
        q.u0 = c->U;
        q.tau = c->tau_active ? c->TAU : nullptr;
        q.Usub = c->U;
        q.norms = c->red;
        for (int m = 0; m < M; ++m)
            for (int j = 0; j < M; ++j) q.cI[m][j] = q.cE[m][j] = dt * c->Q[m + 1][j + 1];
        int rc = launch_quad<1>(c, q, "residual");
        if (rc != SDC_OK) return rc;
    }
    if (type >= SDC_RES_FULL_REL) {
        LaunchTimer lt(c, "amax");
        hipLaunchKernelGGL(k_amax, dim3(grid_for(c->N, 256)), dim3(256), 0, c->stream, c->U, c->N, c->red + 8);
    }
    HIPCHK(c, hipMemcpyAsync(c->red_host, c->red, sizeof(unsigned long long) * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double norms[MAXM], mx = 0.0;
    double f0max = 0.0;
    memcpy(&f0max, &c->red_host[7], sizeof(double));
    for (int m = 0; m < M; ++m) {
        memcpy(&norms[m], &c->red_host[m], sizeof(double));
        if (from_spread) {
            double sq = 0.0;
            for (int j = 1; j <= M; ++j) sq += dt * c->Q[m + 1][j];
            norms[m] = fabs(sq) * f0max;
        }
        if (node_norms) node_norms[m] = norms[m];
        mx = (norms[m] > mx || norms[m] != norms[m]) ? norms[m] : mx;
    }
    double u0n;
    memcpy(&u0n, &c->red_host[8], sizeof(double));
    switch (type) {
        case SDC_RES_FULL_ABS: *residual = mx; break;
        case SDC_RES_LAST_ABS: *residual = norms[M - 1]; break;
        case SDC_RES_FULL_REL: *residual = mx / u0n; break;
        default: *residual = norms[M - 1] / u0n; break;
    }
    return SDC_OK;
}

int sdc_end_point(sdc_ctx* c, double dt, int do_coll_update) {
    if (!c) return SDC_ERR_PARAM;
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    if (!do_coll_update) return sdc_vec_copy(c, c->N, c->U + (size_t)c->M * c->N, c->UEND);
    QuadArgs q;
    quad_base(c, q);
    q.u0 = c->U;
    q.tau = c->tau_active ? c->TAU : nullptr;
    q.tau_row0 = c->M - 1;
    q.nout = 1;
    q.out[0] = c->UEND;
    for (int j = 0; j < c->M; ++j) q.cI[0][j] = q.cE[0][j] = dt * c->weights[j];
    return launch_quad<0>(c, q, "end_point");
}

int sdc_integrate(sdc_ctx* c, double dt, double* const* dst) {
    if (!c || !dst) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    QuadArgs q;
    quad_base(c, q);
    for (int m = 0; m < c->M; ++m) {
        if (!dst[m]) return fail(c, SDC_ERR_PARAM, "null destination %d", m);
        q.out[m] = dst[m];
        for (int j = 0; j < c->M; ++j) q.cI[m][j] = q.cE[m][j] = dt * c->Q[m + 1][j + 1];
    }
    return launch_quad<0>(c, q, "integrate");
}

int sdc_vec_copy(sdc_ctx* c, size_t n, const double* x, double* y) {
    CTX_OR_DEFAULT(c);
    if (!x || !y) return fail(c, SDC_ERR_PARAM, "null pointer");
    LaunchTimer lt(c, "copy");
    HIPCHK(c, hipMemcpyAsync(y, x, n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return SDC_OK;
}

int sdc_vec_fill(sdc_ctx* c, size_t n, double a, double* y) {
    CTX_OR_DEFAULT(c);
    if (!y) return fail(c, SDC_ERR_PARAM, "null pointer");
    LaunchTimer lt(c, "fill");
    hipLaunchKernelGGL(k_fill, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, n, a, y);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

int sdc_vec_axpby(sdc_ctx* c, size_t n, double a, const double* x, double b, const double* y, double* z) {
    CTX_OR_DEFAULT(c);
    if (!z) return fail(c, SDC_ERR_PARAM, "null pointer");
    LaunchTimer lt(c, "axpby");
    hipLaunchKernelGGL(k_axpby, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, n, a, x, b, y, z);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

int sdc_vec_amax(sdc_ctx* c, size_t n, const double* x, double* out) {
    CTX_OR_DEFAULT(c);
    if (!x || !out) return fail(c, SDC_ERR_PARAM, "null pointer");
    HIPCHK(c, hipMemsetAsync(c->red + 9, 0, sizeof(unsigned long long), c->stream));
    {
        LaunchTimer lt(c, "amax");
        hipLaunchKernelGGL(k_amax, dim3(grid_for(n, 256)), dim3(256), 0, c->stream, x, n, c->red + 9);
    }
    HIPCHK(c, hipMemcpyAsync(c->red_host + 9, c->red + 9, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(out, &c->red_host[9], sizeof(double));
    return SDC_OK;
}

int sdc_transfer_apply(void* stream, int ndim, int n_out, int n_in, int width, const int* idx, const double* w,
                       const double* in, double* out) {
    if (ndim < 1 || ndim > 3 || n_out < 1 || n_in < 1 || width < 1 || !idx || !w || !in || !out)
        return fail(nullptr, SDC_ERR_PARAM, "bad transfer arguments");
    // separable: one pass per axis (cost ~ ndim * width per point instead of width^ndim), last axis first so that
    // intermediate fields stay as small as possible when prolonging; two scratch fields ping-pong in between
    static thread_local double* scratch[2] = {nullptr, nullptr};
    static thread_local size_t scratch_len = 0;
    const int nmax = n_out > n_in ? n_out : n_in;
    size_t need = 1;
    for (int d = 0; d < ndim; ++d) need *= (size_t)nmax;
    if (ndim > 1 && need > scratch_len) {
        for (int k = 0; k < 2; ++k) {
            if (scratch[k]) (void)hipFree(scratch[k]);
            if (hipMalloc((void**)&scratch[k], need * sizeof(double)) != hipSuccess) {
                scratch[k] = nullptr;
                scratch_len = 0;
                return fail(nullptr, SDC_ERR_NOMEM, "transfer scratch allocation failed");
            }
        }
        scratch_len = need;
    }
    int dims[3] = {n_in, n_in, n_in};
    const double* src = in;
    for (int pass = 0; pass < ndim; ++pass) {
        const int axis = ndim - 1 - pass;
        XferArgs a;
        a.idx = idx;
        a.w = w;
        a.W = width;
        a.n_in = n_in;
        a.n_out = n_out;
        a.outer = 1;
        a.inner = 1;
        for (int d = 0; d < axis; ++d) a.outer *= (size_t)dims[d];
        for (int d = axis + 1; d < ndim; ++d) a.inner *= (size_t)dims[d];
        a.in = src;
        a.out = pass == ndim - 1 ? out : scratch[pass & 1];
        const size_t total = a.outer * (size_t)n_out * a.inner;
        hipLaunchKernelGGL(k_xfer_axis, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
        dims[axis] = n_out;
        src = a.out;
    }
    HIPCHK(nullptr, hipGetLastError());
    return SDC_OK;
}

int sdc_odd_mirror(sdc_ctx* c, double* field, int n_interior) {
    CTX_OR_DEFAULT(c);
    if (!field || n_interior < 1) return fail(c, SDC_ERR_PARAM, "bad odd-extension arguments");
    hipLaunchKernelGGL(k_odd_mirror, dim3((n_interior + 255) / 256), dim3(256), 0, c->stream, field, n_interior);
    HIPCHK(c, hipGetLastError());
    c->res_valid = false;
    return SDC_OK;
}

int sdc_sync(sdc_ctx* c) {
    if (!c) return SDC_ERR_PARAM;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return SDC_OK;
}

int sdc_timer_begin(sdc_ctx* c) {
    if (!c) return SDC_ERR_PARAM;
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    return SDC_OK;
}

int sdc_timer_end(sdc_ctx* c, double* ms) {
    if (!c || !ms) return SDC_ERR_PARAM;
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float f = 0;
    HIPCHK(c, hipEventElapsedTime(&f, c->ev0, c->ev1));
    *ms = f;
    return SDC_OK;
}

int sdc_profile_enable(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    if (c->profiling) prof_flush(c);
    c->profiling = on != 0;
    if (on) c->prof.clear();
    return SDC_OK;
}

int sdc_profile_read(sdc_ctx* c, int cap, const char** names, double* ms, int* calls, int* count) {
    if (!c || !count) return SDC_ERR_PARAM;
    prof_flush(c);
    c->prof_names.clear();
    int i = 0;
    for (auto& kv : c->prof) {
        if (i < cap) {
            c->prof_names.push_back(kv.first);
            if (ms) ms[i] = kv.second.ms;
            if (calls) calls[i] = kv.second.calls;
        }
        ++i;
    }
    for (int k = 0; k < (int)c->prof_names.size(); ++k)
        if (names) names[k] = c->prof_names[k].c_str();
    *count = i < cap ? i : cap;
    return SDC_OK;
}

}  // extern "C"
