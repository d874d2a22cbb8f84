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
#ifndef SDC_T1024
#define SDC_T1024 8
#endif
#ifndef SDC_XFER_CH
#define SDC_XFER_CH 8
#endif
#ifndef SDC_FUSE_SPECZ
#define SDC_FUSE_SPECZ 1
#endif
#ifndef SDC_SPECZ_PAIRS
#define SDC_SPECZ_PAIRS 1
#endif
#ifndef SDC_FUSE_NODE_RHS
#define SDC_FUSE_NODE_RHS 1   // node-by-node sweeps on symbol operators: the terms of earlier nodes ride on the forward pass
#endif
#ifndef SDC_SPEC_GRID
#define SDC_SPEC_GRID 4096
#endif
#include "context.hpp"
#include "kernels_pointwise.hpp"
#include "kernels_stencil.hpp"
#include "kernels_fft.hpp"
#include "kernels_vdp.hpp"
#include "kernels_transfer.hpp"

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
static int ensure_work(sdc_ctx* c);
// work spectra a context holds: one per node - in 1-D (lines: next to nothing) at least three, so that a problem whose two
// parts are given by symbols gets its solution and both parts out of one transform whatever the number of nodes
static inline int work_fields(const sdc_ctx* c) { return c->ndim == 1 && c->M < 3 ? 3 : c->M; }
// complex columns per tile of the strided passes (x and y): 128-byte row segments up to N = 1024, 64-byte ones at 2048
#ifndef SDC_TSMALL
#define SDC_TSMALL 16   // ... of lines of up to 256 modes: 256-byte row segments (round 6, config 5: +1 - 2 %; 32: the same; 4: -22 %)
#endif
#ifndef SDC_T512
#define SDC_T512 8
#endif
template <int N>
constexpr int strided_cols() { return N >= 2048 ? 4 : (N == 1024 ? SDC_T1024 : (N == 512 ? SDC_T512 : (N <= 256 && (N & (N - 1)) == 0 ? SDC_TSMALL : 8))); }
// the FFT kernels transform lines of length 2^p <= 1024 (2048 in 1-D and 2-D), 3 * 2^p from 24 to 768 and 5 * 2^p from 40 to 640
// (fft.hpp)
static inline bool fourier_ok(const sdc_ctx* c) {
    if (c->n % 3 == 0) return fft_length_ok(c->n) && c->n <= 768;
    if (c->n % 5 == 0) return fft_length_ok(c->n) && c->n <= 640;
    return is_pow2(c->n) && (c->n <= 1024 || (c->n == 2048 && c->ndim <= 2));
}
static inline int grid_for(size_t work, int block) {
    size_t g = (work + block - 1) / block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// the start value where it lies (read-only consumers) / brought home to the U[0] slab (everybody else)
static int spectrum_to_field(sdc_ctx* c, const cd* src, double* out);  // inverse transform of ONE cached spectrum
static int store_spectra(sdc_ctx* c, bool last_only);               // spectra of an iterate that was never stored
static int flush_x(sdc_ctx* c);                                     // residual norms whose last inverse pass was put off
static int trail_reset(sdc_ctx* c);
#define FLUSH_X(c)                              \
    do {                                        \
        int rcx_ = flush_x(c);                  \
        if (rcx_ != SDC_OK) return rcx_;        \
    } while (0)
#define STORE_SPECTRA(c, last)                  \
    do {                                        \
        int rcs_ = store_spectra(c, last);      \
        if (rcs_ != SDC_OK) return rcs_;        \
    } while (0)
static int ensure_u0(sdc_ctx* c) {
    if (c->u0_spec_only) {  // the start value exists as its transform only (sdc_advance after a deferred end value)
        int rcs = spectrum_to_field(c, c->S0, c->U0);
        if (rcs != SDC_OK) return rcs;  // (still "spectrum only": nobody reads a half-written U[0])
        c->u0_spec_only = false;
        return SDC_OK;
    }
    if (c->u0_src) {
        const double* src = c->u0_src;
        c->u0_src = nullptr;
        HIPCHK(c, hipMemcpyAsync(c->U0, src, c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    return SDC_OK;
}
// nullptr when the start value could not be produced (c->u0_rc, c->err say why): callers go through U0R
static inline const double* u0r(sdc_ctx* c) {
    if (c->u0_spec_only && (c->u0_rc = ensure_u0(c)) != SDC_OK) return nullptr;
    return c->u0_src ? c->u0_src : c->U0;
}
#define U0R(c, var)                        \
    const double* var = u0r(c);            \
    if (!var) return (c)->u0_rc
#define ENSURE_U0(c)                       \
    do {                                   \
        int rcu_ = ensure_u0(c);           \
        if (rcu_ != SDC_OK) return rcu_;   \
    } while (0)
// the end value for real: transform the last node's spectrum now if that was put off (sdc_end_point)
static int materialize_uend(sdc_ctx* c) {
    if (!c->uend_pending) return SDC_OK;
    c->uend_pending = false;
    if (!(c->uend_gen >= 0 && c->uend_gen == c->spec_gen)) return SDC_OK;  // (its spectrum is gone: nothing to deliver)
    int rcs = store_spectra(c, true);
    if (rcs != SDC_OK) return rcs;
    c->uend_ev_recorded = false;  // UEND is written NOW: an event recorded for an earlier end value does not cover it
    return spectrum_to_field(c, c->SL, c->UEND);
}
#define MATERIALIZE_UEND(c)                \
    do {                                   \
        int rcv_ = materialize_uend(c);    \
        if (rcv_ != SDC_OK) return rcv_;   \
    } while (0)

// (U / F are node-index bases that point N resp. ncomp N values BEFORE their blocks Un / Fn - index 0 through them is not
// memory of this context - and every block may still be unallocated (need_nodes / need_f0): null then, never base + offset)
static double* slot_ptr(sdc_ctx* c, int slot, int m, int comp) {
    switch (slot) {
        case SDC_SLOT_U:
            if (m == 0) return c->U0;
            return (m >= 1 && m <= c->M && c->Un) ? c->U + (size_t)m * c->N : nullptr;
        case SDC_SLOT_F:
            if (m < 0 || m > c->M || comp < 0 || comp >= c->ncomp) return nullptr;
            if (m == 0) return c->F0 ? c->F0 + (size_t)comp * c->N : nullptr;
            return c->Fn ? c->F + ((size_t)m * c->ncomp + comp) * c->N : nullptr;
        case SDC_SLOT_TAU: return (m >= 0 && m < c->M && c->TAU) ? c->TAU + (size_t)m * c->N : nullptr;
        case SDC_SLOT_UEND: return c->UEND;
        default: return nullptr;
    }
}

template <int MODE>
static int launch_quad(sdc_ctx* c, const QuadArgs& a0, const char* name) {
    NEED_NODES(c);
    QuadArgs a = a0;   // (the bases as they are NOW: the blocks may have been allocated a moment ago)
    a.F = c->F;
    if (a.Usub) a.Usub = c->U;
    LaunchTimer lt(c, name);
    int grid = grid_for(c->N / 2, 256);
    if (MODE >= 1 && grid > 2048) grid = 2048;   // (norms: fewer, longer workgroups - fewer atomics on the M slots)
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

// planes a workgroup of the marching 3-D stencil kernels takes: 64, fewer on small grids so that the launch still has a
// few workgroups per CU (256^3, one field: 128 workgroups with 64 planes each - half the CUs idle; 1024 with 8)
static inline int stencil_xchunk(int n, size_t tiles) {
    int x = n >= 64 ? 64 : n;
    while (x > 8 && tiles * (size_t)(n / x) < 1024) x >>= 1;
    return x;
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
        s3.useE = needE ? 1 : 0;
        s3.n = c->n;
        s3.ntiles = (c->n / 64) * (c->n / (8 * RPT));
        s3.xchunk = stencil_xchunk(c->n, (size_t)s3.ntiles * nf);
        s3.nchunks = c->n / s3.xchunk;
        hipLaunchKernelGGL((k_stencil3d<RPT>), dim3(s3.ntiles * s3.nchunks * nf), dim3(256), 0, c->stream, s3);
        HIPCHK(c, hipGetLastError());
        return SDC_OK;
    }
    hipLaunchKernelGGL(k_stencil, dim3(grid_for(c->N / 2, 256), nf), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

static int eval_nodes_plain(sdc_ctx* c);

// A v for the Krylov solvers and eval_f: the periodic stencil launch, or the banded operator of a bounded grid
static inline size_t problem_size(const sdc_ctx* c) { return c->nb ? c->Nb : c->N; }
static int apply_operator(sdc_ctx* c, const double* v, double* Av) {
    if (c->nb) {
        LaunchTimer lt(c, "banded_apply");
        hipLaunchKernelGGL(k_banded_apply, dim3(grid_for(c->Nb, 256)), dim3(256), 0, c->stream, v, Av, c->ndim, c->nb, c->bw,
                           c->bcols, c->bwts);
        HIPCHK(c, hipGetLastError());
        return SDC_OK;
    }
    const double* in[1] = {v};
    double* out[1] = {Av};
    return run_stencil(c, 1, in, out, nullptr, nullptr);
}

// F[1..M] = A U[1..M] for the new iterate; fused with the residual when the fast 3-D kernel applies
static int eval_nodes(sdc_ctx* c, double dt) {
    const int M = c->M;
    NEED_NODES(c);
    auto three = [](const Stencil& s) { return s.npts == 3 && s.off[0] == -1 && s.off[1] == 0 && s.off[2] == 1; };
    const bool expl = c->ncomp == 2 && c->expl_kind == SDC_EXPL_STENCIL;
    if (c->fuse_residual && c->ndim == 3 && (c->ncomp == 1 || expl) && !c->tau_active && c->n % 64 == 0 &&
        M <= (expl ? 5 : 6) && three(c->st[0]) && (!expl || three(c->st[1]))) {
        StencilResArgs a;
        memset(&a, 0, sizeof a);
        a.U = c->U;
        a.u0 = c->U0;
        a.F = c->F;
        for (int k = 0; k < 3; ++k) {
            a.wI[k] = c->st[0].w[k];
            a.wE[k] = expl ? c->st[1].w[k] : 0.0;
        }
        a.ncomp = c->ncomp;
        for (int m = 0; m < M; ++m)
            for (int j = 0; j < M; ++j) a.cQ[m][j] = dt * c->Q[m + 1][j + 1];
        a.norms = c->res_dev;
        a.n = c->n;
        a.N = c->N;
        a.xchunk = stencil_xchunk(c->n, (size_t)(c->n / 64) * (c->n / 8));
        a.nchunks = c->n / a.xchunk;
        HIPCHK(c, hipMemsetAsync(c->res_dev, 0, sizeof(unsigned long long) * 8, c->stream));
        const unsigned grid = (unsigned)((c->n / 64) * (c->n / 8) * a.nchunks);
        const bool wf = !c->deferred;
        {
            LaunchTimer lt(c, pname(wf ? "stencil_res" : "res_stencil", M));
#define RLAUNCH(MM, EX, WF) \
    hipLaunchKernelGGL((k_stencil3d_res<MM, EX, WF>), dim3(grid), dim3(256), 0, c->stream, a)
#define RCASE(MM)                                \
    case MM:                                     \
        if (expl && wf) RLAUNCH(MM, true, true); \
        else if (expl) RLAUNCH(MM, true, false); \
        else if (wf) RLAUNCH(MM, false, true);   \
        else RLAUNCH(MM, false, false);          \
        break;
            switch (M) { RCASE(1) RCASE(2) RCASE(3) RCASE(4) RCASE(5) RCASE(6) }
#undef RCASE
#undef RLAUNCH
        }
        HIPCHK(c, hipGetLastError());
        c->res_valid = true;
        c->res_dt = dt;
        c->f_pending = !wf;
        return SDC_OK;
    }
    return eval_nodes_plain(c);
}

// F[1..M] = f(U[1..M]) by the plain stencil launch
static int eval_nodes_plain(sdc_ctx* c) {
    const int M = c->M;
    NEED_NODES(c);
    c->f_pending = false;
    if (c->kind == 1) {  // van der Pol ensemble: node by node, not counted again
        for (int m = 1; m <= M; ++m) {
            LaunchTimer lt(c, "vdp_eval");
            hipLaunchKernelGGL(k_vdp_eval, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, c->U + (size_t)m * c->N,
                               c->F + (size_t)m * c->N, c->N / 2, c->vdp_mu, (unsigned long long*)nullptr);
        }
        HIPCHK(c, hipGetLastError());
        return SDC_OK;
    }
    const double* in[MAXM];
    double* oi[MAXM];
    double* oe[MAXM];
    double g[MAXM];
    for (int m = 0; m < M; ++m) {
        in[m] = c->U + (size_t)(m + 1) * c->N;
        oi[m] = c->F + ((size_t)(m + 1) * c->ncomp) * c->N;
        // (forcing: the explicit part is profile * g(t_m) whatever u is - stored again because a 'copy' predictor
        // leaves g(t_0) at every node, imex_1st_order.py:105 re-evaluates it)
        const bool has_expl = c->expl_kind == SDC_EXPL_STENCIL || c->expl_kind == SDC_EXPL_FORCING;
        oe[m] = (c->ncomp == 2 && has_expl) ? oi[m] + c->N : nullptr;
        g[m] = c->gvals[m + 1];
    }
    return run_stencil(c, M, in, oi, (c->expl_kind == SDC_EXPL_STENCIL || c->expl_kind == SDC_EXPL_FORCING) ? oe : nullptr, g);
}

// reaction term riding on a pass of the pipeline (see ReactEpi): where 0 = none, 1 = of the input field `field` (first pass),
// 2 = of the output field `field` (last pass)
struct ReactReq {
    int where = 0, field = 0;   // field < 0: of every field of the pipeline, into outs[f]
    double* out = nullptr;
    double* outs[MAXM] = {};
    double* impl_out = nullptr;      // the implicit part at the solution of field 0 from the solve's own equation (ReactEpi)
    const double* rhs = nullptr;
    double inv_alpha = 0.0;
};
template <int N>
static int fft_pipeline_n(sdc_ctx* c, int nf, const FieldPtrs& p, ZArgs& z, const ReactReq& rq = ReactReq(),
                          const LinTerms* lin = nullptr) {
    const ReactEpi none{nullptr, 0, 0, 0, 0.0, 0.0, {}};
    ReactEpi epi{rq.out, rq.field, c->react_kind, c->react_nu, c->react_p0, c->react_p1, {}};
    for (int f = 0; f < MAXM; ++f) epi.outs[f] = rq.outs[f];
    epi.impl_out = rq.impl_out;
    epi.rhs = rq.rhs;
    epi.inv_alpha = rq.inv_alpha;
    const bool can = c->ndim >= 2 && (rq.out != nullptr || rq.field < 0 || rq.impl_out != nullptr);  // (1-D lines go through k_promote / k_realpart: the caller launches k_reaction)
    constexpr int E = fft_elems(N), P = N / E;
    constexpr int T = strided_cols<N>();  // complex columns per strided tile (128-byte row segments up to N = 1024)
    const size_t lds_str = (size_t)LayStrided<N, T>::doubles(T) * sizeof(double);
    const int n = c->n;
    z.W = c->W;
    z.fstride = c->Nc;
    z.tw = c->tw;
    z.lamI = c->lamI;
    if (z.dup && c->expl_kind == SDC_EXPL_SYMBOL) z.lamE = c->lamE;   // (the explicit operator rides along: ZArgs::dup)
    z.nf = nf;
    z.ndim = c->ndim;
    z.invN = 1.0 / (double)c->N;
    const int nfi = z.dup ? 1 : nf;  // fields that are transformed forward
    size_t lines;
    if (c->ndim == 1) {
        LaunchTimer lt(c, pname("promote", nfi));
        hipLaunchKernelGGL(k_promote, dim3(grid_for(c->N, 256), nfi), dim3(256), 0, c->stream, p, c->W, c->N);
        lines = 1;
    } else {
        const int rest = (int)(c->N / n);
        const int tiles = (rest / 2 + T - 1) / T;
        {
            LaunchTimer lt(c, pname("fft_x_fwd", nfi));
            hipLaunchKernelGGL((k_fftx_fwd<N, T>), dim3(tiles, nfi), dim3(P * T), lds_str, c->stream, p, c->W, c->Nc,
                               rest, c->tw, (can && rq.where == 1) ? epi : none, lin ? *lin : LinTerms{{}, {}, 0, nullptr});
        }
        if (c->ndim == 3) {
            LaunchTimer lt(c, pname("fft_y_fwd", nfi));
            hipLaunchKernelGGL((k_ffty<N, T, -1>), dim3((n + T - 1) / T, n / 2 + 1, nfi), dim3(P * T), lds_str,
                               c->stream, c->W, c->Nc, c->tw);
        }
        lines = (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    }
    {
        LaunchTimer lt(c, pname("fft_z_solve", nf));
        constexpr int LPB = zsolve_lines<N>(), PZ = N / zsolve_elems<N>();
        constexpr int NCH = zsolve_elems<N>() == 16 ? 2 : 1;
        size_t ldsz = (size_t)LayContig<N>::doubles(nf * LPB) * sizeof(double);  // FFT exchange planes
        const size_t solve_sz = (size_t)nf * LPB * (N / NCH) * sizeof(cd);         // node-coupling buffer
        if (solve_sz > ldsz) ldsz = solve_sz;
        hipLaunchKernelGGL((k_fftz_solve<N>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(PZ * LPB * nf), ldsz,
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
        if (can && rq.where == 2 && rq.impl_out)
            hipLaunchKernelGGL((k_fftx_inv<N, T, false, true, false, 0, true>), dim3(tiles, nf), dim3(P * T), lds_str, c->stream, p,
                               c->W, c->Nc, rest, c->tw, nullptr, nullptr, 1, epi);
        else
            hipLaunchKernelGGL((k_fftx_inv<N, T, false, true>), dim3(tiles, nf), dim3(P * T), lds_str, c->stream, p, c->W, c->Nc,
                               rest, c->tw, nullptr, nullptr, 1, (can && rq.where == 2) ? epi : none);
    }
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// forward transform of nf real fields into fully transformed spectra dst[f] (dst + f*fstride)
template <int N>
static int fwd_transform_n(sdc_ctx* c, int nf, const FieldPtrs& p, cd* dst, size_t fstride) {
    constexpr int E = fft_elems(N), P = N / E, T = strided_cols<N>(), LPB = z_lines_per_block<N>();
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

// the inverse passes after the contiguous-axis one: work[f] -> real fields out[f] or (norms != null) max |.|
template <int N>
static int inverse_tail_n(sdc_ctx* c, int nf, cd* work, const FieldPtrs& p, unsigned long long* norms, bool y_done = false,
                          bool defer_x = false) {
    constexpr int E = fft_elems(N), P = N / E, T = strided_cols<N>();
    const size_t lds_str = (size_t)LayStrided<N, T>::doubles(T) * sizeof(double);
    // (the norm-only pass with one wave per column keeps the columns apart in LDS: a few more bytes)
    const size_t lds_x = (SDC_XWAVE && P == 64) ? (size_t)LayCols<N>::doubles(T) * sizeof(double) : lds_str;
    const int n = c->n;
    if (c->ndim == 1) {
        if (norms) return fail(c, SDC_ERR_UNSUPPORTED, "norm-only inverse transform in 1-D");
        LaunchTimer lt(c, pname("realpart", nf));
        hipLaunchKernelGGL(k_realpart, dim3(grid_for(c->N, 256), nf), dim3(256), 0, c->stream, p, work, c->N);
    } else {
        const int rest = (int)(c->N / n);
        const int tiles = (rest / 2 + T - 1) / T;
        if (c->ndim == 3 && !y_done) {
            LaunchTimer lt(c, pname("fft_y_inv", nf));
            hipLaunchKernelGGL((k_ffty<N, T, +1>), dim3((n + T - 1) / T, n / 2 + 1, nf), dim3(P * T), lds_str, c->stream,
                               work, c->Nc, c->tw);
        }
        if (norms && defer_x && !p.out[0] && !y_done) {
            // the last pass waits: for the start value this time-rank is about to receive (one pass then reduces the norms
            // before and after the receive, flush_x) or for somebody who wants the numbers
            HIPCHK(c, hipGetLastError());
            c->xp.pending = true;
            c->xp.has_delta = false;
            c->xp.d_old_spare = false;
            c->xp.nf = nf;
            c->xp.work = work;
            c->xp.norms = norms;
            c->xp.normsA = norms + 8;
            c->xp.tickets.clear();
            return SDC_OK;
        }
        if (norms) {
            if (p.out[0]) {  // the residual fields are kept as well
                LaunchTimer lt(c, pname("fft_x_inv_norm", nf));
                hipLaunchKernelGGL((k_fftx_inv<N, T, true, true>), dim3(tiles, nf), dim3(P * T), lds_str, c->stream, p,
                                   work, c->Nc, rest, c->tw, norms);
            } else {
                LaunchTimer lt(c, pname("fft_x_norm", nf));
                if constexpr (SDC_XHALF && (N == 256 || N == 512 || N == 1024)) {
                    constexpr int H = N / 2, TH = SDC_XHALF_T, PH = H / fft_elems(H);
                    const size_t lds_half = (size_t)LayStrided<H, TH>::doubles(TH) * sizeof(double);
                    hipLaunchKernelGGL((k_fftx_norm_half<N, TH>), dim3((rest + TH - 1) / TH, nf), dim3(PH * TH), lds_half,
                                       c->stream, work, c->Nc, rest, c->tw, c->tw + N, norms);
                } else {
                    hipLaunchKernelGGL((k_fftx_inv<N, T, true, false>), dim3(tiles, nf), dim3(P * T), lds_x, c->stream,
                                       p, work, c->Nc, rest, c->tw, norms);
                }
            }
        } else {
            LaunchTimer lt(c, pname("fft_x_inv", nf));
            hipLaunchKernelGGL((k_fftx_inv<N, T, false, true>), dim3(tiles, nf), dim3(P * T), lds_str, c->stream, p, work,
                               c->Nc, rest, c->tw, nullptr);
        }
    }
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// only the last (x) pass, norm-only, of fields that went through the other passes already, plus one more field `add`
template <int N>
static int inverse_tail_x_only(sdc_ctx* c, int nf, cd* work, const FieldPtrs& p, unsigned long long* norms, const cd* add) {
    constexpr int E = fft_elems(N), P = N / E, T = strided_cols<N>();
    const size_t lds_str = (SDC_XWAVE && P == 64) ? (size_t)LayCols<N>::doubles(T) * sizeof(double)
                                                  : (size_t)LayStrided<N, T>::doubles(T) * sizeof(double);
    const int rest = (int)(c->N / c->n);
    const int tiles = (rest / 2 + T - 1) / T;
    LaunchTimer lt(c, pname("fft_x_norm_add", nf));
    // 8 nf ceil(tiles / 8) workgroups in the XCD-aware order the kernel decodes
    const int groups = (tiles + 7) / 8;
    hipLaunchKernelGGL((k_fftx_inv<N, T, true, false, true>), dim3(groups * 8 * nf), dim3(P * T), lds_str, c->stream, p, work,
                       c->Nc, rest, c->tw, norms, add, nf);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// The difference of two start values through the inverse passes, parked in the arrangement the x pass holds its tiles in
// (dscr), then ONE x pass over the nf residual fields in `work` that reduces max |r| into normsA and max |r + d| into norms.
template <int N>
static int joint_norms_n(sdc_ctx* c, int nf, cd* work, const cd* d_new, const cd* d_old, cd* dbuf, cd* dscr,
                         unsigned long long* norms, unsigned long long* normsA, bool z_done = false) {
    constexpr int E = fft_elems(N), P = N / E, T = strided_cols<N>(), LPB = z_lines_per_block<N>();
    const size_t lds_str = (size_t)LayStrided<N, T>::doubles(T) * sizeof(double);
    const size_t lds_x = (SDC_XWAVE && P == 64) ? (size_t)LayCols<N>::doubles(T) * sizeof(double) : lds_str;
    const int n = c->n;
    const size_t lines = (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    const int rest = (int)(c->N / n);
    const int tiles = (rest / 2 + T - 1) / T;
    const ReactEpi none{nullptr, 0, 0, 0, 0.0, 0.0, {}};
    FieldPtrs p;
    memset(&p, 0, sizeof p);
    if (!z_done) {   // (done: dbuf holds the difference after its z pass - a trail launch transformed it on the way)
        LaunchTimer lt(c, pname("fft_z_diff", 1));
        const size_t ldsz = (size_t)LayContig<N>::doubles(LPB) * sizeof(double);
        hipLaunchKernelGGL((k_fftz_plain<N, +1>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(P * LPB), ldsz, c->stream,
                           d_new, dbuf, c->Nc, c->tw, (unsigned)lines, 1.0 / (double)c->N, (const cd*)nullptr, -1,
                           (const cd*)nullptr, (const cd*)nullptr, 0, d_old);
    }
    if (c->ndim == 3) {
        LaunchTimer lt(c, pname("fft_y_inv", 1));
        hipLaunchKernelGGL((k_ffty<N, T, +1>), dim3((n + T - 1) / T, n / 2 + 1, 1), dim3(P * T), lds_str, c->stream, dbuf,
                           c->Nc, c->tw);
    }
    {
        LaunchTimer lt(c, pname("fft_x_scr", 1));
        hipLaunchKernelGGL((k_fftx_inv<N, T, false, false, false, 1>), dim3(tiles, 1), dim3(P * T), lds_x, c->stream, p, dbuf,
                           c->Nc, rest, c->tw, (unsigned long long*)nullptr, (const cd*)nullptr, 1, none, dscr,
                           (unsigned long long*)nullptr);
    }
    {
        LaunchTimer lt(c, pname("fft_x_norm2", nf));
        const int groups = (tiles + 7) / 8;   // (8 nf ceil(tiles / 8) workgroups in the XCD-aware order the kernel decodes)
        hipLaunchKernelGGL((k_fftx_inv<N, T, true, false, false, 2>), dim3(groups * 8 * nf), dim3(P * T), lds_x, c->stream, p,
                           work, c->Nc, rest, c->tw, norms, (const cd*)nullptr, nf, none, dscr, normsA);
    }
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// inverse transform of nf fully transformed spectra src[f] (src + f*Nc) through the work buffer work[f]
// (work may equal src: in place) into the real fields out[f] - or, with norms != null, into max |.| per field
template <int N>
static int inverse_passes_n(sdc_ctx* c, int nf, const cd* src, cd* work, const FieldPtrs& p, unsigned long long* norms,
                            double scale, const cd* src_one = nullptr, int one = -1) {
    constexpr int P = N / fft_elems(N), LPB = z_lines_per_block<N>();
    const int n = c->n;
    const size_t lines = c->ndim == 1 ? 1 : (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    {
        LaunchTimer lt(c, pname("fft_z_inv", nf));
        const size_t ldsz = (size_t)LayContig<N>::doubles(nf * LPB) * sizeof(double);
        hipLaunchKernelGGL((k_fftz_plain<N, +1>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(P * LPB * nf), ldsz,
                           c->stream, src, work, c->Nc, c->tw, (unsigned)lines, scale, src_one, one);
    }
    return inverse_tail_n<N>(c, nf, work, p, norms);
}

// time-parallel runs: the end value is wanted as early as possible (it is sent while the residual passes run).
// Right after the spectral update S[M-1] is final: transform it into UEND through the spare work buffer.
template <int N>
static int early_end_point_n(sdc_ctx* c, bool norms_only) {
    if (!c->early_uend || !norms_only) return SDC_OK;
    if (c->wire_spectral && c->sl_ev_by_split) {   // (recorded behind the launch that wrote the spectrum by itself: sdc_sweep)
        c->sl_ev_by_split = false;
        return SDC_OK;
    }
    if (c->wire_spectral) {
        // the wire carries the last node's SPECTRUM (final as of now): nothing to transform, only a point in the stream to
        // wait for; the end value itself stays put off (sdc_end_point: uend_pending)
        if (!c->sl_ev) HIPCHK(c, hipEventCreateWithFlags(&c->sl_ev, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->sl_ev, c->stream));
        c->sl_ev_recorded = true;
        return SDC_OK;
    }
    {
        int rcf = uend_write_fence(c);
        if (rcf != SDC_OK) return rcf;
    }
    if (!c->W2) {
        HIPCHK(c, hipMalloc((void**)&c->W2, sizeof(cd) * c->Nc));
        c->bytes += sizeof(cd) * c->Nc;
    }
    if (!c->uend_ev) HIPCHK(c, hipEventCreateWithFlags(&c->uend_ev, hipEventDisableTiming));
    FieldPtrs pe;
    memset(&pe, 0, sizeof pe);
    pe.out[0] = c->UEND;
    int rc = inverse_passes_n<N>(c, 1, c->SL, c->W2, pe, nullptr, 1.0 / (double)c->N);
    if (rc != SDC_OK) return rc;
    c->uend_gen = c->spec_gen;
    HIPCHK(c, hipEventRecord(c->uend_ev, c->stream));
    c->uend_ev_recorded = true;
    return SDC_OK;
}

static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}
// the trail's z launch: the Nyquist modes of all lines by a small launch of their own, then one persistent workgroup per CU
// the table the small launch leaves for the big one: M (+ 1) Nyquist entries per z line (allocated by sdc_sweep, which can fail)
static int ensure_trail_nyq(sdc_ctx* c) {
    if (c->trail_nyq) return SDC_OK;
    const size_t all = c->ndim == 1 ? 1 : (size_t)(c->n / 2 + 1) * (c->ndim == 3 ? c->n : 1);
    HIPCHK(c, hipMalloc((void**)&c->trail_nyq, sizeof(cd) * all * (c->M + 1)));
    c->bytes += sizeof(cd) * all * (c->M + 1);
    return SDC_OK;
}
template <int N, int NF>
static void launch_trail_z(sdc_ctx* c, SpecArgs& a, size_t lines) {
    const unsigned wgs = (unsigned)std::min<size_t>(lines, (size_t)device_cus());
    // (hand-over lines, coefficients, N / 2 twiddles)
    constexpr size_t lds_tail = (((TrailCoef<NF>::COUNT + 1) & ~1) + N) * sizeof(double);
    constexpr bool can_dz = (NF + 1) * (N / specz_elems<N, true>()) <= trail_threads<N, NF>() &&
                            (size_t)2 * (NF + 1) * N * sizeof(double) + lds_tail <= (size_t)160 * 1024;
    if constexpr (can_dz) {
        if (a.dz && a.ns >= 2) {   // ... and the difference of the last two start values as one more line (SpecArgs::dz)
            hipLaunchKernelGGL((k_trail_nyq<NF, true>), dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, c->stream, a, N, (unsigned)lines, c->trail_nyq);
            hipLaunchKernelGGL((k_trail_z<N, NF, true>), dim3(wgs), dim3(trail_threads<N, NF>()), (size_t)2 * (NF + 1) * N * sizeof(double) + lds_tail,
                               c->stream, a, (unsigned)lines, (const cd*)c->trail_nyq);
            c->dz_written = true;
            return;
        }
    }
    hipLaunchKernelGGL((k_trail_nyq<NF, false>), dim3((unsigned)((lines + 255) / 256)), dim3(256), 0, c->stream, a, N, (unsigned)lines, c->trail_nyq);
    hipLaunchKernelGGL((k_trail_z<N, NF, false>), dim3(wgs), dim3(trail_threads<N, NF>()), (size_t)2 * NF * N * sizeof(double) + lds_tail, c->stream, a,
                       (unsigned)lines, (const cd*)c->trail_nyq);
}

template <int N, int NF, bool V>
static void launch_spec_z_cfg(sdc_ctx* c, SpecArgs& a, size_t lines, int mode, size_t launch_lines) {
    constexpr int P = N / specz_elems<N, V>(), LPB = specz_lines<N, V>();
    constexpr int SPAN = LPB * N, CH = specz_chunk(SPAN);
    size_t ldsz = (size_t)LayContig<N>::doubles(NF * LPB) * sizeof(double);
    if ((size_t)NF * CH * sizeof(cd) > ldsz) ldsz = (size_t)NF * CH * sizeof(cd);
    const dim3 grid((unsigned)((launch_lines + LPB - 1) / LPB)), block(P * LPB * NF);
    const int expl = a.lamE ? 1 : (a.SP ? 2 : 0);
#define ZL(M_, E_) hipLaunchKernelGGL((k_spec_z<N, NF, M_, E_>), grid, block, ldsz, c->stream, a, (unsigned)lines)
    if constexpr (V && (N & (N - 1)) != 0) {
        // lines of 3 * 2^p / 5 * 2^p modes: the recomputing launches only (mode pairs where a line fills a workgroup: 768) -
        // no trail, no multiplier table, and stored iterates take the pointwise launch (spec_sweep_n)
        (void)expl;
        (void)mode;
        if constexpr (LPB == 1 && SDC_SPECZ_PAIRS) {
            if (a.real_sym && !a.ns && !a.gmode) ZL(4, 0);
            else if (!a.ns && !a.gmode) ZL(3, 0);
            else fail(c, SDC_ERR_STATE, "no trail / table launch for lines of this length");
        } else {
            if (!a.ns && !a.gmode) ZL(3, 0);
            else fail(c, SDC_ERR_STATE, "no trail / table launch for lines of this length");
        }
    } else if constexpr (V) {  // (iterate recomputed from S0: real symbol, no explicit part - sdc_sweep sees to that)
        (void)expl;
        if constexpr (LPB == 1 && SDC_SPECZ_PAIRS) {
            // real symmetric symbol: the modes kz and N - kz of a line share their node multipliers
            if (a.real_sym && a.ns > 0) {
                if constexpr (N / 2 <= trail_threads<N, NF>()) launch_trail_z<N, NF>(c, a, lines);
                else fail(c, SDC_ERR_STATE, "no trail launch for lines of this length");   // (sdc_sweep: trails only up to n = 1024)
            }
            else if (a.real_sym && a.gmode) ZL(5, 0);
            else if (a.real_sym) ZL(4, 0);
            else ZL(3, 0);
        } else ZL(3, 0);
    } else if constexpr ((N & (N - 1)) == 0) {
#define ZM(E_)                   \
    if (mode == 0) ZL(0, E_);    \
    else if (mode == 1) ZL(1, E_); \
    else ZL(2, E_);
        if (expl == 1) { ZM(1) } else if (expl == 2) { ZM(2) } else { ZM(0) }
#undef ZM
    } else {
        (void)expl;
        fail(c, SDC_ERR_STATE, "stored iterates of lines of this length take the pointwise launch");
    }
#undef ZL
}
template <int N, int NF>
static void launch_spec_z(sdc_ctx* c, SpecArgs& a, size_t lines, int mode, size_t launch_lines = 0) {
    if (mode == 3) launch_spec_z_cfg<N, NF, true>(c, a, lines, mode, launch_lines ? launch_lines : lines);
    else launch_spec_z_cfg<N, NF, false>(c, a, lines, mode, launch_lines ? launch_lines : lines);
}

// spectral sweep; then either the inverse passes into out[f], or (norms != null) only the node norms of the
// collocation residual of the new iterate
// spec_only: the cached transforms are updated and nothing else happens (no residual is wanted, node values stay
// deferred) - one pointwise pass over the spectra
template <int N>
static int spec_sweep_n(sdc_ctx* c, int nf, SpecArgs& a, const FieldPtrs& p, unsigned long long* norms,
                        bool spec_only) {
    const int n = c->n;
    const size_t lines = c->ndim == 1 ? 1 : (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    constexpr bool kPow2 = (N & (N - 1)) == 0;
    if constexpr (SDC_FUSE_SPECZ && N >= 64 && N <= 1024) {  // fused with the first inverse pass (M <= 5)
        // (lines of 3 * 2^p / 5 * 2^p modes: only the launches that recompute the iterate - a.virt - exist)
        if (nf <= 5 && !spec_only && (kPow2 || a.virt)) {
            {
                // after a spread predictor all nodes share S0: that launch does not read S (fewer bytes)
                static const char* const vnames[] = {"spec_z_res_v0", "spec_z_res_v1", "spec_z_res_v2", "spec_z_res_v3",
                                                     "spec_z_res_v4", "spec_z_res_v5", "spec_z_res_v6", "spec_z_res_v7+"};
                static const char* const inames[] = {"spec_z_v0", "spec_z_v1", "spec_z_v2", "spec_z_v3",
                                                     "spec_z_v4", "spec_z_v5", "spec_z_v6", "spec_z_v7+"};
                const char* zbase = a.virt && a.store_last ? (a.virt == 2 ? "spec_z_last" : "spec_z_res_last")
                                    : a.virt && a.gmode == 2 ? (a.virt == 2 ? "spec_z_tab" : "spec_z_res_tab")
                                    : a.virt == 2 ? inames[a.replay < 7 ? a.replay : 7]
                                    : a.virt ? vnames[a.replay < 7 ? a.replay : 7]
                                             : norms ? (a.spread ? "spec_z_res_spread" : "spec_z_res")
                                                     : (a.spread ? "spec_z_spread" : "spec_z");
#define ZCASE(MM) \
    case MM: launch_spec_z<N, MM>(c, a, lines, a.virt ? 3 : (norms ? 1 : 0), glines); break;
                const size_t glines = 0;
                LaunchTimer lt(c, pname(zbase, nf));
                switch (nf) { ZCASE(1) ZCASE(2) ZCASE(3) ZCASE(4) ZCASE(5) }
#undef ZCASE
            }
            HIPCHK(c, hipGetLastError());
            int rce = early_end_point_n<N>(c, norms != nullptr);
            if (rce != SDC_OK) return rce;
            // time-parallel levels with spectra on the wire: the last pass waits for the start value that is on its way
            const bool dx = c->defer_x && c->wire_spectral && c->early_uend && c->ndim >= 2;
            return inverse_tail_n<N>(c, nf, c->W, p, norms, false, dx);
        }
    }
    {
        LaunchTimer lt(c, pname(norms ? "spec_point_res" : (spec_only ? "spec_point_only" : "spec_point"), nf));
        const size_t nmodes = lines * N;
        size_t gblocks = (nmodes + 255) / 256;
        if (gblocks > SDC_SPEC_GRID) gblocks = SDC_SPEC_GRID;
        const dim3 grid((unsigned)gblocks);
#define SCASE(MM)                                                                                             \
    case MM:                                                                                                  \
        if (norms) hipLaunchKernelGGL((k_spec_point<MM, true>), grid, dim3(256), 0, c->stream, a, n, nmodes); \
        else hipLaunchKernelGGL((k_spec_point<MM, false>), grid, dim3(256), 0, c->stream, a, n, nmodes);      \
        break;
        switch (nf) { SCASE(1) SCASE(2) SCASE(3) SCASE(4) SCASE(5) SCASE(6) SCASE(7) SCASE(8) }
#undef SCASE
    }
    HIPCHK(c, hipGetLastError());
    if (spec_only) return early_end_point_n<N>(c, true);
    if (norms) {  // W holds the residual spectra: the contiguous-axis pass runs first, then the early end value
        constexpr int P = N / fft_elems(N), LPB = z_lines_per_block<N>();
        {
            LaunchTimer lt(c, pname("fft_z_inv", nf));
            const size_t ldsz = (size_t)LayContig<N>::doubles(nf * LPB) * sizeof(double);
            hipLaunchKernelGGL((k_fftz_plain<N, +1>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(P * LPB * nf), ldsz,
                               c->stream, c->W, c->W, c->Nc, c->tw, (unsigned)lines, a.invN);
        }
        HIPCHK(c, hipGetLastError());
        int rce = early_end_point_n<N>(c, true);
        if (rce != SDC_OK) return rce;
        return inverse_tail_n<N>(c, nf, c->W, p, norms);
    }
    return inverse_passes_n<N>(c, nf, c->S, c->W, p, norms, a.invN, c->SL, nf - 1);
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
        case 24: return CALL(24);                                                                           \
        case 48: return CALL(48);                                                                           \
        case 96: return CALL(96);                                                                           \
        case 192: return CALL(192);                                                                         \
        case 384: return CALL(384);                                                                         \
        case 768: return CALL(768);                                                                         \
        case 40: return CALL(40);                                                                           \
        case 80: return CALL(80);                                                                           \
        case 160: return CALL(160);                                                                         \
        case 320: return CALL(320);                                                                         \
        case 640: return CALL(640);                                                                         \
        default: return fail(c, SDC_ERR_UNSUPPORTED, "spectral solve needs n = 2^p <= 2048, 3 * 2^p in 24 .. 768 or 5 * 2^p in 40 .. 640, got %d", (c)->n); \
    }

static int fwd_transform(sdc_ctx* c, int nf, const FieldPtrs& p, cd* dst, size_t fstride) {
#define CALL(NN) fwd_transform_n<NN>(c, nf, p, dst, fstride)
    N_DISPATCH(c, CALL)
#undef CALL
}
static int spec_sweep(sdc_ctx* c, int nf, SpecArgs& a, const FieldPtrs& p, unsigned long long* norms,
                      bool spec_only = false) {
    {
        int rw = ensure_work(c);
        if (rw != SDC_OK) return rw;
        a.W = c->W;
    }
#define CALL(NN) spec_sweep_n<NN>(c, nf, a, p, norms, spec_only)
    N_DISPATCH(c, CALL)
#undef CALL
}
// node norms of the collocation residual of the cached iterate against the current U[0], reduced into norms[0..M)
template <int N>
static int spec_residual_n(sdc_ctx* c, SpecArgs& a, unsigned long long* norms) {
    const int n = c->n, nf = c->M;
    const size_t lines = c->ndim == 1 ? 1 : (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    FieldPtrs p0;
    memset(&p0, 0, sizeof p0);
    if constexpr (SDC_FUSE_SPECZ && N >= 64 && N <= 1024 && (N & (N - 1)) == 0) {
        if (nf <= 5) {
            {
                LaunchTimer lt(c, pname("spec_z_resid", nf));
#define ZCASE(MM) \
    case MM: launch_spec_z<N, MM>(c, a, lines, 2); break;
                switch (nf) { ZCASE(1) ZCASE(2) ZCASE(3) ZCASE(4) ZCASE(5) }
#undef ZCASE
            }
            HIPCHK(c, hipGetLastError());
            return inverse_tail_n<N>(c, nf, c->W, p0, norms);
        }
    }
    {
        LaunchTimer lt(c, pname("spec_residual", nf));
        const size_t nmodes = lines * N;
        size_t gblocks = (nmodes + 255) / 256;
        if (gblocks > SDC_SPEC_GRID) gblocks = SDC_SPEC_GRID;
        const dim3 grid((unsigned)gblocks);
#define SCASE(MM) \
    case MM: hipLaunchKernelGGL((k_spec_residual<MM>), grid, dim3(256), 0, c->stream, a, n, nmodes); break;
        switch (nf) { SCASE(1) SCASE(2) SCASE(3) SCASE(4) SCASE(5) SCASE(6) SCASE(7) SCASE(8) }
#undef SCASE
    }
    HIPCHK(c, hipGetLastError());
    FieldPtrs p;
    memset(&p, 0, sizeof p);
    return inverse_passes_n<N>(c, nf, c->W, c->W, p, norms, a.invN);
}
static int fwd_transform(sdc_ctx* c, int nf, const FieldPtrs& p, cd* dst, size_t fstride);
// u-independent forcing profile * g(t): SP = transform of the profile (cached), cP[m] = dt sum_j Q[m][j] g(t_j)
static int forcing_spectrum(sdc_ctx* c, SpecArgs& a, double dt) {
    if (c->expl_kind != SDC_EXPL_FORCING) return SDC_OK;
    if (!c->profile) return fail(c, SDC_ERR_STATE, "forcing profile not set");
    if (!c->SP) {
        HIPCHK(c, hipMalloc((void**)&c->SP, sizeof(cd) * c->Nc));
        c->bytes += sizeof(cd) * c->Nc;
        c->specP_valid = false;
    }
    if (!c->specP_valid) {
        FieldPtrs pp;
        memset(&pp, 0, sizeof pp);
        pp.in[0] = c->profile;
        int rc = fwd_transform(c, 1, pp, c->SP, 0);
        if (rc != SDC_OK) return rc;
        c->specP_valid = true;
    }
    a.SP = c->SP;
    for (int m = 0; m < c->M; ++m) {
        double s = 0.0;
        for (int j = 0; j < c->M; ++j) s += dt * c->Q[m + 1][j + 1] * c->gvals[j + 1];
        a.cP[m] = s;
    }
    return SDC_OK;
}
static int spec_residual(sdc_ctx* c, double dt, unsigned long long* norms) {
    int rw = ensure_work(c);
    if (rw != SDC_OK) return rw;
    STORE_SPECTRA(c, false);  // (before S0 is replaced: an iterate that was not stored is a function of the OLD one)
    {
        int rcn = need_node_spectra(c);
        if (rcn != SDC_OK) return rcn;
    }
    if (!c->spec0_valid) {
        FieldPtrs p0;
        memset(&p0, 0, sizeof p0);
        U0R(c, u0p);
        p0.in[0] = u0p;
        int rc0 = fwd_transform(c, 1, p0, c->S0, 0);
        if (rc0 != SDC_OK) return rc0;
        c->spec0_valid = true;
    }
    SpecArgs a;
    memset(&a, 0, sizeof a);
    int rcf = forcing_spectrum(c, a, dt);
    if (rcf != SDC_OK) return rcf;
    a.S = c->S;
    a.SL = c->SL;
    a.real_sym = c->sym_real[0] ? 1 : 0;
    a.fstride = c->Nc;
    a.S0 = c->S0;
    a.W = c->W;
    a.tw = c->tw;
    a.lamI = c->lamI;
    a.lamE = c->expl_kind == SDC_EXPL_STENCIL ? c->lamE : nullptr;
    a.invN = 1.0 / (double)c->N;
    a.nf = c->M;
    a.ndim = c->ndim;
    for (int m = 0; m < c->M; ++m)
        for (int j = 0; j < c->M; ++j) a.rQ[m][j] = dt * c->Q[m + 1][j + 1];
#define CALL(NN) spec_residual_n<NN>(c, a, norms)
    N_DISPATCH(c, CALL)
#undef CALL
}
// Iterates that were never stored (c->spec_virtual sweeps since a spread predictor, all with the coefficients c->vcoef):
// write the spectra of the current one to S / SL now - all of them, or only the last node's (the end value).  Everything
// that reads the cache, and everything that is about to change S0, comes through here first.
static int store_spectra(sdc_ctx* c, bool last_only) {
    if (!c->spec_valid || c->spec_virtual <= 0) return SDC_OK;
    if (last_only && c->sl_stored) return SDC_OK;
    SpecArgs a;
    memset(&a, 0, sizeof a);
    const SpecCoef& v = c->vcoef;
    memcpy(a.gI, v.gI, sizeof a.gI);
    memcpy(a.gE, v.gE, sizeof a.gE);
    memcpy(a.cI, v.cI, sizeof a.cI);
    memcpy(a.cE, v.cE, sizeof a.cE);
    memcpy(a.alpha, v.alpha, sizeof a.alpha);
    a.coupled = v.coupled;
    a.real_sym = v.real_sym;
    a.lamE = v.has_e ? c->lamE : nullptr;
    a.S = c->S;
    a.SL = c->SL;
    a.fstride = c->Nc;
    a.S0 = c->S0;
    a.lamI = c->lamI;
    a.nf = c->M;
    a.ndim = c->ndim;
    a.last_only = last_only ? 1 : 0;
    if (!last_only) {
        int rcn = need_node_spectra(c);
        if (rcn != SDC_OK) return rcn;
        a.S = c->S;
    }
    const int n = c->n;
    const size_t lines = c->ndim == 1 ? 1 : (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    if (c->trail_ns > 0) {
        // the iterate depends on several start values (time-parallel levels): trail_iterate per mode pair
        a.ns = c->trail_ns;
        a.nsw = c->spec_virtual;
        for (int i = 0; i < c->trail_ns; ++i) a.src[i] = c->trail_src[i];
        memcpy(a.vsrc, c->vsrc, sizeof a.vsrc);
        for (int k = 0; k < a.nsw; ++k) a.scnt[a.vsrc[k]]++;
        const size_t nitems = lines * (size_t)(n / 2 + 1);
        size_t tb = (nitems + 255) / 256;
        if (tb > SDC_SPEC_GRID) tb = SDC_SPEC_GRID;
        {
            LaunchTimer lt(c, pname(last_only ? "trail_store_last" : "trail_store", c->M));
#define SCASE(MM) \
    case MM: hipLaunchKernelGGL((k_trail_store<MM>), dim3((unsigned)tb), dim3(256), 0, c->stream, a, n, nitems); break;
            switch (c->M) { SCASE(1) SCASE(2) SCASE(3) SCASE(4) SCASE(5) SCASE(6) SCASE(7) SCASE(8) }
#undef SCASE
        }
        HIPCHK(c, hipGetLastError());
        c->sl_stored = true;
        if (!last_only) {
            c->spec_virtual = 0;
            return trail_reset(c);
        }
        return SDC_OK;
    }
    if (c->Gm && c->g_sweeps > 0 && c->g_sweeps == c->spec_virtual) {  // the multipliers of this very iterate are on record
        a.G = c->Gm;
        a.gmode = 2;
    }
    const size_t nmodes = lines * n;
    size_t gblocks = (nmodes + 255) / 256;
    if (gblocks > SDC_SPEC_GRID) gblocks = SDC_SPEC_GRID;
    {
        LaunchTimer lt(c, pname(last_only ? "spec_store_last" : "spec_store", c->M));
        if (a.real_sym && !a.lamE && c->ndim >= 2 && n >= 4 && SDC_SPECZ_PAIRS) {
            const size_t npairs = lines * (size_t)(n / 2);
            size_t pb = (npairs + 255) / 256;
            if (pb > SDC_SPEC_GRID) pb = SDC_SPEC_GRID;
#define SCASE(MM) \
    case MM: hipLaunchKernelGGL((k_spec_store_pairs<MM>), dim3((unsigned)pb), dim3(256), 0, c->stream, a, n, npairs, c->spec_virtual); break;
            switch (c->M) { SCASE(1) SCASE(2) SCASE(3) SCASE(4) SCASE(5) SCASE(6) SCASE(7) SCASE(8) }
#undef SCASE
        } else {
#define SCASE(MM) \
    case MM: hipLaunchKernelGGL((k_spec_store<MM>), dim3((unsigned)gblocks), dim3(256), 0, c->stream, a, n, nmodes, c->spec_virtual); break;
            switch (c->M) { SCASE(1) SCASE(2) SCASE(3) SCASE(4) SCASE(5) SCASE(6) SCASE(7) SCASE(8) }
#undef SCASE
        }
    }
    HIPCHK(c, hipGetLastError());
    c->sl_stored = true;
    if (!last_only) c->spec_virtual = 0;
    return SDC_OK;
}

// real fields out[f] from the cached spectra S[first .. first + nf)
static int inverse_from_cache(sdc_ctx* c, int first, int nf, const FieldPtrs& p) {
    int rw = ensure_work(c);
    if (rw != SDC_OK) return rw;
    STORE_SPECTRA(c, first == c->M - 1 && nf == 1);
    if (!(first == c->M - 1 && nf == 1)) {
        int rcn = need_node_spectra(c);
        if (rcn != SDC_OK) return rcn;
    }
    const double invN = 1.0 / (double)c->N;
    // (the last node's spectrum lives behind its own pointer)
    const bool has_last = first + nf == c->M;
#define CALL(NN) \
    inverse_passes_n<NN>(c, nf, c->S + (size_t)first * c->Nc, c->W, p, nullptr, invN, has_last ? c->SL : nullptr, has_last ? nf - 1 : -1)
    N_DISPATCH(c, CALL)
#undef CALL
}

static int spectrum_to_field(sdc_ctx* c, const cd* src, double* out) {
    int rw = ensure_work(c);
    if (rw != SDC_OK) return rw;
    FieldPtrs p;
    memset(&p, 0, sizeof p);
    p.out[0] = out;
    const double invN = 1.0 / (double)c->N;
#define CALL(NN) inverse_passes_n<NN>(c, 1, src, c->W, p, nullptr, invN)
    N_DISPATCH(c, CALL)
#undef CALL
}

// max |f(u)| of the field whose transform is src, for right-hand sides that are linear with a stencil symbol: norm-only
// inverse transform of (symbol * src) - three passes over ONE spectrum, nothing stored in real space
template <int N>
static int symbol_norm_n(sdc_ctx* c, const cd* src, unsigned long long* slot) {
    constexpr int P = N / fft_elems(N), LPB = z_lines_per_block<N>();
    const int n = c->n;
    const size_t lines = (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    {
        LaunchTimer lt(c, pname("fft_z_sym", 1));
        const size_t ldsz = (size_t)LayContig<N>::doubles(LPB) * sizeof(double);
        hipLaunchKernelGGL((k_fftz_plain<N, +1, true>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(P * LPB), ldsz, c->stream, src,
                           c->W, c->Nc, c->tw, (unsigned)lines, 1.0 / (double)c->N, (const cd*)nullptr, -1, c->lamI,
                           c->expl_kind == SDC_EXPL_STENCIL ? c->lamE : (const cd*)nullptr, c->ndim);
    }
    HIPCHK(c, hipGetLastError());
    FieldPtrs p;
    memset(&p, 0, sizeof p);
    return inverse_tail_n<N>(c, 1, c->W, p, slot);
}
static int symbol_norm(sdc_ctx* c, const cd* src, unsigned long long* slot) {
    int rw = ensure_work(c);
    if (rw != SDC_OK) return rw;
#define CALL(NN) symbol_norm_n<NN>(c, src, slot)
    N_DISPATCH(c, CALL)
#undef CALL
}

// (I - alpha_f A) out_f = in_f + sum_{j<f} (cI[f][j] A + cE[f][j] B) out_j for f = 0..nf-1
static int fft_pipeline(sdc_ctx* c, int nf, const FieldPtrs& p, ZArgs& z, const ReactReq& rq = ReactReq(),
                        const LinTerms* lin = nullptr) {
    if (!c->have_stencil[0]) return fail(c, SDC_ERR_STATE, "implicit operator not set (sdc_set_stencil)");
    if (!fourier_ok(c))
        return fail(c, SDC_ERR_UNSUPPORTED,
                    "spectral solve needs n = 2^p <= 1024 per dimension (<= 2048 in 1-D / 2-D), 3 * 2^p in 24 .. 768 or 5 * 2^p in 40 .. 640, got %d", c->n);
    {
        int rw = ensure_work(c);
        if (rw != SDC_OK) return rw;
    }
    switch (c->n) {
        case 2: return fft_pipeline_n<2>(c, nf, p, z, rq, lin);
        case 4: return fft_pipeline_n<4>(c, nf, p, z, rq, lin);
        case 8: return fft_pipeline_n<8>(c, nf, p, z, rq, lin);
        case 16: return fft_pipeline_n<16>(c, nf, p, z, rq, lin);
        case 32: return fft_pipeline_n<32>(c, nf, p, z, rq, lin);
        case 64: return fft_pipeline_n<64>(c, nf, p, z, rq, lin);
        case 128: return fft_pipeline_n<128>(c, nf, p, z, rq, lin);
        case 256: return fft_pipeline_n<256>(c, nf, p, z, rq, lin);
        case 512: return fft_pipeline_n<512>(c, nf, p, z, rq, lin);
        case 1024: return fft_pipeline_n<1024>(c, nf, p, z, rq, lin);
        case 2048: return fft_pipeline_n<2048>(c, nf, p, z, rq, lin);
        case 24: return fft_pipeline_n<24>(c, nf, p, z, rq, lin);
        case 48: return fft_pipeline_n<48>(c, nf, p, z, rq, lin);
        case 96: return fft_pipeline_n<96>(c, nf, p, z, rq, lin);
        case 192: return fft_pipeline_n<192>(c, nf, p, z, rq, lin);
        case 384: return fft_pipeline_n<384>(c, nf, p, z, rq, lin);
        case 768: return fft_pipeline_n<768>(c, nf, p, z, rq, lin);
        case 40: return fft_pipeline_n<40>(c, nf, p, z, rq, lin);
        case 80: return fft_pipeline_n<80>(c, nf, p, z, rq, lin);
        case 160: return fft_pipeline_n<160>(c, nf, p, z, rq, lin);
        case 320: return fft_pipeline_n<320>(c, nf, p, z, rq, lin);
        case 640: return fft_pipeline_n<640>(c, nf, p, z, rq, lin);
    }
    return fail(c, SDC_ERR_UNSUPPORTED, "n = %d", c->n);
}

static int build_symbol(sdc_ctx* c, int which) {
    const int n = c->n;
    std::vector<cd> lam(n);
    const Stencil& s = c->st[which];
    // a symmetric stencil (w_{-o} = w_o: the heat operator) has a real symbol; say so exactly instead of leaving the
    // 1e-20 the two sine terms of a pair do not cancel to in floating point
    bool symmetric = true;
    for (int q = 0; q < s.npts && symmetric; ++q) {
        bool found = s.off[q] == 0;
        for (int r = 0; r < s.npts && !found; ++r) found = s.off[r] == -s.off[q] && s.w[r] == s.w[q];
        symmetric = found;
    }
    c->sym_real[which] = symmetric;
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
        lam[k] = cd{(double)re, symmetric ? 0.0 : (double)im};
    }
    if (symmetric)  // ... and an even one, to the last bit (modes k and n - k share their node multipliers in k_spec_z)
        for (int k = n / 2 + 1; k < n; ++k) lam[k].x = lam[n - k].x;
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

template <int N>
static int residual_shift_n(sdc_ctx* c, cd* dbuf, const cd* newS0, const cd* oldS0) {
    constexpr int P = N / fft_elems(N), LPB = z_lines_per_block<N>();
    const int n = c->n;
    const size_t lines = (size_t)(n / 2 + 1) * (c->ndim == 3 ? n : 1);
    {
        // d = new - old start value, through the contiguous-axis inverse pass (dbuf may be the old spectrum itself: in place)
        LaunchTimer lt(c, pname("fft_z_diff", 1));
        const size_t ldsz = (size_t)LayContig<N>::doubles(LPB) * sizeof(double);
        hipLaunchKernelGGL((k_fftz_plain<N, +1>), dim3((unsigned)((lines + LPB - 1) / LPB)), dim3(P * LPB), ldsz, c->stream,
                           newS0, dbuf, c->Nc, c->tw, (unsigned)lines, 1.0 / (double)c->N, (const cd*)nullptr, -1,
                           (const cd*)nullptr, (const cd*)nullptr, 0, oldS0);
    }
    HIPCHK(c, hipGetLastError());
    if (c->ndim == 3) {
        constexpr int E = fft_elems(N), PS = N / E, T = strided_cols<N>();
        const size_t lds_str = (size_t)LayStrided<N, T>::doubles(T) * sizeof(double);
        LaunchTimer lt(c, pname("fft_y_inv", 1));
        hipLaunchKernelGGL((k_ffty<N, T, +1>), dim3((n + T - 1) / T, n / 2 + 1, 1), dim3(PS * T), lds_str, c->stream, dbuf,
                           c->Nc, c->tw);
        HIPCHK(c, hipGetLastError());
    }
    FieldPtrs p;
    memset(&p, 0, sizeof p);
    return inverse_tail_x_only<N>(c, c->M, c->W, p, c->res_dev, dbuf);
}

// publishing launch of one residual ticket (see k_publish_residual below)
static int publish_ticket(sdc_ctx* c, const PendingTicket& t, const unsigned long long* norms);

// Residual norms whose last inverse pass was put off (inverse_tail_n, defer_x): run it now - alone, or, when the start value
// has been replaced since (sdc_replace_u0_spectrum), as the pass that reduces the norms before AND after the receive - and
// publish the tickets that wait for it.  Everything that is about to overwrite W, the norm slots or the spectra involved
// comes through here first.
static int flush_x(sdc_ctx* c) {
    PendingX& xp = c->xp;
    if (!xp.pending) return SDC_OK;
    xp.pending = false;
    const unsigned long long* before = xp.norms;
    FieldPtrs p0;
    memset(&p0, 0, sizeof p0);
    int rc = SDC_OK;
    if (xp.has_delta) {
        const bool z_done = xp.dz != nullptr;
        cd* dbuf = z_done ? xp.dz : spool_get(c);
        cd* dscr = dbuf ? spool_get(c) : nullptr;
        xp.dz = nullptr;
        if (!dbuf || !dscr) {
            rc = SDC_ERR_NOMEM;   // (spool_get has left the message)
        } else if (hipMemsetAsync(xp.normsA, 0, sizeof(unsigned long long) * 8, c->stream) != hipSuccess) {
            rc = fail(c, SDC_ERR_HIP, "hipMemsetAsync of the norm slots failed");
        } else {
#define CALL(NN) joint_norms_n<NN>(c, xp.nf, xp.work, xp.d_new, xp.d_old, dbuf, dscr, xp.norms, xp.normsA, z_done)
            rc = [&]() -> int { N_DISPATCH(c, CALL) }();
#undef CALL
        }
        spool_put(c, dbuf);   // (stream-ordered: whoever takes them next works behind these launches)
        spool_put(c, dscr);
        if (xp.d_old_spare) spool_put(c, xp.d_old);
        before = xp.normsA;
        if (xp.work == c->W) c->rlines_valid = false;   // W + d is what the current norms describe
    } else {
#define CALL(NN) inverse_tail_n<NN>(c, xp.nf, xp.work, p0, xp.norms, true)
        rc = [&]() -> int { N_DISPATCH(c, CALL) }();
#undef CALL
    }
    if (rc != SDC_OK) {
        // the numbers could not be made: whoever holds a ticket of this pass reads NaN (all bits set) - never a record of an
        // older residual, never a wait without end - and the caller gets the error
        (void)hipMemsetAsync(xp.norms, 0xFF, sizeof(unsigned long long) * 8, c->stream);
        if (xp.has_delta) (void)hipMemsetAsync(xp.normsA, 0xFF, sizeof(unsigned long long) * 8, c->stream);
        (void)hipGetLastError();
    }
    for (const PendingTicket& t : xp.tickets) {
        const int rcp = publish_ticket(c, t, (xp.has_delta && !t.after) ? before : xp.norms);
        if (rc == SDC_OK) rc = rcp;
    }
    xp.tickets.clear();
    xp.has_delta = false;
    xp.d_old_spare = false;
    xp.d_old = xp.d_new = nullptr;
    return rc;
}

// the trail of unstored sweeps is over (its iterate was stored, or its state is gone): the start values it depended on -
// all but the current one - are spare buffers again
static int trail_reset(sdc_ctx* c) {
    if (c->trail_ns == 0) return SDC_OK;
    if (c->xp.pending && c->xp.has_delta) FLUSH_X(c);   // (reads two of them)
    for (int i = 0; i < c->trail_ns; ++i) {
        const cd* b = c->trail_src[i];
        if (b != c->S0 && b != c->SL && b != c->Sin) spool_put(c, b);
        c->trail_src[i] = nullptr;
    }
    c->trail_ns = 0;
    return SDC_OK;
}

extern "C" {

int sdc_version(void) { return 103; }  // 101: sdc_work_counters writes out[5]; 102: residual post / wait, batched eval_f, accumulating transfer; sdc_set_pipeline_groups gone; 103: sdc_transfer_apply_nested, SDC_EXPL_SYMBOL (sdc_set_symbol which = 1)

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
        if (const char* lm = getenv("SDC_LAZY_MIN_BYTES")) c->lazy_min_bytes = (size_t)strtoull(lm, nullptr, 10);
        // U[0] is there from the start; U[1..M] and F are mapped when something first touches them in real space (the sweeps
        // that stay in Fourier space never do)
        c->bytes = 0;
        HIPCHK(nullptr, hipMalloc((void**)&c->U0, fb));
        HIPCHK(nullptr, hipMemsetAsync(c->U0, 0, fb, c->stream));
        c->bytes += fb;
        if (fb < c->lazy_min_bytes) {   // small fields: everything from the start (same blocks, same addressing)
            int rcs = need_nodes(c);
            if (rcs != SDC_OK) {
                g_create_err = c->err;
                return rcs;
            }
        }
        HIPCHK(nullptr, hipMalloc((void**)&c->UEND, fb));
        HIPCHK(nullptr, hipMalloc((void**)&c->red, sizeof(unsigned long long) * 16));
        HIPCHK(nullptr, hipHostMalloc((void**)&c->red_host, sizeof(unsigned long long) * 16));
        HIPCHK(nullptr, hipHostMalloc((void**)&c->ring, sizeof(ResRecord) * RES_RING, hipHostMallocMapped | hipHostMallocCoherent));
        memset(c->ring, 0, sizeof(ResRecord) * RES_RING);
        HIPCHK(nullptr, hipHostGetDevicePointer((void**)&c->ring_dev, c->ring, 0));
        c->bytes += fb;
        HIPCHK(nullptr, hipMalloc((void**)&c->counters, sizeof(unsigned long long) * 4));
        HIPCHK(nullptr, hipMalloc((void**)&c->res_bank[0], sizeof(unsigned long long) * 32));
        c->res_bank[1] = c->res_bank[0] + 16;
        c->res_dev = c->res_bank[0];
        c->res_devA = c->res_dev + 8;
        HIPCHK(nullptr, hipMemsetAsync(c->counters, 0, sizeof(unsigned long long) * 4, c->stream));
        HIPCHK(nullptr, hipMemsetAsync(c->res_bank[0], 0, sizeof(unsigned long long) * 32, c->stream));
        HIPCHK(nullptr, hipMemsetAsync(c->UEND, 0, fb, c->stream));
        HIPCHK(nullptr, hipEventCreate(&c->ev0));
        HIPCHK(nullptr, hipEventCreate(&c->ev1));
        HIPCHK(nullptr, hipEventCreate(&c->pev0));
        HIPCHK(nullptr, hipEventCreate(&c->pev1));
        if (is_pow2(n) || fft_length_ok(n)) {
            std::vector<cd> tw(n);
            for (int m = 0; m < n; ++m) {
                const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)m / (long double)n;
                tw[m] = cd{(double)cosl(ang), (double)(-sinl(ang))};
            }
            // behind it the table of the half-length transform (k_fftx_norm_half): e^{-2 pi i m/(n/2)}, m < n/2
            for (int m = 0; m < n / 2; ++m) tw.push_back(tw[2 * m]);
            HIPCHK(nullptr, hipMalloc((void**)&c->tw, sizeof(cd) * tw.size()));
            HIPCHK(nullptr, hipMemcpyAsync(c->tw, tw.data(), sizeof(cd) * tw.size(), hipMemcpyHostToDevice, c->stream));
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
    (void)flush_x(c);
    comm_free(c);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->U0);
    (void)hipFree(c->Un);
    (void)hipFree(c->Fn);
    (void)hipFree(c->F0);
    (void)hipFree(c->Sn);
    for (cd* b : c->spool_owned) (void)hipFree(b);
    (void)hipFree(c->TAU);
    (void)hipFree(c->UEND);
    (void)hipFree(c->W);
    (void)hipFree(c->W2);
    (void)hipFree(c->trail_nyq);
    (void)hipFree(c->cgw);
    (void)hipFree(c->bcols);
    (void)hipFree(c->bwts);
    (void)hipFree(c->gmw);
    (void)hipFree(c->SP);
    if (c->uend_ev) (void)hipEventDestroy(c->uend_ev);
    (void)hipFree(c->Sx);
    (void)hipFree(c->Wend);
    if (c->ring) (void)hipHostFree(c->ring);
    for (double* b : c->odd_buf) (void)hipFree(b);
    (void)hipFree(c->profile_compact);
    (void)hipFree(c->Gm);
    if (c->sl_ev) (void)hipEventDestroy(c->sl_ev);
    (void)hipFree(c->UEND2);
    (void)hipFree(c->tw);
    (void)hipFree(c->lamI);
    (void)hipFree(c->lamE);
    (void)hipFree(c->profile);
    (void)hipFree(c->red);
    (void)hipFree(c->counters);
    (void)hipFree(c->res_bank[0]);
    (void)hipFree(c->Wb);
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
    STORE_SPECTRA(c, false);  // (an iterate that was not stored is a function of the OLD symbol)
    c->g_sweeps = 0;          // (and so are the node multipliers on record)
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

int sdc_set_banded_operator(sdc_ctx* c, int n_interior, int width, const int* cols, const double* weights) {
    if (!c || !cols || !weights || n_interior < 1 || width < 1) return fail(c, SDC_ERR_PARAM, "bad banded operator");
    size_t Nb = 1;
    for (int d = 0; d < c->ndim; ++d) Nb *= (size_t)n_interior;
    if (Nb > c->N) return fail(c, SDC_ERR_PARAM, "the interior (%d per axis) does not fit the slab fields (%d per axis)", n_interior, c->n);
    for (size_t i = 0; i < (size_t)n_interior * width; ++i)
        if (cols[i] >= n_interior || cols[i] < -1) return fail(c, SDC_ERR_PARAM, "column %d out of range", cols[i]);
    (void)hipFree(c->bcols);
    (void)hipFree(c->bwts);
    c->bcols = nullptr;
    c->bwts = nullptr;
    const size_t cnt = (size_t)n_interior * width;
    HIPCHK(c, hipMalloc((void**)&c->bcols, cnt * sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->bwts, cnt * sizeof(double)));
    HIPCHK(c, hipMemcpyAsync(c->bcols, cols, cnt * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->bwts, weights, cnt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nb = n_interior;
    c->bw = width;
    c->Nb = Nb;
    return SDC_OK;
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

int sdc_set_vdp_block_solver(sdc_ctx* c, int kind) {
    if (!c) return SDC_ERR_PARAM;
    if (kind != 0 && kind != 1) return fail(c, SDC_ERR_PARAM, "block solver %d: 0 = closed form on the vector ALUs, 1 = MFMA", kind);
    c->vdp_block_solver = kind;
    return SDC_OK;
}

int sdc_work_counters(sdc_ctx* c, unsigned long long* out) {
    if (!c || !out) return SDC_ERR_PARAM;
    HIPCHK(c, hipMemcpyAsync(c->red_host + 12, c->counters, sizeof(unsigned long long) * 3, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 3; ++k) out[k] = c->red_host[12 + k];
    out[1] += c->rhs_host;
    out[3] = c->cg_iters;
    out[4] = c->gmres_iters;
    return SDC_OK;
}

int sdc_set_solver(sdc_ctx* c, int kind, double rtol, int maxiter) {
    if (!c) return SDC_ERR_PARAM;
    if (kind < 0 || kind > 2) return fail(c, SDC_ERR_PARAM, "solver kind %d (0: direct, 1: CG, 2: GMRES)", kind);
    c->solver_kind = kind;
    c->cg_rtol = rtol;
    c->cg_maxiter = maxiter;
    return SDC_OK;
}

static int vdp_check_failures(sdc_ctx* c) {
    unsigned long long v[5];
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
    STORE_SPECTRA(c, false);
    c->sym_real[which] = false;  // (a user-given table is used as it is)
    cd** dst = which == 0 ? &c->lamI : &c->lamE;
    if (!*dst) {
        HIPCHK(c, hipMalloc((void**)dst, sizeof(cd) * c->n));
        c->bytes += sizeof(cd) * c->n;
    }
    HIPCHK(c, hipMemcpyAsync(*dst, table, sizeof(cd) * c->n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_stencil[which] = true;
    if (which == 1) {
        if (c->ncomp != 2) return fail(c, SDC_ERR_PARAM, "explicit part needs ncomp == 2");
        c->expl_kind = SDC_EXPL_SYMBOL;
    }
    if (which == 0) {
        c->spectral_op = true;
        c->sym_table_real = true;
        c->sym_absmax = 0.0;
        for (int k = 0; k < c->n; ++k) {
            if (table[2 * k + 1] != 0.0) c->sym_table_real = false;
            const double m = hypot(table[2 * k], table[2 * k + 1]);
            if (m > c->sym_absmax) c->sym_absmax = m;
        }
    }
    c->spec_valid = c->spec0_valid = false;
    c->g_sweeps = 0;  // (multipliers of another symbol)
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
    if (!c || kind < 0 || kind > 4) return fail(c, SDC_ERR_PARAM, "bad explicit kind");
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
    c->specP_valid = false;
    if (c->profile_compact) {   // (made from the old profile)
        (void)hipFree(c->profile_compact);
        c->profile_compact = nullptr;
        c->bytes -= c->N * sizeof(double);
    }
    return SDC_OK;
}

int sdc_set_forcing_values(sdc_ctx* c, const double* g) {
    if (!c || !g) return fail(c, SDC_ERR_PARAM, "null forcing values");
    for (int m = 0; m <= c->M; ++m) c->gvals[m] = g[m];
    return SDC_OK;
}

extern "C" int sdc_invalidate_spectra(sdc_ctx* c, int which);
extern "C" int sdc_transfer_apply_batch(void* stream, int nfields, int ndim, int n_out, int n_in, int width, const int* idx,
                                        const double* w, const double* in, double* out);
extern "C" int sdc_materialize(sdc_ctx* c, int slot, int m);
extern "C" int sdc_solve(sdc_ctx* c, const double* rhs, double factor, const double* guess, double* out);
extern "C" int sdc_eval_f(sdc_ctx* c, const double* u, double g_t, double* f_impl, double* f_expl);

static int ensure_work(sdc_ctx* c) {
    FLUSH_X(c);               // (residual lines that still wait for their last pass)
    c->rlines_valid = false;  // whoever asks for the work spectra is about to overwrite them (a sweep says so again afterwards)
    if (!c->W) {
        const int nw = work_fields(c);
        HIPCHK(c, hipMalloc((void**)&c->W, sizeof(cd) * c->Nc * nw));
        c->bytes += sizeof(cd) * c->Nc * nw;
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

static int ensure_spec_cache(sdc_ctx* c) {
    if (!c->Sx) {
        // the start value's and the last node's spectrum are there from the start (they trade places from step to step); the
        // other node spectra are allocated when an iterate is first STORED (need_node_spectra) - sweeps that recompute theirs
        // never do
        HIPCHK(c, hipMalloc((void**)&c->Sx, sizeof(cd) * c->Nc));
        c->S0 = c->Sx;
        c->SL = spool_get(c);   // (its own buffer: it takes turns with the start value's spectrum)
        if (!c->SL) return SDC_ERR_NOMEM;
        c->bytes += sizeof(cd) * c->Nc;
        c->spec_valid = c->spec0_valid = false;
        if (sizeof(cd) * c->Nc < c->lazy_min_bytes) {
            int rcs = need_node_spectra(c);
            if (rcs != SDC_OK) return rcs;
        }
        c->S = c->Sn;
    }
    // the second end-value buffer of sdc_advance, allocated with the cache (not inside a time loop) - unless this is a
    // time-parallel level (keep_rfields / early_uend are switched on before its first sweep): those advance in place only as
    // spectra (sdc_advance allocates the buffer itself should it ever need it)
    if (!c->UEND2 && !c->keep_rfields && !c->early_uend) {
        HIPCHK(c, hipMalloc((void**)&c->UEND2, c->N * sizeof(double)));
        c->bytes += c->N * sizeof(double);
    }
    return SDC_OK;
}

static int launch_spread(sdc_ctx* c, int guess, double fill_u, double fill_f, bool reduce_f0) {
    SpreadArgs a;
    memset(&a, 0, sizeof a);
    NEED_NODES(c);
    U0R(c, u0p);
    a.u0 = u0p;
    a.f0 = c->F0;
    a.profile = c->profile;
    if (c->odd_n && c->profile && c->expl_kind == SDC_EXPL_FORCING) {
        // compact interior fields (sdc_set_odd_interior): the fill needs the profile in THEIR layout - the interior of the
        // extension-shaped one, extracted once
        if (!c->profile_compact) {
            HIPCHK(c, hipMalloc((void**)&c->profile_compact, c->N * sizeof(double)));
            HIPCHK(c, hipMemsetAsync(c->profile_compact, 0, c->N * sizeof(double), c->stream));
            c->bytes += c->N * sizeof(double);
            int rcx = sdc_odd_extract(c, c->profile, c->profile_compact, c->odd_n, c->ndim);
            if (rcx != SDC_OK) return rcx;
        }
        a.profile = c->profile_compact;
    }
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
    if (reduce_f0) a.f0max = c->res_dev + 7;
    {
        LaunchTimer lt(c, "spread");
        hipLaunchKernelGGL(k_spread, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, a);
    }
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// bring deferred real-space state up to date before it is read (or partially overwritten)
static int materialize_at(sdc_ctx* c, bool need_u, bool need_f, int line) {
    if ((need_u || need_f) && !c->Un && getenv("SDC_TRACE_LAZY")) fprintf(stderr, "[sdcmi] materialize called from line %d\n", line);
    if (need_u || need_f) NEED_NODES(c);
    if (c->spread_pending && (need_u || need_f)) {
        if (c->f0_pending) {  // the copies are copies of F[0]
            U0R(c, u0p);
            c->f0_pending = false;
            int rc0 = sdc_eval_f(c, u0p, c->gvals[0], c->F0, c->ncomp == 2 ? c->F0 + c->N : nullptr);
            if (rc0 != SDC_OK) return rc0;
        }
        c->spread_pending = false;
        c->f_pending = false;
        return launch_spread(c, SDC_GUESS_SPREAD, 0.0, 0.0, false);
    }
    if (c->u_pending && (need_u || (need_f && c->f_pending))) {
        FieldPtrs p;
        memset(&p, 0, sizeof p);
        for (int m = 0; m < c->M; ++m) p.out[m] = c->U + (size_t)(m + 1) * c->N;
        c->u_pending = false;
        c->rfields_valid = false;
        int rc = inverse_from_cache(c, 0, c->M, p);
        if (rc != SDC_OK) return rc;
    }
    if (c->f_pending && need_f) return eval_nodes_plain(c);
    return SDC_OK;
}

#define materialize(c, u, f) materialize_at(c, u, f, __LINE__)   // (SDC_TRACE_LAZY: who made the node fields real)
int sdc_materialize(sdc_ctx* c, int slot, int m) {
    if (!c) return SDC_ERR_PARAM;
    if (slot == SDC_SLOT_UEND) return materialize_uend(c);  // (readers of the end value: views, sends)
    if (slot < 0) MATERIALIZE_UEND(c);
    if (slot == SDC_SLOT_U && m == 0) {
        // whoever gets the address of U[0] may read or overwrite it: the start value has to BE there, and a pending
        // spread and a pending F[0] = f(U[0]) refer to the value it holds NOW
        ENSURE_U0(c);
        if (c->spread_pending) {
            int rc0 = materialize(c, true, false);
            if (rc0 != SDC_OK) return rc0;
        }
        if (c->f0_pending) {
            int rcf = need_f0(c);
            if (rcf != SDC_OK) return rcf;
            c->f0_pending = false;
            return sdc_eval_f(c, c->U0, c->gvals[0], c->F0, c->ncomp == 2 ? c->F0 + c->N : nullptr);
        }
        return SDC_OK;
    }
    if (slot == SDC_SLOT_F && m == 0) {
        {
            int rcf = need_f0(c);
            if (rcf != SDC_OK) return rcf;
        }
        if (!c->f0_pending) return SDC_OK;
        U0R(c, u0p);
        c->f0_pending = false;
        return sdc_eval_f(c, u0p, c->gvals[0], c->F0, c->ncomp == 2 ? c->F0 + c->N : nullptr);
    }
    NEED_NODES(c);
    if (slot < 0) ENSURE_U0(c);  // "everything": all of the real-space state is about to be used as it is stored
    if (slot < 0 && c->f0_pending) {
        c->f0_pending = false;
        int rc0 = sdc_eval_f(c, c->U0, c->gvals[0], c->F0, c->ncomp == 2 ? c->F0 + c->N : nullptr);
        if (rc0 != SDC_OK) return rc0;
    }
    // a node value that is handed out may be overwritten by the holder: F[1..M] = f(U[1..M]) of the CURRENT node
    // values must be stored before that can happen (the reference's L.f[m] does not follow a later L.u[m] = x)
    return materialize(c, slot == SDC_SLOT_U || slot < 0, slot == SDC_SLOT_F || slot < 0 || c->f_pending);
}

int sdc_set_deferred(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    c->deferred = on != 0;
    return on ? SDC_OK : sdc_materialize(c, -1, -1);
}

int sdc_defer_f0(sdc_ctx* c) {
    if (!c) return SDC_ERR_PARAM;
    c->f0_pending = true;
    return c->deferred ? SDC_OK : sdc_materialize(c, SDC_SLOT_F, 0);
}

void* sdc_uend_address(sdc_ctx* c) { return c ? (void*)c->UEND : nullptr; }

long long sdc_end_value_generation(sdc_ctx* c) {
    // > 0 while the end value (UEND, or its put-off transform) still IS the last node of the cached iterate, i.e. while
    // sdc_advance could hand it over; changes with every sweep.  A caller that gave a COPY of the end value away can tell
    // later whether the engine still holds exactly that state (pysdc_amd/controller.py: consecutive run() calls).
    if (!c || !c->Sx || !c->spec_valid || c->uend_gen < 0 || c->uend_gen != c->spec_gen) return -1;
    return c->spec_gen + 1;
}

void* sdc_slot_ptr(sdc_ctx* c, int slot, int m, int comp) {
    if (!c) return nullptr;
    // whoever asks for the address of a node field is about to read or write it
    if ((slot == SDC_SLOT_U || slot == SDC_SLOT_F) && sdc_materialize(c, slot, m) != SDC_OK) return nullptr;
    if ((slot == SDC_SLOT_U && m != 0) || (slot == SDC_SLOT_F && m != 0)) {
        if (!c->Un && getenv("SDC_TRACE_LAZY")) fprintf(stderr, "[sdcmi] node fields allocated by sdc_slot_ptr(%d, %d)\n", slot, m);
        if (need_nodes(c) != SDC_OK) return nullptr;
    }
    if (slot == SDC_SLOT_F && m == 0 && need_f0(c) != SDC_OK) return nullptr;
    if (slot == SDC_SLOT_TAU && ensure_tau(c) != SDC_OK) return nullptr;
    if (slot == SDC_SLOT_UEND) {
        if (materialize_uend(c) != SDC_OK) return nullptr;
        c->uend_gen = -1;  // the holder may write it
        if (uend_write_fence(c) != SDC_OK) return nullptr;
    }
    if (slot == SDC_SLOT_WORK) return c->W;
    return slot_ptr(c, slot, m, comp);
}

int sdc_invalidate_spectra(sdc_ctx* c, int which) {
    if (!c) return SDC_ERR_PARAM;
    if ((which & 2) && c->u_pending)
        return fail(c, SDC_ERR_STATE, "U[1..M] were deferred: sdc_materialize before writing through a kept pointer");
    if (which & ~8) {  // (the end value does not enter the residual)
        c->res_valid = false;
        c->res_spread = false;
    }
    if (which & 1) {
        c->spec0_valid = false;
        c->spec_spread = false;  // "all nodes equal U[0]" no longer holds for the new U[0]
    }
    if (which & 2) {
        c->spec_valid = false;
        c->spec_spread = false;
        // a node value was replaced but not (yet) its right-hand side: the reference's next sweep integrates the
        // stored f (generic_implicit.py:75), so gather on the F slab instead of assuming F = f(U)
        c->force_gather = true;
    }
    if (which & 4) c->force_gather = true;  // some F[m >= 1] no longer equals f(U[m]): gather on F itself
    if (which & 8) {  // UEND was overwritten
        c->uend_gen = -1;
        c->uend_pending = false;
    }
    return SDC_OK;
}

int sdc_set_fused_residual(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    c->fuse_residual = on != 0;
    c->res_valid = false;
    return SDC_OK;
}

int sdc_set_virtual_sweeps(sdc_ctx* c, int max_sweeps) {
    if (!c || max_sweeps < 0) return fail(c, SDC_ERR_PARAM, "bad number of sweeps");
    if (max_sweeps == 0) STORE_SPECTRA(c, false);
    c->virt_max = max_sweeps;
    return SDC_OK;
}

int sdc_set_lazy_predictor_residual(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    c->lazy_f0norm = on != 0;
    return SDC_OK;
}

int sdc_residual_deferred(sdc_ctx* c) {
    if (!c) return SDC_ERR_PARAM;
    return (c->res_spread && c->f0norm_pending && !c->res_valid) ? 1 : 0;
}

int sdc_set_multiplier_table(sdc_ctx* c, int from_sweep) {
    if (!c || from_sweep < 0) return fail(c, SDC_ERR_PARAM, "multiplier table: first sweep that uses it (0: never)");
    c->g_from = from_sweep;
    c->g_sweeps = 0;
    return SDC_OK;
}

int sdc_set_skip_residual(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    c->skip_residual = on != 0;
    return SDC_OK;
}

int sdc_set_unlocked(sdc_ctx* c, int unlocked) {
    if (!c) return SDC_ERR_PARAM;
    c->unlocked = unlocked != 0;
    return SDC_OK;
}

int sdc_set_spectral_reuse(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    MATERIALIZE_UEND(c);                    // ... and of a put-off end value
    if (c->spread_pending || c->u_pending) {   // the cache may be the only holder of U[1..M]
        int rcm = materialize(c, true, false);
        if (rcm != SDC_OK) return rcm;
    }
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

int sdc_set_odd_interior(sdc_ctx* c, int n_interior) {
    if (!c) return SDC_ERR_PARAM;
    if (n_interior != 0 && (c->ndim < 2 || c->n != 2 * (n_interior + 1)))
        return fail(c, SDC_ERR_PARAM, "odd extension of %d interior points needs an engine grid of %d per axis in 2-D / 3-D, got %d (%d-D)",
                    n_interior, 2 * (n_interior + 1), c->n, c->ndim);
    c->odd_n = n_interior;
    return SDC_OK;
}
static int odd_scratch(sdc_ctx* c, int k) {
    if (!c->odd_buf[k]) {
        HIPCHK(c, hipMalloc((void**)&c->odd_buf[k], c->N * sizeof(double)));
        c->bytes += c->N * sizeof(double);
    }
    return SDC_OK;
}
struct OddScope {   // the calls between construction and destruction see extension-sized fields
    sdc_ctx* c;
    explicit OddScope(sdc_ctx* c_) : c(c_) { c->odd_busy = true; }
    ~OddScope() { c->odd_busy = false; }
};

int sdc_eval_f(sdc_ctx* c, const double* u, double g_t, double* f_impl, double* f_expl) {
    if (!c || !u || !f_impl) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (c->odd_n && !c->odd_busy) {   // compact interior fields: through the odd extension and back
        for (int k = 0; k < (f_expl ? 3 : 2); ++k) {
            int rs = odd_scratch(c, k);
            if (rs != SDC_OK) return rs;
        }
        int rc = sdc_odd_extend(c, u, c->odd_buf[0], c->odd_n, c->ndim);
        if (rc != SDC_OK) return rc;
        {
            OddScope scope(c);
            rc = sdc_eval_f(c, c->odd_buf[0], g_t, c->odd_buf[1], f_expl ? c->odd_buf[2] : nullptr);
        }
        if (rc != SDC_OK) return rc;
        rc = sdc_odd_extract(c, c->odd_buf[1], f_impl, c->odd_n, c->ndim);
        if (rc == SDC_OK && f_expl) rc = sdc_odd_extract(c, c->odd_buf[2], f_expl, c->odd_n, c->ndim);
        return rc;
    }
    if (c->kind == 1) {
        LaunchTimer lt(c, "vdp_eval");
        hipLaunchKernelGGL(k_vdp_eval, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, u, f_impl, c->N / 2,
                           c->vdp_mu, c->counters);
        HIPCHK(c, hipGetLastError());
        return SDC_OK;
    }
    if (c->nb) {  // bounded grid, row-dependent stencils; the only explicit part there is: a u-independent forcing profile(x) g(t)
        if (f_expl) {
            if (c->expl_kind != SDC_EXPL_FORCING || !c->profile)
                return fail(c, SDC_ERR_UNSUPPORTED, "the banded operator evaluates the implicit part (and a forcing profile) only");
            // (compact fields: Nb values - the caller's f_expl need not be longer than that)
            int rcp = sdc_vec_axpby(c, problem_size(c), g_t, c->profile, 0.0, nullptr, f_expl);
            if (rcp != SDC_OK) return rcp;
        }
        return apply_operator(c, u, f_impl);
    }
    if (!c->have_stencil[0]) return fail(c, SDC_ERR_STATE, "implicit operator not set (sdc_set_stencil)");
    if (c->expl_kind == SDC_EXPL_FORCING && !c->profile) return fail(c, SDC_ERR_STATE, "forcing profile not set");
    if (c->spectral_op || c->expl_kind == SDC_EXPL_REACTION) {
        bool react_done = false;
        if (c->spectral_op) {
            FieldPtrs p;
            memset(&p, 0, sizeof p);
            ZArgs z;
            memset(&z, 0, sizeof z);
            p.in[0] = u;
            p.out[0] = f_impl;
            z.apply = 1;
            ReactReq rq;
            if (f_expl && c->expl_kind == SDC_EXPL_REACTION && c->ndim >= 2 && f_expl != u) {
                rq.where = 1;  // the explicit part is a function of the very values the first pass reads
                rq.out = f_expl;
                react_done = true;
            }
            int nfe = 1;
            if (f_expl && c->expl_kind == SDC_EXPL_SYMBOL) {   // both parts from ONE forward transform
                if (!c->lamE) return fail(c, SDC_ERR_STATE, "explicit symbol not set (sdc_set_symbol which=1)");
                if (work_fields(c) < 2) return fail(c, SDC_ERR_UNSUPPORTED, "two work spectra needed (num_nodes >= 2)");
                p.out[1] = f_expl;
                z.dup = 1;
                nfe = 2;
            }
            int rc0 = fft_pipeline(c, nfe, p, z, rq);
            if (rc0 != SDC_OK) return rc0;
        } else {
            const double* in1[1] = {u};
            double* oi1[1] = {f_impl};
            int rc0 = run_stencil(c, 1, in1, oi1, nullptr, nullptr);
            if (rc0 != SDC_OK) return rc0;
        }
        if (f_expl && c->expl_kind == SDC_EXPL_REACTION && !react_done) {
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

int sdc_eval_f_batch(sdc_ctx* c, int nf, const double* const* u, const double* g_t, double* const* f_impl,
                     double* const* f_expl) {
    if (!c || !u || !f_impl || nf < 1) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (nf > c->M) return fail(c, SDC_ERR_PARAM, "at most num_nodes = %d fields per call (the work spectra), got %d", c->M, nf);
    for (int f = 0; f < nf; ++f)
        if (!u[f] || !f_impl[f]) return fail(c, SDC_ERR_PARAM, "null field pointer");
    const bool batched_spectral = c->kind == 0 && !c->nb && !c->odd_n && c->spectral_op && c->ndim >= 2 &&
                                  (c->expl_kind == SDC_EXPL_NONE || c->expl_kind == SDC_EXPL_REACTION);
    const bool batched_stencil = c->kind == 0 && !c->nb && !c->odd_n && !c->spectral_op && c->have_stencil[0] &&
                                 c->expl_kind != SDC_EXPL_REACTION && !(c->expl_kind == SDC_EXPL_FORCING && !c->profile);
    if (batched_spectral) {
        // ONE transform round trip for all fields: forward passes, the symbol on the contiguous axis, inverse passes; the
        // pointwise reaction term of every field rides on the pass that reads it
        FieldPtrs p;
        memset(&p, 0, sizeof p);
        ZArgs z;
        memset(&z, 0, sizeof z);
        ReactReq rq;
        bool react = c->expl_kind == SDC_EXPL_REACTION && f_expl != nullptr;
        for (int f = 0; f < nf; ++f) {
            p.in[f] = u[f];
            p.out[f] = f_impl[f];
            if (react && (!f_expl[f] || f_expl[f] == u[f])) react = false;
        }
        z.apply = 1;
        if (react) {
            rq.where = 1;
            rq.field = -1;
            for (int f = 0; f < nf; ++f) rq.outs[f] = f_expl[f];
        }
        int rc0 = fft_pipeline(c, nf, p, z, rq);
        if (rc0 != SDC_OK) return rc0;
        if (!react && f_expl && c->expl_kind == SDC_EXPL_REACTION)
            for (int f = 0; f < nf; ++f)
                if (f_expl[f]) {
                    LaunchTimer lt(c, "reaction");
                    hipLaunchKernelGGL(k_reaction, dim3(grid_for(c->N, 256)), dim3(256), 0, c->stream, u[f], f_expl[f], c->N,
                                       c->react_kind, c->react_p0, c->react_p1, c->react_nu);
                    HIPCHK(c, hipGetLastError());
                }
        return SDC_OK;
    }
    if (batched_stencil) {
        const double* in[MAXM];
        double *oi[MAXM], *oe[MAXM], g[MAXM];
        bool expl = f_expl != nullptr;
        for (int f = 0; f < nf; ++f) {
            in[f] = u[f];
            oi[f] = f_impl[f];
            oe[f] = f_expl ? f_expl[f] : nullptr;
            g[f] = g_t ? g_t[f] : 0.0;
            if (expl && !oe[f]) expl = false;
        }
        if (expl || !f_expl) return run_stencil(c, nf, in, oi, expl ? oe : nullptr, g);
    }
    for (int f = 0; f < nf; ++f) {   // every other kind of level: field by field
        int rc = sdc_eval_f(c, u[f], g_t ? g_t[f] : 0.0, f_impl[f], f_expl ? f_expl[f] : nullptr);
        if (rc != SDC_OK) return rc;
    }
    return SDC_OK;
}

int sdc_predict(sdc_ctx* c, double t, double dt, int guess, double fill_u, double fill_f) {
    (void)t;
    (void)dt;
    if (!c) return SDC_ERR_PARAM;
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    if (guess < 0 || guess > 3) return fail(c, SDC_ERR_PARAM, "initial_guess option %d not implemented", guess);
    FLUSH_X(c);
    // a put-off end value is the inverse transform of the iterate this predictor is about to drop (the reference's
    // predict leaves L.uend alone, core/sweeper.py:125-162)
    MATERIALIZE_UEND(c);
    // all nodes equal u0 and f does not depend on t: every f_j equals f(u0), so the node residuals are
    // dt * |sum_j Q[m][j]| * max|f(u0)| and the fill kernel can reduce max|f(u0)| on the way
    const bool spread_res = guess == SDC_GUESS_SPREAD && c->expl_kind != SDC_EXPL_FORCING && !c->tau_active;
    if (spread_res) HIPCHK(c, hipMemsetAsync(c->res_dev, 0, sizeof(unsigned long long) * 8, c->stream));
    c->spread_pending = c->f_pending = c->u_pending = c->f0_pending = c->rfields_valid = false;
    c->f0norm_pending = false;
    // (van der Pol ensemble: the first sweep reads U[0] alone, sdc_sweep)
    const bool lazy_spread = c->deferred && (c->kind == 0 || (c->kind == 1 && !c->tau_active)) && guess == SDC_GUESS_SPREAD &&
                             c->expl_kind != SDC_EXPL_FORCING;
    if (!lazy_spread || c->kind == 1) ENSURE_U0(c);  // (the deferred spread only READS the start value: wherever it lies)
    // with the 3-D three-point kernel even f(u0) itself is not stored: one pass over u0 that only reduces max|f(u0)|
    auto three = [](const Stencil& s) { return s.npts == 3 && s.off[0] == -1 && s.off[1] == 0 && s.off[2] == 1; };
    const bool explS = c->expl_kind == SDC_EXPL_STENCIL;
    const bool f0_by_norm = lazy_spread && spread_res && !c->spectral_op && !c->odd_n && c->ndim == 3 && c->n % 64 == 0 &&
                            c->have_stencil[0] && three(c->st[0]) &&
                            (c->expl_kind == SDC_EXPL_NONE || (explS && c->have_stencil[1] && three(c->st[1])));
    int rc = SDC_OK;
    bool f0_max_done = false;
    if (f0_by_norm && c->u0_spec_only && c->spec0_valid && c->S0 && c->expl_kind != SDC_EXPL_FORCING) {
        // the start value exists as its transform only: max |f(u0)| from the norm-only inverse transform of symbol * S0 -
        // when somebody asks for the residual of the predictor's state (sdc_residual; sdc_residual_deferred tells whether
        // asking costs that transform): three passes over one field that a run with fixed sweep counts never needs
        if (c->lazy_f0norm) {
            c->f0norm_pending = true;
        } else {
            rc = symbol_norm(c, c->S0, c->res_dev + 7);
            if (rc != SDC_OK) return rc;
        }
        c->f0_pending = true;
    } else if (f0_by_norm) {
        constexpr int RPT = 4;
        Stencil3Args s3;
        memset(&s3, 0, sizeof s3);
        U0R(c, u0p);
        s3.in[0] = u0p;
        for (int k = 0; k < 3; ++k) {
            s3.wI[k] = c->st[0].w[k];
            s3.wE[k] = explS ? c->st[1].w[k] : 0.0;
        }
        s3.useE = explS ? 1 : 0;
        s3.fmax = c->res_dev + 7;
        s3.n = c->n;
        s3.ntiles = (c->n / 64) * (c->n / (8 * RPT));
        s3.xchunk = stencil_xchunk(c->n, (size_t)s3.ntiles);
        s3.nchunks = c->n / s3.xchunk;
        {
            LaunchTimer lt(c, "stencil_max");
            hipLaunchKernelGGL((k_stencil3d<RPT>), dim3(s3.ntiles * s3.nchunks), dim3(256), 0, c->stream, s3);
        }
        HIPCHK(c, hipGetLastError());
        c->f0_pending = true;
    } else {
        U0R(c, u0p);
        if (c->kind == 1 && lazy_spread && spread_res) {  // max |f(u0)| on the way (no second pass over F[0])
            rc = need_f0(c);
            if (rc != SDC_OK) return rc;
            LaunchTimer lt(c, "vdp_eval");
            hipLaunchKernelGGL(k_vdp_eval, dim3(grid_for(c->N / 4, 256)), dim3(256), 0, c->stream, u0p, c->F0, c->N / 2,
                               c->vdp_mu, c->counters, c->res_dev + 7);
            HIPCHK(c, hipGetLastError());
            f0_max_done = true;
        } else {
            rc = need_f0(c);
            if (rc == SDC_OK) rc = sdc_eval_f(c, u0p, c->gvals[0], c->F0, c->ncomp == 2 ? c->F0 + c->N : nullptr);
            if (rc != SDC_OK) return rc;
        }
    }
    if (lazy_spread) {
        // the node copies are not stored until somebody reads them (materialize); only max|f(u0)| is needed now
        if (spread_res && !f0_by_norm && !f0_max_done) {
            LaunchTimer lt(c, "amax");
            hipLaunchKernelGGL(k_amax_sum, dim3(grid_for(c->N, 256)), dim3(256), 0, c->stream, c->F0,
                               c->ncomp == 2 ? c->F0 + c->N : nullptr, c->N, c->res_dev + 7);
            HIPCHK(c, hipGetLastError());
        }
        c->spread_pending = true;
    } else {
        rc = launch_spread(c, guess, fill_u, fill_f, spread_res);
        if (rc != SDC_OK) return rc;
    }
    // 'spread' evaluates f at every node in the reference (core/sweeper.py:142-143); the engine copies F[0]
    if (c->kind == 1 && guess == SDC_GUESS_SPREAD) c->rhs_host += (unsigned long long)c->M * (c->N / 2);
    c->unlocked = true;
    c->res_valid = false;
    c->res_spread = spread_res;
    // forcing: the transformed sweep takes the explicit values of node j to be profile * g(t_j), which the stored ones
    // are after a spread predictor or any sweep - not after 'copy' / 'zero' / constant fills
    if (c->expl_kind == SDC_EXPL_FORCING && guess != SDC_GUESS_SPREAD) c->force_gather = true;
    c->spec_valid = false;
    c->spec_spread = (guess == SDC_GUESS_SPREAD || guess == SDC_GUESS_COPY);  // all nodes equal U[0]
    return SDC_OK;
}

// Node-by-node sweep on the device for right-hand sides that are not linear in u (pointwise reaction terms) or
// whose implicit operator is given by its symbol only: the reference's loop (imex_1st_order.py:57-108 /
// generic_implicit.py:51-103) with every step a kernel on the stream - gather for all nodes, then per node
// right-hand side, solve (FFT pipeline), f evaluation (operator by FFT or stencil + explicit part).
// keep_guess: an iterative solver starts from the previous node value (generic_ND_FD.py:252-260, x0 = u0), so the
// right-hand sides are gathered into scratch instead of over the old iterate.
static int sweep_nodewise(sdc_ctx* c, double dt, bool keep_guess = false) {
    const int M = c->M;
    const bool imex = c->ncomp == 2;
    int rcm = materialize(c, true, true);
    if (rcm != SDC_OK) return rcm;
    double* G = nullptr;
    if (keep_guess) {
        rcm = ensure_work(c);
        if (rcm != SDC_OK) return rcm;
        G = reinterpret_cast<double*>(c->W);  // M spectra hold more than M real fields
    }
    QuadArgs q;
    quad_base(c, q);
    q.u0 = c->U0;
    q.tau = c->tau_active ? c->TAU : nullptr;
    for (int m = 0; m < M; ++m) {
        // without keep_guess the old iterate is only a solver guess: its storage takes the right-hand side
        q.out[m] = keep_guess ? G + (size_t)m * c->N : c->U + (size_t)(m + 1) * c->N;
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
        double* rhs = keep_guess ? G + (size_t)m * c->N : um;
        const double alpha = dt * c->QI[m + 1][m + 1];
        // operator given by its symbol, direct solve: solve and implicit evaluation share one forward transform - and the
        // terms of the nodes before this one are added by that transform's first pass as it reads the gathered field
        // (a complex symbol in more than one dimension: solve and evaluation each go through real space - the projection onto
        // real fields between them is part of the result, see k_fftz_solve)
        const bool shared_fwd = c->spectral_op && !keep_guess && c->solver_kind == 0 && work_fields(c) >= 2 &&  // (two work spectra needed)
                                (c->ndim == 1 || (c->sym_table_real && c->expl_kind != SDC_EXPL_SYMBOL));
        const bool rhs_on_the_way = shared_fwd && c->ndim >= 2 && SDC_FUSE_NODE_RHS;
        LinTerms lin;
        memset(&lin, 0, sizeof lin);
        if (m > 0) {
            LinArgs la;
            memset(&la, 0, sizeof la);
            la.out = rhs;
            la.base = rhs;
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
            if (la.nterms > 0 && rhs_on_the_way) {
                for (int k = 0; k < la.nterms; ++k) {
                    lin.x[k] = la.x[k];
                    lin.c[k] = la.c[k];
                }
                lin.n = la.nterms;
            } else if (la.nterms > 0) {
                LaunchTimer lt(c, "node_rhs");
                hipLaunchKernelGGL(k_lincomb, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, la);
                HIPCHK(c, hipGetLastError());
            }
        }
#ifndef SDC_IMPL_FROM_SOLVE
#define SDC_IMPL_FROM_SOLVE 1
#endif
        // ... and when the symbol is real and alpha |symbol|max >= 1, the implicit part at the new value is (u - rhs) / alpha:
        // the solve's own equation, evaluated by the last pass as it writes u - ONE spectrum through the z / y passes instead
        // of two (rounding error eps |u| / alpha <= eps |symbol|max |u|: that of the transformed evaluation).  The first pass
        // stores the completed right-hand side where the last one reads it.
        const bool impl_from_solve = SDC_IMPL_FROM_SOLVE && shared_fwd && c->ndim >= 2 && c->sym_table_real && alpha > 0.0 &&
                                     alpha * c->sym_absmax * c->ndim >= 1.0 && (!imex || c->expl_kind == SDC_EXPL_REACTION);
        if (impl_from_solve) {
            FieldPtrs p1;
            memset(&p1, 0, sizeof p1);
            ZArgs z1;
            memset(&z1, 0, sizeof z1);
            p1.in[0] = rhs;
            p1.out[0] = um;
            z1.alpha[0] = alpha;
            ReactReq rq1;
            rq1.where = 2;
            if (imex) rq1.out = c->F + ((size_t)(m + 1) * c->ncomp + 1) * c->N;
            rq1.impl_out = c->F + ((size_t)(m + 1) * c->ncomp) * c->N;
            rq1.rhs = rhs;
            rq1.inv_alpha = 1.0 / alpha;
            if (lin.n) lin.wb = rhs;
            rc = fft_pipeline(c, 1, p1, z1, rq1, lin.n ? &lin : nullptr);
            if (rc != SDC_OK) return rc;
            continue;
        }
        if (shared_fwd) {
            // operator given by its symbol: the solve and the evaluation of the implicit part at the new value share
            // one forward transform (u_hat and symbol * u_hat leave the spectral pass together)
            FieldPtrs p2;
            memset(&p2, 0, sizeof p2);
            ZArgs z2;
            memset(&z2, 0, sizeof z2);
            p2.in[0] = rhs;
            p2.out[0] = um;
            p2.out[1] = c->F + ((size_t)(m + 1) * c->ncomp) * c->N;
            z2.alpha[0] = alpha;
            z2.dup = 1;
            const bool esym = imex && c->expl_kind == SDC_EXPL_SYMBOL;
            if (imex && c->expl_kind != SDC_EXPL_REACTION && !esym)
                return fail(c, SDC_ERR_UNSUPPORTED, "explicit part of a symbol-only operator must be a reaction term or a symbol");
            if (esym && work_fields(c) < 3) {
                // (fewer than three work spectra: solve + implicit part now, the explicit operator by an evaluation of its own)
                rc = fft_pipeline(c, 2, p2, z2, ReactReq(), lin.n ? &lin : nullptr);
                if (rc == SDC_OK) {
                    FieldPtrs pe;
                    memset(&pe, 0, sizeof pe);
                    ZArgs ze;
                    memset(&ze, 0, sizeof ze);
                    pe.in[0] = um;
                    pe.out[0] = p2.out[1];   // (the implicit part once more: same bits) ...
                    pe.out[1] = c->F + ((size_t)(m + 1) * c->ncomp + 1) * c->N;   // ... and the explicit one
                    ze.apply = 1;
                    ze.dup = 1;
                    rc = fft_pipeline(c, 2, pe, ze);
                }
                if (rc != SDC_OK) return rc;
                continue;
            }
            if (esym) p2.out[2] = c->F + ((size_t)(m + 1) * c->ncomp + 1) * c->N;
            ReactReq rq2;
            if (imex && !esym && c->ndim >= 2) {  // f_expl(u_m) leaves the last inverse pass together with u_m
                rq2.where = 2;
                rq2.out = c->F + ((size_t)(m + 1) * c->ncomp + 1) * c->N;
            }
            rc = fft_pipeline(c, esym ? 3 : 2, p2, z2, rq2, lin.n ? &lin : nullptr);
            if (rc != SDC_OK) return rc;
            if (imex && !esym && rq2.where == 0) {
                LaunchTimer lt(c, "reaction");
                hipLaunchKernelGGL(k_reaction, dim3(grid_for(c->N, 256)), dim3(256), 0, c->stream, um,
                                   c->F + ((size_t)(m + 1) * c->ncomp + 1) * c->N, c->N, c->react_kind, c->react_p0,
                                   c->react_p1, c->react_nu);
                HIPCHK(c, hipGetLastError());
            }
            continue;
        }
        if (alpha != 0.0 || (keep_guess && imex)) {  // imex_1st_order always solves (imex_1st_order.py:98-103)
            rc = sdc_solve(c, rhs, alpha, um, um);
            if (rc != SDC_OK) return rc;
        } else if (keep_guess) {
            rc = sdc_vec_copy(c, c->N, rhs, um);
            if (rc != SDC_OK) return rc;
        }
        rc = sdc_eval_f(c, um, c->gvals[m + 1], c->F + ((size_t)(m + 1) * c->ncomp) * c->N,
                        imex ? c->F + ((size_t)(m + 1) * c->ncomp + 1) * c->N : nullptr);
        if (rc != SDC_OK) return rc;
    }
    return SDC_OK;
}

#ifndef SDC_STORE_LAST_IN_SWEEP
#define SDC_STORE_LAST_IN_SWEEP 1
#endif
int sdc_sweep(sdc_ctx* c, double t, double dt) {
    (void)t;
    if (!c) return SDC_ERR_PARAM;
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    if (!c->unlocked) return fail(c, SDC_ERR_STATE, "level is locked: predict first (assert L.status.unlocked)");
    const int M = c->M;
    c->res_valid = false;
    c->res_host_valid = false;
    c->res_spread = false;
    c->rfields_valid = false;
    if (c->kind == 1) {
        NEED_NODES(c);
        VdpSweepArgs a;
        memset(&a, 0, sizeof a);
        a.U = c->U;
        a.u0 = c->U0;
        a.F = c->F;
        a.tau = c->tau_active ? c->TAU : nullptr;
        a.T = c->N / 2;
        a.mu = c->vdp_mu;
        a.dt = dt;
        a.tol = c->vdp_tol;
        a.maxiter = c->vdp_maxiter;
        a.counters = c->counters;
        a.norms = c->fuse_residual ? c->res_dev : nullptr;
        if (a.norms) HIPCHK(c, hipMemsetAsync(c->res_dev, 0, sizeof(unsigned long long) * 8, c->stream));
        for (int m = 0; m < M; ++m)
            for (int j = 0; j < M; ++j) {
                a.Q[m][j] = dt * c->Q[m + 1][j + 1];
                a.QI[m][j] = dt * c->QI[m + 1][j + 1];
            }
#ifndef SDC_VDP_FULL_GRID
#define SDC_VDP_FULL_GRID 0  // 1: one trajectory per thread (no grid-stride loop)
#endif
#ifndef SDC_VDP_GRID_CAP
#define SDC_VDP_GRID_CAP 4096  // workgroups of the sweep launch at most (grid-stride loop over the trajectories)
#endif
        int grid = SDC_VDP_FULL_GRID ? (int)((a.T + 255) / 256) : grid_for(a.T, 256);
        if (!SDC_VDP_FULL_GRID && grid > SDC_VDP_GRID_CAP) grid = SDC_VDP_GRID_CAP;
        // F[1..M] = f(U[1..M]) unless somebody overwrote an F field (force_gather): recompute instead of reading,
        // and leave the new values to sdc_materialize
        if (c->force_gather) {
            int rcm = materialize(c, false, true);
            if (rcm != SDC_OK) return rcm;
        }
        const bool lazyf = c->deferred && !c->force_gather;
        c->force_gather = false;
        if (c->spread_pending) {
            // node copies of a spread predictor that were never written: the closed-form kernel takes U[0] for all of them
            if (lazyf && c->vdp_block_solver != 1) {
                a.spread = 1;
                c->spread_pending = false;
            } else {
                int rcm = materialize(c, true, true);
                if (rcm != SDC_OK) return rcm;
            }
        }
        if (c->vdp_block_solver == 1 && lazyf) {
            LaunchTimer lt(c, "vdp_sweep_mfma");
            const size_t ldsv = (size_t)4 * 64 * 10 * sizeof(double);
#define VMCASE(MM) \
    case MM: hipLaunchKernelGGL((k_vdp_sweep_mfma<MM>), dim3(grid), dim3(256), ldsv, c->stream, a); break;
            switch (M) {
                VMCASE(1) VMCASE(2) VMCASE(3) VMCASE(4) VMCASE(5) VMCASE(6) VMCASE(7) VMCASE(8)
            }
#undef VMCASE
        } else {
            LaunchTimer lt(c, lazyf ? "vdp_sweep_lazyf" : "vdp_sweep");
#define VCASE(MM)                                                                                       \
    case MM:                                                                                            \
        if (lazyf) hipLaunchKernelGGL((k_vdp_sweep<MM, true>), dim3(grid), dim3(256), 0, c->stream, a); \
        else hipLaunchKernelGGL((k_vdp_sweep<MM, false>), dim3(grid), dim3(256), 0, c->stream, a);      \
        break;
            switch (M) {
                VCASE(1) VCASE(2) VCASE(3) VCASE(4) VCASE(5) VCASE(6) VCASE(7) VCASE(8)
            }
#undef VCASE
        }
        HIPCHK(c, hipGetLastError());
        if (a.norms) {
            c->res_valid = true;
            c->res_dt = dt;
        }
        c->f_pending = lazyf;
        if (a.norms) {  // one round trip for the failure count and the node norms (sdc_residual then needs none)
            HIPCHK(c, hipMemcpyAsync(c->red_host, c->res_dev, sizeof(unsigned long long) * 8, hipMemcpyDeviceToHost, c->stream));
            int rcv = vdp_check_failures(c);  // (synchronises the stream)
            memcpy(c->res_host, c->red_host, sizeof(unsigned long long) * 8);
            c->res_host_valid = rcv == SDC_OK;
            return rcv;
        }
        return vdp_check_failures(c);
    }
    if (c->expl_kind == SDC_EXPL_STENCIL && !c->have_stencil[1])
        return fail(c, SDC_ERR_STATE, "explicit operator not set (sdc_set_stencil which=1)");
    MATERIALIZE_UEND(c);  // (a put-off end value belongs to the iterate this sweep is about to replace)
    {
        // only a sweep that stays in Fourier space and stores nothing in real space leaves the start value where it
        // lies (u0_src); every other data flow reads the U[0] slab
        const bool fourier_only = c->reuse && !c->force_gather && !c->tau_active && c->have_stencil[0] && fourier_ok(c) &&
                                  c->expl_kind != SDC_EXPL_REACTION && !c->spectral_op && c->solver_kind == 0 && !c->odd_n &&
                                  c->deferred && c->ndim >= 2 && (c->skip_residual || c->fuse_residual);
        if (!fourier_only) ENSURE_U0(c);
    }
    if (c->expl_kind == SDC_EXPL_REACTION || c->spectral_op) return sweep_nodewise(c, dt);
    if (c->nb) return sweep_nodewise(c, dt, true);   // bounded grid, row table: iterative node solves (guess = old node value)
    if (c->odd_n) return sweep_nodewise(c, dt, c->solver_kind != 0);   // compact fields, solves / evaluations through the odd extension
    if (c->solver_kind != 0 || !fourier_ok(c)) return sweep_nodewise(c, dt, true);
    const bool gather_once = c->force_gather;
    c->force_gather = false;
    if (c->reuse && !gather_once && !c->tau_active && c->have_stencil[0] &&
        fourier_ok(c)) {
        // ---- spectral reuse: f is linear in u, so the gather happens on the cached transforms ----
        {
            int rca = ensure_spec_cache(c);
            if (rca != SDC_OK) return rca;
        }
        FieldPtrs p;
        memset(&p, 0, sizeof p);
        if (!c->spec_valid) c->spec_virtual = 0;
        if (!c->spec0_valid) {
            STORE_SPECTRA(c, false);  // (an iterate that was not stored is a function of the OLD S0)
            U0R(c, u0p);
            p.in[0] = u0p;
            int rc0 = fwd_transform(c, 1, p, c->S0, 0);
            if (rc0 != SDC_OK) return rc0;
            c->spec0_valid = true;
        }
        if (!c->spec_valid && !c->spec_spread) {
            int rcm = materialize(c, true, false);
            if (rcm != SDC_OK) return rcm;
            // the first M-1 nodes into the strided block, the last one to wherever its spectrum lives right now
            for (int m = 0; m < M - 1; ++m) p.in[m] = c->U + (size_t)(m + 1) * c->N;
            if (M > 1) {
                int rcn = need_node_spectra(c);   // (the iterate is stored in the node spectra from here on)
                if (rcn != SDC_OK) return rcn;
                int rc0 = fwd_transform(c, M - 1, p, c->S, c->Nc);
                if (rc0 != SDC_OK) return rc0;
            }
            memset(&p, 0, sizeof p);
            p.in[0] = c->U + (size_t)M * c->N;
            int rc1 = fwd_transform(c, 1, p, c->SL, 0);
            if (rc1 != SDC_OK) return rc1;
            memset(&p, 0, sizeof p);
            c->spec_gen++;
            c->spec_valid = true;
        }
        SpecArgs a;
        memset(&a, 0, sizeof a);
        int rcf = forcing_spectrum(c, a, dt);
        if (rcf != SDC_OK) return rcf;
        a.S = c->S;
        a.SL = c->SL;
#ifdef SDC_NO_REALSYM
        a.real_sym = 0;
#else
        a.real_sym = c->sym_real[0] ? 1 : 0;
#endif
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
        // nobody will ask for the residual of this iterate (sdc_set_skip_residual): only the cached transforms move
        const bool spec_only = c->deferred && c->skip_residual && c->ndim >= 2 && M <= 8;
        const bool norms_only = !spec_only && c->deferred && c->fuse_residual && c->ndim >= 2;
        if (!(spec_only || (norms_only && !c->keep_rfields))) NEED_NODES(c);   // (before their addresses are formed)
        for (int m = 0; m < M; ++m) {
            // norms only: nothing is stored in real space - unless the residual FIELDS are wanted (keep_rfields),
            // which then occupy the U[1..M] slab while the iterate itself lives in the cache
            p.out[m] = (spec_only || (norms_only && !c->keep_rfields)) ? nullptr : c->U + (size_t)(m + 1) * c->N;
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
        // Iterates that are not stored.  After a spread predictor the iterate of a linear problem is a function of the
        // transform of u0 alone, and so is every later one while u0 and the coefficients stay what they are: a sweep
        // then reads S0 only, repeats the earlier sweeps in registers (a.replay), and stores nothing but the residual
        // lines - 6 instead of 16 spectrum passes.  store_spectra writes the iterate out when somebody needs it.
        bool go = false;
        {
            const bool fused_z = SDC_FUSE_SPECZ && c->n >= 64 && c->n <= 1024 && is_pow2(c->n) && M <= 5;
            // the launches that recompute the iterate exist for lines of 3 * 2^p / 5 * 2^p modes from 80 on as well (round 6)
            const bool virt_z = fused_z || (SDC_FUSE_SPECZ && c->n >= 80 && c->n <= 768 && fourier_ok(c) && M <= 5);
            // (with a residual to deliver this pays for a REAL symbol without explicit part - heat: real multipliers, and
            // the modes kz / N - kz of a line share them; complex multipliers cost more than reading the stored iterate:
            // advection-diffusion 512^3 3.2 -> 3.4 ms per launch)
            // (a sweep that stores the node values - eager fields - hands the iterate itself to the transform: mode pairs only)
            const bool pairs_z = SDC_SPECZ_PAIRS && virt_z && (is_pow2(c->n) ? c->n >= 512 : c->n == 768) && a.real_sym && !a.lamE;
            const bool iter_out = !spec_only && !norms_only && pairs_z;
            // Time-parallel levels (spectra on the wire, the end value wanted early): u[0] is replaced between the sweeps, so the
            // unstored iterate becomes a function of ALL the start values the slice has had since its predictor - a trail
            // (trail_iterate), kept while their number stays below trail_max; the launch always writes the last node's spectrum
            const bool trail = c->wire_spectral && c->early_uend && c->trail_max > 0 && norms_only && pairs_z && is_pow2(c->n) && !c->keep_rfields;
            go = c->virt_max > 0 && (is_pow2(c->n) || virt_z) && (c->deferred || iter_out) && !c->keep_rfields && (!c->early_uend || trail) && c->ndim >= 2 &&
                      c->expl_kind != SDC_EXPL_FORCING &&
                      (spec_only || iter_out || (norms_only && virt_z && a.real_sym && !a.lamE));
            SpecCoef now;
            memset(&now, 0, sizeof now);
            memcpy(now.gI, a.gI, sizeof now.gI);
            memcpy(now.gE, a.gE, sizeof now.gE);
            memcpy(now.cI, a.cI, sizeof now.cI);
            memcpy(now.cE, a.cE, sizeof now.cE);
            memcpy(now.alpha, a.alpha, sizeof now.alpha);
            now.coupled = a.coupled;
            now.real_sym = a.real_sym;
            now.has_e = a.lamE ? 1 : 0;
            // the multiplier table takes over where replaying gets dearer than 2 M doubles of traffic per mode pair - if the
            // replayed sweeps get that far (virt_max) and the table can be had
            bool table = go && !trail && pairs_z && is_pow2(c->n) && !spec_only && c->g_from > 0 && c->virt_max > c->g_from && c->spec_virtual >= c->g_from;
            if (table && !c->Gm) {
                const size_t glines = (size_t)(c->n / 2 + 1) * (c->ndim == 3 ? c->n : 1);
                const size_t gbytes = sizeof(double) * glines * (size_t)M * (size_t)(c->n / 2 + 1);
                if (hipMalloc((void**)&c->Gm, gbytes) == hipSuccess) {
                    c->bytes += gbytes;
                    c->g_sweeps = 0;
                } else {  // no room: sweeps are replayed up to virt_max and stored from there on, as without the table
                    (void)hipGetLastError();
                    c->Gm = nullptr;
                    c->g_from = 0;
                    table = false;
                }
            }
            if (go && a.spread) {
                if (memcmp(&now, &c->vcoef, sizeof now) != 0) c->g_sweeps = 0;
                memcpy(&c->vcoef, &now, sizeof now);
                a.replay = 0;
                int rct = trail_reset(c);   // (start values of a trail that is over)
                if (rct != SDC_OK) return rct;
                if (trail) {
                    c->trail_ns = 1;
                    c->trail_src[0] = c->S0;
                }
            } else if (go && c->spec_valid && c->spec_virtual > 0 && (c->spec_virtual < c->virt_max || table) &&
                       (c->trail_ns > 0) == trail && (!trail || (c->spec_virtual < MAXVSWEEPS && c->trail_ns <= TRAIL_S)) &&
                       memcmp(&now, &c->vcoef, sizeof now) == 0) {
                a.replay = c->spec_virtual;
                a.spread = 1;
                if (table) {
                    a.G = c->Gm;
                    a.gmode = c->g_sweeps == c->spec_virtual ? 2 : 1;
                }
            } else {
                go = false;
                STORE_SPECTRA(c, false);
            }
            if (go) {
                a.virt = iter_out ? 2 : 1;
                c->spec_virtual = a.replay + 1;
                if (a.gmode && !spec_only) c->g_sweeps = c->spec_virtual;
                // the step before took as many sweeps as this one is about to have had: probably the last of the step - its
                // launch writes the last node's spectrum on the way (one store per mode of a launch bound by arithmetic)
                // instead of leaving it to spec_store_last (S0 read and the multipliers recomputed: 4.2 ms at 1024^3)
                a.store_last = (pairs_z && !spec_only && a.gmode == 0 && c->prev_sweeps > 0 &&
                                c->spec_virtual == c->prev_sweeps && SDC_STORE_LAST_IN_SWEEP) ? 1 : 0;
                if (c->trail_ns > 0) {
                    // this sweep starts from the current start value: the last one on the trail
                    c->vsrc[c->spec_virtual - 1] = (unsigned char)(c->trail_ns - 1);
                    {
                        int rcq = ensure_trail_nyq(c);
                        if (rcq != SDC_OK) return rcq;
                    }
                    a.ns = c->trail_ns;
                    a.nsw = c->spec_virtual;
                    for (int i = 0; i < c->trail_ns; ++i) a.src[i] = c->trail_src[i];
                    // (the launch fetches TRAIL_S start values whatever their number - a fixed number of loads in flight lets
                    // it wait for exactly the ones it needs; the entries beyond the trail repeat the last one: cache hits)
                    for (int i = c->trail_ns; i < MAXTRAIL; ++i) a.src[i] = c->trail_src[c->trail_ns - 1];
                    memcpy(a.vsrc, c->vsrc, sizeof a.vsrc);
                    for (int k = 0; k < a.nsw; ++k) a.scnt[a.vsrc[k]]++;
                    for (int i = 0; i < c->trail_ns; ++i) a.scnt_packed |= (unsigned long long)(a.scnt[i] & 255) << (8 * i);
                    for (int m = 0; m < M; ++m) {
                        a.gIrow[m] = 0.0;
                        for (int j = 0; j < M; ++j) a.gIrow[m] += a.gI[m][j];
                    }
                    a.store_last = 1;
                    for (int m = 0; m < M; ++m)
                        for (int j = 0; j < M; ++j) a.rQ[m][j] = dt * c->Q[m + 1][j + 1];
                    if (c->split_send) {
                        // the last node's spectrum by a launch of its own, FIRST: the message leaves while the passes that were
                        // put off for the previous iterate (flush_x) and this sweep's own residual passes run
                        SpecArgs al = a;
                        al.last_only = 1;
                        const size_t tl = c->ndim == 1 ? 1 : (size_t)(c->n / 2 + 1) * (c->ndim == 3 ? c->n : 1);
                        const size_t nitems = tl * (size_t)(c->n / 2 + 1);
                        size_t tb = (nitems + 255) / 256;
                        if (tb > SDC_SPEC_GRID) tb = SDC_SPEC_GRID;
                        {
                            LaunchTimer lt(c, pname("trail_send", M));
#define SCASE(MM) \
    case MM: hipLaunchKernelGGL((k_trail_store<MM>), dim3((unsigned)tb), dim3(256), 0, c->stream, al, c->n, nitems); break;
                            switch (M) { SCASE(1) SCASE(2) SCASE(3) SCASE(4) SCASE(5) }
#undef SCASE
                        }
                        HIPCHK(c, hipGetLastError());
                        if (!c->sl_ev) HIPCHK(c, hipEventCreateWithFlags(&c->sl_ev, hipEventDisableTiming));
                        HIPCHK(c, hipEventRecord(c->sl_ev, c->stream));
                        c->sl_ev_recorded = true;
                        a.store_last = 0;
                        c->sl_ev_by_split = true;   // (early_end_point_n: the event stands)
                    }
                }
                c->sl_stored = a.store_last != 0 || (c->trail_ns > 0);
                if (spec_only) {  // no residual wanted: nothing to launch at all - the sweep is remembered
                    c->spec_gen++;
                    c->spec_valid = true;
                    c->spec_spread = false;
                    c->spread_pending = false;
                    c->u_pending = c->f_pending = true;
                    c->res_valid = false;
                    c->rfields_valid = false;
                    return SDC_OK;
                }
            }
        }
        // nothing downstream needs the node values in real space to keep sweeping: only the node norms of the
        // residual are produced (from its transform), U and F stay deferred
        if (!go) {   // the iterate is (about to be) stored in the node spectra
            int rcn = need_node_spectra(c);
            if (rcn != SDC_OK) return rcn;
            a.S = c->S;
        }
        // Norms of the previous iterate that still wait for their last pass (time-parallel levels, nobody has asked for the
        // numbers): this sweep's z / y launches go FIRST - into the other set of work spectra, with the other bank of norm
        // slots - so that the last node's spectrum is final, and on the wire, one launch after the start value arrived; the
        // passes that were put off follow and run while the message travels.  Anything else: they run now.
        bool pipelined = c->pipeline_x && c->xp.pending && norms_only && c->wire_spectral && c->early_uend && c->defer_x &&
                         SDC_FUSE_SPECZ && c->n >= 64 && c->n <= 1024 && is_pow2(c->n) && M <= 5;
        if (pipelined && !c->Wb) {
            // the second set of work spectra, by its first use; a GPU without room for it (43 GB at 1024^3 x 5) keeps the
            // put-off passes in front of the sweep from now on - same numbers, less overlap
            if (hipMalloc((void**)&c->Wb, sizeof(cd) * c->Nc * c->M) == hipSuccess) {
                c->bytes += sizeof(cd) * c->Nc * c->M;
            } else {
                (void)hipGetLastError();
                c->Wb = nullptr;
                c->pipeline_x = false;
                pipelined = false;
            }
        }
        PendingX put_off;
        c->dz_written = false;
        if (pipelined) {
            put_off = c->xp;
            c->xp = PendingX();
            // the put-off pass waits for the difference of the last two start values: this sweep's trail launch reads both
            // anyway and transforms the difference line on the way (no fft_z_diff: 4.5 ms at 1024^3 for 1.5 in the launch)
            if (put_off.has_delta && !put_off.dz && a.ns >= 2 && c->ndim >= 2 && put_off.d_new == a.src[a.ns - 1] &&
                put_off.d_old == a.src[a.ns - 2] && !getenv("SDC_NO_TRAIL_DZ")) {
                a.dz = spool_get(c);
                if (!a.dz) (void)hipGetLastError();   // (no room for one more spectrum: the difference goes through fft_z_diff)
            }
            std::swap(c->W, c->Wb);
            c->res_bank_now ^= 1;
            c->res_dev = c->res_bank[c->res_bank_now];
            c->res_devA = c->res_dev + 8;
        } else {
            FLUSH_X(c);
        }
        if (norms_only) {
            for (int m = 0; m < M; ++m)
                for (int j = 0; j < M; ++j) a.rQ[m][j] = dt * c->Q[m + 1][j + 1];
            HIPCHK(c, hipMemsetAsync(c->res_dev, 0, sizeof(unsigned long long) * 8, c->stream));
        }
        c->spec_gen++;
        int rc0 = spec_sweep(c, M, a, p, norms_only ? c->res_dev : nullptr, spec_only);
        if (pipelined) {   // (also when the sweep failed: the tickets of the previous iterate are answered)
            PendingX mine = c->xp;
            if (a.dz) {
                if (c->dz_written && rc0 == SDC_OK) put_off.dz = a.dz;
                else spool_put(c, a.dz);   // (a launch that could not take the line along: the difference goes through fft_z_diff)
            }
            c->xp = put_off;
            int rcp = flush_x(c);
            c->xp = mine;
            if (rc0 == SDC_OK) rc0 = rcp;
        }
        if (rc0 != SDC_OK) return rc0;
        c->spec_valid = true;
        c->spec_spread = false;
        c->spread_pending = false;
        if (spec_only) {  // a residual asked for after all comes from the cache (sdc_residual, spectral route)
            c->u_pending = c->f_pending = true;
            c->res_valid = false;
            c->rfields_valid = false;
            return SDC_OK;
        }
        if (norms_only) {
            c->u_pending = c->f_pending = true;
            c->res_valid = true;
            c->res_dt = dt;
            c->rfields_valid = c->keep_rfields;
            c->rlines_valid = a.virt == 0 && !c->keep_rfields;  // (W: the residual after its z / y inverse passes)
            return SDC_OK;
        }
        c->u_pending = false;  // U[1..M] hold the new iterate; F follows in eval_nodes
        return eval_nodes(c, dt);
    }
    {
        int rcm = materialize(c, false, true);  // the gather reads F[1..M]
        if (rcm != SDC_OK) return rcm;
    }
    c->spec_valid = false;
    c->spec_spread = false;
    c->u_pending = false;  // U[1..M] are overwritten below
    // 1. gather u0 + dt (Q - QI) F_impl + dt (Q - QE) F_expl (+ tau) for all nodes into U[1..M]
    QuadArgs q;
    quad_base(c, q);
    q.u0 = c->U0;
    q.tau = c->tau_active ? c->TAU : nullptr;
    const bool forcing = c->expl_kind == SDC_EXPL_FORCING;
    for (int m = 0; m < M; ++m) {
        q.out[m] = c->U + (size_t)(m + 1) * c->N;
        for (int j = 0; j < M; ++j) {
            q.cI[m][j] = dt * (c->Q[m + 1][j + 1] - c->QI[m + 1][j + 1]);
            q.cE[m][j] = dt * (c->Q[m + 1][j + 1] - c->QE[m + 1][j + 1]);
        }
    }
    int rc = launch_quad<0>(c, q, "gather");
    if (rc != SDC_OK) return rc;
    if (forcing) {
        // the strictly lower QE add-back of imex_1st_order.py:94 uses the NEW explicit values, which for a u-independent
        // forcing are known before any solve: profile * g(t_j).  (The stored old ones may be something else - a
        // 'copy' predictor leaves g(t_0) everywhere - so they only enter through the gather above.)
        for (int m = 1; m < M; ++m) {
            double cm = 0.0;
            for (int j = 0; j < m; ++j) cm += dt * c->QE[m + 1][j + 1] * c->gvals[j + 1];
            if (cm == 0.0) continue;
            LinArgs la;
            memset(&la, 0, sizeof la);
            la.out = c->U + (size_t)(m + 1) * c->N;
            la.base = la.out;
            la.n = c->N;
            la.x[0] = c->profile;
            la.c[0] = cm;
            la.nterms = 1;
            LaunchTimer lt(c, "node_rhs");
            hipLaunchKernelGGL(k_lincomb, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, la);
        }
        HIPCHK(c, hipGetLastError());
    }
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

// scipy.sparse.linalg.cg as the reference calls it (generic_ND_FD.py:252-260): x0 = guess, rtol = lintol, atol = 0,
// maxiter = liniter, no preconditioner; every iteration counts (the reference's callback).  rhs must not alias out.
static int cg_solve(sdc_ctx* c, const double* b, double factor, const double* guess, double* x) {
    if (!c->have_stencil[0] && !c->nb) return fail(c, SDC_ERR_STATE, "implicit operator not set (sdc_set_stencil)");
    if (b == x) return fail(c, SDC_ERR_PARAM, "conjugate gradients: right-hand side and solution share storage");
    constexpr int NB = 2048;
    const size_t N = problem_size(c);
    if (!c->cgw) {
        HIPCHK(c, hipMalloc((void**)&c->cgw, sizeof(double) * (4 * N + NB + 16)));
        c->bytes += sizeof(double) * (4 * N + NB + 16);
    }
    double *r = c->cgw, *p = r + N, *q = p + N, *Ap = q + N, *part = Ap + N, *scal = part + NB;
    const int nb = grid_for(N, 256) < NB ? grid_for(N, 256) : NB;
    auto launch = [&](int mode, const double* a0, const double* a1, const double* a2, double* o0, double* o1, double s,
                      double* result) -> int {
        CgArgs a{a0, a1, a2, o0, o1, s, N, result ? part : nullptr, mode};
        hipLaunchKernelGGL(k_cg, dim3(nb), dim3(256), 0, c->stream, a);
        HIPCHK(c, hipGetLastError());
        if (result) {
            hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(256), 0, c->stream, part, nb, scal);
            HIPCHK(c, hipMemcpyAsync(c->red_host + 15, scal, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            memcpy(result, &c->red_host[15], sizeof(double));
        }
        return SDC_OK;
    };
    auto matvec = [&](const double* v) -> int { return apply_operator(c, v, Ap); };  // Ap = A v
    LaunchTimer lt(c, "cg_solve");
    int rc;
    if (!guess) HIPCHK(c, hipMemsetAsync(x, 0, N * sizeof(double), c->stream));
    else if (guess != x) HIPCHK(c, hipMemcpyAsync(x, guess, N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    double bb = 0.0, rr = 0.0;
    if ((rc = launch(4, b, nullptr, nullptr, nullptr, nullptr, 0.0, &bb)) != SDC_OK) return rc;
    if (bb == 0.0) {  // scipy returns b itself
        HIPCHK(c, hipMemcpyAsync(x, b, N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return SDC_OK;
    }
    const double atol = c->cg_rtol * sqrt(bb);
    if ((rc = matvec(x)) != SDC_OK) return rc;
    if ((rc = launch(0, b, x, Ap, r, nullptr, factor, &rr)) != SDC_OK) return rc;
    if (sqrt(rr) < atol || c->cg_maxiter <= 0) return SDC_OK;
    // The iteration itself runs without the host: alpha, beta, the stopping test and the iteration count live in `scal`
    // (k_cg_scalars), every launch of an iteration that follows the stop is a no-op, and the host looks at the flag once per
    // batch of iterations instead of twice per iteration (round 2: two blocking 8-byte copies per iteration, a fifth of the
    // time of a 256^3 solve).  Same operations in the same order: identical iterates and iteration counts.
    {
        double init[9] = {0.0, rr, 0.0, 0.0, 0.0, atol, 0.0, 0.0, (double)c->cg_maxiter};
        HIPCHK(c, hipMemcpyAsync(scal, init, sizeof init, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (init lives on this stack frame)
    }
    auto dev_launch = [&](int mode, const double* a0, const double* a1, double* o0, double* o1, double s_host, int sidx,
                          int step) -> int {
        CgArgs a{a0, a1, nullptr, o0, o1, s_host, N, step >= 0 ? part : nullptr, mode, sidx >= 0 ? scal + sidx : nullptr,
                 scal + CGS_DONE};
        hipLaunchKernelGGL(k_cg, dim3(nb), dim3(256), 0, c->stream, a);
        if (step >= 0) hipLaunchKernelGGL(k_cg_scalars, dim3(1), dim3(256), 0, c->stream, part, nb, scal, step);
        HIPCHK(c, hipGetLastError());
        return SDC_OK;
    };
    // batches: two fewer iterations than the previous solve with this factor took (one factor per collocation node: the
    // counts differ from node to node, little from sweep to sweep), then two at a time
    int enqueued = 0;
    int batch = 8;
    {
        auto it = c->cg_hist.find(factor);
        if (it != c->cg_hist.end() && it->second > 3) batch = it->second - 2;
    }
    for (;;) {
        if (batch > c->cg_maxiter - enqueued) batch = c->cg_maxiter - enqueued;
        for (int k = 0; k < batch; ++k, ++enqueued) {
            if (enqueued > 0) {
                if ((rc = dev_launch(1, r, nullptr, p, nullptr, 0.0, CGS_BETA, -1)) != SDC_OK) return rc;  // p = r + beta p
            } else {
                HIPCHK(c, hipMemcpyAsync(p, r, N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            }
            if ((rc = matvec(p)) != SDC_OK) return rc;
            if ((rc = dev_launch(2, p, Ap, q, nullptr, factor, -1, 0)) != SDC_OK) return rc;   // q = p - factor A p, p.q
            if ((rc = dev_launch(3, p, q, x, r, 0.0, CGS_ALPHA, 1)) != SDC_OK) return rc;      // x += alpha p, r -= alpha q
        }
        HIPCHK(c, hipMemcpyAsync(c->red_host + 14, scal + CGS_DONE, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        double flag[2];
        memcpy(flag, &c->red_host[14], sizeof flag);
        if (flag[0] != 0.0 || enqueued >= c->cg_maxiter) {
            c->cg_iters += (unsigned long long)flag[1];
            if (c->cg_hist.size() > 64) c->cg_hist.clear();
            c->cg_hist[factor] = (int)flag[1];
            break;
        }
        batch = 2;
    }
    return SDC_OK;
}

// scipy.sparse.linalg.gmres as the reference calls it (generic_ND_FD.py:241-250): x0 = guess, rtol = lintol, atol = 0,
// restart = 20, no preconditioner, callback_type 'legacy' - one count per INNER iteration, and maxiter = liniter counts
// inner iterations too.  Restated from scipy 1.15 (_isolve/iterative.py: Arnoldi with modified Gram-Schmidt, Givens
// rotations by LAPACK's lartg, the inner tolerance control of gh-8400), vector work on the device with dot products
// reduced in a fixed order, the small Hessenberg problem on the host.  rhs must not alias out.
static void host_lartg(double f, double g, double* c, double* s, double* r) {  // LAPACK 3.10 dlartg, values in safe range
    if (g == 0.0) {
        *c = 1.0;
        *s = 0.0;
        *r = f;
    } else if (f == 0.0) {
        *c = 0.0;
        *s = g < 0 ? -1.0 : 1.0;
        *r = fabs(g);
    } else {
        const double d = sqrt(f * f + g * g);
        *c = fabs(f) / d;
        *r = f < 0 ? -d : d;
        *s = g / *r;
    }
}

static int gmres_solve(sdc_ctx* c, const double* b, double factor, const double* guess, double* x) {
    if (!c->have_stencil[0] && !c->nb) return fail(c, SDC_ERR_STATE, "implicit operator not set (sdc_set_stencil)");
    if (b == x) return fail(c, SDC_ERR_PARAM, "GMRES: right-hand side and solution share storage");
    constexpr int NB = 2048;
    const size_t N = problem_size(c);
    int restart = 20;
    if ((size_t)restart > N) restart = (int)N;
    const size_t need = ((size_t)restart + 4) * N + NB + 8;
    if (!c->gmw || c->gmw_len < need) {
        if (c->gmw) {
            (void)hipFree(c->gmw);
            c->bytes -= c->gmw_len * sizeof(double);
        }
        HIPCHK(c, hipMalloc((void**)&c->gmw, sizeof(double) * need));
        c->gmw_len = need;
        c->bytes += sizeof(double) * need;
    }
    double *V = c->gmw, *w = V + (size_t)(restart + 1) * N, *Av = w + N, *tsum = Av + N, *part = tsum + N, *scal = part + NB;
    const int nb = grid_for(N, 256) < NB ? grid_for(N, 256) : NB;
    auto launch = [&](int mode, const double* a0, const double* a1, const double* a2, double* o0, double s,
                      double* result) -> int {
        CgArgs a{a0, a1, a2, o0, nullptr, s, N, result ? part : nullptr, mode};
        hipLaunchKernelGGL(k_cg, dim3(nb), dim3(256), 0, c->stream, a);
        HIPCHK(c, hipGetLastError());
        if (result) {
            hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(256), 0, c->stream, part, nb, scal);
            HIPCHK(c, hipMemcpyAsync(c->red_host + 15, scal, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            memcpy(result, &c->red_host[15], sizeof(double));
        }
        return SDC_OK;
    };
    auto apply_A = [&](const double* v) -> int { return apply_operator(c, v, Av); };  // Av = A v
    LaunchTimer lt(c, "gmres_solve");
    int rc;
    if (!guess) HIPCHK(c, hipMemsetAsync(x, 0, N * sizeof(double), c->stream));
    else if (guess != x) HIPCHK(c, hipMemcpyAsync(x, guess, N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    double bb = 0.0;
    if ((rc = launch(4, b, nullptr, nullptr, nullptr, 0.0, &bb)) != SDC_OK) return rc;
    const double bnrm2 = sqrt(bb);
    if (bnrm2 == 0.0) {  // scipy returns b itself
        HIPCHK(c, hipMemcpyAsync(x, b, N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return SDC_OK;
    }
    const double atol = c->cg_rtol * bnrm2, eps = 2.220446049250313e-16;
    const int maxiter = c->cg_maxiter;
    double ptol_max_factor = 1.0, ptol = bnrm2 * std::min(ptol_max_factor, atol / bnrm2), presid = 0.0;
    std::vector<double> h((size_t)restart * (restart + 1), 0.0), giv((size_t)restart * 2, 0.0), S(restart + 1), y(restart);
    int inner_iter = 0;
    double* r = w;  // the residual is formed where w will live; v[0] takes it over before w is used
    double rr = 0.0;  // |r|^2 of the current residual
    for (int iteration = 0; iteration < maxiter; ++iteration) {
        if (iteration == 0) {
            // r = b - (x - factor A x); (x0 = 0: scipy copies b - same values)
            if ((rc = apply_A(x)) != SDC_OK) return rc;
            if ((rc = launch(0, b, x, Av, r, factor, &rr)) != SDC_OK) return rc;
            if (sqrt(rr) < atol) return SDC_OK;
        }
        // v[0] = r / |r|
        double tmp = sqrt(rr);
        if ((rc = launch(9, r, nullptr, nullptr, V, 1.0 / tmp, nullptr)) != SDC_OK) return rc;
        std::fill(S.begin(), S.end(), 0.0);
        S[0] = tmp;
        bool breakdown = false;
        int col = 0;
        for (col = 0; col < restart; ++col) {
            double* vc = V + (size_t)col * N;
            if ((rc = apply_A(vc)) != SDC_OK) return rc;
            double ww = 0.0, dot = 0.0;
            if ((rc = launch(6, vc, Av, nullptr, w, factor, &ww)) != SDC_OK) return rc;  // w = (I - factor A) v_col
            const double h0 = sqrt(ww);
            // modified Gram-Schmidt; the axpy of step k and the dot product of step k + 1 share a pass
            if ((rc = launch(7, V, w, nullptr, nullptr, 0.0, &dot)) != SDC_OK) return rc;
            for (int k = 0; k <= col; ++k) {
                h[(size_t)col * (restart + 1) + k] = dot;
                const double* next = k < col ? V + (size_t)(k + 1) * N : w;
                double nd = 0.0;
                if ((rc = launch(8, V + (size_t)k * N, next, nullptr, w, dot, &nd)) != SDC_OK) return rc;
                dot = nd;  // <v_{k+1}, w> or, after the last step, |w|^2
            }
            const double h1 = sqrt(dot);
            double* hc = &h[(size_t)col * (restart + 1)];
            hc[col + 1] = h1;
            if (h1 <= eps * h0) {  // exact solution indicator
                hc[col + 1] = 0.0;
                breakdown = true;
                HIPCHK(c, hipMemcpyAsync(V + (size_t)(col + 1) * N, w, N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            } else {
                if ((rc = launch(9, w, nullptr, nullptr, V + (size_t)(col + 1) * N, 1.0 / h1, nullptr)) != SDC_OK) return rc;
            }
            for (int k = 0; k < col; ++k) {  // past rotations on the new column
                const double cc = giv[2 * k], ss = giv[2 * k + 1], n0 = hc[k], n1 = hc[k + 1];
                hc[k] = cc * n0 + ss * n1;
                hc[k + 1] = -ss * n0 + cc * n1;
            }
            double cc, ss, mag;
            host_lartg(hc[col], hc[col + 1], &cc, &ss, &mag);
            giv[2 * col] = cc;
            giv[2 * col + 1] = ss;
            hc[col] = mag;
            hc[col + 1] = 0.0;
            const double t2 = -ss * S[col];
            S[col] = cc * S[col];
            S[col + 1] = t2;
            presid = fabs(t2);
            ++inner_iter;
            c->gmres_iters++;  // the reference's legacy callback
            if (inner_iter == maxiter) break;
            if (presid <= ptol || breakdown) break;
        }
        if (col == restart) col = restart - 1;  // (the loop ran out: Python leaves col at its last value)
        // back substitution on the rotated Hessenberg matrix (rows of h are ITS columns), singular case as in scipy
        if (h[(size_t)col * (restart + 1) + col] == 0.0) S[col] = 0.0;
        for (int k = 0; k <= col; ++k) y[k] = S[k];
        for (int k = col; k > 0; --k) {
            if (y[k] != 0.0) {
                y[k] /= h[(size_t)k * (restart + 1) + k];
                const double tk = y[k];
                for (int q = 0; q < k; ++q) y[q] -= tk * h[(size_t)k * (restart + 1) + q];
            }
        }
        if (y[0] != 0.0) y[0] /= h[0];
        // x += y @ v[:col+1]: the combination first, then ONE addition to x
        for (int k = 0; k <= col; ++k)
            if ((rc = launch(10, V + (size_t)k * N, k ? tsum : nullptr, nullptr, tsum, y[k], nullptr)) != SDC_OK) return rc;
        if ((rc = launch(11, tsum, nullptr, nullptr, x, 0.0, nullptr)) != SDC_OK) return rc;
        if ((rc = apply_A(x)) != SDC_OK) return rc;
        if ((rc = launch(0, b, x, Av, r, factor, &rr)) != SDC_OK) return rc;
        const double rnorm = sqrt(rr);
        if (inner_iter == maxiter) return SDC_OK;  // legacy exit
        if (rnorm <= atol || breakdown) break;
        if (presid <= ptol) ptol_max_factor = std::max(eps, 0.25 * ptol_max_factor);
        else ptol_max_factor = std::min(1.0, 1.5 * ptol_max_factor);
        ptol = presid * std::min(ptol_max_factor, atol / rnorm);
    }
    return SDC_OK;
}

int sdc_solve(sdc_ctx* c, const double* rhs, double factor, const double* guess, double* out) {
    if (!c || !rhs || !out) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (c->odd_n && !c->odd_busy) {   // compact interior fields: the solve on the odd extension is the sine-transform solve
        const bool with_guess = guess != nullptr && c->solver_kind != 0;
        for (int k = 0; k < (with_guess ? 3 : 2); ++k) {
            int rs = odd_scratch(c, k);
            if (rs != SDC_OK) return rs;
        }
        int rc = sdc_odd_extend(c, rhs, c->odd_buf[0], c->odd_n, c->ndim);
        if (rc == SDC_OK && with_guess) rc = sdc_odd_extend(c, guess, c->odd_buf[2], c->odd_n, c->ndim);
        if (rc != SDC_OK) return rc;
        {
            OddScope scope(c);
            rc = sdc_solve(c, c->odd_buf[0], factor, with_guess ? c->odd_buf[2] : nullptr, c->odd_buf[1]);
        }
        if (rc != SDC_OK) return rc;
        return sdc_odd_extract(c, c->odd_buf[1], out, c->odd_n, c->ndim);
    }
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
    if (c->nb && c->solver_kind == 0) {
        // 'direct' on a bounded grid with shifted boundary stencils (non-symmetric, not diagonal in any transform we
        // have): GMRES to round-off stands in for the reference's sparse LU
        const double keep_rtol = c->cg_rtol;
        const int keep_maxiter = c->cg_maxiter;
        const unsigned long long keep_gm = c->gmres_iters;
        c->cg_rtol = 1e-14;
        c->cg_maxiter = 100000;
        int rcg = gmres_solve(c, rhs, factor, guess, out);
        c->cg_rtol = keep_rtol;
        c->cg_maxiter = keep_maxiter;
        c->gmres_iters = keep_gm;
        return rcg;
    }
    if (c->solver_kind == 1 && !c->spectral_op) return cg_solve(c, rhs, factor, guess, out);
    if (c->solver_kind == 2 && !c->spectral_op) return gmres_solve(c, rhs, factor, guess, out);
    if (!fourier_ok(c) && !c->spectral_op) {
        // a grid the FFT kernels do not transform (n not a power of two, or too long): conjugate gradients to
        // round-off instead of the exact solve - for symmetric operators only
        const Stencil& s = c->st[0];
        bool symmetric = c->have_stencil[0];
        for (int a = 0; a < s.npts && symmetric; ++a) {
            bool found = false;
            for (int b = 0; b < s.npts; ++b)
                if (s.off[b] == -s.off[a] && s.w[b] == s.w[a]) found = true;
            symmetric = found;
        }
        const double keep_rtol = c->cg_rtol;
        const int keep_maxiter = c->cg_maxiter;
        if (!symmetric) {
            // ... and GMRES to round-off for the others (advection: non-symmetric)
            const unsigned long long keep_gm = c->gmres_iters;
            c->cg_rtol = 1e-14;
            c->cg_maxiter = 100000;
            int rcg = gmres_solve(c, rhs, factor, guess, out);
            c->cg_rtol = keep_rtol;
            c->cg_maxiter = keep_maxiter;
            c->gmres_iters = keep_gm;  // not the user's solver: nothing to count
            return rcg;
        }
        const unsigned long long keep_iters = c->cg_iters;
        c->cg_rtol = 1e-14;
        c->cg_maxiter = 100000;
        int rcg = cg_solve(c, rhs, factor, guess, out);
        c->cg_rtol = keep_rtol;
        c->cg_maxiter = keep_maxiter;
        c->cg_iters = keep_iters;  // not the user's solver: nothing to count
        return rcg;
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

int sdc_solve_jacobian(sdc_ctx* c, const double* rhs, double dt, const double* u, double* out) {
    if (!c || !rhs || !u || !out) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (c->kind != 1) return fail(c, SDC_ERR_UNSUPPORTED, "solve_jacobian is the van der Pol ensemble's (sdc_set_problem_vdp)");
    LaunchTimer lt(c, "vdp_jac_solve");
    hipLaunchKernelGGL(k_vdp_jac_solve, dim3(grid_for(c->N / 2, 256)), dim3(256), 0, c->stream, rhs, u, out, c->N / 2, dt,
                       c->vdp_mu);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

// the last launch of a residual: finishes the number on the device and leaves the record in pinned host memory
struct PublishArgs {
    const unsigned long long* norms;   // M node norms (bit patterns of non-negative doubles; NaN sorts above everything)
    const unsigned long long* f0max;   // spread predictor: max |f(u0)|, the node norms are scale[m] * it (or null)
    const unsigned long long* u0norm;  // relative residual types: max |u0| (or null)
    double scale[MAXM];
    double restol;
    int M, type;
};

__global__ void k_publish_residual(PublishArgs a, ResRecord* dst, unsigned long long seq) {
    if (threadIdx.x != 0) return;
    double mx = 0.0, last = 0.0;
    for (int m = 0; m < a.M; ++m) {
        double v;
        if (a.f0max) {
            v = fabs(a.scale[m]) * __longlong_as_double((long long)*a.f0max);
        } else {
            v = __longlong_as_double((long long)a.norms[m]);
        }
        dst->norms[m] = v;
        mx = (v > mx || v != v) ? v : mx;   // (a NaN stays, like np.max / max() over the reference's list)
        last = v;
    }
    const double u0n = a.u0norm ? __longlong_as_double((long long)*a.u0norm) : 1.0;
    double r;
    switch (a.type) {
        case SDC_RES_FULL_ABS: r = mx; break;
        case SDC_RES_LAST_ABS: r = last; break;
        case SDC_RES_FULL_REL: r = mx / u0n; break;
        default: r = last / u0n; break;
    }
    dst->residual = r;
    dst->converged = (a.restol >= 0.0 && r <= a.restol) ? 1 : 0;
    __threadfence_system();
    *(volatile unsigned long long*)&dst->seq = seq;
}

}  // extern "C"
static int publish_ticket(sdc_ctx* c, const PendingTicket& t, const unsigned long long* norms) {
    PublishArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.M = c->M;
    pa.type = t.type;
    pa.restol = t.restol;
    pa.norms = norms;
    hipLaunchKernelGGL(k_publish_residual, dim3(1), dim3(64), 0, c->stream, pa, c->ring_dev + t.seq % RES_RING, t.seq);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}
extern "C" {

int sdc_set_restol(sdc_ctx* c, double restol) {
    if (!c) return SDC_ERR_PARAM;
    c->restol = restol;
    return SDC_OK;
}

int sdc_residual_post(sdc_ctx* c, double dt, int type, unsigned long long* ticket) {
    return sdc_residual_post_integrals(c, dt, type, nullptr, nullptr, ticket);
}

int sdc_residual_post_integrals(sdc_ctx* c, double dt, int type, double* const* integrals, int* wrote, unsigned long long* ticket) {
    if (!c || !ticket) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (wrote) *wrote = 0;
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    if (type < 0 || type > 3)
        return fail(c, SDC_ERR_PARAM,
                    "residual_type = %d not implemented, choose full_abs, last_abs, full_rel or last_rel instead", type);
    const int M = c->M;
    PublishArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.M = M;
    pa.type = type;
    pa.restol = c->restol;
    const unsigned long long seq = c->ring_seq + 1;
    ResRecord* slot = c->ring + seq % RES_RING;
    // the record this ticket replaces (RES_RING tickets ago) may still be on its way - its publishing launch queued behind a
    // long run of posts nobody waited for: let it land before the slot is written again (host or device), or it would
    // overwrite the newer record
    if (seq > RES_RING && slot->seq != seq - RES_RING) HIPCHK(c, hipStreamSynchronize(c->stream));
    // the sweep that reduced these norms ended with a look at its counters anyway (van der Pol: Newton failures are reported
    // by the sweep) and brought the norms along: nothing to launch, copy or wait for - the host writes the record itself
    const bool on_host = c->kind == 1 && c->res_valid && c->res_dt == dt && c->res_host_valid && type < SDC_RES_FULL_REL;
    if (on_host) {
        double mx = 0.0, last = 0.0;
        for (int m = 0; m < M; ++m) {
            double v;
            memcpy(&v, &c->res_host[m], sizeof(double));
            slot->norms[m] = v;
            mx = (v > mx || v != v) ? v : mx;
            last = v;
        }
        slot->residual = type == SDC_RES_FULL_ABS ? mx : last;
        slot->converged = (c->restol >= 0.0 && slot->residual <= c->restol) ? 1 : 0;
        slot->seq = seq;
        c->ring_seq = seq;
        *ticket = seq;
        return SDC_OK;
    }
    if (c->xp.pending) {
        if (c->res_valid && c->res_dt == dt && type < SDC_RES_FULL_REL) {
            // the norms of this very state wait for their last inverse pass (flush_x): the ticket waits with them - for the
            // start value the rank is about to receive, the next sweep, or somebody who asks (sdc_residual_wait)
            slot->seq = 0;
            c->xp.tickets.push_back(PendingTicket{seq, type, c->xp.has_delta ? 1 : 0, c->restol});
            c->ring_seq = seq;
            *ticket = seq;
            return SDC_OK;
        }
        FLUSH_X(c);
    }
    if (c->res_valid && c->res_dt == dt) {
        // the sweep's fused kernels already reduced the node norms of this very state
        pa.norms = c->res_dev;
    } else if (c->res_spread) {
        if (c->f0norm_pending) {  // (put off by sdc_predict; res_spread: no sweep since - S0 and the work buffer are as it left them)
            c->f0norm_pending = false;
            if (!(c->spec0_valid && c->S0)) return fail(c, SDC_ERR_STATE, "the transform of u[0] is gone: predict again");
            HIPCHK(c, hipMemsetAsync(c->res_dev + 7, 0, sizeof(unsigned long long), c->stream));
            int rcn = symbol_norm(c, c->S0, c->res_dev + 7);
            if (rcn != SDC_OK) return rcn;
        }
        pa.f0max = c->res_dev + 7;
        for (int m = 0; m < M; ++m) {
            double sq = 0.0;
            for (int j = 1; j <= M; ++j) sq += dt * c->Q[m + 1][j];
            pa.scale[m] = sq;
        }
    } else {
        HIPCHK(c, hipMemsetAsync(c->red, 0, sizeof(unsigned long long) * 16, c->stream));
        pa.norms = c->red;
        if (c->u_pending && !c->spread_pending && c->spec_valid && !c->tau_active && c->ndim >= 2) {
            // the iterate lives in the spectral cache (u[0] was replaced after the sweep, or dt differs): reduce the
            // residual from its transform instead of bringing U and F back to real space
            int rcs = spec_residual(c, dt, c->red);
            if (rcs != SDC_OK) return rcs;
        } else {
            ENSURE_U0(c);
            int rcm = materialize(c, true, true);
            if (rcm != SDC_OK) return rcm;
            QuadArgs q;
            quad_base(c, q);
            q.u0 = c->U0;
            q.tau = c->tau_active ? c->TAU : nullptr;
            q.Usub = c->U;
            q.norms = c->red;
            for (int m = 0; m < M; ++m)
                for (int j = 0; j < M; ++j) q.cI[m][j] = q.cE[m][j] = dt * c->Q[m + 1][j + 1];
            bool with_integrals = integrals != nullptr;
            for (int m = 0; m < M && with_integrals; ++m) with_integrals = integrals[m] != nullptr;
            if (with_integrals) {   // the quadrature sums are wanted as fields too (sdc_integrate's result): same pass
                for (int m = 0; m < M; ++m) q.out[m] = integrals[m];
                int rc = launch_quad<2>(c, q, "residual_integrate");
                if (rc != SDC_OK) return rc;
                if (wrote) *wrote = 1;
            } else {
                int rc = launch_quad<1>(c, q, "residual");
                if (rc != SDC_OK) return rc;
            }
        }
    }
    if (type >= SDC_RES_FULL_REL) {
        U0R(c, u0p);
        HIPCHK(c, hipMemsetAsync(c->red + 8, 0, sizeof(unsigned long long), c->stream));
        LaunchTimer lt(c, "amax");
        hipLaunchKernelGGL(k_amax, dim3(grid_for(c->N, 256)), dim3(256), 0, c->stream, u0p, c->N, c->red + 8);
        pa.u0norm = c->red + 8;
    }
    slot->seq = 0;   // (a record of RES_RING tickets ago: sdc_residual_wait no longer answers for it)
    hipLaunchKernelGGL(k_publish_residual, dim3(1), dim3(64), 0, c->stream, pa, c->ring_dev + seq % RES_RING, seq);
    HIPCHK(c, hipGetLastError());
    c->ring_seq = seq;
    *ticket = seq;
    return SDC_OK;
}

int sdc_residual_route(sdc_ctx* c, double dt) {
    if (!c) return SDC_ERR_PARAM;
    if (c->res_valid && c->res_dt == dt) return 0;
    if (c->res_spread) return 1;
    if (c->u_pending && !c->spread_pending && c->spec_valid && !c->tau_active && c->ndim >= 2) return 2;
    return 3;
}

unsigned long long sdc_residual_last_ticket(sdc_ctx* c) { return c ? c->ring_seq : 0; }

int sdc_residual_wait(sdc_ctx* c, unsigned long long ticket, int block, double* node_norms, double* residual, int* converged,
                      int* ready) {
    if (!c || !ready) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (ticket == 0 || ticket > c->ring_seq || ticket + RES_RING <= c->ring_seq)
        return fail(c, SDC_ERR_STATE, "residual ticket %llu is not in flight (last %llu, %d records kept)", ticket, c->ring_seq,
                    RES_RING);
    volatile ResRecord* r = c->ring + ticket % RES_RING;
    *ready = r->seq == ticket;
    if (!*ready && block && c->xp.pending) FLUSH_X(c);   // (the record may be waiting for a pass that was put off)
    if (!*ready && block) {
        // the record arrives by itself when the stream gets there; look at it for a while, then wait on the stream (which
        // also surfaces a launch that failed)
        for (int spin = 0; spin < 2000000 && r->seq != ticket; ++spin) __builtin_ia32_pause();
        if (r->seq != ticket) HIPCHK(c, hipStreamSynchronize(c->stream));
        if (r->seq != ticket) return fail(c, SDC_ERR_HIP, "the residual record of ticket %llu never arrived", ticket);
        *ready = 1;
    }
    if (*ready) {
        std::atomic_thread_fence(std::memory_order_acquire);
        if (node_norms)
            for (int m = 0; m < c->M; ++m) node_norms[m] = r->norms[m];
        if (residual) *residual = r->residual;
        if (converged) *converged = r->converged;
    }
    return SDC_OK;
}

int sdc_residual(sdc_ctx* c, double dt, int type, double* node_norms, double* residual) {
    if (!c || !residual) return fail(c, SDC_ERR_PARAM, "null pointer");
    unsigned long long ticket = 0;
    int rc = sdc_residual_post(c, dt, type, &ticket);
    if (rc != SDC_OK) return rc;
    int ready = 0;
    return sdc_residual_wait(c, ticket, 1, node_norms, residual, nullptr, &ready);
}

int sdc_end_point(sdc_ctx* c, double dt, int do_coll_update) {
    if (!c) return SDC_ERR_PARAM;
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    if (!do_coll_update) {
        if (c->u_pending && !c->spread_pending && c->uend_gen >= 0 && c->uend_gen == c->spec_gen)
            return SDC_OK;  // the sweep already produced it (sdc_set_early_end_point)
    }
    {
        int rcf = uend_write_fence(c);
        if (rcf != SDC_OK) return rcf;
    }
    c->uend_ev_recorded = false;  // (whatever is written below is covered by a fresh event, sdc_stream_wait_uend)
    if (!do_coll_update) {
        c->uend_gen = -1;
        c->uend_pending = false;
        if (c->u_pending && !c->spread_pending) {  // only the last node is needed
            if (c->deferred && c->kind == 0 && (!c->early_uend || c->wire_spectral) && !c->keep_rfields && c->ndim >= 2) {
                // ... and not even that until somebody reads it: the end value IS the inverse transform of SL
                // (materialize_uend); a following sdc_advance hands the spectrum over and never needs it in real space
                c->uend_pending = true;
                c->uend_gen = c->spec_gen;
                return SDC_OK;
            }
            FieldPtrs p;  // transform it straight into UEND
            memset(&p, 0, sizeof p);
            p.out[0] = c->UEND;
            int rci = inverse_from_cache(c, c->M - 1, 1, p);
            if (rci == SDC_OK) c->uend_gen = c->spec_gen;
            return rci;
        }
        // a pending spread means U[M] equals U[0]
        if (!c->spread_pending) {
            NEED_NODES(c);
            return sdc_vec_copy(c, c->N, c->U + (size_t)c->M * c->N, c->UEND);
        }
        U0R(c, u0p);
        return sdc_vec_copy(c, c->N, u0p, c->UEND);
    }
    c->uend_gen = -1;
    c->uend_pending = false;
    ENSURE_U0(c);
    int rcm = materialize(c, false, true);
    if (rcm != SDC_OK) return rcm;
    QuadArgs q;
    quad_base(c, q);
    q.u0 = c->U0;
    q.tau = c->tau_active ? c->TAU : nullptr;
    q.tau_row0 = c->M - 1;
    q.nout = 1;
    q.out[0] = c->UEND;
    for (int j = 0; j < c->M; ++j) q.cI[0][j] = q.cE[0][j] = dt * c->weights[j];
    return launch_quad<0>(c, q, "end_point");
}

int sdc_advance(sdc_ctx* c) {
    if (!c) return SDC_ERR_PARAM;
    FLUSH_X(c);
    int rcm = materialize(c, c->spread_pending, false);  // pending copies of the OLD u[0] are stored first
    if (rcm != SDC_OK) return rcm;
    c->res_valid = false;
    c->res_spread = false;
    c->spec_spread = false;
    // UEND is the inverse transform of the last node's spectrum: that spectrum is the transform of the new u[0]
    const bool handover = c->Sx && c->spec_valid && c->uend_gen >= 0 && c->uend_gen == c->spec_gen;
    c->prev_sweeps = (c->spec_valid && c->spec_virtual > 0) ? c->spec_virtual : 0;
    if (handover) STORE_SPECTRA(c, true);  // (SL has to BE there; S0 changes below)
    if (handover) {
        // the step is over (every handover branch below drops its iterate): the start values its trail depended on are spare
        // buffers again - all but the current one and the last node's spectrum, which trade places
        int rct = trail_reset(c);
        if (rct != SDC_OK) return rct;
    }
    if (handover && c->uend_pending && c->kind == 0 && c->deferred) {
        // the end value was never transformed back: the start value of the new step exists as its transform only
        std::swap(c->S0, c->SL);
        c->spec0_valid = true;
        c->spec_valid = false;
        c->u_pending = c->f_pending = c->rfields_valid = false;
        c->uend_pending = false;
        c->uend_gen = -1;
        c->u0_src = nullptr;
        c->u0_spec_only = true;
        return SDC_OK;
    }
    MATERIALIZE_UEND(c);
    if (handover && c->kind == 0 && c->deferred) {
        // Fourier-space data flow: nothing is copied.  The spectrum of the last node and the spectrum of u[0] trade
        // places (the first sweep after the predictor rewrites every node spectrum), the end-value buffer becomes the
        // place where the new start value lies, and the other buffer of the pair takes the next end value.
        if (!c->UEND2) {
            HIPCHK(c, hipMalloc((void**)&c->UEND2, c->N * sizeof(double)));
            c->bytes += c->N * sizeof(double);
        }
        int rcf = uend_write_fence(c);  // (a send that still reads the old end value of the buffer we are about to reuse)
        if (rcf != SDC_OK) return rcf;
        c->u0_src = c->UEND;
        std::swap(c->UEND, c->UEND2);
        std::swap(c->S0, c->SL);
        c->spec0_valid = true;
        c->spec_valid = false;  // the node values of the step that just ended are gone (reset_level, core/level.py:110-131)
        c->u_pending = c->f_pending = c->rfields_valid = false;
        c->uend_gen = -1;       // UEND now names the other buffer
        return SDC_OK;
    }
    c->u0_src = nullptr;  // U[0] is overwritten as a whole
    c->u0_spec_only = false;
    HIPCHK(c, hipMemcpyAsync(c->U0, c->UEND, c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    if (handover) {
        HIPCHK(c, hipMemcpyAsync(c->S0, c->SL, sizeof(cd) * c->Nc, hipMemcpyDeviceToDevice, c->stream));
        c->spec0_valid = true;
        if (c->spec_virtual > 0) {  // an iterate that was never stored was a function of the OLD S0: gone with it (the
            c->spec_valid = false;  // node values themselves are in U - this path is not the deferred one)
            c->spec_virtual = 0;
        }
    } else {
        c->spec0_valid = false;
    }
    return SDC_OK;
}

int sdc_set_timeslice_options(sdc_ctx* c, int trail_sources, int defer_last_pass, int split_send) {
    if (!c || trail_sources < 0) return fail(c, SDC_ERR_PARAM, "trail: a number of start values >= 0");
    // (the recomputing launch holds TRAIL_S start values in registers; one more may be on the trail: the one received after the
    // last sweep of a step, which only the residual against it sees - as a difference of two start values)
    if (trail_sources > TRAIL_S + 1) trail_sources = TRAIL_S + 1;
    FLUSH_X(c);
    if (trail_sources < c->trail_max) STORE_SPECTRA(c, false);
    c->trail_max = trail_sources;
    c->defer_x = defer_last_pass != 0;
    c->pipeline_x = defer_last_pass == 1;   // (2: put off, but never behind the next sweep's launches)
    c->split_send = split_send != 0;
    return SDC_OK;
}

int sdc_set_early_end_point(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    c->early_uend = on != 0;
    c->uend_ev_recorded = false;
    return SDC_OK;
}

int sdc_stream_wait_uend(sdc_ctx* c, void* other_stream) {
    if (!c) return SDC_ERR_PARAM;
    MATERIALIZE_UEND(c);
    if (!c->uend_ev) HIPCHK(c, hipEventCreateWithFlags(&c->uend_ev, hipEventDisableTiming));
    if (!(c->uend_ev_recorded && c->uend_gen >= 0 && c->uend_gen == c->spec_gen)) {
        // UEND was written by the latest work on the engine's stream (sdc_end_point): everything so far
        HIPCHK(c, hipEventRecord(c->uend_ev, c->stream));
    }
    c->uend_ev_recorded = false;
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)other_stream, c->uend_ev, 0));
    return SDC_OK;
}

int sdc_set_keep_residual_fields(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    c->keep_rfields = on != 0;
    if (!on) c->rfields_valid = false;
    return SDC_OK;
}

int sdc_replace_u0(sdc_ctx* c, const double* src) {
    if (!c || !src) return fail(c, SDC_ERR_PARAM, "null pointer");
    FLUSH_X(c);
    ENSURE_U0(c);  // (the update of the kept residual fields reads the old start value)
    int rcm = materialize(c, c->spread_pending, false);  // pending copies of the OLD u[0] are stored first
    if (rcm != SDC_OK) return rcm;
    const bool fast = c->rfields_valid && c->res_valid && c->u_pending && !c->tau_active;
    if (fast) {
        HIPCHK(c, hipMemsetAsync(c->res_dev, 0, sizeof(unsigned long long) * 8, c->stream));
        LaunchTimer lt(c, pname("replace_u0", c->M));
        const int grid = grid_for(c->N / 2, 256);
#define RCASE(MM) \
    case MM: hipLaunchKernelGGL((k_replace_u0<MM>), dim3(grid), dim3(256), 0, c->stream, src, c->U0, c->U, c->N, c->res_dev); break;
        switch (c->M) { RCASE(1) RCASE(2) RCASE(3) RCASE(4) RCASE(5) RCASE(6) RCASE(7) RCASE(8) }
#undef RCASE
        HIPCHK(c, hipGetLastError());
        c->rfields_valid = false;  // the stored fields belong to the old u[0]
    } else {
        HIPCHK(c, hipMemcpyAsync(c->U0, src, c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        c->res_valid = false;
    }
    c->res_spread = false;
    c->spec0_valid = false;
    c->spec_spread = false;
    return SDC_OK;
}

// ---- start and end values as SPECTRA (time-parallel runs whose levels sweep in Fourier space) ---------------------
// Between such levels the forward hand-over uend -> u[0] (controller_MPI.py:218-305) needs neither the inverse transform
// of the sender's last node nor the forward transform of the receiver's new start value: the last node's spectrum IS the
// message.  sdc_comm_set_format(ctx, 1) turns that on; the three calls below are what the communicator uses.
static bool spectral_level(const sdc_ctx* c) {
    return c->kind == 0 && c->reuse && c->deferred && c->ndim >= 2 && fourier_ok(c) && c->have_stencil[0] && !c->spectral_op &&
           c->expl_kind != SDC_EXPL_REACTION && c->solver_kind == 0 && c->fuse_residual;
}

int sdc_spectral_handover_ok(sdc_ctx* c) { return c && spectral_level(c) ? 1 : 0; }

// what sdc_comm_set_format does to the engine, without a communicator (a caller that moves the spectra itself)
int sdc_set_wire_spectral(sdc_ctx* c, int on) {
    if (!c) return SDC_ERR_PARAM;
    if (on && !spectral_level(c)) return fail(c, SDC_ERR_STATE, "this level does not sweep in Fourier space");
    c->wire_spectral = on != 0;
    return SDC_OK;
}

// device address of the half spectrum (Nc complex values) a received start value is to be written to
void* sdc_spectrum_inbox(sdc_ctx* c) {
    if (!c || ensure_spec_cache(c) != SDC_OK) return nullptr;
    if (!c->Sin) {
        c->Sin = spool_get(c);
        if (!c->Sin) return nullptr;
    }
    // (a pass that was put off reads the old start value, which - off a trail - is the buffer the next message lands in)
    if (c->xp.pending && c->xp.has_delta && (c->xp.d_old == c->Sin || c->xp.d_new == c->Sin) && flush_x(c) != SDC_OK) return nullptr;
    return c->Sin;
}

// device address of the spectrum of the end value (sdc_end_point has been called), and - stream != null - make that
// stream wait until it is complete (and for nothing queued on the engine's stream after it)
void* sdc_end_spectrum(sdc_ctx* c, void* stream) {
    if (!c || !c->have_coeffs) return nullptr;
    const cd* src = nullptr;
    hipEvent_t ev = nullptr;
    if (c->Sx && c->spec_valid && c->uend_gen >= 0 && c->uend_gen == c->spec_gen) {
        if (store_spectra(c, true) != SDC_OK) return nullptr;  // (an iterate that was never stored: now its last node is)
        src = c->SL;
        if (c->sl_ev_recorded) ev = c->sl_ev;
    } else {
        // any other end value (a predictor's copy of u[0], a collocation update): forward transform of UEND
        if (materialize_uend(c) != SDC_OK) return nullptr;
        if (!c->Wend) {
            if (hipMalloc((void**)&c->Wend, sizeof(cd) * c->Nc) != hipSuccess) {
                fail(c, SDC_ERR_NOMEM, "end spectrum: out of device memory");
                return nullptr;
            }
            c->bytes += sizeof(cd) * c->Nc;
        }
        FieldPtrs p;
        memset(&p, 0, sizeof p);
        p.in[0] = c->UEND;
        if (ensure_work(c) != SDC_OK || fwd_transform(c, 1, p, c->Wend, 0) != SDC_OK) return nullptr;
        src = c->Wend;
    }
    if (stream) {
        if (!c->sl_ev && hipEventCreateWithFlags(&c->sl_ev, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (!ev) {
            if (hipEventRecord(c->sl_ev, c->stream) != hipSuccess) return nullptr;
            ev = c->sl_ev;
        }
        c->sl_ev_recorded = false;
        if (hipStreamWaitEvent((hipStream_t)stream, ev, 0) != hipSuccess) return nullptr;
    }
    return (void*)src;
}

// u[0] <- the field whose spectrum has been written to sdc_spectrum_inbox().  While the work spectra still hold the
// residual of the cached iterate after its z / y inverse passes (rlines_valid: the last sweep only reduced norms), the
// node norms against the new start value follow from ONE more field through those passes - the residual of every node
// changes by d = new - old (core/sweeper.py:186-199 is linear in u[0]) - and the last pass over W + d.
int sdc_replace_u0_spectrum(sdc_ctx* c) {
    if (!c || !c->Sin) return fail(c, SDC_ERR_STATE, "no spectrum inbox (sdc_spectrum_inbox)");
    if (!spectral_level(c)) return fail(c, SDC_ERR_STATE, "this level does not sweep in Fourier space");
    // An iterate that was not stored is a function of the start values so far: on a trail the old one simply stays on record
    // (the received one joins it); otherwise the iterate is written out before its only source goes
    const bool on_trail = c->spec_valid && c->spec_virtual > 0 && c->trail_ns > 0 && c->trail_ns < c->trail_max &&
                          c->trail_ns < MAXTRAIL && c->trail_src[c->trail_ns - 1] == c->S0 && !c->tau_active;
    if (!on_trail) STORE_SPECTRA(c, false);
    int rcm = materialize(c, c->spread_pending, false);  // pending copies of the OLD u[0] are stored first
    if (rcm != SDC_OK) return rcm;
    const bool linear_shift = c->res_valid && c->u_pending && c->spec0_valid && c->spec_valid && !c->tau_active &&
                              c->expl_kind != SDC_EXPL_FORCING;
    bool keep_old = false;
    if (c->xp.pending && !c->xp.has_delta && linear_shift && c->n >= 64) {
        // the residual lines of the sweep still wait for their last pass: it will reduce the norms before and after this
        // receive in one go (flush_x) - the residual of every node changes by d = new - old (core/sweeper.py:186-199 is linear
        // in u[0]).  Nothing is launched here.
        c->xp.has_delta = true;
        c->xp.d_old = c->S0;
        c->xp.d_new = c->Sin;
        keep_old = !on_trail;   // (off a trail the old spectrum would become the next inbox: it has to outlive the put-off pass)
    } else {
        FLUSH_X(c);
        const bool fast = c->rlines_valid && linear_shift;
        if (fast) {
            HIPCHK(c, hipMemsetAsync(c->res_dev, 0, sizeof(unsigned long long) * 8, c->stream));
            // the difference goes through the passes in place over the old spectrum - or, when that one stays on the trail,
            // through a spare buffer
            cd* dbuf = on_trail ? spool_get(c) : c->S0;
            if (!dbuf) return SDC_ERR_NOMEM;
#define CALL(NN) residual_shift_n<NN>(c, dbuf, c->Sin, c->S0)
            int rcd = [&]() -> int { N_DISPATCH(c, CALL) }();
#undef CALL
            if (on_trail) spool_put(c, dbuf);
            if (rcd != SDC_OK) return rcd;
            c->rlines_valid = false;  // W + d is what the norms describe now, W alone is not
        } else {
            c->res_valid = false;
        }
    }
    if (on_trail) {
        cd* next_inbox = spool_get(c);
        if (!next_inbox) return SDC_ERR_NOMEM;
        c->trail_src[c->trail_ns++] = c->Sin;
        c->S0 = c->Sin;
        c->Sin = next_inbox;
    } else if (keep_old) {
        cd* next_inbox = spool_get(c);
        if (!next_inbox) return SDC_ERR_NOMEM;
        c->xp.d_old_spare = true;   // (flush_x hands it back)
        c->S0 = c->Sin;
        c->Sin = next_inbox;
    } else {
        std::swap(c->S0, c->Sin);
    }
    c->spec0_valid = true;
    c->u0_spec_only = true;  // U[0] is produced from the spectrum when somebody reads it there
    c->u0_src = nullptr;
    c->res_spread = false;
    c->spec_spread = false;
    c->f0_pending = false;
    return SDC_OK;
}

// A new block starts from the value whose spectrum lies in the inbox (the end value of the previous block, broadcast as a
// spectrum: sdc_comm_bcast_end_spectrum): what sdc_advance does on the rank that owns that end value, for everybody else.
// Nothing of the finished step survives (reset_level, core/level.py:110-131).
int sdc_start_from_spectrum(sdc_ctx* c) {
    if (!c || !c->Sin) return fail(c, SDC_ERR_STATE, "no spectrum inbox (sdc_spectrum_inbox)");
    if (!spectral_level(c)) return fail(c, SDC_ERR_STATE, "this level does not sweep in Fourier space");
    FLUSH_X(c);   // (norms of the finished block that still wait for their last pass read the spectra that move below)
    {
        int rct = trail_reset(c);
        if (rct != SDC_OK) return rct;
    }
    std::swap(c->S0, c->Sin);
    c->spec0_valid = true;
    c->u0_spec_only = true;
    c->u0_src = nullptr;
    c->spec_valid = c->spec_spread = false;
    c->spec_virtual = 0;
    c->u_pending = c->f_pending = c->spread_pending = c->f0_pending = false;
    c->rfields_valid = c->rlines_valid = c->res_valid = c->res_spread = false;
    c->uend_pending = false;
    c->uend_gen = -1;
    return SDC_OK;
}

int sdc_integrate(sdc_ctx* c, double dt, double* const* dst) {
    if (!c || !dst) return fail(c, SDC_ERR_PARAM, "null pointer");
    if (!c->have_coeffs) return fail(c, SDC_ERR_STATE, "coefficients not set (sdc_set_coeffs)");
    int rcm = materialize(c, false, true);
    if (rcm != SDC_OK) return rcm;
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

int sdc_vec_box(sdc_ctx* c, int ndim, const long long* shape, const long long* start, const long long* step,
                const long long* count, double* field, double* compact, int direction, double value) {
    CTX_OR_DEFAULT(c);
    if (ndim < 1 || ndim > 4 || !shape || !start || !step || !count || !field || direction < 0 || direction > 2 ||
        (direction != 2 && !compact))
        return fail(c, SDC_ERR_PARAM, "bad box arguments");
    BoxArgs a;
    memset(&a, 0, sizeof a);
    // right-aligned into four axes; strides of the C-ordered field
    long long stride = 1, total = 1;
    for (int d = 0; d < 4; ++d) {
        a.c[d] = 1;
        a.s[d] = 0;
    }
    for (int d = ndim - 1; d >= 0; --d) {
        const int q = 4 - ndim + d;
        if (shape[d] < 1 || count[d] < 0 || step[d] == 0) return fail(c, SDC_ERR_PARAM, "bad box arguments (axis %d)", d);
        if (count[d] > 0) {
            const long long last = start[d] + (count[d] - 1) * step[d];
            if (start[d] < 0 || start[d] >= shape[d] || last < 0 || last >= shape[d])
                return fail(c, SDC_ERR_PARAM, "index out of bounds on axis %d (size %lld)", d, shape[d]);
        }
        a.c[q] = count[d];
        a.s[q] = step[d] * stride;
        a.off += start[d] * stride;
        stride *= shape[d];
        total *= count[d];
    }
    if (total == 0) return SDC_OK;
    a.field = field;
    a.compact = compact;
    a.value = value;
    a.dir = direction;
    LaunchTimer lt(c, "box");
    hipLaunchKernelGGL(k_box, dim3(grid_for((size_t)total, 256)), dim3(256), 0, c->stream, a);
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
    return sdc_transfer_apply_batch(stream, 1, ndim, n_out, n_in, width, idx, w, in, out);
}

int sdc_transfer_apply_batch(void* stream, int nfields, int ndim, int n_out, int n_in, int width, const int* idx,
                             const double* w, const double* in, double* out) {
    return sdc_transfer_apply_batch_acc(stream, nfields, ndim, n_out, n_in, width, idx, w, in, out, 0);
}

int sdc_transfer_apply_batch_acc(void* stream, int nfields, int ndim, int n_out, int n_in, int width, const int* idx,
                                 const double* w, const double* in, double* out, int accumulate) {
    if (nfields < 1 || ndim < 1 || ndim > 3 || n_out < 1 || n_in < 1 || width < 1 || !idx || !w || !in || !out)
        return fail(nullptr, SDC_ERR_PARAM, "bad transfer arguments");
    // separable: one pass per axis (cost ~ ndim * width per point instead of width^ndim), last axis first so that
    // intermediate fields stay as small as possible when prolonging; two scratch fields ping-pong in between
    struct Scratch {  // per host thread (= per context user), released when the thread ends
        double* p[2] = {nullptr, nullptr};
        size_t len = 0;
        ~Scratch() {
            for (int k = 0; k < 2; ++k)
                if (p[k]) (void)hipFree(p[k]);
        }
    };
    static thread_local Scratch sc;
    double** scratch = sc.p;
    size_t& scratch_len = sc.len;
    const int nmax = n_out > n_in ? n_out : n_in;
    size_t need = (size_t)nfields;  // the fields are one more (outermost) dimension of every pass
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
#ifndef SDC_XFER_FUSED3
#define SDC_XFER_FUSED3 1
#endif
    if (SDC_XFER_FUSED3 && ndim == 3 && n_out < n_in && width <= 3 && !accumulate && n_out <= 65535 &&
        (size_t)n_out * nfields <= 65535) {
        // coarsening with a narrow table (rorder 2: three entries per row): all three axes in one launch, same bits
        XferArgs a;
        memset(&a, 0, sizeof a);
        a.idx = idx;
        a.w = w;
        a.W = width;
        a.n_in = n_in;
        a.n_out = n_out;
        a.in = in;
        a.out = out;
        const unsigned bx = n_out >= 256 ? 256u : (unsigned)((n_out + 63) / 64 * 64);
        const dim3 g((unsigned)((n_out + bx - 1) / bx), (unsigned)n_out, (unsigned)n_out * (unsigned)nfields), blk(bx);
        switch (width) {
        case 1: hipLaunchKernelGGL((k_xfer_fused3<1>), g, blk, 0, (hipStream_t)stream, a, (unsigned)nfields); break;
        case 2: hipLaunchKernelGGL((k_xfer_fused3<2>), g, blk, 0, (hipStream_t)stream, a, (unsigned)nfields); break;
        default: hipLaunchKernelGGL((k_xfer_fused3<3>), g, blk, 0, (hipStream_t)stream, a, (unsigned)nfields); break;
        }
        HIPCHK(nullptr, hipGetLastError());
        return SDC_OK;
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
        a.outer = (size_t)nfields;
        a.inner = 1;
        for (int d = 0; d < axis; ++d) a.outer *= (size_t)dims[d];
        for (int d = axis + 1; d < ndim; ++d) a.inner *= (size_t)dims[d];
        a.in = src;
        a.out = pass == ndim - 1 ? out : scratch[pass & 1];
        a.accumulate = (accumulate && pass == ndim - 1) ? 1 : 0;
        const size_t total = a.outer * (size_t)n_out * a.inner;
        const size_t in_total = a.outer * (size_t)n_in * a.inner;
        const bool small = total < 0xffffffffull && in_total < 0xffffffffull;
        // strided axis: output row from the block index (uniform table loads), 16-byte accesses along the contiguous
        // direction, a few consecutive output rows per block when refining (their input rows overlap: L1)
        const bool refine = n_out > n_in;
        const size_t rows_y = a.outer * (size_t)(refine ? (n_out + 3) / 4 : n_out);
        const bool aligned = (((uintptr_t)a.in | (uintptr_t)a.out) & 15) == 0;
        const size_t nq = (a.inner / 2 + 255) / 256;
        if (a.inner >= 2 && a.inner % 2 == 0 && nq * rows_y < 0x7fffffffull && aligned) {
            const dim3 grid((unsigned)(nq * rows_y));
            if (refine) hipLaunchKernelGGL((k_xfer_axis_rows<4>), grid, dim3(256), 0, (hipStream_t)stream, a, (unsigned)nq);
            else hipLaunchKernelGGL((k_xfer_axis_rows<1>), grid, dim3(256), 0, (hipStream_t)stream, a, (unsigned)nq);
        } else if (a.inner == 1) {
            const dim3 g(grid_for(total, 256));
#define XL(WT_)                                                                                                \
    if (small) hipLaunchKernelGGL((k_xfer_line<unsigned, WT_>), g, dim3(256), 0, (hipStream_t)stream, a);      \
    else hipLaunchKernelGGL((k_xfer_line<size_t, WT_>), g, dim3(256), 0, (hipStream_t)stream, a)
            switch (width) {
            case 1: XL(1); break;
            case 2: XL(2); break;
            case 3: XL(3); break;
            case 4: XL(4); break;
            case 6: XL(6); break;
            case 8: XL(8); break;
            default: XL(0); break;
            }
#undef XL
        } else if (!refine) {
            if (small) hipLaunchKernelGGL((k_xfer_axis<unsigned, 1>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((k_xfer_axis<size_t, 1>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
        } else {
            constexpr int CH = SDC_XFER_CH;
            const size_t threads = a.outer * (size_t)((n_out + CH - 1) / CH) * a.inner;
            if (small) hipLaunchKernelGGL((k_xfer_axis<unsigned, CH>), dim3(grid_for(threads, 256)), dim3(256), 0, (hipStream_t)stream, a);
            else hipLaunchKernelGGL((k_xfer_axis<size_t, CH>), dim3(grid_for(threads, 256)), dim3(256), 0, (hipStream_t)stream, a);
        }
        dims[axis] = n_out;
        src = a.out;
    }
    HIPCHK(nullptr, hipGetLastError());
    return SDC_OK;
}

int sdc_transfer_apply_nested(void* stream, int nfields, int ndim, int n_out, int n_in, int width, const int* idx,
                              const double* w, const double* in, const double* in_minus, double* out,
                              const double* out_minus, int accumulate) {
    if (nfields < 1 || ndim < 1 || ndim > 3 || n_out < 1 || n_in < 1 || width < 1 || !idx || !w || !in || !out)
        return fail(nullptr, SDC_ERR_PARAM, "bad transfer arguments");
#ifndef SDC_XFER_NESTED
#define SDC_XFER_NESTED 1
#endif
    hipStream_t st = (hipStream_t)stream;
    NestedArgs a;
    memset(&a, 0, sizeof a);
    a.in = in;
    a.in_minus = in_minus;
    a.out = out;
    a.out_minus = out_minus;
    a.idx = idx;
    a.w = w;
    a.n_out = n_out;
    a.n_in = n_in;
    a.W = width;
    a.accumulate = accumulate;
    const bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    // (threads along k: the largest power of two that divides n_out, at most 256)
    unsigned kx = (unsigned)(n_out & -n_out);
    if (kx > 256u) kx = 256u;
    const unsigned jy = kx ? 256u / kx : 1u;
    if (SDC_XFER_NESTED && ndim == 3 && n_in == 2 * n_out && width == 3 && !accumulate && !in_minus && kx >= 16 &&
        (unsigned)n_out % jy == 0 && n_out % 8 == 0 && nfields <= 65535 && aligned) {
        // coarse planes a thread walks over: the longer the walk, the fewer fine planes are read twice (1 + 1 / (2 TI)) - but the
        // launch wants a few thousand workgroups
        const unsigned kx_blocks = (unsigned)n_out / kx;
        const size_t per_plane = (size_t)kx_blocks * ((unsigned)n_out / jy) * (size_t)nfields;
        // (measured at 3 x 256^3 -> 128^3, scripts/probes/xfer_nested_probe.hip: TI = 1: 146 us, 2: 125, 4: 133 - 139, 8: 146,
        // 16: 172 - few, long workgroups leave CUs idle at the end; rows per workgroup 1 / 2 / 4 / 8: 138 / 138 / 153 / 156)
        const int TI = per_plane * (n_out / 8) >= 65536 ? 8 : (per_plane * (n_out / 4) >= 16384 ? 4 : 2);
        const dim3 g(kx_blocks * ((unsigned)n_out / jy), (unsigned)n_out / TI, (unsigned)nfields), blk(kx, jy);
        if (TI == 8) hipLaunchKernelGGL((k_restrict3_nested<8>), g, blk, 0, st, a, kx_blocks);
        else if (TI == 4) hipLaunchKernelGGL((k_restrict3_nested<4>), g, blk, 0, st, a, kx_blocks);
        else hipLaunchKernelGGL((k_restrict3_nested<2>), g, blk, 0, st, a, kx_blocks);
        HIPCHK(nullptr, hipGetLastError());
        return SDC_OK;
    }
    if (SDC_XFER_NESTED && ndim == 3 && n_out == 2 * n_in && !out_minus && n_in % 8 == 0 && n_in >= 2 * (8 + width) &&
        (width == 2 || width == 4 || width == 6 || width == 8) && (size_t)(n_in / 8) * nfields <= 65535) {
        const unsigned tiles = (unsigned)n_in / 8;
        const dim3 g(tiles, tiles, tiles * (unsigned)nfields), blk(256);
        switch (width) {
        case 2: hipLaunchKernelGGL((k_prolong3_nested<2>), g, blk, 0, st, a); break;
        case 4: hipLaunchKernelGGL((k_prolong3_nested<4>), g, blk, 0, st, a); break;
        case 6: hipLaunchKernelGGL((k_prolong3_nested<6>), g, blk, 0, st, a); break;
        default: hipLaunchKernelGGL((k_prolong3_nested<8>), g, blk, 0, st, a); break;
        }
        HIPCHK(nullptr, hipGetLastError());
        return SDC_OK;
    }
    // everything else: the separable passes, the differences by launches of their own (a scratch field per host thread)
    size_t len_in = (size_t)nfields, len_out = (size_t)nfields;
    for (int d = 0; d < ndim; ++d) {
        len_in *= (size_t)n_in;
        len_out *= (size_t)n_out;
    }
    struct Tmp {
        double* p = nullptr;
        size_t len = 0;
        ~Tmp() {
            if (p) (void)hipFree(p);
        }
    };
    static thread_local Tmp tmp;
    const double* src = in;
    if (in_minus) {
        if (tmp.len < len_in) {
            if (tmp.p) (void)hipFree(tmp.p);
            tmp.p = nullptr;
            tmp.len = 0;
            if (hipMalloc((void**)&tmp.p, len_in * sizeof(double)) != hipSuccess)
                return fail(nullptr, SDC_ERR_NOMEM, "transfer scratch allocation failed");
            tmp.len = len_in;
        }
        hipLaunchKernelGGL(k_axpby, dim3(grid_for(len_in, 256)), dim3(256), 0, st, len_in, 1.0, in, -1.0, in_minus, tmp.p);
        src = tmp.p;
    }
    int rc = sdc_transfer_apply_batch_acc(stream, nfields, ndim, n_out, n_in, width, idx, w, src, out, accumulate);
    if (rc != SDC_OK) return rc;
    if (out_minus) {
        if (accumulate) return fail(nullptr, SDC_ERR_PARAM, "out_minus and accumulate exclude each other");
        hipLaunchKernelGGL(k_axpby, dim3(grid_for(len_out, 256)), dim3(256), 0, st, len_out, 1.0, out, -1.0, out_minus, out);
    }
    HIPCHK(nullptr, hipGetLastError());
    return SDC_OK;
}

int sdc_fft_prolong(sdc_ctx* coarse, sdc_ctx* fine, const double* src, double* dst, double factor) {
    if (!coarse || !fine || !src || !dst) return fail(fine, SDC_ERR_PARAM, "null pointer");
    if (coarse->ndim != fine->ndim)
        return fail(fine, SDC_ERR_UNSUPPORTED, "Fourier prolongation between grids of one dimension (got %d-D -> %d-D)",
                    coarse->ndim, fine->ndim);
    if (coarse->kind != 0 || fine->kind != 0 || !fourier_ok(coarse) || !fourier_ok(fine) || fine->n < coarse->n ||
        coarse->n < 2 || fine->n > 1024)
        return fail(fine, SDC_ERR_UNSUPPORTED, "Fourier prolongation needs n = 2^p <= 1024 on both grids (%d -> %d)",
                    coarse->n, fine->n);
    int rc = ensure_work(coarse);
    if (rc == SDC_OK) rc = ensure_work(fine);
    if (rc != SDC_OK) return rc;
    FieldPtrs p;
    memset(&p, 0, sizeof p);
    p.in[0] = src;
    rc = fwd_transform(coarse, 1, p, coarse->W, 0);
    if (rc != SDC_OK) {
        fine->err = coarse->err;
        return rc;
    }
    if (coarse->stream != fine->stream) {
        HIPCHK(fine, hipEventRecord(coarse->ev0, coarse->stream));
        HIPCHK(fine, hipStreamWaitEvent(fine->stream, coarse->ev0, 0));
    }
    const int nc = coarse->n, nf = fine->n;
    {
        LaunchTimer lt(fine, "pad_spectrum");
        if (fine->ndim == 1)
            hipLaunchKernelGGL(k_pad_spectrum_1d, dim3(grid_for(nf, 256)), dim3(256), 0, fine->stream, coarse->W, fine->W,
                               nc, nf);
        else if (fine->ndim == 2)
            hipLaunchKernelGGL(k_pad_spectrum_2d, dim3(grid_for((size_t)(nf / 2 + 1) * nf, 256)), dim3(256), 0,
                               fine->stream, coarse->W, fine->W, nc, nf);
        else
            hipLaunchKernelGGL(k_pad_spectrum_3d, dim3(grid_for((size_t)(nf / 2 + 1) * nf * nf, 256)), dim3(256), 0,
                               fine->stream, coarse->W, fine->W, nc, nf);
    }
    HIPCHK(fine, hipGetLastError());
    memset(&p, 0, sizeof p);
    p.out[0] = dst;
    const double scale = factor / (double)fine->N;
#define CALL(NN) inverse_passes_n<NN>(fine, 1, fine->W, fine->W, p, nullptr, scale)
    N_DISPATCH(fine, CALL)
#undef CALL
}

int sdc_odd_mirror(sdc_ctx* c, double* field, int n_interior) {
    CTX_OR_DEFAULT(c);
    if (!field || n_interior < 1) return fail(c, SDC_ERR_PARAM, "bad odd-extension arguments");
    hipLaunchKernelGGL(k_odd_mirror, dim3((n_interior + 255) / 256), dim3(256), 0, c->stream, field, n_interior);
    HIPCHK(c, hipGetLastError());
    c->res_valid = false;
    return SDC_OK;
}

int sdc_odd_extend(sdc_ctx* c, const double* interior, double* ext, int n_interior, int ndim) {
    CTX_OR_DEFAULT(c);
    if (!interior || !ext || n_interior < 1 || ndim < 1 || ndim > 3) return fail(c, SDC_ERR_PARAM, "bad odd-extension arguments");
    size_t total = 1;
    for (int d = 0; d < ndim; ++d) total *= (size_t)(2 * (n_interior + 1));
    hipLaunchKernelGGL(k_odd_extend_nd, dim3(grid_for(total, 256)), dim3(256), 0, c->stream, interior, ext, n_interior, ndim);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

int sdc_odd_extract(sdc_ctx* c, const double* ext, double* interior, int n_interior, int ndim) {
    CTX_OR_DEFAULT(c);
    if (!interior || !ext || n_interior < 1 || ndim < 1 || ndim > 3) return fail(c, SDC_ERR_PARAM, "bad odd-extension arguments");
    size_t total = 1;
    for (int d = 0; d < ndim; ++d) total *= (size_t)n_interior;
    hipLaunchKernelGGL(k_odd_extract_nd, dim3(grid_for(total, 256)), dim3(256), 0, c->stream, ext, interior, n_interior, ndim);
    HIPCHK(c, hipGetLastError());
    return SDC_OK;
}

int sdc_sync(sdc_ctx* c) {
    if (!c) return SDC_ERR_PARAM;
    FLUSH_X(c);
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

#include "comm.hpp"
