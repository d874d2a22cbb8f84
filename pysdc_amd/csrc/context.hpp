// libsdcmi: context of one level (device slabs, coefficients, operator tables) and launch timing.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>
#include "../../include/sdcmi.h"
#include "fft.hpp"
#define MAXM 8
#define MAXSTEN 12

// ------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------
// coefficients of a sweep in the transformed domain (inner M x M blocks, scaled by dt; see SpecArgs)
struct SpecCoef {
    double gI[MAXM][MAXM], gE[MAXM][MAXM], cI[MAXM][MAXM], cE[MAXM][MAXM], alpha[MAXM];
    int coupled, real_sym, has_e, pad_;
};
struct Stencil {
    int npts = 0;
    int off[MAXSTEN];
    double w[MAXSTEN];
};

struct ProfEntry {
    double ms = 0;
    int calls = 0;
};

struct CommState;  // comm.hpp: RCCL communicator, message stream, inbox

#define MAXTRAIL 8     // start values a trail of unstored sweeps may depend on (time-parallel runs)
#define MAXVSWEEPS 32  // sweeps whose start-value indices are on record

// Residual norms whose last inverse pass (x) has not run yet: the residual lines sit in W after their z / y passes and wait -
// for the start value the time-rank is about to receive (then ONE x pass reduces the norms before and after the receive,
// the difference field added on the way), or for somebody who wants the numbers.
struct PendingTicket {
    unsigned long long seq;
    int type, after;       // after = 1: the residual against the start value received since the sweep
    double restol;
};
struct PendingX {
    bool pending = false;      // W holds nf residual fields after their z / y inverse passes, norms not reduced yet
    bool has_delta = false;    // ... and u[0] has been replaced since: d_new - d_old is added for the second set of norms
    int nf = 0;
    const cd *d_old = nullptr, *d_new = nullptr;
    bool d_old_spare = false;          // d_old was kept alive for this pass only: a spare buffer again afterwards
    cd* dz = nullptr;                  // d_new - d_old after its z pass already (the next sweep's trail launch made it on the way)
    cd* work = nullptr;                // the work spectra that hold the residual lines
    unsigned long long *norms = nullptr, *normsA = nullptr;   // slots of the norms after / before the receive
    std::vector<PendingTicket> tickets;
};

// one finished residual as the device leaves it in pinned host memory (sdc_residual_post)
#define RES_RING 256
struct ResRecord {
    unsigned long long seq;    // ticket this record answers; written LAST
    double residual;           // the number compute_residual stores in L.status.residual (core/sweeper.py:200-215)
    double norms[MAXM];        // node-wise max norms of the collocation residual
    int converged;             // residual <= restol (check_convergence.py:72-75), taken on the device
    int pad;
};

struct sdc_ctx {
    CommState* comm = nullptr;
    int device = 0, ndim = 0, n = 0, M = 0, ncomp = 1;
    size_t N = 0;       // n^ndim
    size_t Nc = 0;      // complex entries of one spectrum field
    hipStream_t stream = nullptr;
    double *U = nullptr, *F = nullptr, *TAU = nullptr, *UEND = nullptr, *profile = nullptr;
    // The node fields live in blocks of their own that are allocated when something first touches them in real space: U0 = u[0]
    // (always there), Un = U[1..M], F0 = f[0][ncomp], Fn = f[1..M][ncomp], Sn = the node spectra S[0..M-2] (the last node's has
    // its own buffer).  The sweeps that stay in Fourier space never touch Un, Fn, Sn - 120 GB at 1024^3 that are simply never
    // allocated.  `U` / `F` are the bases the node index counts from: U + m N = U[m], F + (m ncomp + comp) N = f[m].comp for
    // m >= 1 (Un - N, Fn - ncomp N); u[0] and f[0] are reached through U0 / F0 ONLY.
    double *U0 = nullptr, *Un = nullptr, *F0 = nullptr, *Fn = nullptr;
    cd* Sn = nullptr;
    size_t lazy_min_bytes = (size_t)64 << 20;   // contexts with smaller fields allocate everything at once (SDC_LAZY_MIN_BYTES)
    cd* W = nullptr;
    cd *S = nullptr, *S0 = nullptr;  // spectral cache: transforms of U[1..M] and of U[0] (lazy)
    // The transform of the LAST node and the transform of u[0] trade places from one time step to the next (the end
    // value of a step is the start value of the following one, sdc_advance): S0 and SL are, in either order, the last
    // slot of the S block and the separately allocated spectrum Sx.  Fields 0..M-2 are S + m*Nc, field M-1 is SL.
    cd *SL = nullptr, *Sx = nullptr;
    // Time-parallel runs whose wire carries SPECTRA (sdc_comm_set_format): a third spectrum buffer Sy joins the rotation -
    // Sin names the one a received start value lands in; when it is taken (replace_u0_spectrum) it becomes S0 and the old
    // S0 buffer the next inbox.  sl_ev: recorded right after the launch that leaves the last node's spectrum final, so
    // that a message stream can pick it up while the residual passes still run.  rlines_valid: the work spectra W still
    // hold the residual of the current iterate after its z / y inverse passes (the x pass only reduced a norm), which lets
    // a new start value update the node norms by ONE more field through the pipeline instead of M.
    cd* Sin = nullptr;
    bool dz_written = false;   // the trail launch of the sweep in progress took the difference line along (SpecArgs::dz)
    cd* trail_nyq = nullptr;   // [lines][M]: the Nyquist mode's residual of every z line (k_trail_nyq -> k_trail_z)
    // spectrum-sized buffers nobody uses right now (sources of a finished trail, difference spectra, ...) and the ones this
    // context allocated one by one (freed with it)
    std::vector<cd*> spool, spool_owned;
    // Trail (time-parallel levels, spectra on the wire): the iterate after spec_virtual unstored sweeps as a function of the
    // start values the slice has had since its spread predictor - trail_src[0] the block's, trail_src[i] the i-th one received -
    // and of which sweep started from which (vsrc).  trail_ns = 0: no trail (the iterate depends on S0 alone, as in serial runs).
    int trail_max = 0, trail_ns = 0;   // (0: iterates are stored - measured faster than the recomputing launch, profiles/r05)
    const cd* trail_src[MAXTRAIL] = {};
    unsigned char vsrc[MAXVSWEEPS] = {};
    PendingX xp;
    unsigned long long* res_devA = nullptr;   // norms BEFORE the receive when one x pass delivers both sets
    bool defer_x = true;                      // spectra on the wire: the x pass of a sweep's residual waits for the receive
    bool split_send = false;                  // ... and the last node's spectrum is produced by a launch of its own, first
    bool sl_ev_by_split = false;
    // ... and when the NEXT sweep arrives while that pass still waits (nobody asked for the numbers: a fixed number of sweeps),
    // the sweep's z / y launches go first - into a second set of work spectra - and the passes put off for the previous
    // iterate follow them: the last node's spectrum is final (and on the wire) one z launch after the receive, and the
    // put-off passes run while the message travels
    bool pipeline_x = true;
    cd* Wb = nullptr;                         // the other set of work spectra
    unsigned long long* res_bank[2] = {nullptr, nullptr};   // norm slots: two banks (one per iterate in flight), 16 each
    int res_bank_now = 0;
    bool wire_spectral = false, rlines_valid = false, sl_ev_recorded = false;
    hipEvent_t sl_ev = nullptr;
    cd* Wend = nullptr;  // spectrum of an end value that is not the cached last node (forward transform of UEND on demand)
    // Same for the real-space pair: UEND and UEND2 alternate as the end-value buffer; after sdc_advance the start value
    // of the new step is still where the old step left its end value (u0_src) and reaches the U[0] slab only when
    // somebody needs it THERE (ensure_u0); read-only consumers take it where it lies (u0r).
    double* UEND2 = nullptr;
    const double* u0_src = nullptr;
    // One step further: while sweeps stay in Fourier space the end value is not even transformed back until somebody reads
    // it (uend_pending: UEND = inverse transform of SL, valid while uend_gen == spec_gen), and after sdc_advance the start
    // value of the new step may exist as its spectrum S0 only (u0_spec_only) - the predictor's residual then takes
    // max |f(u0)| from the norm-only inverse transform of symbol * S0 instead of a stencil pass over u0.
    bool uend_pending = false, u0_spec_only = false;
    // ... and that norm itself waits until the residual of the predictor's state is asked for (sdc_set_lazy_predictor_residual)
    bool lazy_f0norm = false, f0norm_pending = false;
    // node norms of the residual that a sweep brought to the host together with its counters (valid with res_valid)
    bool res_host_valid = false;
    unsigned long long res_host[8] = {};
    int u0_rc = SDC_OK;  // why u0r() could not deliver the start value
    bool spec_valid = false, spec0_valid = false, spec_spread = false, reuse = true, force_gather = false;
    // Iterates that are not stored: spec_virtual > 0 (while spec_valid) = the cached iterate is the result of that many
    // sweeps with the coefficients vcoef, started from "all nodes equal u0" - a function of S0 alone, recomputed by whoever
    // needs it (store_spectra).  sl_stored: the last node's spectrum has been written to SL all the same.
    int spec_virtual = 0, virt_max = 16;
    bool sl_stored = false;
    int prev_sweeps = 0;  // sweeps the step before this one took (its last sweep wrote SL itself: sdc_sweep, store_last)
    // Long runs of sweeps (mode pairs): from the g_from-th sweep of a step on the node multipliers are kept in a table
    // (Gm, M doubles per mode pair) and advanced by one sweep per launch instead of being recomputed from 1 every time;
    // g_sweeps = the sweeps its content stands for (with the coefficients vcoef), 0 = nothing in it.  g_from 0: no table.
    double* Gm = nullptr;
    int g_from = 8, g_sweeps = 0;
    SpecCoef vcoef;
    cd *tw = nullptr, *lamI = nullptr, *lamE = nullptr;
    bool sym_table_real = false;   // a symbol table given by the user (sdc_set_symbol, which = 0) whose imaginary parts are all zero ...
    double sym_absmax = 0.0;       // ... and the largest |entry| of it (the operator's symbol is the sum over the axes)
    bool sym_real[2] = {false, false};  // the stencil is symmetric: its Fourier symbol is real (imaginary parts stored as 0)
    unsigned long long* red = nullptr;  // reduction slots (device)
    unsigned long long* red_host = nullptr;
    // Residuals that the host does not wait for (sdc_residual_post / sdc_residual_wait): a one-workgroup launch at the end of
    // the residual's device work finishes the number (node norms -> residual, residual <= restol) and writes the record into
    // pinned host memory; its sequence number goes last, behind a system-scope fence.  The host compares that number with
    // its ticket - no stream synchronisation, no copy.  RES_RING records in flight; the host side of a ticket (what the
    // record must be read as) lives in ring_meta.
    ResRecord* ring = nullptr;          // host address of the pinned records
    ResRecord* ring_dev = nullptr;      // the same memory as the device sees it
    unsigned long long ring_seq = 0;    // last ticket handed out
    double restol = -1.0;               // tolerance the device-side flag is taken against (sdc_set_restol)
    bool tau_active = false, have_coeffs = false, have_stencil[2] = {false, false}, unlocked = false;
    int expl_kind = SDC_EXPL_NONE;
    bool res_spread = false;  // state = spread predictor of an autonomous f: residual_m = dt |sum_j Q[m][j]| max|f(u0)|
    bool fuse_residual = true;
    bool skip_residual = false;  // nobody asks for the residual after sweeps (skip_residual_computation)
    // deferred real-space state (sdc_set_deferred): the spectral-reuse sweep reads neither F[1..M] nor the node
    // copies of a spread predictor, so they are only written when somebody asks for them (sdc_materialize)
    bool deferred = true;
    bool spread_pending = false;  // U[1..M] = U[0], F[1..M] = F[0] not stored yet
    bool f_pending = false;       // F[1..M] = f(U[1..M]) not stored yet
    cd* SP = nullptr;             // transform of the forcing profile (heatNd_forced), valid while specP_valid
    bool specP_valid = false;
    // bounded grids with row-dependent stencils (sdc_set_banded_operator): fields are compact nb^ndim arrays at the start of
    // the slab fields; the operator is applied axis by axis from a row table, the solve is iterative
    int nb = 0, bw = 0;
    size_t Nb = 0;
    int* bcols = nullptr;
    double* bwts = nullptr;
    // dirichlet-zero in 2-D / 3-D by the odd extension (sdc_set_odd_interior): level fields are compact odd_n^ndim arrays at the
    // start of the slab fields (the interior is strided inside its extension of 2 (odd_n + 1) points per axis, which is the
    // engine's grid); sdc_eval_f / sdc_solve pack them into extension-sized scratch, run the periodic kernels, extract
    int odd_n = 0;
    bool odd_busy = false;        // inside such a call: the pointers ARE extensions
    double* odd_buf[3] = {nullptr, nullptr, nullptr};
    double* profile_compact = nullptr;   // the forcing profile in the layout of the compact fields (predictor fill)
    int solver_kind = 0;          // 0: exact solve in Fourier space, 1: conjugate gradients (solver_type='CG'), 2: GMRES
    double cg_rtol = 1e-12;
    int cg_maxiter = 10000;
    unsigned long long cg_iters = 0;
    std::map<double, int> cg_hist;  // iterations of the previous solve per factor (size of the first batch of the next one)
    double* cgw = nullptr;        // r, p, q, A p + partial sums
    double* gmw = nullptr;        // GMRES: Krylov basis (restart + 1 fields), w, A v, the update + partial sums
    size_t gmw_len = 0;
    unsigned long long gmres_iters = 0;
    bool early_uend = false;      // sweeps produce UEND right after the spectral update (before the residual passes)
    hipEvent_t uend_ev = nullptr;  // recorded when UEND is complete
    bool uend_ev_recorded = false;
    cd* W2 = nullptr;             // one-spectrum work buffer of that early transform (W is busy)
    bool keep_rfields = false;    // sweeps that only reduce the residual also store its fields in the U[1..M] slab
    bool rfields_valid = false;   // ... and they are there now (U[1..M] themselves live in S: u_pending)
    bool f0_pending = false;      // F[0] = f(U[0]) not evaluated yet (no sweep reads it)
    long long spec_gen = 0;       // bumped whenever the contents of S change
    long long uend_gen = -1;      // spec_gen at which UEND was produced as the inverse transform of S[M-1], or -1
    bool u_pending = false;       // U[1..M] live in the spectral cache S only (inverse transform not done yet)
    bool res_valid = false;   // node norms of the residual were produced by the fused stencil kernel
    double res_dt = 0.0;
    unsigned long long* res_dev = nullptr;  // device slots of those norms
    bool spectral_op = false;  // implicit operator given by its Fourier symbol only (no stencil): eval_f by FFT
    int react_kind = 0, react_nu = 2;
    double react_p0 = 0, react_p1 = 0;
    int kind = 0;  // 0: periodic finite differences, 1: van der Pol ensemble (N = 2 * ntraj, SoA)
    double vdp_mu = 0, vdp_tol = 1e-9;
    int vdp_maxiter = 100;
    int vdp_block_solver = 0;  // 0: closed-form 2x2 inverse applied on the vector ALUs, 1: applied on the matrix cores (MFMA)
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
    int prof_open = 0;                       // LaunchTimer brackets that are open right now (nested: a solver around its launches)
    std::map<std::string, ProfEntry> prof;
    std::vector<std::string> prof_names;
    std::string err;
};

static int uend_write_fence(sdc_ctx* c);  // comm.hpp: a send that still reads UEND goes first
static void comm_free(sdc_ctx* c);

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

// ---- node fields that are allocated on first touch -------------------------------------------------------------------------
static int lazy_block(sdc_ctx* c, void** dst, size_t bytes) {
    if (*dst || bytes == 0) return SDC_OK;
    HIPCHK(c, hipMalloc(dst, bytes));
    HIPCHK(c, hipMemsetAsync(*dst, 0, bytes, c->stream));
    c->bytes += bytes;
    return SDC_OK;
}
// f[0] is about to be read or written (a block of its own: a run that stays in Fourier space touches f(u0), never f[1..M])
static inline int need_f0(sdc_ctx* c) {
    return c->F0 ? SDC_OK : lazy_block(c, (void**)&c->F0, c->N * sizeof(double) * (size_t)c->ncomp);
}
// U[1..M] and F (all of it) are about to be read or written in real space
static inline int need_nodes(sdc_ctx* c) {
    if (c->Un && c->Fn && c->F0) return SDC_OK;
    int rc = lazy_block(c, (void**)&c->Un, c->N * sizeof(double) * (size_t)c->M);
    if (rc != SDC_OK) return rc;
    c->U = c->Un - c->N;   // (node index base: U + m N for m >= 1)
    rc = lazy_block(c, (void**)&c->Fn, c->N * sizeof(double) * (size_t)c->M * c->ncomp);
    if (rc != SDC_OK) return rc;
    c->F = c->Fn - (size_t)c->ncomp * c->N;   // (node index base: F + (m ncomp + comp) N for m >= 1; f[0] lives in F0)
    return need_f0(c);
}
// the node spectra S[0..M-2] (an iterate is stored in Fourier space)
static inline int need_node_spectra(sdc_ctx* c) {
    if (c->Sn || c->M < 2) return SDC_OK;
    int rc = lazy_block(c, (void**)&c->Sn, sizeof(cd) * c->Nc * (size_t)(c->M - 1));
    c->S = c->Sn;
    return rc;
}
#define NEED_NODES(c)                                                                                              \
    do {                                                                                                           \
        if (!(c)->Un && getenv("SDC_TRACE_LAZY")) fprintf(stderr, "[sdcmi] node fields allocated by %s:%d\n", __func__, __LINE__); \
        int rcn_ = need_nodes(c);                                                                                  \
        if (rcn_ != SDC_OK) return rcn_;                                                                           \
    } while (0)

// ---- spare spectra ---------------------------------------------------------------------------------------------------
static cd* spool_get(sdc_ctx* c) {
    if (!c->spool.empty()) {
        cd* b = c->spool.back();
        c->spool.pop_back();
        return b;
    }
    cd* b = nullptr;
    if (hipMalloc((void**)&b, sizeof(cd) * c->Nc) != hipSuccess) {
        (void)hipGetLastError();
        fail(c, SDC_ERR_NOMEM, "out of device memory for one more spectrum (%zu bytes)", sizeof(cd) * c->Nc);
        return nullptr;
    }
    c->spool_owned.push_back(b);
    c->bytes += sizeof(cd) * c->Nc;
    return b;
}
static void spool_put(sdc_ctx* c, const cd* b) {
    if (!b) return;
    for (cd* x : c->spool)
        if (x == b) return;
    c->spool.push_back(const_cast<cd*>(b));
}

// Per-kernel device time: a pair of events from a pool is recorded around every launch on the context's
// stream; nothing synchronises until the pool is full or the profile is read, so the timed region of bench.py
// is not perturbed.
static void prof_flush(sdc_ctx* c) {
    if (c->pool_used == 0) return;
    for (size_t i = 0; i < c->pool_used; ++i) {
        float ms = 0;
        (void)hipEventSynchronize(c->pool[2 * i + 1]);  // (pairs of the second stream end in their own order)
        if (hipEventElapsedTime(&ms, c->pool[2 * i], c->pool[2 * i + 1]) == hipSuccess) {
            ProfEntry& e = c->prof[c->pool_names[i]];
            e.ms += ms;
            e.calls += 1;
        } else {
            (void)hipGetLastError();  // (a pair that cannot be read is dropped; the error must not surface in a later check)
        }
    }
    c->pool_used = 0;
}

struct LaunchTimer {
    sdc_ctx* c;
    size_t slot = 0;
    bool on;
    hipStream_t s;
    LaunchTimer(sdc_ctx* c_, const char* n, hipStream_t other = nullptr, bool use_other = false)
        : c(c_), on(c_->profiling), s(use_other ? other : c_->stream) {
        if (!on) return;
        constexpr size_t kPairs = 2048;
        if (c->pool.empty()) {
            c->pool.resize(2 * kPairs);
            c->pool_names.resize(kPairs);
            for (auto& e : c->pool) (void)hipEventCreate(&e);
        }
        if (c->pool_used == c->pool_names.size()) {
            // full: read the pairs out - unless an enclosing timer is still open (a solver's launches inside its own
            // bracket), whose end is not recorded yet: then the pool grows instead
            if (c->prof_open == 0) {
                prof_flush(c);
            } else {
                const size_t old = c->pool_names.size();
                c->pool.resize(2 * (old + kPairs));
                c->pool_names.resize(old + kPairs);
                for (size_t i = 2 * old; i < c->pool.size(); ++i) (void)hipEventCreate(&c->pool[i]);
            }
        }
        slot = c->pool_used++;
        c->pool_names[slot] = n;
        ++c->prof_open;
        (void)hipEventRecord(c->pool[2 * slot], s);
    }
    ~LaunchTimer() {
        if (!on) return;
        (void)hipEventRecord(c->pool[2 * slot + 1], s);
        --c->prof_open;
    }
};

static inline int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// profile names that carry the number of fields of the launch, e.g. "fft_x_fwd[5]" - and, for a launch that covers one of
// G groups of kx planes, "[5/G]" (interned, static lifetime)
static const char* pname(const char* base, int nf, int groups = 1) {
    static std::map<std::string, std::string> table;
    static std::mutex guard;  // contexts of several host threads (in-process ranks of the tests) intern concurrently
    std::lock_guard<std::mutex> lock(guard);
    std::string key = std::string(base) + "[" + std::to_string(nf) + (groups > 1 ? "/" + std::to_string(groups) : "") + "]";
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

