/* sdcmi.h - C-ABI of libsdcmi.so, the MI355X (gfx950) SDC sweep engine.
 *
 * The reference (pySDC) has no FFI on this path: its plug-in boundary is two Python classes named in the
 * description dict (sweeper_class / problem_class, pySDC/core/level.py:87-88) plus the datatype the problem
 * names (pySDC/implementations/problem_classes/generic_ND_FD.py:81-82).  The entry points below are what the
 * host-side mirror of those classes (pysdc_amd/) binds through ctypes; each one cites the reference method
 * it replaces.  Plain pointers and sizes only; every function returns 0 on success and a negative
 * sdc_status otherwise, with text from sdc_last_error().
 *
 * State of one level lives on the device as slabs (f64, C order, spatial index fastest):
 *   U[(M+1)][N]  F[(M+1)][ncomp][N]  TAU[M][N]  UEND[N]        N = n^ndim, M = collocation nodes
 * replacing the reference's Python lists u[], f[], tau[] (pySDC/core/level.py:96-106).  u[0] and f[0] always exist; the
 * blocks U[1..M] and F[1..M] are allocated the first time something reads or writes a node value in real space (contexts
 * whose fields are smaller than 64 MB allocate everything at once): sdc_slot_ptr() of such a slot makes them real, pointers
 * handed out earlier stay valid.
 *
 * Environment switches the library itself reads: SDC_LAZY_MIN_BYTES (field size from which node blocks are allocated on first
 * touch; default 64 MB), SDC_TRACE_LAZY=1 (stderr: which entry point made them real), SDC_NO_TRAIL_DZ=1 (time slices: the
 * difference of two start values always through a launch of its own - A/B runs), SDC_PIPE_CHUNK (doubles per slot of the
 * host-memory pipe of two-rank runs), SDC_COMM_TIMEOUT (seconds a receive waits).
 */
#ifndef SDCMI_H
#define SDCMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sdc_ctx sdc_ctx;

enum sdc_status {
    SDC_OK = 0,
    SDC_ERR_PARAM = -1,    /* -> ParameterError / ProblemError */
    SDC_ERR_HIP = -2,      /* HIP runtime failure */
    SDC_ERR_STATE = -3,    /* level not unlocked, coefficients / operator not set (assert L.status.unlocked) */
    SDC_ERR_UNSUPPORTED = -4,
    SDC_ERR_NOMEM = -5,
    SDC_ERR_NEWTON = -6,   /* Newton failure: pySDC raises ProblemError (Van_der_Pol_implicit.py:179-186) */
    SDC_ERR_COMM = -7      /* RCCL failure / librccl missing: CommunicationError (core/errors.py) */
};

enum sdc_slot { SDC_SLOT_U = 0, SDC_SLOT_F = 1, SDC_SLOT_TAU = 2, SDC_SLOT_UEND = 3, SDC_SLOT_WORK = 4 };
enum sdc_residual_type { SDC_RES_FULL_ABS = 0, SDC_RES_LAST_ABS = 1, SDC_RES_FULL_REL = 2, SDC_RES_LAST_REL = 3 };
enum sdc_guess { SDC_GUESS_SPREAD = 0, SDC_GUESS_COPY = 1, SDC_GUESS_ZERO = 2, SDC_GUESS_CONST = 3 };
enum sdc_expl_kind {
    SDC_EXPL_NONE = 0,    /* fully implicit problem (generic_implicit) */
    SDC_EXPL_STENCIL = 1, /* f.expl = B u with a periodic FD stencil (advection part of config 3) */
    SDC_EXPL_FORCING = 2, /* f.expl = P(x) g(t), independent of u (heatNd_forced, HeatEquation_ND_FD.py:162-204) */
    SDC_EXPL_REACTION = 3, /* f.expl = pointwise nonlinear function of u (Allen-Cahn); swept node by node */
    SDC_EXPL_SYMBOL = 4    /* f.expl = B u with B given by its Fourier symbol (sdc_set_symbol which = 1): the explicit advection
                            * part -c d/dx of AdvectionDiffusionEquation_1D_FFT.py:82-87 beside the implicit diffusion symbol */
};

/* ---- context ----------------------------------------------------------------------------------------- */
/* ndim in 1..3, n points per dimension (square grid as generic_ND_FD.py:131-132 requires), num_nodes = M,
 * ncomp = 1 (dtype_f = mesh) or 2 (dtype_f = imex_mesh).  `stream` is a hipStream_t (0 = default stream).
 * Allocates the slabs.  Replaces Level.__init__ container set-up (pySDC/core/level.py:96-106). */
int sdc_ctx_create(sdc_ctx** out, int device, int ndim, int n, int num_nodes, int ncomp, void* stream);
int sdc_ctx_destroy(sdc_ctx* ctx);
const char* sdc_last_error(const sdc_ctx* ctx); /* ctx may be NULL: error of the last failed create */
size_t sdc_ctx_bytes(const sdc_ctx* ctx);       /* device bytes held by the context */

/* Collocation and Q-Delta matrices in pySDC layout: (M+1)x(M+1) row-major with zero first row/column
 * (pySDC/core/collocation.py:88-97, pySDC/core/sweeper.py:100-123).  QE may be NULL when ncomp == 1.
 * Called again by updateVariableCoeffs (pySDC/core/sweeper.py:262-276). */
int sdc_set_coeffs(sdc_ctx* ctx, const double* Qmat, const double* QI, const double* QE, const double* nodes,
                   const double* weights);

/* Periodic finite-difference operator A = sum over axes of the 1-D stencil sum_s w[s] u[i+off[s]]
 * (weights already carry coeff / dx^derivative): the matrix generic_ND_FD.py:140-149 assembles through
 * helpers/problem_helper.py:83-242.  which = 0: implicit part (solved), 1: explicit part (SDC_EXPL_STENCIL). */
int sdc_set_stencil(sdc_ctx* ctx, int which, int npts, const int* offsets, const double* weights);
/* Implicit (which = 0) / explicit (1) operator given directly by its 1-D Fourier symbol, n complex values
 * (re, im interleaved); the N-D symbol is the sum over the axes.  Used for the pseudo-spectral Laplacian
 * -(2 pi k / L)^2 of AllenCahn_2D_FFT.py:84-93 / generic_MPIFFT_Laplacian.py:113-124 and for the operators nu d2/dx2
 * (implicit) and -c d/dx (explicit, which = 1: sets SDC_EXPL_SYMBOL; needs the implicit symbol too) of
 * AdvectionDiffusionEquation_1D_FFT.py:63-72; eval_f then applies the operator(s) through the FFT pipeline instead of a
 * stencil.  The table covers ALL n modes of the complex transform the engine uses (k and n - k conjugate for a real
 * operator); like numpy's irfft the result is the real part, so the imaginary part a Nyquist entry produces is dropped. */
int sdc_set_symbol(sdc_ctx* ctx, int which, const double* table);
/* Pointwise explicit term (SDC_EXPL_REACTION): kind 1: p0 * u * (1 - u^nu)  (AllenCahn_2D_FFT.py:140-141, p0 =
 * 1/eps^2); kind 2: p0 * u (1-u)(1-2u) - p1 * u (1-u)  (AllenCahn_MPIFFT.py:83-85, p0 = -2/eps^2, p1 = 6 dw). */
int sdc_set_reaction(sdc_ctx* ctx, int kind, double p0, double p1, int nu);
/* Explicit part = profile[N] * g(t) with the profile given on the host (SDC_EXPL_FORCING). */
int sdc_set_expl_kind(sdc_ctx* ctx, int kind);
int sdc_set_forcing_profile(sdc_ctx* ctx, const double* host_profile);
/* g(t) at the left point and at the M node times of the current step: g[0..M] (host). */
int sdc_set_forcing_values(sdc_ctx* ctx, const double* g);

/* ---- slab access -------------------------------------------------------------------------------------- */
/* Device pointer of one field: U[m] (m = 0..M), F[m][comp], TAU[m] (m = 0..M-1), UEND.  Non-owning; valid
 * until sdc_ctx_destroy.  These back the L.u[m] / L.f[m] views (SURVEY 8b "Level/data surface"). */
void* sdc_slot_ptr(sdc_ctx* ctx, int slot, int m, int comp);
/* Address of the buffer that holds (or will hold) the end value RIGHT NOW, without the side effects of sdc_slot_ptr
 * (which assumes its caller may write and therefore forgets that UEND is the transform of the last node).  The end
 * value alternates between two buffers from step to step (sdc_advance), so the address is only good until the next
 * sdc_advance; writers must report through sdc_invalidate_spectra(ctx, 8). */
void* sdc_uend_address(sdc_ctx* ctx);
/* > 0 while the end value still is the last node of the cached iterate (sdc_advance could hand it over without copies);
 * a new value after every sweep; -1 otherwise. */
long long sdc_end_value_generation(sdc_ctx* ctx);
/* NOTE: while sweeps stay in Fourier space sdc_end_point may put the inverse transform of the last node off until the
 * end value is read (sdc_slot_ptr / sdc_download / sdc_stream_wait_uend / sdc_materialize(ctx, SDC_SLOT_UEND, 0) do it); a
 * reader that holds the plain address calls sdc_materialize(ctx, SDC_SLOT_UEND, 0) first.  sdc_advance then hands the
 * spectrum over without ever producing the field, and the predictor's residual takes max |f(u0)| from the norm-only
 * inverse transform of symbol x spectrum. */
int sdc_upload(sdc_ctx* ctx, int slot, int m, int comp, const double* host);
int sdc_download(sdc_ctx* ctx, int slot, int m, int comp, double* host);
int sdc_set_tau_active(sdc_ctx* ctx, int active); /* L.tau[m] is None  <->  0 */
/* Spectral reuse: for linear right-hand sides the engine keeps the Fourier transforms of U[0] and
 * of U[1..M] between sweeps and gathers on them instead of re-transforming M fields every sweep.  Anything
 * that writes a U field behind the engine's back (datatype operations on slab views, RCCL receives) must say
 * so: which = 1: U[0] changed, 2: some U[m >= 1] changed, 4: some F[m >= 1] was overwritten (the next sweep then
 * gathers on the F slab like the reference does), 8: UEND was overwritten, 16: some TAU[m] changed (every bit but 8
 * drops the cached residual norms); bits combine.  sdc_upload / sdc_predict do it themselves. */
int sdc_invalidate_spectra(sdc_ctx* ctx, int which);
int sdc_set_spectral_reuse(sdc_ctx* ctx, int on);
/* The 3-D sweep evaluates f at all nodes and the node norms of the collocation residual in ONE kernel; a
 * following sdc_residual with the same dt then returns those norms without another pass (default on). */
int sdc_set_fused_residual(sdc_ctx* ctx, int on);
/* The sweeper parameter skip_residual_computation (core/sweeper.py:176-179: compute_residual returns at once in the
 * listed stages).  When it covers every stage that follows a sweep, nobody reads the residual of the new iterate:
 * with on = 1 a sweep that gathers on the cached transforms then ONLY updates those transforms (one pointwise pass, no
 * inverse transform at all).  A residual asked for later is still answered, from the cache (sdc_residual). */
int sdc_set_skip_residual(sdc_ctx* ctx, int on);
/* Iterates that are not stored.  After a 'spread' predictor (core/sweeper.py:129-143: every node starts from u[0]) the
 * iterate of a linear problem is a function of the transform of u[0] alone, and so is every later one while u[0] and the
 * coefficients stay what they are (generic_implicit.py:51-103 applied to the same data again and again).  A sweep that
 * stays in Fourier space then reads the transform of u[0] only, repeats the earlier sweeps of the step in registers and
 * stores nothing but what the residual norm needs; whoever needs the iterate itself (node values, the end value, a new
 * u[0], other coefficients) has it written out first.  max_sweeps: sweeps per step that may be repeated that way before
 * the iterate is stored after all (arithmetic grows with each; default 16), 0: every sweep stores its iterate. */
int sdc_set_virtual_sweeps(sdc_ctx* ctx, int max_sweeps);
/* Long runs of such sweeps (a step iterated to a residual tolerance: restol of the level, core/level.py:16-21, tens of
 * sweeps).  For a real symbol the iterate of a mode is the start value times M real node multipliers that the modes kz and
 * N - kz of a line share; from the from_sweep-th sweep of a step on they are kept in a table (M doubles per mode pair, about
 * a quarter of one node's spectrum per node), read, advanced by one sweep and written back by every launch instead of being
 * recomputed from 1 - constant cost per sweep where replaying grows with every sweep, and no switch to stored iterates
 * at max_sweeps.  Same arithmetic in the same order as the replay.  Used when the replayed sweeps get that far
 * (max_sweeps > from_sweep) and the table can be allocated; from_sweep 0: never (default 8: where one more replayed sweep costs what the table traffic does at 1024^3). */
int sdc_set_multiplier_table(sdc_ctx* ctx, int from_sweep);
/* The residual of the state a 'spread' predictor leaves (core/sweeper.py:164-215 with every node equal to u[0]) is
 * dt |sum_j Q[m][j]| max|f(u[0])|.  When u[0] exists as its transform only (the step before handed it over in Fourier space),
 * max|f(u[0])| costs a norm-only inverse transform of one field.  on = 1: sdc_predict puts that transform off until
 * sdc_residual is called for this state; sdc_residual_deferred returns 1 while a call of sdc_residual would have to do it, so
 * that a caller whose convergence test cannot depend on the value (controller_nonMPI.py:493 with restol < 0,
 * check_convergence.py:60-92) may postpone the call until somebody reads L.status.residual.  Default 0: computed by sdc_predict. */
int sdc_set_lazy_predictor_residual(sdc_ctx* ctx, int on);
int sdc_residual_deferred(sdc_ctx* ctx);
/* Residuals the host does not wait for (compute_residual + CheckConvergence on the device).
 * sdc_residual_post queues everything sdc_residual does on the context's stream and ends with a one-workgroup launch that
 * finishes the number on the device - node norms -> L.status.residual by residual_type (core/sweeper.py:200-215) and the
 * test `residual <= restol` of check_convergence.py:72-75 against the tolerance given by sdc_set_restol (< 0: never
 * converged, the reference's default) - and writes the record into pinned host memory, its ticket last.  It returns the
 * ticket without synchronising.  sdc_residual_wait(ticket, block, ...) looks at that memory: ready = 1 and the values once
 * the record is there; with block = 1 it waits for it (polling the host memory, no stream synchronisation on the fast
 * path).  A caller whose control flow cannot depend on the value (restol < 0, a fixed number of sweeps) never waits: it
 * collects the values when somebody reads them (pysdc_amd.engine.ResidualFuture).  The last 256 tickets can be asked for.
 * sdc_residual = post + wait(block). */
int sdc_set_restol(sdc_ctx* ctx, double restol);
int sdc_residual_post(sdc_ctx* ctx, double dt, int residual_type, unsigned long long* ticket);
/* ... and, when the residual is reduced from F in real space anyway (node-by-node levels), the quadrature sums themselves -
 * dt sum_j Q[m][j] f_j, what sdc_integrate returns - into integrals[0..M-1] in the SAME pass over F: a fine level's
 * compute_residual is followed by the restriction's integrate() (core/base_transfer.py:120-127).  *wrote = 1 if they were
 * written (0: the residual came from a cache or from Fourier space; call sdc_integrate). */
int sdc_residual_post_integrals(sdc_ctx* ctx, double dt, int residual_type, double* const* integrals, int* wrote,
                                unsigned long long* ticket);
int sdc_residual_wait(sdc_ctx* ctx, unsigned long long ticket, int block, double* node_norms, double* residual, int* converged,
                      int* ready);
/* Where the residual of the current state would come from if it were asked for now (with this dt): 0 = node norms a sweep
 * already reduced, 1 = the closed form of a spread predictor's state, 2 = reduced from the spectrum of the cached iterate,
 * 3 = one pass over u[0], U[1..M], F[1..M] in real space - the only route on which sdc_residual_post_integrals writes the
 * quadrature sums (a caller allocates their M fields only then).  Negative: error code. */
int sdc_residual_route(sdc_ctx* ctx, double dt);
/* the last ticket handed out by sdc_residual_post / sdc_residual (0: none yet) */
unsigned long long sdc_residual_last_ticket(sdc_ctx* ctx);
/* Deferred node fields (default on).  The spectral-reuse sweep reads neither F[1..M] nor the M copies a 'spread'
 * predictor makes (core/sweeper.py:140-146): the engine therefore leaves them unwritten until somebody needs
 * them.  sdc_slot_ptr / sdc_upload / sdc_download / sdc_integrate / sdc_end_point / sdc_residual and the
 * non-reuse sweeps bring them up to date themselves; a caller that KEEPS a pointer from sdc_slot_ptr across a
 * sweep or predict calls sdc_materialize(ctx, slot, m) (slot = SDC_SLOT_U, SDC_SLOT_F, or -1 for both; m = the
 * node it is about to touch, -1 for all) before it dereferences it again.  Values are the ones the reference
 * stores in L.u[m] / L.f[m].  With the mode on, sweeps that gather on the cached transforms do not leave Fourier
 * space at all: they return the node norms of the collocation residual (reduced from the inverse transform of
 * its spectrum) and keep U[1..M] in the cache; sdc_end_point transforms only the last node. */
int sdc_set_deferred(sdc_ctx* ctx, int on);
int sdc_materialize(sdc_ctx* ctx, int slot, int m);
/* f[0] = f(u[0]) after u[0] was replaced by a receive (controller_MPI.py:233, controller_nonMPI.py:284).  No sweep,
 * residual or end point reads f[0] (all node loops start at 1); the evaluation happens when F[0] is asked for
 * (sdc_slot_ptr / sdc_materialize / views), or at once when the deferred mode is off. */
int sdc_defer_f0(sdc_ctx* ctx);
/* Time-parallel runs replace u[0] between sweeps (the receive of controller_MPI.py:218-233) and then ask for the
 * residual against the new value (:592).  With sdc_set_keep_residual_fields(ctx, 1) a sweep that stays in Fourier
 * space also stores the residual fields r_m it reduces (in the U[1..M] slab, free while the iterate lives in the
 * cache); sdc_replace_u0(ctx, src) then copies the new value in and updates the node norms in the same pass
 * (r_m changes by new - old for every m).  Without kept fields it is a plain copy. src: device field of N doubles. */
int sdc_set_keep_residual_fields(sdc_ctx* ctx, int on);
/* Sending the end value while the residual passes still run: with sdc_set_early_end_point(ctx, 1) a sweep that
 * stays in Fourier space transforms its last node into UEND right after the spectral update (a following
 * sdc_end_point without collocation update is then free); sdc_stream_wait_uend(ctx, stream) makes another HIP
 * stream (the one the message is posted on) wait until UEND is complete - and for nothing queued after it. */
int sdc_set_early_end_point(sdc_ctx* ctx, int on);
/* Time-parallel levels that sweep in Fourier space with spectra on the wire (sdc_comm_set_format): how the engine deals
 * with a u[0] that is replaced between sweeps (controller_MPI.py:218-233, :574-583).
 *  trail_sources (default 0 = off; at most 5): iterates are not stored; a sweep recomputes its iterate from the start values the slice
 *    has had since its spread predictor (the sweep is linear: u^k = sum_i C_i(lambda) u0_i per Fourier mode with real node
 *    multipliers), reads those <= trail_sources spectra and writes only the residual lines and the last node's spectrum.
 *    One more start value than that, or sdc_set_virtual_sweeps' limit, and the iterate is stored after all.
 *  defer_last_pass (default 1): the last inverse pass of a sweep's residual waits until the new start value has arrived;
 *    ONE pass over the residual lines then reduces the node norms before AND after the receive (they differ by one field, the
 *    difference of the two start values).  Residuals posted meanwhile (sdc_residual_post) are published by that pass;
 *    a blocking wait or anything else that needs the work spectra runs it at once.  When the NEXT sweep arrives while it
 *    still waits (a fixed number of sweeps: nobody asked), that sweep's z / y launches go first, into a second set of work
 *    spectra, and the put-off passes follow them: the last node's spectrum is on the wire one launch after the receive and the
 *    put-off passes run while it travels (2: put off, but never behind the next sweep; 0: every pass at once).
 *  split_send (default 0): a sweep first writes the last node's spectrum by a small launch of its own, so that the message
 *    leaves before the passes put off for the previous iterate and the sweep's own residual passes run. */
int sdc_set_timeslice_options(sdc_ctx* ctx, int trail_sources, int defer_last_pass, int split_send);
int sdc_stream_wait_uend(sdc_ctx* ctx, void* other_stream);
int sdc_replace_u0(sdc_ctx* ctx, const double* src);
/* Start and end values as SPECTRA.  Between levels that sweep in Fourier space (periodic finite differences, exact solve,
 * deferred node fields: sdc_spectral_handover_ok) the forward hand-over of a time-parallel run (controller_MPI.py:218-305)
 * needs neither the inverse transform of the sender's last node nor the forward transform of the receiver's new u[0]: the
 * last node's half spectrum (Nc = (n/2+1) n^(ndim-1) complex values) IS the message.
 *   sdc_end_spectrum(ctx, stream)    device address of the spectrum of the end value (after sdc_end_point); `stream`, if not
 *                                    null, is made to wait until it is complete - right after the first launch of the sweep
 *                                    with sdc_set_early_end_point - and for nothing queued on the engine's stream later
 *   sdc_spectrum_inbox(ctx)          device address a received spectrum is to be written to
 *   sdc_replace_u0_spectrum(ctx)     u[0] <- the field whose spectrum lies in the inbox (the buffers trade places, nothing is
 *                                    copied; U[0] itself is produced when somebody reads it).  If the last sweep only reduced
 *                                    the residual norms, the norms against the new u[0] come from ONE more field through the
 *                                    inverse passes: r_m changes by new - old for every node m (core/sweeper.py:186-199). */
int sdc_spectral_handover_ok(sdc_ctx* ctx);
int sdc_set_wire_spectral(sdc_ctx* ctx, int on); /* the engine side of sdc_comm_set_format, for callers that move the spectra themselves */
void* sdc_end_spectrum(sdc_ctx* ctx, void* stream);
void* sdc_spectrum_inbox(sdc_ctx* ctx);
int sdc_replace_u0_spectrum(sdc_ctx* ctx);
/* a new block starts from the field whose spectrum lies in the inbox (nothing of the finished step is kept): what sdc_advance
 * is for the rank that owns the end value, for the ranks that received it (sdc_comm_bcast_end_spectrum) */
int sdc_start_from_spectrum(sdc_ctx* ctx);

/* ---- time-rank communication (RCCL over xGMI; host mailboxes as the rehearsal wire) --------------------------
 * The forward transfer uend -> u[0] of the next time rank (controller_MPI.py:218-305 send_full / recv_full; mesh.py:85-125
 * isend / irecv / bcast) and the end-of-block broadcast (controller_MPI.py:125-130), modelled on the reference's NCCL
 * wrapper helpers/NCCL_communicator.py:12-20 (unique id made by rank 0 and distributed by the host-side communicator,
 * one NCCL communicator per process) and :128-135 (Bcast).  One process per GPU; librccl is bound at run time on the
 * first call.  Messages travel on a stream of their own, ordered against the engine's stream by events only (no host
 * block): a send starts once UEND is complete (sdc_end_point, or the early end value of sdc_set_early_end_point) and
 * UEND is not rewritten before it has left; a received value lands in an inbox and reaches the level through
 * sdc_replace_u0 on the engine's stream.  MPI tags (level*100 + iter) are replaced by the strict order of the calls.
 * A unique id that starts with "shm:" selects the second wire instead: single-slot mailboxes in POSIX shared memory
 * named after the rest of the id (same calls, same stream ordering, data staged through host memory, host-blocking) -
 * for ranks that RCCL cannot connect: several ranks on ONE GPU, ranks that are threads of one process.
 *   sdc_comm_unique_id(out)          128 bytes, called on ONE rank; ship them to the others on the host side
 *   sdc_comm_init(ctx, uid, P, r)    ncclCommInitRank on the context's device (collective over the P ranks)
 *   sdc_comm_attach(ctx, owner)      a coarser level of the same time rank shares the owner's communicator and stream
 *   sdc_comm_exchange(ctx, to, from) send UEND to rank `to` and / or receive the new u[0] from rank `from` as ONE
 *                                    group (both directions progress together); a peer < 0 skips that direction
 *   sdc_send_uend / sdc_recv_u0      the two halves on their own
 *   sdc_comm_handover_post(ctx, P)   lock-step runs: the hand-over uend(r) -> u[0](r+1) of ALL P active ranks, posted behind
 *                                    the completion of UEND only (it travels while the residual passes run); with more than
 *                                    two ranks every message is cut into P pieces that travel over two hops through all
 *                                    ranks (each xGMI link carries 1/P per phase) unless sdc_comm_set_relay(ctx, 0)
 *   sdc_comm_handover_complete(ctx)  the engine's stream waits for that hand-over; the received value becomes u[0]
 *   sdc_bcast(ctx, slot, m, root)    one slab field of rank `root` to all, in place (more than two ranks and relay on:
 *                                    scatter + all-gather over the mesh, same bits)
 *   sdc_comm_bcast_buffer(ctx, p, n, root)  the same for any device buffer of n doubles
 *   sdc_comm_bcast_end_spectrum(ctx, root)  the end value of a block as its half spectrum: from the root's cache into the other
 *                                    ranks' spectrum inboxes (sdc_advance on the root, sdc_start_from_spectrum elsewhere)
 *   sdc_comm_set_chunk(ctx, n)       cut every message into pieces of n doubles inside its group (0 = one piece)
 *   sdc_comm_set_format(ctx, 1)      lock-step hand-overs carry half spectra instead of fields (sdc_end_spectrum ->
 *                                    sdc_spectrum_inbox -> sdc_replace_u0_spectrum); same choice on every rank
 *   sdc_comm_set_relay(ctx, on)      two-hop hand-over / mesh broadcast for more than two ranks (default on)
 *   sdc_comm_info(ctx, ...)          rank, size, counts of two-hop hand-overs and mesh broadcasts, wire kind ("rccl" / "shm")
 *   sdc_comm_sync(ctx)               host waits for the messages posted so far
 *   sdc_comm_selftest(job, P, r, n, what, arg, rounds)   the exchange patterns (0 direct, 1 two-hop among the first `arg`
 *                                    ranks; 2 mesh broadcast from rank `arg`) on plain HOST buffers over a mailbox wire named
 *                                    `job`, called by P threads or processes; checks every received value.  No GPU needed. */
int sdc_comm_unique_id(char* out128);
int sdc_comm_init(sdc_ctx* ctx, const char* uid128, int nranks, int rank);
int sdc_comm_attach(sdc_ctx* ctx, sdc_ctx* owner);
int sdc_comm_destroy(sdc_ctx* ctx);
int sdc_comm_exchange(sdc_ctx* ctx, int send_peer, int recv_peer);
int sdc_send_uend(sdc_ctx* ctx, int peer);
int sdc_recv_u0(sdc_ctx* ctx, int peer);
int sdc_comm_handover_post(sdc_ctx* ctx, int nactive);
int sdc_comm_handover_complete(sdc_ctx* ctx);
int sdc_bcast(sdc_ctx* ctx, int slot, int m, int root);
int sdc_comm_bcast_buffer(sdc_ctx* ctx, double* buf, size_t n, int root);
int sdc_comm_bcast_end_spectrum(sdc_ctx* ctx, int root);
int sdc_comm_set_chunk(sdc_ctx* ctx, size_t doubles_per_piece);
int sdc_comm_set_relay(sdc_ctx* ctx, int on);
/* Two ranks: the direct message r -> r + 1 uses ONE xGMI link (8.6 GB at 1024^3: ~134 ms) while the host link idles.  With
 * share > 0 that fraction of every lock-step hand-over (its tail) travels through pinned host memory instead - a ring of slots
 * in POSIX shared memory registered with the HIP runtime, filled by the sender and drained by the receiver on helper threads
 * and streams of their own, beside the direct transfer of the head.  0 (default): off.  Same choice on both ranks. */
int sdc_comm_set_host_share(sdc_ctx* ctx, double share);
int sdc_comm_set_format(sdc_ctx* ctx, int spectra);
int sdc_comm_info(sdc_ctx* ctx, int* rank, int* size, unsigned long long* two_hop_calls, unsigned long long* mesh_bcast_calls,
                  char* kind16);
int sdc_comm_selftest(const char* job, int nranks, int rank, size_t n, int what, int arg, int rounds);
int sdc_comm_sync(sdc_ctx* ctx);
/* L.status.unlocked: set by sdc_predict; a coarse level is unlocked by the restriction instead
 * (pySDC/core/base_transfer.py:166) - the host mirrors that here. */
int sdc_set_unlocked(sdc_ctx* ctx, int unlocked); /* default on; 0 = transform the gathered fields every sweep */
/* Synthetic input generated on the device (no multi-GB host arrays): dst[i] = prod_d sin(pi*freq[d]*x_d) on
 * the grid of generic_ND_FD.py:171-180 (the u_exact(0) of HeatEquation_ND_FD.py:103-132) + amp * g(i), g a
 * standard normal from splitmix64(seed, i) + Box-Muller; host equivalent: pysdc_amd.synth.init_field. */
int sdc_init_field(sdc_ctx* ctx, double* dst, const int* freq, double amp, unsigned long long seed);

/* ---- the sweep path ----------------------------------------------------------------------------------- */
/* Sweeper.predict (pySDC/core/sweeper.py:125-162): F[0] = f(U[0], t); nodes filled per `guess`;
 * fill_u / fill_f are the constants for SDC_GUESS_CONST ('random' draws them on the host). */
int sdc_predict(sdc_ctx* ctx, double t, double dt, int guess, double fill_u, double fill_f);
/* One sweep over all nodes: generic_implicit.update_nodes (generic_implicit.py:51-103) when ncomp == 1,
 * imex_1st_order.update_nodes (imex_1st_order.py:57-108) when ncomp == 2, including the M implicit solves
 * (generic_ND_FD.py:208-264, solver_type 'direct') and the M right-hand side evaluations (:188-206). */
int sdc_sweep(sdc_ctx* ctx, double t, double dt);
/* Sweeper.compute_residual (pySDC/core/sweeper.py:164-215).  node_norms[M] receives abs(residual[m]),
 * *residual the value stored in L.status.residual for `type`.  Synchronises the stream. */
int sdc_residual(sdc_ctx* ctx, double dt, int type, double* node_norms, double* residual);
/* compute_end_point (generic_implicit.py:105-131 / imex_1st_order.py:110-137) into UEND. */
int sdc_end_point(sdc_ctx* ctx, double dt, int do_coll_update);
/* integrate() (generic_implicit.py:29-49 / imex_1st_order.py:37-55): dst[m] = dt sum_j Q[m+1][j] f[j],
 * dst = M device pointers (called by BaseTransfer.restrict, pySDC/core/base_transfer.py:134,137). */
/* Next time step on the same level: u[0] <- uend (controller_nonMPI.py:148 hands the end value of a block to
 * its first step; core/step.py:271).  When UEND is the inverse transform of the cached spectrum of the last node
 * and nothing changed since, that spectrum becomes the transform of the new u[0] (no forward transform). */
int sdc_advance(sdc_ctx* ctx);
int sdc_integrate(sdc_ctx* ctx, double dt, double* const* dst);

/* ---- problem-level operations on raw device fields (the non-fused plug-in path) ------------------------ */
/* eval_f (generic_ND_FD.py:188-206; HeatEquation_ND_FD.py:162-204 for the IMEX variant).  g_t is the value
 * of the forcing's time factor g(t) (only read for SDC_EXPL_FORCING); f_expl may be NULL. */
int sdc_eval_f(sdc_ctx* ctx, const double* u, double g_t, double* f_impl, double* f_expl);
/* eval_f of nf <= num_nodes fields in ONE pass of the launches sdc_eval_f takes per field (spectral operators: one transform
 * round trip for all of them, the pointwise reaction term of each riding on the pass that reads it; stencil operators: one
 * marching launch).  This is the loop of core/base_transfer.py:207-213 (`F.f[m] = P.eval_f(F.u[m], ...)` for every node after
 * a prolongation) and of the predictors' node evaluations.  g_t: one forcing factor per field, or null; f_expl: null or one
 * pointer per field.  Same values as nf calls of sdc_eval_f. */
int sdc_eval_f_batch(sdc_ctx* ctx, int nf, const double* const* u, const double* g_t, double* const* f_impl,
                     double* const* f_expl);
/* solve_system(rhs, factor, u0, t) (generic_ND_FD.py:208-264, 'direct'): (I - factor*A) out = rhs; `guess` is
 * the reference's u0 argument (unused by the direct solver, may be NULL; the Newton solver requires it). */
int sdc_solve(sdc_ctx* ctx, const double* rhs, double factor, const double* guess, double* out);

/* ---- datatype operations (mesh arithmetic, datatype_classes/mesh.py:12-125) ----------------------------
 * ctx may be NULL: the operation then runs on the null stream of the current device. */
int sdc_vec_copy(sdc_ctx* ctx, size_t n, const double* x, double* y);
int sdc_vec_fill(sdc_ctx* ctx, size_t n, double a, double* y);
int sdc_vec_axpby(sdc_ctx* ctx, size_t n, double a, const double* x, double b, const double* y, double* z);
int sdc_vec_amax(sdc_ctx* ctx, size_t n, const double* x, double* out);
/* A box of a C-ordered field of `shape` (ndim <= 4) - per axis start, step (may be negative), count: what basic indexing of
 * the reference's ndarray datatype selects (datatype_classes/mesh.py:12-60: `u[i]`, `u[..., j]`, `u[1:-1]`).  direction 0:
 * compact <- box (a gather into count[0] * ... * count[ndim-1] doubles), 1: box <- compact (scatter), 2: box <- value.
 * Runs on the context's stream (NULL: the default context of the calling thread, like the other sdc_vec_* calls). */
int sdc_vec_box(sdc_ctx* ctx, int ndim, const long long* shape, const long long* start, const long long* step,
                const long long* count, double* field, double* compact, int direction, double value); /* abs(): max-norm, synchronises */

/* ---- van der Pol ensemble (BASELINE config 4; Van_der_Pol_implicit.py:106-201) -------------------------
 * ntraj independent trajectories are ONE level with N = 2 * ntraj unknowns, SoA fields [x1[ntraj], x2[ntraj]]
 * (create the context with ndim = 1, n = 2 * ntraj, ncomp = 1).  After this call sdc_eval_f is the van der Pol
 * right-hand side, sdc_solve the Newton solve with the closed-form 2x2 inverse (guess = previous iterate) and
 * sdc_sweep one generic_implicit sweep for every trajectory in one launch; predict / residual / end point /
 * integrate are the generic kernels.  A failed Newton solve makes the call return SDC_ERR_NEWTON
 * (ProblemError in the reference, Van_der_Pol_implicit.py:179-186). */
int sdc_set_problem_vdp(sdc_ctx* ctx, double mu, double newton_tol, int newton_maxiter);
/* How the Newton step applies the inverse of the dense local Jacobian block (I - dt J), 2x2 per trajectory
 * (Van_der_Pol_implicit.py:190-201 solve_jacobian): 0 = closed form on the vector ALUs (default), 1 = the same inverse
 * applied on the matrix cores, two trajectories per 4x4 block of v_mfma_f64_4x4x4_4b_f64 (BASELINE.json north_star).
 * Same Newton iteration, same stopping rule; the products are fused multiply-adds on the matrix path. */
int sdc_set_vdp_block_solver(sdc_ctx* ctx, int kind);
/* out[0] = Newton iterations, out[1] = right-hand side evaluations, out[2] = failed solves (pending), summed
 * over trajectories since context creation (work_counters of Van_der_Pol_implicit.py:71-73). */
/* out = (dg/du)^{-1} rhs at u for g(u) = u - dt f(u), every trajectory (Van_der_Pol_implicit.py:190-201). */
int sdc_solve_jacobian(sdc_ctx* ctx, const double* rhs, double dt, const double* u, double* out);
int sdc_work_counters(sdc_ctx* ctx, unsigned long long* out); /* out[5]; out[3] = CG, out[4] = GMRES iterations (sdc_set_solver) */
/* solver_type of GenericNDimFinDiff (generic_ND_FD.py:238-262).  kind 0 ('direct'): the exact solve in Fourier space
 * (default; satisfies any lintol).  kind 1 ('CG'): scipy.sparse.linalg.cg as the reference calls it - x0 = the previous
 * node value, rtol = lintol, atol = 0, maxiter = liniter, every iteration counted (work_counters['CG']); sweeps then run
 * node by node on the device like the reference's loop.  Dot products are reduced in a fixed order, so the counts are
 * reproducible.  kind 2 ('GMRES'): scipy.sparse.linalg.gmres as the reference calls it (generic_ND_FD.py:241-250) -
 * restart 20, x0 = the previous node value, rtol = lintol, atol = 0, callback_type 'legacy': every INNER iteration is
 * counted (work_counters['GMRES']) and maxiter = liniter counts inner iterations; Arnoldi with modified Gram-Schmidt on
 * the device, the Hessenberg least-squares problem (Givens rotations) on the host. */
int sdc_set_solver(sdc_ctx* ctx, int kind, double rtol, int maxiter);
/* Bounded grids whose stencil depends on the row: dirichlet-zero with stencils of order >= 4, where the reference shifts
 * one-sided stencils into the rows next to the boundary (helpers/problem_helper.py:143-224) - a non-symmetric banded matrix
 * per axis, Kronecker-summed over the axes (:226-237).  cols[n_interior][width] (-1 = unused) and weights[n_interior][width]
 * give the 1-D rows.  From then on sdc_eval_f / sdc_solve of this context work on COMPACT fields of n_interior^ndim values
 * (the start of a slab field is such a field): eval_f applies the operator, sdc_solve runs the configured Krylov solver
 * (sdc_set_solver: CG / GMRES with the user's tolerance, counted) or, for 'direct', GMRES to round-off in place of the
 * reference's sparse LU.  sdc_sweep on such levels runs the reference's node loop inside the engine (gather for all nodes,
 * then per node right-hand side, Krylov solve from the old node value, operator application).  Buffers handed to
 * sdc_eval_f / sdc_solve on such a context need n_interior^ndim doubles (f_expl of a forced level included). */
int sdc_set_banded_operator(sdc_ctx* ctx, int n_interior, int width, const int* cols, const double* weights);

/* ---- space transfer between two grids --------------------------------------------------------------------
 * mesh_to_mesh (transfer_classes/TransferMesh.py:9-218): Pspace / Rspace are Kronecker products of ONE 1-D
 * sparse matrix (helpers/transfer_helper.py:140-242).  The host builds that 1-D matrix exactly as the reference
 * does and hands its rows over as fixed-width tables on the device: out[i] = sum_j w[j][i] * in[idx[j][i]] per
 * axis, zero-padded to `width` (entry-major: entry j of row i at [j * n_out + i]; padded entries carry weight 0 and
 * any valid index).  Applied as a tensor product over ndim axes in one launch.  Context-free (two
 * levels are involved); errors are reported through sdc_last_error(NULL). */
int sdc_transfer_apply(void* stream, int ndim, int n_out, int n_in, int width, const int* idx, const double* w,
                       const double* in, double* out);
/* The same for nfields fields that lie one behind the other (in: nfields * n_in^ndim doubles, out likewise): the node
 * values U[1..M] of a slab, a set of quadrature integrals - one launch per axis for all of them. */
int sdc_transfer_apply_batch(void* stream, int nfields, int ndim, int n_out, int n_in, int width, const int* idx,
                             const double* w, const double* in, double* out);
/* ... with accumulate = 1 the last pass ADDS the result to `out` instead of storing it: the coarse-grid correction
 * u_F[m] += P (u_G[m] - uold_G[m]) of core/base_transfer.py:196-205 without a field for the prolonged difference and
 * without the pass that adds it (same bits: x + P d either way). */
int sdc_transfer_apply_batch_acc(void* stream, int nfields, int ndim, int n_out, int n_in, int width, const int* idx,
                                 const double* w, const double* in, double* out, int accumulate);
/* ... between NESTED periodic grids that differ by a factor of two per axis (n_out = 2 n_in or n_in = 2 n_out), where the host
 * has checked that every table entry of output row i lies in the window of that row (coarsening: columns 2i-1, 2i, 2i+1
 * modulo n_in, width 3 - rorder 2; refinement: columns i/2 - width/2 + 1 .. i/2 + width/2 modulo n_in, even rows a single
 * entry - TransferMesh.py:49-146 with helpers/transfer_helper.py:153-186, periodic / equidist_nested): the three axes in
 * ONE launch that streams every input once (coarsening) or stages a coarse tile with its halo in LDS (refinement), the
 * sums nested and accumulated in the order of the separable passes.  The differences FAS forms around a transfer ride
 * along (core/base_transfer.py:120-147, :196-205): in_minus != NULL transfers in - in_minus (the coarse-grid correction
 * u_G - uold_G), out_minus != NULL stores result - out_minus (tau = R(Q_F f_F) - Q_G f_G).  Anything the fused launches
 * do not cover (1-D / 2-D, other widths, grids that are not a multiple of the tile) takes sdc_transfer_apply_batch_acc's
 * passes and forms the differences with launches of their own: same results. */
int sdc_transfer_apply_nested(void* stream, int nfields, int ndim, int n_out, int n_in, int width, const int* idx,
                              const double* w, const double* in, const double* in_minus, double* out,
                              const double* out_minus, int accumulate);

/* Fourier prolongation between two periodic grids held by two contexts (the levels' engines):
 * mesh_to_mesh_fft (1-D, transfer_classes/TransferMesh_FFT.py:36-57: rfft, low modes + Nyquist copied, irfft,
 * factor = ratio) and mesh_to_mesh_fft2d (2-D, TransferMesh_FFT2D.py:58-77: fft2, four corner blocks, real part of
 * ifft2, factor = 2 * ratio - the reference's constant).  dst = factor * IFFT_fine(pad(FFT_coarse(src))) with the
 * reference's index conventions.  src / dst are plain device fields of the two grids; the restriction of both
 * classes is the injection F[::ratio] (sdc_transfer_apply with width 1). */
int sdc_fft_prolong(sdc_ctx* coarse, sdc_ctx* fine, const double* src, double* dst, double factor);

/* Dirichlet-zero boundaries in 1-D (generic_ND_FD.py:99-133 'dirichlet-zero', order 2): the n interior values
 * live inside their odd extension [0, u_0..u_{n-1}, 0, -u_{n-1}..-u_0] of length 2(n+1) = 2^p, on which the
 * Dirichlet 3-point operator IS the periodic one, so every kernel of the periodic engine applies unchanged
 * (the Fourier solve becomes the sine-transform solve).  This rebuilds the end points and the mirrored half of
 * one field after its interior was written.  ctx may be NULL. */
int sdc_odd_mirror(sdc_ctx* ctx, double* field, int n_interior);
/* dirichlet-zero in 2-D / 3-D with the centred order-2 operator (generic_ND_FD.py:130, helpers/problem_helper.py:143-160 with
 * zero boundary values): the level's fields are the n_interior^ndim interior points, stored compactly at the start of the
 * context's slab fields; the context's grid is the odd extension, 2 (n_interior + 1) points per axis, on which the operator
 * is the periodic one and the Fourier solve is the sine-transform solve.  After this call sdc_eval_f / sdc_solve take and
 * return COMPACT fields (they pack into extension-sized scratch, run, extract), and sdc_sweep runs the node loop on the
 * device.  n_interior = 0 switches the mode off. */
int sdc_set_odd_interior(sdc_ctx* ctx, int n_interior);
/* The same idea in 2-D / 3-D (generic_ND_FD.py:99-133 'dirichlet-zero', order 2; helpers/problem_helper.py:143-224): the
 * interior n^ndim values are not contiguous inside their odd extension of (2(n+1))^ndim points, so fields stay compact and
 * are packed into / extracted from an extension-sized scratch field around eval_f and solve (problem level: the sweep then
 * runs node by node).  interior: n^ndim doubles, ext: (2(n+1))^ndim doubles, axis order as stored (last axis fastest). */
int sdc_odd_extend(sdc_ctx* ctx, const double* interior, double* ext, int n_interior, int ndim);
int sdc_odd_extract(sdc_ctx* ctx, const double* ext, double* interior, int n_interior, int ndim);

/* ---- stream / timing ------------------------------------------------------------------------------------ */
int sdc_sync(sdc_ctx* ctx);
/* hipEvent timing on the context's stream (GPUTimings analogue, hooks/log_timings.py:328-342):
 * begin records an event, end records another, synchronises and returns elapsed milliseconds. */
int sdc_timer_begin(sdc_ctx* ctx);
int sdc_timer_end(sdc_ctx* ctx, double* ms);
/* Per-kernel accumulated device time since the last reset, measured with events around every launch when
 * profiling is enabled (adds launch serialisation; off by default).  names/ms/calls arrays of length cap. */
int sdc_profile_enable(sdc_ctx* ctx, int on);
int sdc_profile_read(sdc_ctx* ctx, int cap, const char** names, double* ms, int* calls, int* count);

int sdc_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SDCMI_H */
