"""Benchmark of the SDC sweep path (BASELINE.json metric): time-steps/s and SDC-iterations/s of the 3-D heat
equation, finite differences, M=5 Gauss-Radau nodes, implicit (generic_implicit) sweeps, f64.

  python bench.py --gpus N --steps K --warmup W

N > 1 starts by itself: the parent process - before anything touches a GPU - starts one rank process per GPU (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), passes rank 0's JSON line on and exits non-zero with a JSON
`error` line if a rank fails or the job times out.  Launched under torch.distributed.run (WORLD_SIZE already in the
environment) every process is a rank right away.  `--backend gloo --same-device` puts all ranks on GPU 0 with the
shared-memory wire of the C-ABI communicator (RCCL refuses two ranks on one device): the whole multi-rank path rehearsed
on a one-GPU box.

One "step" = one block of N time steps (one time-slice per GPU; N = 1: one time step): predict, then 4 SDC
iterations (update_nodes + compute_residual + convergence check each) and the end point, all through the
drop-in plug-in path  controller -> sweeper_class/problem_class -> C-ABI -> HIP kernels.  Inputs are generated
on the device and are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
F64_VALU_PEAK_TFLOPS = 78.6   # f64 vector peak: half the guide's 157.3 TFLOP/s FP32 vector figure (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
VDP_VALU_PER_WAVE = 1328      # measured: SQ_INSTS_VALU per wave of k_vdp_sweep<5, lazy F> on the bench input (profiles/r03)


def _kernel_bytes(name, n, M, ncomp=1):
    """ALGORITHMIC bytes of one launch of each kernel of the sweep (DESIGN.md 'kernels'): fields read + written
    once, f64; one half spectrum = (n/2+1) n^2 complex.  Names carry the number of fields, e.g. fft_x_fwd[5]."""
    N = n**3
    field = 8.0 * N
    spec = 16.0 * (n // 2 + 1) * n * n
    base, nf, groups = name, M, 1
    if '[' in name:   # "fft_y_inv[5]": five fields; "fft_y_inv[5/32]": five fields, one of 32 groups of kx planes per launch
        base, inner = name[:name.index('[')], name[name.index('[') + 1:-1]
        nf, groups = (int(v) for v in inner.split('/')) if '/' in inner else (int(inner), 1)
    table = {
        'gather': (1 + M * ncomp + M) * field,          # u0 + F[1..M] -> R[1..M]
        'fft_x_fwd': nf * (field + spec),               # real tiles in, half spectra out
        'fft_y_fwd': 2 * nf * spec,
        'fft_z_fwd': 2 * nf * spec,
        'fft_z_solve': 2 * nf * spec,
        'spec_point': (1 + nf) * spec + nf * spec,      # S0 + S[1..M] in, S[1..M] out (in place)
        'spec_point_res': (1 + nf) * spec + 2 * nf * spec,  # ... + the residual spectra out
        'spec_point_only': (1 + nf) * spec + nf * spec,
        'spec_z_res': (1 + nf) * spec + 2 * nf * spec,      # spectral sweep + first inverse pass in one launch:
        'spec_z': (1 + nf) * spec + 2 * nf * spec,          # S0 + S in, S and the line-transformed field out
        'spec_z_res_spread': (1 + 2 * nf) * spec,           # first sweep after a spread predictor: only S0 is read
        'spec_z_spread': (1 + 2 * nf) * spec,
        # iterate recomputed from the transform of u0 (virtual sweeps): S0 in, the residual's transformed lines out
        # (a time slice on the trail reads one start value per receive more and may write the difference line: not counted -
        # the fractions of --gpus N lines are on the low side)
        **{f'spec_z_res_v{r}': (1 + nf) * spec for r in ('0', '1', '2', '3', '4', '5', '6', '7+')},
        **{f'spec_z_v{r}': (1 + nf) * spec for r in ('0', '1', '2', '3', '4', '5', '6', '7+')},   # (the iterate itself out)
        # long runs of sweeps: the real node multipliers of a mode pair (nf doubles per two modes) are read from a table,
        # advanced by one sweep and written back instead of being recomputed by replaying every earlier sweep
        # the sweep expected to be the last of its step also writes the last node's spectrum (end value / next start value)
        'spec_z_res_last': (2 + nf) * spec,
        'spec_z_last': (2 + nf) * spec,
        'spec_z_res_tab': (1 + nf) * spec + nf * spec / 2,
        'spec_z_tab': (1 + nf) * spec + nf * spec / 2,
        'spec_store': (1 + nf) * spec,                      # ... and its transforms written out when somebody needs them
        'spec_store_last': 2 * spec,                        # (only the last node's: the end value / next start value)
        'fft_x_norm': nf * spec,
        'fft_x_norm2': nf * spec + field,               # the norms before and after a receive in one pass: + the parked difference field
        'fft_x_scr': spec + field,                      # ... which this launch parks (one field)
        'fft_z_diff': 3 * nf * spec,                    # two start spectra in, their transformed difference out
        'trail_store': (4 + nf) * spec,                 # (typical: four start values in, every node's spectrum out)
        'trail_store_last': (4 + 1) * spec,
        'trail_send': (4 + 1) * spec,                   # the last node's spectrum alone, ahead of everything else (split send)
        'fft_x_inv_norm': nf * (field + spec),          # norms and the residual fields (time-parallel runs)
        'spec_z_resid': (1 + nf) * spec + nf * spec,        # residual spectrum of the cached iterate, no update
        'replace_u0': (3 + nf) * field,                 # new + old u0 in, u0 out, M residual fields in                        # half spectra in, max norms out
        'fft_z_inv': 2 * nf * spec,
        'fft_z_sym': 2 * nf * spec,                     # the same pass with the operator symbol multiplied in
        'fft_y_inv': 2 * nf * spec,
        'fft_x_inv': nf * (field + spec),
        'stencil': 2 * nf * field if ncomp == 1 else 3 * nf * field,  # IMEX: one read, impl + expl written
        'stencil_res': (1 + (1 + ncomp) * nf) * field,  # u0 + U[1..M] in, F[1..M] (impl, expl) out, residual norms
        'res_stencil': (1 + nf) * field,                # same launch with F deferred: u0 + U[1..M] in, norms out
        'amax': ncomp * field,
        'stencil_max': field,                           # f(u0) evaluated for its max norm only (predictor)
        'residual': (1 + M * ncomp + M) * field,        # u0, F[1..M], U[1..M] -> M norms
        'spread': (2 + 2 * M) * field,
        'copy': 2 * field,
    }
    b = table.get(base)
    return b / groups if b is not None else None


def _cpu_sample(sample_n, M, dt_ref_n, target_n, nsweeps, solver='CG'):
    """(solver='fft': the same sweep with the oracle's exact Fourier solve, oracle.spectral_solve - the algorithm of the GPU
    path - instead of the reference's CG)
    one bounded sample of the oracle (NumPy/SciPy restatement of the reference's path, oracle/sdc_oracle.py): heat 3-D
    sample_n^3, M nodes, 1 time step = nsweeps sweeps, CG(rtol 1e-12) like the reference's feasible 3-D configuration
    (BASELINE.md 3), same dt*nu/dx^2 stiffness as the GPU workload.  Runs in a child interpreter (no GPU, no torch)."""
    import numpy as np

    from oracle import sdc_oracle as O
    from pysdc_amd.coeffs import CollBase, QDELTA_GENERATORS
    from pysdc_amd.synth import init_field

    c = CollBase(M, 0, 1, 'LEGENDRE', 'RADAU-RIGHT')
    QI = np.zeros_like(c.Qmat)
    QI[1:, 1:] = QDELTA_GENERATORS['IE'](qGen=c.generator, tLeft=0).genCoeffs()
    coll = O.Coll(c.nodes, c.weights, c.Qmat, QI)
    nv = (sample_n,) * 3
    dt = dt_ref_n * (target_n / sample_n) ** 2
    t0 = time.perf_counter()
    if solver == 'fft':
        class HeatFourierSolve(O.HeatUnforced):
            def solve_system(self, rhs, factor, u0, t):
                return O.spectral_solve(self, rhs, factor)

        prob = HeatFourierSolve(nv, 0.1, 2)
    else:
        prob = O.HeatUnforced(nv, 0.1, 2, solver_type='CG', lintol=1e-12)
    setup = time.perf_counter() - t0
    u0 = init_field(nv, 2, 1e-3, 0)
    t0 = time.perf_counter()
    O.run_sdc(lambda: O.Level(prob, coll, dt, restol=-1.0), u0, 0.0, dt, maxiter=nsweeps)
    el = time.perf_counter() - t0
    return {'n': sample_n, 'seconds': el, 'setup_seconds': setup, 'sweeps': nsweeps, 'solver': solver,
            'cg_iterations': prob.work_counters['CG'].niter if 'CG' in prob.work_counters else 0}


def _spawn_cpu_sample(sample_n, M, dt_ref_n, target_n, nsweeps, solver='CG'):
    """a fresh interpreter per sample (a child process, never an exec of this one: the GPU is initialised here);
    single-threaded BLAS / OpenMP so that `cores` means what it says"""
    import subprocess

    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1', HIP_VISIBLE_DEVICES='')
    code = (f'import sys, json; sys.path.insert(0, {ROOT!r}); import bench; '
            f'print(json.dumps(bench._cpu_sample({sample_n}, {M}, {dt_ref_n!r}, {target_n}, {nsweeps}, {solver!r})))')
    return subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env,
                            cwd='/tmp')


def _collect(proc, timeout=600):
    out, _ = proc.communicate(timeout=timeout)
    return json.loads(out.decode().strip().splitlines()[-1])


class CpuBaseline:
    """kind = "port": the oracle (NumPy / SciPy restatement of the reference's path, CG rtol 1e-12 like the reference's
    feasible 3-D configuration) timed on the GPU box's host cores on bounded samples of the headline workload, every
    sample a child interpreter with single-threaded BLAS.  The figures are anchored on the LARGEST samples run:
      * `value`: `cores` concurrent copies of ONE sweep at 128^3 (the reference's path is single-threaded, so throughput
        over cores = independent copies; at most 32: sparse mat-vecs are bound by host memory bandwidth and more copies
        make the CPU look worse), time-steps/s = copies / (slowest copy x sweeps per step), scaled to the target by DOF;
      * `one_core`: ONE sweep at 256^3 on one core (it runs beside the GPU sub-records, which leave the host idle),
        scaled the same way;
      * `one_core_64_dof_scaled`: one core, a whole time step at 64^3, scaled by DOF - the optimistic bound for one
        core (the cost per DOF grows with the grid: memory hierarchy, the CG iteration counts stay)."""

    def __init__(self, M, dt_ref_n, target_n=1024, nsweeps=4, max_workers=32, big_n=256, mid_n=128):
        self.M, self.dt, self.target, self.nsweeps = M, dt_ref_n, target_n, nsweeps
        self.ncpu = os.cpu_count() or 1
        self.workers = max(1, min(self.ncpu, max_workers))
        self.big_n, self.mid_n = big_n, mid_n
        self.r64 = _collect(_spawn_cpu_sample(64, M, dt_ref_n, target_n, nsweeps))   # alone on the host
        self.big = _spawn_cpu_sample(big_n, M, dt_ref_n, target_n, 1) if big_n else None   # collected in finish()
        # the SAME algorithm as the GPU path beside it: one oracle sweep with the exact Fourier solve, one core, big_n^3
        self.big_fft = _spawn_cpu_sample(big_n, M, dt_ref_n, target_n, 1, 'fft') if big_n else None

    def finish(self):
        M, nsweeps, target = self.M, self.nsweeps, self.target
        rbig = None
        if self.big is not None:
            try:   # (a host much slower than the boxes seen so far: the line then goes without the one-core sample)
                rbig = _collect(self.big, timeout=600)
            except Exception:  # noqa: BLE001
                try:
                    self.big.kill()
                except OSError:
                    pass
        t0 = time.perf_counter()
        procs = [_spawn_cpu_sample(self.mid_n, M, self.dt, target, 1) for _ in range(self.workers)]
        rs = [_collect(p) for p in procs]
        wall = time.perf_counter() - t0
        slowest = max(r['seconds'] for r in rs)
        sc = lambda n: (float(n) / target) ** 3       # noqa: E731  (DOF scaling of a rate measured at n^3)
        all_cores = self.workers / (slowest * nsweeps) * sc(self.mid_n)
        v64 = 1.0 / self.r64['seconds'] * sc(64)
        out = {'value': all_cores, 'unit': 'time-steps/s', 'cores': self.workers, 'kind': 'port',
               'sample': (f'{self.workers} concurrent copies (host: {self.ncpu} cores) of ONE oracle sweep, heat 3-D '
                          f'{self.mid_n}^3 f64 M={M}, CG rtol 1e-12 ({rs[0]["cg_iterations"]} it), slowest {slowest:.1f} s '
                          f'({wall:.0f} s wall); x{nsweeps} sweeps/step, scaled to {target}^3 by DOF')[:200],
               'one_core_64_dof_scaled': v64}
        if rbig is not None:
            out['one_core'] = {'value': 1.0 / (rbig['seconds'] * nsweeps) * sc(self.big_n), 'n': self.big_n,
                               'seconds_per_sweep': rbig['seconds'], 'cg_iterations': rbig['cg_iterations']}
        # same algorithm (Fourier solve): all cores = concurrent copies of one sweep at mid_n^3, one core = one sweep at big_n^3
        same, rsf, rbf = None, None, None
        try:
            procs = [_spawn_cpu_sample(self.mid_n, M, self.dt, target, 1, 'fft') for _ in range(self.workers)]
            rsf = [_collect(p) for p in procs]
            slow_f = max(r['seconds'] for r in rsf)
            same = {'value': self.workers / (slow_f * nsweeps) * sc(self.mid_n), 'unit': 'time-steps/s', 'cores': self.workers,
                    'n': self.mid_n, 'seconds_per_sweep': slow_f,
                    'note': "oracle sweep with its exact Fourier solve (numpy rfftn, one thread per copy): the GPU path's "
                            'algorithm, not the CG of the reference\'s feasible 3-D configuration'}
            if self.big_fft is not None:
                rbf = _collect(self.big_fft, timeout=600)
                same['one_core'] = {'value': 1.0 / (rbf['seconds'] * nsweeps) * sc(self.big_n), 'n': self.big_n,
                                    'seconds_per_sweep': rbf['seconds']}
        except Exception as e:  # noqa: BLE001  (never takes the line down)
            same = same or {'error': repr(e)[:120]}
        out['same_algorithm'] = same
        detail = {'r64': self.r64, 'big': rbig, 'mid': rs, 'mid_wall_seconds': wall, 'mid_fft': rsf, 'big_fft': rbf}
        return out, detail


def allencahn_ref2d_extruded():
    """--ac-variant ref2d: the product's 3-D pseudo-spectral two-level problem carrying the reaction term and the start value of
    the REFERENCE's 2-D class (AllenCahn_2D_FFT.py:140-141 1/eps^2 u (1 - u^nu); :171-174 the circle on [-L/2, L/2]^2),
    constant along z - every z-plane then evolves exactly like the reference's allencahn2d_imex, whose golden two-level runs
    (tests/golden/runs_cfg5.npz: MLSDC, and PFASST on 8 processes) pin iteration counts and end value of BASELINE config 5."""
    import numpy as np

    from pysdc_amd.problems import allencahn_imex

    class allencahn2d_extruded(allencahn_imex):
        def __init__(self, nvars=None, nu=2, eps=0.04, radius=0.25):
            super().__init__(nvars=nvars, eps=eps, radius=radius)
            self._nu2d = nu

        def configure_engine(self, engine):
            engine.set_symbol(0, self._symbol())
            engine.set_reaction(1, 1.0 / self.eps**2, 0.0, int(self._nu2d))

        def u_exact(self, t, **kwargs):
            assert t == 0, 'ERROR: u_exact only valid for t=0'
            n = self.nvars[0]
            x = np.array([i * self.L / n - self.L / 2.0 for i in range(n)])
            xv, yv = np.meshgrid(x, x, indexing='ij')
            plane = np.tanh((self.radius - np.sqrt(xv**2 + yv**2)) / (np.sqrt(2) * self.eps))
            return self._from_host(np.ascontiguousarray(np.broadcast_to(plane[:, :, None], self.nvars)))

    return allencahn2d_extruded


def stream_reference(torch, eng, nbytes=1 << 32):
    """what plain streaming operations reach on THIS GPU (GB/s), measured after the timed region, to read the
    roofline fraction against: write-only fill, read-only max reduction, device-to-device copy (1 read : 1 write)"""
    import ctypes as C
    try:
        n = nbytes // 8
        x = torch.empty(n, dtype=torch.float64, device='cuda')
        y = torch.empty(n, dtype=torch.float64, device='cuda')
    except Exception:  # noqa: BLE001  (no room left beside the slabs)
        return None
    lib, out, res = eng.lib, C.c_double(), {}
    for name, fn, moved in (('fill', lambda: lib.sdc_vec_fill(None, n, 1.0, x.data_ptr()), nbytes),
                            ('amax', lambda: lib.sdc_vec_amax(None, n, x.data_ptr(), C.byref(out)), nbytes),
                            ('copy', lambda: y.copy_(x), 2 * nbytes)):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        res[name] = moved * 5 / (time.perf_counter() - t) / 1e9
    return res


def run_workload(args, world, rank, use_dist, with_stream_reference=True):
    """one measured run of one workload; returns the record on rank 0 (None elsewhere)"""
    import numpy as np
    import torch
    import torch.distributed as dist

    import ctypes as C

    from pysdc_amd import lib as Lb
    from pysdc_amd.controller import controller_nonMPI, controller_dist
    from pysdc_amd.problems import heatNd_unforced, advectiondiffusionNd_imex, vanderpol_ensemble, allencahn_imex
    from pysdc_amd.transfer import mesh_to_mesh
    from pysdc_amd.sweepers import generic_implicit, imex_1st_order
    import pysdc_amd.level as _level
    _level.LAZY_PREDICTOR_RESIDUAL = bool(args.lazy_predictor_residual)   # default: K+1 residuals per step, like the reference
    from pysdc_amd.stats import get_sorted

    M, K = args.nodes, (args.sweeps if args.restol < 0 else 50)
    ncomp = 1
    if args.workload == 'heat':
        n = args.n or 1024
        # 1024^3 with M = 5: the sweeps that stay in Fourier space hold ~15 fields of 8.6 GB (U[0], two end-value buffers, M work
        # spectra, the start value's and the last node's spectrum, start / end value objects of the runs; U[1..M], F and the node
        # spectra are mapped only when something touches them in real space - the eager-fields sub-record does: ~31 fields); a
        # time slice adds the start values of its trail, the spectrum inbox, two spare spectra, a second set of M work spectra
        # (the put-off passes of one iterate run behind the next sweep's first launches) and the relay staging: 215 GB = 25
        # fields measured by scripts/emulate_timeslice.py.  The second set of work spectra is given up when it does not fit
        # (sdc_sweep: same numbers, less overlap), so 24 fields are asked for.  A GPU that cannot hold that ends the job with an
        # error: the grid is never changed behind the caller's back, so that the lines of --gpus 1 and --gpus N are always
        # about the same workload
        need = ((31.0 if args.eager_fields else 17.0) if world == 1 else 24.0) * 8.0 * n**3
        free = torch.cuda.mem_get_info()[0]
        if free < need:
            raise MemoryError(f'heat {n}^3 needs {need / 1e9:.0f} GB of HBM on every GPU, {free / 1e9:.0f} GB free on rank {rank} '
                              f'(pass --n explicitly for a smaller grid)')

        dt = 1e-3 * (512.0 / n) ** 2  # dt*nu*12/dx^2 ~ 315 at every size
        desc = dict(problem_class=heatNd_unforced, problem_params=dict(nvars=(n, n, n), nu=0.1, freq=2, order=2, solver_type=args.solver_type,
                                                                         lintol=1e-12, liniter=10000),
                    sweeper_class=generic_implicit,
                    sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI=args.qi),
                    level_params=dict(dt=dt, restol=args.restol, nsweeps=1), step_params=dict(maxiter=K))
        wl = (f'heatNd_unforced {n}^3 periodic order-2 FD, nu=0.1, M={M} LEGENDRE RADAU-RIGHT, QI={args.qi}, '
              f'generic_implicit')
        unit = 'time-steps/s'
    elif args.workload == 'advdiff':
        n = args.n or 512
        ncomp = 2
        dt = 1e-3 * (512.0 / n) ** 2
        desc = dict(problem_class=advectiondiffusionNd_imex,
                    problem_params=dict(nvars=(n, n, n), nu=0.02, c=1.0, freq=2, order=2),
                    sweeper_class=imex_1st_order,
                    sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI=args.qi, QE='EE'),
                    level_params=dict(dt=dt, restol=args.restol, nsweeps=1), step_params=dict(maxiter=K))
        wl = (f'advectiondiffusionNd_imex {n}^3 periodic order-2 FD, nu=0.02 (implicit), c=1 (explicit), M={M} '
              f'RADAU-RIGHT, QI={args.qi}, QE=EE, imex_1st_order')
        unit = 'time-steps/s'
    elif args.workload == 'allencahn':
        n = args.n or 256
        ncomp = 2
        M = 3
        dt = 1e-3
        ref2d = getattr(args, 'ac_variant', 'sphere') == 'ref2d'
        desc = dict(problem_class=allencahn_ref2d_extruded() if ref2d else allencahn_imex,
                    problem_params=(dict(nvars=[(n, n, n), (n // 2, n // 2, n // 2)], nu=2, eps=0.04, radius=0.25) if ref2d else
                                    dict(nvars=[(n, n, n), (n // 2, n // 2, n // 2)], eps=0.04, radius=0.25, init_type='sphere')),
                    sweeper_class=imex_1st_order,
                    sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='LU', QE='EE'),
                    level_params=dict(dt=dt, restol=args.restol, nsweeps=1), step_params=dict(maxiter=K),
                    space_transfer_class=mesh_to_mesh, space_transfer_params=dict(iorder=6, rorder=2, periodic=True))
        wl = ((f'allencahn2d_imex extruded along z (reference reaction term and circle, z-invariant) {n}^3 / {n // 2}^3, two-level '
               if ref2d else f'allencahn_imex {n}^3 / {n // 2}^3 (pseudo-spectral, eps=0.04, sphere), two-level ') +
              f'{"PFASST" if world > 1 else "MLSDC"}, M={M} RADAU-RIGHT on both levels, QI=LU, QE=EE, '
              f'mesh_to_mesh iorder 6 / rorder 2')
        unit = 'time-steps/s'
    else:
        n = 0
        dt = 0.05
        rng = np.random.default_rng(0)
        u0h = rng.uniform(-2.0, 2.0, (2, args.ntraj))
        desc = dict(problem_class=vanderpol_ensemble,
                    problem_params=dict(ntraj=args.ntraj, u0=u0h, mu=5.0, newton_tol=1e-9, newton_maxiter=100,
                                        block_solver='mfma' if getattr(args, 'mfma', False) else 'closed_form'),
                    sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='LU'),
                    level_params=dict(dt=dt, restol=args.restol, nsweeps=1), step_params=dict(maxiter=K))
        wl = (f'vanderpol ensemble, {args.ntraj} trajectories, mu=5, M={M} RADAU-RIGHT, QI=LU, Newton tol 1e-9, block solver '
              + ('MFMA (v_mfma_f64_4x4x4)' if getattr(args, 'mfma', False) else 'closed form (VALU)'))
        unit = 'trajectory-steps/s'
        if world > 1:
            raise SystemExit('the ensemble shards trivially over GPUs (independent trajectories); run --gpus 1')
    if args.skip_residual:
        desc['sweeper_params']['skip_residual_computation'] = ('IT_CHECK', 'IT_FINE', 'IT_DOWN', 'IT_UP', 'IT_COARSE')
    cparams = dict(logger_level=40, mssdc_jac=args.mssdc == 'jacobi')
    if args.workload == 'allencahn' and world > 1:
        cparams['predict_type'] = 'pfasst_burnin'
    if not use_dist:
        ctrl = controller_nonMPI(1, cparams, desc)
        step = ctrl.MS[0]
    else:
        cparams['comm_wire'] = args.wire
        ctrl = controller_dist(cparams, desc)
        step = ctrl.S
    L = step.levels[0]
    eng = L.engine  # allocates the device slabs
    eng.set_spectral_reuse(not args.no_spectral_reuse)
    eng.set_deferred(not args.eager_fields)
    if args.virtual_sweeps is not None and hasattr(eng, 'set_virtual_sweeps'):
        eng.set_virtual_sweeps(args.virtual_sweeps)
    if args.multiplier_table is not None and hasattr(eng, 'set_multiplier_table'):
        eng.set_multiplier_table(args.multiplier_table)
    if args.workload == 'allencahn' and getattr(args, 'start_plane_file', None):
        # (tests: a 2-D start value extruded along z - e.g. the perturbed circle a golden run of the reference started from)
        plane = np.load(args.start_plane_file)
        u0 = L.prob.u_init
        u0[:] = np.ascontiguousarray(np.broadcast_to(plane[:, :, None], tuple(L.prob.nvars)))
    elif args.workload in ('vdp', 'allencahn'):
        u0 = L.prob.u_exact(0.0)
    else:
        # synthetic input on the device: sin mode (freq 2) + 1e-3 * seeded noise
        u0 = L.prob.u_init
        freq = (C.c_int * 3)(2, 2, 2)
        Lb.check(eng.lib.sdc_init_field(eng.ctx, u0.ptr, freq, 1e-3, 0), eng.ctx)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    block = dt * world  # one bench step advances `world` time steps
    uend, _ = ctrl.run(u0, 0.0, block * args.warmup) if args.warmup > 0 else (u0, None)
    # run() hands out a fresh end-value object per call like the reference; let the caching allocator hold a block of that
    # size already, so that the timed region contains the run and not a first-time 8.6 GB hipMalloc (set-up, not work)
    if world == 1:
        spare = torch.empty(int(np.prod(uend.shape)), dtype=torch.float64, device='cuda')
        del spare
    del u0
    sync()
    # HIP events around every launch of the timed region give the kernel table (roofline).  A configuration made of ~75
    # launches of 10 - 300 us per iteration (the two-level Allen-Cahn run) pays ~10 us per launch for them - a tenth of its
    # wall time: there the timed region runs WITHOUT events and the same steps run once more WITH them for the table
    # (`kernel_events`: 'timed region' | 'second run of the same steps' | 'none'; the second run's wall time is reported too).
    events_mode = getattr(args, 'kernel_events', None) or ('separate' if args.workload == 'allencahn' else 'timed')
    if getattr(args, 'no_kernel_events', False):
        events_mode = 'none'
    coarser = [Lc.engine for Lc in step.levels[1:] if hasattr(Lc, 'engine')]

    def set_events(on):
        eng.profile_enable(on)
        for ec in coarser:
            ec.profile_enable(on)

    set_events(events_mode == 'timed')
    t0 = time.perf_counter()
    uend, stats = ctrl.run(uend, block * args.warmup, block * (args.warmup + args.steps))
    sync()
    el = time.perf_counter() - t0
    el_events = None
    if events_mode == 'separate':
        set_events(True)
        t1 = time.perf_counter()
        uend2, _ = ctrl.run(uend, block * (args.warmup + args.steps), block * (args.warmup + 2 * args.steps))
        sync()
        el_events = time.perf_counter() - t1
        del uend2
    prof = eng.profile_read()
    prof_coarse = [(ec.n, ec.profile_read()) for ec in coarser]
    set_events(False)

    el_own = el
    elt = torch.tensor([el], dtype=torch.float64)   # (host tensor: the process group is gloo)
    if use_dist:
        dist.all_reduce(elt, op=dist.ReduceOp.MAX)
    el = float(elt.item())
    niter = [v for _, v in get_sorted(stats, type='niter')]
    assert all(v <= K for v in niter) and len(niter) > 0, niter
    sweeps_done = sum(niter)  # over the ranks' own steps; equals steps * K for a fixed number of sweeps
    finite = bool(np.isfinite(abs(uend)))

    def kernel_bytes(name, n_, M_):  # noqa: F811  (workload-aware wrapper)
        if args.workload == 'vdp':
            T = args.ntraj
            return {'vdp_sweep': 8.0 * T * (2 + 2 * M_ + 2 * M_ + 4 * M_), 'vdp_eval': 8.0 * T * 4,
                    'vdp_sweep_lazyf': 8.0 * T * (2 + 2 * M_ + 2 * M_),   # u0 + old nodes in, new nodes out (F deferred)
                    'vdp_sweep_mfma': 8.0 * T * (2 + 2 * M_ + 2 * M_),
                    'residual': 8.0 * 2 * T * (1 + 2 * M_), 'spread': 8.0 * 2 * T * (2 + 2 * M_),
                    'copy': 8.0 * 2 * T * 2}.get(name.split('[')[0])
        return _kernel_bytes(name, n_, M_, ncomp)

    per_rank = None
    if use_dist:
        # every rank's own view, so that a scaling run explains itself: wall time, iterations, time spent in the
        # hand-over (hooks: pre_comm .. post_comm, host clock), device time of its kernels
        comm_s = sum(v for _, v in get_sorted(stats, type='timing_comm'))
        mine = {'rank': rank, 'seconds': el_own, 'niter': niter, 'ms_per_iteration': 1e3 * el_own / max(1, sum(niter)),
                'comm_ms_per_iteration': 1e3 * comm_s / max(1, sum(niter)),
                'kernel_ms_per_iteration': sum(v[0] for v in prof.values()) / max(1, sum(niter)),
                'two_hop_exchanges': getattr(ctrl, 'two_hop_calls', None),
                'mesh_broadcasts': getattr(ctrl, 'bcast_two_hop_calls', None),
                'wire': (ctrl._comms[0].info()['wire'] if getattr(ctrl, '_comms', None) else None),
                'message_bytes': 8 * int(np.prod(uend.shape))}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        steps_total = args.steps * world
        sweeps_total = steps_total * K if args.restol < 0 else sweeps_done * world
        units = args.ntraj if args.workload == 'vdp' else 1
        # dominant kernel of the timed region, from HIP events on the engine's stream
        # (the launches of the virtual sweep are ONE kernel - named by the number of sweeps they repeat - and count together)
        merged = {}
        for k, v in prof.items():
            base = k.split('[')[0]
            key = ('spec_z_res_v*' + k[len(base):]) if base.startswith(('spec_z_res_v', 'spec_z_v')) else k
            t, c_ = merged.get(key, (0.0, 0))
            merged[key] = (t + v[0], c_ + v[1])
        dom = max(merged.items(), key=lambda kv: kv[1][0]) if merged else (None, (0.0, 0))
        if dom[0] is not None and dom[0].startswith('spec_z_res_v*'):
            dom = (dom[0].replace('v*', 'v0'), dom[1])      # (bytes of any of them; the time is their average)
        kern = {k: {'ms_per_launch': v[0] / v[1], 'launches': v[1],
                    'gbs': (kernel_bytes(k, n, M) or 0) / (v[0] / v[1]) / 1e6}
                for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]) if v[1] > 0}
        roof = None
        if dom[0] is not None and kernel_bytes(dom[0], n, M):
            ach = kernel_bytes(dom[0], n, M) / (dom[1][0] / dom[1][1]) / 1e6
            traffic = None
            tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tfile):
                tr = json.load(open(tfile)).get(f'{dom[0].split("[")[0]}@{n}')
                traffic = tr['hbm_bytes_per_launch'] if tr else None  # PMC passes of the same build (profiles/)
                if traffic is not None and '/' in dom[0]:   # one of G groups per launch (PMC passes run ungrouped)
                    traffic /= int(dom[0][dom[0].index('/') + 1:-1])
            roof = {'kernel': dom[0], 'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': ach / HBM_PEAK_GBS, 'traffic': traffic,
                    'traffic_source': ('profiles/traffic.json: rocprofv3 --pmc passes of this kernel run by the builder '
                                       '(scripts/profile_bench.sh), not measured in this run') if traffic is not None else None,
                    'algorithmic_bytes_per_launch': kernel_bytes(dom[0], n, M),
                    'ms_per_launch': dom[1][0] / dom[1][1],
                    'stream_reference_gbs': stream_reference(torch, eng) if with_stream_reference else None}
            if args.workload == 'vdp' and dom[0].startswith('vdp_sweep'):
                # the ensemble sweep is bound by its f64 vector instructions, not by its 1.76 GB (SQ counters under profiles/:
                # VDP_VALU_PER_WAVE wave instructions per 64 trajectories, Newton iterations of this input included): the
                # bound is the f64 vector peak of the guide, one 64-lane instruction priced as 64 fused multiply-adds
                flops = 2.0 * 64 * VDP_VALU_PER_WAVE * (args.ntraj / 64.0)
                tf = flops / (dom[1][0] / dom[1][1]) / 1e9
                roof.update({'bound': 'f64-valu', 'achieved': tf, 'peak': F64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': tf / F64_VALU_PEAK_TFLOPS, 'hbm_gbs': ach,
                             'note': f'{VDP_VALU_PER_WAVE} VALU wave-instructions per 64 trajectories and sweep (SQ_INSTS_VALU, '
                                     'profiles/r03/vdp_block_solver_counters.json) priced as 64-lane FMAs'})
        if args.workload == 'allencahn' and sweeps_total:
            # a multi-level iteration is many short launches: the figure that says something is the iteration as a whole -
            # algorithmic bytes of every launch the engines of both levels timed (quadratures, transforms, node right-hand
            # sides) plus the bytes of the space-transfer launches, over the WALL time of an iteration
            tab_extra = {'node_rhs': 4.0, 'axpby': 3.0, 'copy': 2.0, 'fill': 1.0, 'reaction': 2.0, 'integrate': 1 + 2 * M + M,
                         'end_point': 2 + 2 * M}
            def it_bytes(pr, n_):
                tot = 0.0
                for k_, v_ in pr.items():
                    b_ = _kernel_bytes(k_, n_, M, ncomp)
                    if b_ is None and k_.split('[')[0] in tab_extra:
                        b_ = tab_extra[k_.split('[')[0]] * 8.0 * n_**3
                    tot += (b_ or 0.0) * v_[1]
                return tot
            moved = it_bytes(prof, n) + sum(it_bytes(pc, nc_) for nc_, pc in prof_coarse)
            kernel_ms = sum(v_[0] for v_ in prof.values()) + sum(v_[0] for _, pc in prof_coarse for v_ in pc.values())
            iters = max(1, sweeps_total // world)
            # the space transfers run on the stream outside the engines' timers; since round 6 each is ONE launch whose
            # algorithmic bytes are its fields (sdc_transfer_apply_nested): per iteration the restriction of U[1..M] and of the
            # M quadrature sums (with the coarse sums subtracted on the way out) and the prolongation of the M coarse
            # corrections (u_G and uold_G read, the fine node values read and written)
            nc_ = n // 2
            xfer_per_iter = 8.0 * (n**3 * (M + M) + nc_**3 * (M + 2 * M)) + 8.0 * (nc_**3 * 2 * M + n**3 * 2 * M)
            moved += xfer_per_iter * iters
            wall_ms = 1e3 * el / iters
            roof = {'kernel': 'iteration (fine sweep, FAS down / coarse sweep / up, re-evaluation; transfers included)', 'bound': 'hbm',
                    'achieved': moved / iters / wall_ms / 1e6, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': moved / iters / wall_ms / 1e6 / HBM_PEAK_GBS, 'traffic': None,
                    'algorithmic_bytes_per_launch': moved / iters, 'ms_per_launch': wall_ms,
                    'engine_kernel_ms_per_iteration': kernel_ms / iters,
                    'launches_per_iteration': (sum(v_[1] for v_ in prof.values())
                                               + sum(v_[1] for _, pc in prof_coarse for v_ in pc.values())) / iters}
        in_sweep = ('gather', 'fft_x_fwd', 'fft_y_fwd', 'fft_z_fwd', 'fft_z_solve', 'spec_point', 'spec_point_res', 'spec_point_only', 'spec_z', 'spec_z_res', 'spec_z_spread', 'spec_z_res_spread',
                    'spec_z_res_v0', 'spec_z_res_v1', 'spec_z_res_v2', 'spec_z_res_v3', 'spec_z_res_v4', 'spec_z_res_v5', 'spec_z_res_v6', 'spec_z_res_v7+', 'spec_store', 'spec_z_res_tab', 'spec_z_tab', 'spec_z_res_last', 'spec_z_last',
                    'spec_z_v0', 'spec_z_v1', 'spec_z_v2', 'spec_z_v3', 'spec_z_v4', 'spec_z_v5', 'spec_z_v6', 'spec_z_v7+',
                    'fft_x_norm', 'fft_x_inv_norm', 'fft_x_norm2', 'fft_x_scr', 'fft_z_diff', 'trail_send', 'trail_store', 'fft_x_norm_add',
                    'fft_z_inv', 'fft_y_inv',
                    'fft_x_inv', 'stencil', 'stencil_res', 'res_stencil', 'vdp_sweep', 'vdp_sweep_lazyf', 'vdp_sweep_mfma')
        def of_sweep(k):   # (one-field launches beside M-field ones are the predictor's norm-only transform, once per step)
            return k.split('[')[0] in in_sweep and not (k.endswith('[1]') and M > 1 and args.workload in ('heat', 'advdiff'))

        sweep_ms = sum(v[0] for k, v in prof.items() if of_sweep(k)) / max(1, sweeps_total // world)
        # all launches of a sweep together: the bytes they actually move (their algorithmic bytes; PMC traffic agrees to
        # 1.00x, profiles/) and SURVEY 8(d)'s floor 8 N (3M+1) [IMEX: 8 N (5M+1)], both over the kernel time of a sweep
        sweep_bytes = sum((kernel_bytes(k, n, M) or 0) * v[1] for k, v in prof.items() if of_sweep(k)) / max(1, sweeps_total // world)
        floor_bytes = 8.0 * n**3 * ((3 if ncomp == 1 else 5) * M + 1) if n else None
        roof_sweep = None
        if sweep_ms and n and sweep_bytes:
            launches = sorted(((k, v[0] / max(1, sweeps_total // world)) for k, v in prof.items() if of_sweep(k)),
                              key=lambda kv: -kv[1])
            roof_sweep = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'ms_per_sweep': sweep_ms,
                          'bytes_moved_per_sweep': sweep_bytes, 'achieved': sweep_bytes / sweep_ms / 1e6,
                          'frac': sweep_bytes / sweep_ms / 1e6 / HBM_PEAK_GBS,
                          'floor_bytes_per_sweep': floor_bytes, 'achieved_on_floor': floor_bytes / sweep_ms / 1e6,
                          'frac_on_floor': floor_bytes / sweep_ms / 1e6 / HBM_PEAK_GBS,
                          'launches_ms_per_sweep': {k: round(t, 3) for k, t in launches[:8]}}
        out = {
            'metric': {'heat': 'time-steps/s (HeatND 3-D FD, M=5, implicit SDC sweeps)',
                       'advdiff': 'time-steps/s (advection-diffusion 3-D FD IMEX, M=5)',
                       'vdp': 'trajectory-steps/s (van der Pol ensemble, M=5)',
                       'allencahn': 'time-steps/s (Allen-Cahn 3-D two-level MLSDC / PFASST, M=3)'}[args.workload],
            'value': units * steps_total / el, 'unit': unit, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * el / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'{wl}, ' + (f'{K} sweeps/step (restol=-1, maxiter={K})' if args.restol < 0 else f'restol={args.restol:g} (maxiter={K})') + f', dt={dt:g}, '
                                   f'solver={"direct (Fourier)" if args.solver_type == "direct" else "CG rtol 1e-12 on the device"}, spectral_reuse={not args.no_spectral_reuse}, '
                                   f'deferred_node_fields={not args.eager_fields}'
                                   + (', skip_residual_computation=all stages' if args.skip_residual else '')
                                   + (', residual of the predictor\'s state put off until somebody reads it (--lazy-predictor-residual)'
                                      if (_level.LAZY_PREDICTOR_RESIDUAL and args.restol < 0) else ''),
                       'time_parallel': f'{world} time-slice(s), one per GPU, '
                                        + ('two-level PFASST' if (args.workload == 'allencahn' and world > 1) else
                                           f'multi-step SDC ({"Jacobi" if args.mssdc == "jacobi" else "Gauss-Seidel"})')
                                        + (f'; slice flow (trail, put-off pass, split send) = {tuple(int(v) for v in ctrl.timeslice_flow)}'
                                           if getattr(ctrl, 'timeslice_flow', None) else '')
                                        + (f'; wire {args.wire}, mode: {getattr(args, "wire_mode", "default")}'
                                           + (f' ({"64^3 / 32^3 two-level" if args.workload == "allencahn" else "64^3 and 512^2"} checks vs serial emulation: {args.wire_check:.1e})'
                                              if getattr(args, 'wire_check', None) is not None else '') if world > 1 else '')},
            'sdc_iters_per_s': units * sweeps_total / el,
            'niter': niter,
            'work_counters': {k: v.niter for k, v in L.prob.work_counters.items()},
            'sweep_kernels_ms': sweep_ms,
            'sweep_floor_gbs': (8.0 * n**3 * ((3 if ncomp == 1 else 5) * M + 1) / sweep_ms / 1e6
                                if sweep_ms and n else None),
            'roofline': roof, 'roofline_sweep': roof_sweep, 'kernels': kern, 'finite': finite,
            'device_bytes_per_gpu': eng.device_bytes,
            'params': {'M': M, 'dt': dt, 'n': n}, 'restol': args.restol,
            'kernel_events': {'timed': 'timed region', 'separate': 'second run of the same steps', 'none': 'none'}[events_mode],
            'ms_per_step_with_events': (1e3 * el_events / args.steps) if el_events is not None else None,
        }
        if per_rank is not None:
            out['per_rank'] = per_rank
        if args.dump_end_value:
            np.save(args.dump_end_value, uend.get())
        if hasattr(ctrl, 'close'):
            ctrl.close()
        return out
    if hasattr(ctrl, 'close'):
        ctrl.close()
    return None


SUB_PLAN = [
    # (short title (<= 40 characters, goes into the line), argument overrides)
    ('heat 1024^3 eager U,F (reference flow)', dict(eager_fields=True, steps=3, warmup=1)),
    ('heat 1024^3 lazy predictor residual', dict(lazy_predictor_residual=True, steps=3, warmup=1)),
    ('heat 1024^3 restol 1e-10', dict(restol=1e-10, steps=2, warmup=1)),
    ('heat 1024^3 QI=MIN-SR-S', dict(qi='MIN-SR-S', steps=3, warmup=1)),
    ('heat 256^3 CG rtol 1e-12 on device', dict(n=256, solver_type='CG', steps=2, warmup=1)),
    ('cfg2 heat 512^3', dict(n=512, steps=10, warmup=2)),
    ('cfg3 advdiff IMEX 512^3', dict(workload='advdiff', n=512, steps=10, warmup=2)),
    ('cfg4 vdp 1e7 trajectories', dict(workload='vdp', steps=10, warmup=2)),
    ('cfg5 Allen-Cahn 256^3/128^3 MLSDC', dict(workload='allencahn', steps=10, warmup=2)),
]


def extras(args):
    """short runs of the other BASELINE configurations and of the headline workload's variants, in this process, so
    that they are measured by the same command the driver times.  Returns the full records (side file); the line
    carries compact_sub() of each."""
    import argparse as ap
    import gc

    import torch

    recs = []
    for title, over in SUB_PLAN:
        gc.collect()
        torch.cuda.empty_cache()
        a = ap.Namespace(**{**vars(args), **over})
        try:
            r = run_workload(a, 1, 0, False, with_stream_reference=False)
            r['title'] = title
            recs.append(r)
        except Exception as e:  # noqa: BLE001  a sub-record must never take the headline line down
            recs.append({'title': title, 'error': repr(e)[:200]})
    gc.collect()
    torch.cuda.empty_cache()
    return recs


def _r(x, digits=5):
    """numbers of the line: a few significant digits, never NaN / Infinity (not JSON)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        return float(f'{x:.{digits}g}')
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def compact_sub(r):
    """one sub-value of the line: title, value, ms per step, roofline fraction of its dominant kernel / of its sweep"""
    if 'error' in r:
        return {'title': r['title'][:40], 'error': r['error'][:120]}
    roof, rs = r.get('roofline') or {}, r.get('roofline_sweep') or {}
    c = {'title': r['title'][:40], 'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'],
         'kernel': roof.get('kernel'), 'bound': roof.get('bound'), 'frac': roof.get('frac'), 'sweep_frac': rs.get('frac')}
    if len(set(r['niter'])) > 1 or r.get('restol', -1) > 0:
        c['niter'] = r['niter'][:8]
    if r.get('ms_per_step_with_events') is not None:   # (timed without HIP events; the kernel table from a second run with them)
        c['ms_per_step_with_events'] = r['ms_per_step_with_events']
    return _r(c, 4)


def compact_line(out, subs=None, cpu=None, details_path=None, sustained=None):
    """THE line the driver parses: the contract's keys, `roofline` and `cpu_baseline`, the sweep's fraction, one short list
    of sub-values.  Everything else - per-kernel tables, per-step iteration lists, long descriptions, the sub-records'
    own rooflines - goes to the side file.  Kept far below the 8 KB of standard output the driver retains
    (tests/test_bench_line.py)."""
    roof, rs = out.get('roofline'), out.get('roofline_sweep')
    niter = out.get('niter') or []
    line = {k: out[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                'scaling', 'vs_baseline', 'dtype', 'data')}
    line['config'] = {'workload': out['config']['workload'][:400], 'time_parallel': out['config']['time_parallel'][:260]}
    line['sdc_iters_per_s'] = out['sdc_iters_per_s']
    line['niter'] = niter[0] if len(set(niter)) == 1 else {'min': min(niter), 'max': max(niter), 'sum': sum(niter)}
    line['finite'] = out['finite']
    if roof:
        line['roofline'] = {k: roof.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic',
                                                     'traffic_source', 'algorithmic_bytes_per_launch', 'ms_per_launch')}
    else:
        line['roofline'] = None
    if rs:
        line['roofline_sweep'] = {k: rs.get(k) for k in ('bound', 'peak', 'unit', 'ms_per_sweep', 'bytes_moved_per_sweep',
                                                          'achieved', 'frac', 'floor_bytes_per_sweep', 'frac_on_floor')}
    if cpu is not None:
        line['cpu_baseline'] = cpu
    if subs:
        line['sub'] = [compact_sub(r) for r in subs]
        for r in subs:   # the two variants of the headline workload the verdicts compare it with, by name
            if 'error' not in r and r['title'].startswith('heat 1024^3 eager'):
                line['value_eager_fields'] = r['value']
            if 'error' not in r and r['title'].startswith('heat 1024^3 lazy'):
                line['value_lazy_predictor_residual'] = r['value']
    if sustained is not None:
        # the same configuration re-timed at the end of the process (after the sub-records): what the part sustains
        if 'error' in sustained:
            line['value_sustained'] = None
            line['sustained'] = {'error': sustained['error'][:120]}
        else:
            rf, rsw = sustained.get('roofline') or {}, sustained.get('roofline_sweep') or {}
            line['value_sustained'] = sustained['value']
            line['sustained'] = {'steps': sustained['steps'], 'ms_per_step': sustained['ms_per_step'],
                                 'roofline': {'kernel': rf.get('kernel'), 'frac': rf.get('frac'),
                                              'ms_per_launch': rf.get('ms_per_launch')},
                                 'sweep_frac': rsw.get('frac'), 'ms_per_sweep': rsw.get('ms_per_sweep')}
    if out.get('per_rank'):
        pr = out['per_rank']
        line['per_rank'] = {'seconds': [p['seconds'] for p in pr],
                            'ms_per_iteration': [p['ms_per_iteration'] for p in pr],
                            'comm_ms_per_iteration': [p['comm_ms_per_iteration'] for p in pr],
                            'kernel_ms_per_iteration': [p['kernel_ms_per_iteration'] for p in pr],
                            'sweeps': [sum(p['niter']) for p in pr],
                            'wire': pr[0].get('wire'), 'message_bytes': pr[0].get('message_bytes')}
    if details_path:
        line['details'] = details_path
    return _r(line)


def write_details(path, record):
    """the full record (per-kernel tables, sub-records with their own rooflines, CPU samples) beside the line; under
    gpurun_out/ on a GPU box so that it travels back.  Never fatal."""
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(record, f, default=str)
        return os.path.relpath(path, ROOT)
    except OSError:
        return None


def preflight(torch, dist, rank, world, wire):
    """before any 8.6 GB buffer exists: a host collective, then a neighbour hand-over and a broadcast of a small field
    through the C-ABI communicator on the chosen wire, so that a broken fabric / IPC set-up ends the job here (non-zero
    exit) instead of hanging the timed run"""
    import numpy as np

    from pysdc_amd import lib as Lb
    from pysdc_amd.comm import DeviceComm, torch_host_bcast
    from pysdc_amd.engine import SweepEngine

    t = torch.ones(1)
    dist.all_reduce(t)
    if int(t.item()) != world:
        raise RuntimeError(f'all_reduce returned {t.item()} on {world} ranks')
    if world > 1:
        e = SweepEngine((64, 64, 64), 1, 1)
        comm = DeviceComm(e, world, rank, wire=wire, host_bcast=lambda uid: torch_host_bcast(uid, 0, None, dist))
        e.upload(Lb.SLOT_UEND, 0, np.full(e.nvars, float(rank + 1)))
        e.upload(Lb.SLOT_U, 0, np.zeros(e.nvars))
        comm.handover_post(world)
        comm.handover_complete()
        got = e.download(Lb.SLOT_U, 0)
        want = float(rank) if rank >= 1 else 0.0
        if got.min() != want or got.max() != want:
            raise RuntimeError(f'neighbour message arrived corrupted on rank {rank}: [{got.min()}, {got.max()}], expected {want}')
        comm.bcast(Lb.SLOT_UEND, 0, root=world - 1)
        got = e.download(Lb.SLOT_UEND)
        if got.min() != float(world) or got.max() != float(world):
            raise RuntimeError(f'broadcast arrived corrupted on rank {rank}')
        info = comm.info()
        comm.close()
        e.close()
        return info
    return None


WIRE_MODES = [   # tried in this order until one reproduces the serial emulation on a small grid (validate_wire)
    ('default', {}),
    ('fields on the wire, every hand-over sent', {'PYSDC_AMD_SPECTRAL_WIRE': '0', 'PYSDC_AMD_SKIP_FIRST': '0'}),
    ('direct messages only (no two-hop relay, no host path), fields, every hand-over sent',
     {'PYSDC_AMD_SPECTRAL_WIRE': '0', 'PYSDC_AMD_SKIP_FIRST': '0', 'PYSDC_AMD_RELAY': '0', 'PYSDC_AMD_HOST_SHARE': '0'}),
]


def validate_wire(args, torch, dist, rank, world):
    """Before the 8.6 GB fields exist: the multi-rank run itself on a 64^3 and a 512^2 grid - same controller, same wire, same options
    (spectra on the wire, two-hop relay, skipped first hand-over) - against controller_nonMPI emulating the ranks in this
    process.  The RCCL wire has carried one rank before this job (no multi-GPU box was reachable during development), so a
    mode that does not reproduce the emulation is dropped for the next, more conservative one; the line says which ran.
    A hang ends with the watchdog's JSON error.  Returns (mode name, max relative difference)."""
    import numpy as np

    from pysdc_amd.controller import controller_nonMPI, controller_dist
    from pysdc_amd.problems import heatNd_unforced, allencahn_imex
    from pysdc_amd.sweepers import generic_implicit, imex_1st_order
    from pysdc_amd.transfer import mesh_to_mesh

    M, blocks = 5, 2
    cpar = dict(logger_level=40, mssdc_jac=args.mssdc == 'jacobi')
    if args.workload == 'allencahn':
        cpar['predict_type'] = 'pfasst_burnin'
    # two grids: 64^3 (the bench's dimension; iterates stored) and 512^2 with four sweeps per step - the smallest grid on
    # which a slice recomputes its iterates from the start values it has received (csrc: trail) and puts the last pass of a
    # residual off, i.e. the data flow of the 1024^3 run itself, at 2 MB per field
    checks = []
    if args.workload == 'allencahn':
        # the two-level flow of config 5 (PFASST with burn-in: fine and coarse hand-overs, FAS transfers between them) at
        # 64^3 / 32^3, a fixed number of iterations and one to a tolerance (the `done` chain, controller_MPI.py:574-807)
        plan = [(dict(problem_class=allencahn_imex,
                      problem_params=dict(nvars=[(64, 64, 64), (32, 32, 32)], eps=0.04, radius=0.25, init_type='sphere'),
                      sweeper_class=imex_1st_order, sweeper_params=dict(num_nodes=3, quad_type='RADAU-RIGHT', QI='LU', QE='EE'),
                      level_params=dict(dt=1e-3, restol=restol, nsweeps=1), step_params=dict(maxiter=K),
                      space_transfer_class=mesh_to_mesh, space_transfer_params=dict(iorder=6, rorder=2, periodic=True)), 1e-3)
                for restol, K in ((-1.0, 3), (1e-7, 50))]
    else:
        plan = []
        for nvars, K in (((64, 64, 64), 3), ((512, 512), 4)):
            dt = 1e-3 * (512.0 / nvars[0]) ** 2
            plan.append((dict(problem_class=heatNd_unforced, problem_params=dict(nvars=nvars, nu=0.1, freq=2, order=2),
                              sweeper_class=generic_implicit, sweeper_params=dict(num_nodes=M, quad_type='RADAU-RIGHT', QI='IE'),
                              level_params=dict(dt=dt, restol=-1.0, nsweeps=1), step_params=dict(maxiter=K)), dt))
    for desc, dt in plan:
        serial = controller_nonMPI(world, cpar, desc)
        ref, _ = serial.run(serial.MS[0].levels[0].prob.u_exact(0.0), 0.0, blocks * world * dt)
        checks.append((desc, blocks * world * dt, ref.get()))
        del serial
    last = None
    # (tests: the first so many modes are treated as if they had not reproduced the emulation, so that every fallback - the
    # environment it sets, its restoration, the agreement between the ranks - runs once before a real fabric ever needs it)
    forced = int(os.environ.get('PYSDC_BENCH_FORCE_WIRE_MISMATCH', '0'))
    for idx_mode, (name, env) in enumerate(WIRE_MODES):
        # (the pinned-host share of two-rank hand-overs is for messages of tens of MB: on the small grids of this check it is
        # switched on for every message, so that the full-size run is not its first use)
        env = dict(env, SDC_PIPE_MIN_BYTES='0')
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        ok = 0
        try:
            err = 0.0
            for desc, tend, ref in checks:
                ctrl = controller_dist(dict(cpar, comm_wire=args.wire), desc)
                got, _ = ctrl.run(ctrl.S.levels[0].prob.u_exact(0.0), 0.0, tend)
                err = max(err, float(np.max(np.abs(got.get() - ref)) / np.max(np.abs(ref))))
                ctrl.close() if hasattr(ctrl, 'close') else None
            ok = 1 if err <= 1e-10 else 0
            last = f'{name}: differs from the serial emulation by {err:.2e}'
            if idx_mode < forced:
                ok, last = 0, f'{name}: mismatch forced by PYSDC_BENCH_FORCE_WIRE_MISMATCH (real difference {err:.2e})'
        except Exception as e:  # noqa: BLE001
            err, last = float('nan'), f'{name}: {e!r}'
        t = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)     # every rank must agree, and every rank takes the same branch
        if int(t.item()) == 1:
            if saved.get('SDC_PIPE_MIN_BYTES') is None:      # (the mode's own settings stay for the run; this one was the check's)
                os.environ.pop('SDC_PIPE_MIN_BYTES', None)
            else:
                os.environ['SDC_PIPE_MIN_BYTES'] = saved['SDC_PIPE_MIN_BYTES']
            return name, err
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    raise RuntimeError(f'no wire mode reproduces the serial emulation on rank {rank} ({last})')


def launch_ranks(args, argv):
    """the parent of a multi-GPU run: it never touches a GPU (no torch.cuda call, no HIP library loaded); it starts one
    fresh interpreter per rank, relays rank 0's standard output (the JSON line) and turns any failure - a rank that exits
    non-zero, a job that exceeds --job-timeout - into ONE JSON error line and a non-zero exit code"""
    import socket
    import subprocess
    import tempfile

    with socket.socket() as sk:   # a free port for the rendezvous on the loopback interface
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    def die_with_parent():   # a rank never outlives the launcher (e.g. when the launcher itself is killed on a timeout)
        import ctypes
        import signal

        ctypes.CDLL('libc.so.6').prctl(1, signal.SIGKILL)   # PR_SET_PDEATHSIG

    procs, logs = [], []
    out0_file = tempfile.TemporaryFile(mode='w+')
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', GLOO_SOCKET_IFNAME='lo',
                   PYSDC_BENCH_CHILD='1')
        log = tempfile.TemporaryFile(mode='w+')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0_file if r == 0 else log, stderr=log, text=True,
                                      start_new_session=True, preexec_fn=die_with_parent))
    deadline = time.time() + args.job_timeout
    out0, failed = None, None
    try:
        # all ranks are watched together: the first one that exits non-zero ends the job (the others would only sit in a
        # collective or a stream sync until their own timeouts)
        while failed is None:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f'rank {bad[0][0]} exited with code {bad[0][1]}'
            elif all(c == 0 for c in codes):
                break
            elif time.time() > deadline:
                failed = f'job exceeded --job-timeout {args.job_timeout:.0f} s'
            else:
                time.sleep(0.2)
    finally:
        for p in procs:   # never leave a rank behind that may hold a GPU (each one is its own process group)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
        import glob
        for f in glob.glob(f'/dev/shm/sdcmi.{procs[0].pid}-*'):   # mailboxes of the shared-memory wire a killed rank left behind
            try:
                os.unlink(f)
            except OSError:
                pass
    out0_file.seek(0)
    out0 = out0_file.read()
    line = None
    for ln in (out0 or '').splitlines():
        if ln.startswith('{'):
            line = ln
    if failed is None and line is None:
        failed = 'rank 0 printed no JSON line'
    if failed is None and 'error' in json.loads(line) and 'metric' not in json.loads(line):
        failed = json.loads(line)['error']
    if failed is not None:
        tails = {}
        for r, log in enumerate(logs):
            log.seek(0)
            tails[f'rank{r}'] = log.read()[-1500:]
        print(json.dumps({'error': failed, 'n_gpus': args.gpus, 'rank_output_tails': tails}), flush=True)
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='heat', choices=['heat', 'advdiff', 'vdp', 'allencahn'],
                    help='heat = BASELINE metric (default); advdiff = config 3 (IMEX); vdp = config 4 (ensemble); '
                         'allencahn = config 5 (two-level MLSDC / PFASST)')
    ap.add_argument('--n', type=int, default=None, help='grid points per dimension (heat: 1024 = BASELINE metric; advdiff: 512)')
    ap.add_argument('--ntraj', type=int, default=10_000_000)
    ap.add_argument('--nodes', type=int, default=5)
    ap.add_argument('--sweeps', type=int, default=4)
    ap.add_argument('--qi', default='IE')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true',
                    help='use the one-step-per-rank controller and torch.distributed even with one GPU (self test)')
    ap.add_argument('--no-spectral-reuse', action='store_true',
                    help='transform the gathered fields in every sweep instead of gathering on cached transforms')
    ap.add_argument('--solver-type', default='direct', choices=['direct', 'CG'],
                    help="heat: 'direct' = exact solve in Fourier space (headline); 'CG' = the reference's conjugate "
                         "gradients (rtol 1e-12) on the device, node by node")
    ap.add_argument('--restol', type=float, default=-1.0,
                    help='> 0: iterate to this residual instead of a fixed number of sweeps (maxiter 50); niter is reported')
    ap.add_argument('--eager-fields', action='store_true',
                    help='store F[1..M] and the predictor copies in every sweep / predict even when nothing reads them')
    ap.add_argument('--mssdc', default='jacobi', choices=['jacobi', 'gs'],
                    help='multi-step SDC over the time ranks (--gpus > 1): Jacobi (mssdc_jac=True, the default of the '
                         'reference: all slices sweep concurrently) or Gauss-Seidel (receive, sweep, blocking send)')
    ap.add_argument('--virtual-sweeps', type=int, default=None,
                    help='sdc_set_virtual_sweeps: sweeps per step whose iterate is recomputed from the transform of u0 instead '
                         'of stored (0: every sweep stores its iterate, the round-1 data flow; default: the library\'s)')
    ap.add_argument('--multiplier-table', type=int, default=None,
                    help='sdc_set_multiplier_table: first sweep of a step that takes the node multipliers of a mode pair from '
                         'the table instead of replaying the earlier sweeps (0: never; default: the engine\'s, 8)')
    ap.add_argument('--lazy-predictor-residual', action='store_true',
                    help='put the residual of the state the spread predictor leaves off until somebody reads L.status.residual - '
                         'with restol < 0 nobody does.  Default: computed when compute_residual is called, like the reference '
                         '(K+1 residuals per step)')
    ap.add_argument('--details-file', default=os.path.join(ROOT, 'gpurun_out', 'bench_details.json'),
                    help='where the full record goes (per-kernel tables, sub-records, CPU samples); the line names it')
    ap.add_argument('--cpu-big-n', type=int, default=256, help='cpu_baseline: grid of the one-core sample (0: none)')
    ap.add_argument('--cpu-mid-n', type=int, default=128, help='cpu_baseline: grid of the all-cores sample')
    ap.add_argument('--skip-residual', action='store_true',
                    help="sweeper parameter skip_residual_computation for every stage (the reference's switch for runs with a "
                         'fixed number of sweeps): no residual is computed; NOT the headline configuration')
    ap.add_argument('--mfma', action='store_true',
                    help='vdp: apply the 2x2 Newton block inverses on the matrix cores (v_mfma_f64_4x4x4, two trajectories per '
                         'block) instead of the vector ALUs')
    ap.add_argument('--no-extras', action='store_true',
                    help='only the headline record (default at one GPU: the line also carries short sub-records of the other '
                         'BASELINE configurations and of the eager / restol variants, measured in the same process)')
    ap.add_argument('--p2p-chunk-mb', type=float, default=0.0,
                    help='time-parallel runs: cut the forward message into pieces of this size (0 = one piece)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help="--gpus > 1: 'nccl' = state vectors over RCCL / xGMI (one GPU per rank); 'gloo' = over the "
                         'shared-memory wire of the C-ABI communicator (host staging; needed with --same-device).  The '
                         'torch.distributed process group itself (rendezvous, flags, counts) is gloo in both cases')
    ap.add_argument('--same-device', action='store_true',
                    help='all ranks on GPU 0 (rehearsal of the multi-rank path on a one-GPU box; needs --backend gloo)')
    ap.add_argument('--no-wire-check', action='store_true',
                    help='--gpus > 1: skip the small-grid run over the wire that is compared with the serial emulation')
    ap.add_argument('--job-timeout', type=float, default=1500.0,
                    help='--gpus > 1 started without a launcher: seconds after which the parent ends the job with an error')
    ap.add_argument('--ac-variant', default='sphere', choices=['sphere', 'ref2d'],
                    help="--workload allencahn: 'sphere' = allencahn_imex with a sphere (the bench default); 'ref2d' = the reference's "
                         '2-D allencahn2d_imex extruded along z (its reaction term and circle), whose golden runs pin config 5 '
                         '(tests/golden/runs_cfg5.npz): with --restol 1e-8 --steps 1 --warmup 0 --gpus 8 the run IS the golden '
                         'PFASST case')
    ap.add_argument('--start-plane-file', default=None,
                    help='--workload allencahn: .npy file with a 2-D start value, extruded along z (tests: the start value of a golden run)')
    ap.add_argument('--kernel-events', default=None, choices=['timed', 'separate'],
                    help="HIP events around every launch: 'timed' = inside the timed region (default; the headline's roofline "
                         "is measured there), 'separate' = the timed region runs without them and the same steps run once "
                         'more with them for the kernel table (default for --workload allencahn, whose ~75 short launches '
                         'per iteration pay a tenth of their wall time for the events)')
    ap.add_argument('--no-kernel-events', action='store_true',
                    help='no HIP events around the launches of the timed region (no kernel table, no roofline): what the '
                         'event records themselves cost a launch-bound configuration')
    ap.add_argument('--dump-end-value', default=None,
                    help='rank 0 saves the end value of the timed run to this .npy file (tests)')
    args = ap.parse_args()
    args.wire = 'rccl' if args.backend == 'nccl' else 'shm'
    if args.same_device and args.backend != 'gloo':
        print(json.dumps({'error': '--same-device needs --backend gloo (RCCL refuses two ranks on one device)'}), flush=True)
        raise SystemExit(2)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # not under a launcher: be the launcher - BEFORE anything in this process initialises the GPU
        raise SystemExit(launch_ranks(args, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        print(json.dumps({'error': f'--gpus {args.gpus} but WORLD_SIZE={world} in the environment'}), flush=True)
        raise SystemExit(2)
    try:
        torch.cuda.set_device(0 if args.same_device else local_rank)
    except Exception as e:  # noqa: BLE001  (fewer GPUs than ranks: say so in the one line the caller parses)
        print(json.dumps({'error': f'rank {rank}: no GPU {local_rank} ({e!r}); {torch.cuda.device_count()} visible'}), flush=True)
        raise SystemExit(3)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # one node by contract: rendezvous and the gloo side group (1-byte flags) on the loopback interface - the container's
        # host name may not resolve, which gloo would otherwise try
        os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')   # (RCCL's bootstrap: the unique id carries an address of this interface)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        import datetime

        os.environ.setdefault('SDC_COMM_TIMEOUT', '300')
        # a watchdog of the rank's own (under a launcher that has none): a job stuck in a collective ends with a JSON error
        # line from rank 0 and a non-zero exit code instead of hanging until somebody else's limit
        import threading

        def give_up():
            if rank == 0:
                print(json.dumps({'error': f'rank 0: job exceeded --job-timeout {args.job_timeout:.0f} s', 'n_gpus': world}), flush=True)
            os._exit(4)

        watchdog = threading.Timer(args.job_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            # the process group only carries host data (rendezvous, unique id, 1-byte flags, counts): gloo; the state
            # vectors travel through the C-ABI communicator (RCCL or the shared-memory wire)
            dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            preflight(torch, dist, rank, world, args.wire)
            args.wire_mode, args.wire_check = 'default (not checked)', None
            if world > 1 and not args.no_wire_check:
                def stuck():   # a hand-over that never completes on the small grid: say so in seconds, not after the job's limit
                    if rank == 0:
                        print(json.dumps({'error': f'the {args.wire} wire did not complete a 64^3 multi-rank run within 300 s '
                                                   '(validate_wire)', 'n_gpus': world}), flush=True)
                    os._exit(4)

                guard = threading.Timer(300.0, stuck)
                guard.daemon = True
                guard.start()
                try:
                    args.wire_mode, args.wire_check = validate_wire(args, torch, dist, rank, world)
                finally:
                    guard.cancel()
        except Exception as e:  # noqa: BLE001  fail fast and loudly: no hang, no partial line
            print(json.dumps({'error': f'rank {rank}: preflight of the {args.wire} wire failed: {e!r}'[:1500]}), flush=True)
            raise SystemExit(3)
        if args.p2p_chunk_mb > 0:
            os.environ['PYSDC_AMD_P2P_CHUNK'] = str(int(args.p2p_chunk_mb * (1 << 20) // 8))

    try:
        out = run_workload(args, world, rank, use_dist)
    except MemoryError as e:
        if rank == 0:
            print(json.dumps({'error': str(e), 'n_gpus': world}), flush=True)
        else:
            print(f'rank {rank}: {e}', file=sys.stderr, flush=True)
        os._exit(5)   # (other ranks reach the same verdict on their own GPU; nobody waits in a collective)
    if rank == 0:
        subs = cpu = cpu_detail = cb = None
        headline = world == 1 and args.workload == 'heat' and not args.force_dist
        if headline and not args.no_cpu_baseline:
            try:   # the 64^3 sample alone on the host, then the one-core 256^3 sweep in the background (collected below)
                pr = out['params']
                cb = CpuBaseline(pr['M'], pr['dt'], target_n=pr['n'], nsweeps=args.sweeps,
                                 big_n=args.cpu_big_n, mid_n=args.cpu_mid_n)
            except Exception as e:  # pragma: no cover
                cpu = {'error': repr(e)[:200]}
        sustained = None
        if headline and not args.no_extras and args.n is None:
            subs = extras(args)
            # the headline configuration once more at the END of the process, >= 20 timed steps: clocks under sustained load
            # (the first seconds of a process run ~7 % faster than everything after them) - `value_sustained`
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            try:
                a2 = argparse.Namespace(**{**vars(args), 'steps': max(20, args.steps), 'warmup': 2})
                sustained = run_workload(a2, 1, 0, False, with_stream_reference=False)
            except Exception as e:  # noqa: BLE001  (never takes the line down)
                sustained = {'error': repr(e)[:200]}
        if cb is not None:
            try:
                cpu, cpu_detail = cb.finish()
            except Exception as e:  # pragma: no cover
                cpu = {'error': repr(e)[:200]}
        details = write_details(args.details_file, {'headline': out, 'sub_records': subs, 'cpu_baseline': cpu,
                                                    'cpu_samples': cpu_detail, 'sustained': sustained, 'argv': sys.argv[1:]})
        line = json.dumps(compact_line(out, subs, cpu, details, sustained), allow_nan=False)
        assert len(line) < 6000, len(line)
        sys.stderr.flush()
        print(line, flush=True)
    if use_dist:
        watchdog.cancel()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
