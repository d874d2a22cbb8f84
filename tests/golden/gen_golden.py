"""Generate golden vectors by running the REFERENCE (pySDC, /root/reference) in the build container.

Usage (build container only; the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 \
      PYTHONPATH=/root/repo/oracle/qmat_shim:/root/repo:/root/reference \
      python /root/repo/tests/golden/gen_golden.py

That regenerates the sweep / run / multi-level / Allen-Cahn / dirichlet files.  The files added later have their
own entry points, switched on by environment variables (or call the function after ``runpy.run_path``):
GOLDEN_FFT=1 fft_transfer_main (transfer_fft.npz, runs_ac_fft.npz), GOLDEN_RELAY=1 relay_main (runs_relay.npz),
GOLDEN_CG=1 cg_main (sweeps_cg.npz), GOLDEN_VDPJ=1 vdp_jacobian_main (vdp_jacobian.npz), GOLDEN_DML=1
dirichlet_ml_main (transfer_dirichlet.npz, runs_ml_dirichlet.npz), GOLDEN_GUESS=1 guess_main (sweeps_guess.npz), GOLDEN_SKIP=1 skip_main (runs_skip.npz), GOLDEN_RELAY8=1 relay8_main (runs_relay8.npz),
GOLDEN_BIG3D=1 big3d_main (sweeps_big3d.npz), GOLDEN_CFG5=1 cfg5_main (runs_cfg5.npz), GOLDEN_ML8=1 ml8_main (runs_ml8.npz), GOLDEN_DND=1 dirichlet_nd_main (sweeps_dirichlet_nd.npz, runs_dirichlet_nd.npz), GOLDEN_NSW2=1 nsweeps2_main (runs_nsweeps2.npz), GOLDEN_PIN1024=1 pin1024_main (sweeps_pin1024.npz, 20 minutes), GOLDEN_GMRES=1 gmres_main (sweeps_gmres.npz), GOLDEN_DHO=1 dirichlet_ho_main (sweeps_dirichlet_ho.npz, runs_dirichlet_ho.npz, dirichlet_ho_matrices.npz), GOLDEN_BC=1 boundary_main (boundary_matrices.npz, sweeps_neumann.npz, runs_neumann.npz), GOLDEN_PIN512=1 pin512_main (sweeps_pin512.npz), GOLDEN_R3=1 radix3_main (sweeps_radix3.npz, runs_radix3.npz), GOLDEN_R5=1 radix5_main (sweeps_radix5.npz, runs_radix5.npz), GOLDEN_T3D=1 transfer3d_main (transfer3d.npz), GOLDEN_AD1D=1 ad1d_main (sweeps_ad1d.npz, runs_ad1d.npz);
GOLDEN_ML=0 skips the multi-level block.

``qmat`` (third-party, absent here) is replaced by oracle/qmat_shim, which forwards to
pysdc_amd.coeffs; the coefficient matrices that were actually used are stored with every case.
Outputs: tests/golden/*.npz (inputs + expected outputs only - data, no reference source).
"""
import json
import os

import numpy as np

from pySDC.core.step import Step
from pySDC.implementations.problem_classes.HeatEquation_ND_FD import heatNd_unforced, heatNd_forced
from pySDC.implementations.problem_classes.AdvectionEquation_ND_FD import advectionNd
from pySDC.implementations.problem_classes.Van_der_Pol_implicit import vanderpol
from pySDC.implementations.sweeper_classes.generic_implicit import generic_implicit
from pySDC.implementations.sweeper_classes.imex_1st_order import imex_1st_order
from pySDC.implementations.controller_classes.controller_nonMPI import controller_nonMPI
from pySDC.implementations.datatype_classes.mesh import mesh, imex_mesh
from pySDC.core.problem import Problem
from pySDC.helpers.stats_helper import get_sorted

OUT = os.path.dirname(os.path.abspath(__file__))


class advdiff_composite(Problem):
    """test-only composite: impl part / solve delegate to a reference
    heatNd_unforced, expl part to a reference advectionNd."""

    dtype_u = mesh
    dtype_f = imex_mesh

    def __init__(self, nvars=64, nu=0.02, c=1.0, freq=2, order=2, stencil_type='center', bc='periodic',
                 solver_type='direct', lintol=1e-12, liniter=10000):
        self.diff = heatNd_unforced(nvars=nvars, nu=nu, freq=freq, order=order, bc=bc, solver_type=solver_type,
                                    lintol=lintol, liniter=liniter)
        self.adv = advectionNd(nvars=nvars, c=c, freq=freq, order=order, stencil_type=stencil_type, bc=bc)
        super().__init__(init=self.diff.init)
        self.nvars = self.diff.nvars
        self.work_counters = self.diff.work_counters

    def eval_f(self, u, t):
        f = self.dtype_f(self.init)
        f.impl[:] = self.diff.eval_f(u, t)
        f.expl[:] = self.adv.eval_f(u, t)
        return f

    def solve_system(self, rhs, factor, u0, t):
        return self.diff.solve_system(rhs, factor, u0, t)

    def u_exact(self, t):
        return self.diff.u_exact(0.0)


PROBS = {'heat_unforced': heatNd_unforced, 'heat_forced': heatNd_forced, 'advection': advectionNd,
         'advdiff': advdiff_composite, 'vanderpol': vanderpol}
SWEEPERS = {'generic_implicit': generic_implicit, 'imex_1st_order': imex_1st_order}


def coll_dict(sw):
    d = dict(nodes=sw.coll.nodes, weights=sw.coll.weights, Qmat=sw.coll.Qmat, QI=sw.QI,
             right_is_node=np.array(sw.coll.right_is_node), left_is_node=np.array(sw.coll.left_is_node))
    if hasattr(sw, 'QE'):
        d['QE'] = sw.QE
    return d


def fstack(f):
    return np.asarray(f) if not hasattr(f, 'impl') else np.stack([np.asarray(f.impl), np.asarray(f.expl)])


def sweep_case(name, prob, prob_params, sweeper, sweeper_params, dt, t0=0.1, nsweeps=3, seed=0, tau_seed=None,
               u0_kind='randn'):
    """one level: predict, then nsweeps x (update_nodes, compute_residual for all four residual types,
    compute_end_point for both do_coll_update values)."""
    desc = dict(problem_class=PROBS[prob], problem_params=dict(prob_params), sweeper_class=SWEEPERS[sweeper],
                sweeper_params=dict(sweeper_params), level_params=dict(dt=dt), step_params=dict(maxiter=50))
    S = Step(desc)
    L = S.levels[0]
    P = L.prob
    M = L.sweep.coll.num_nodes
    rng = np.random.default_rng(seed)
    if prob == 'vanderpol':
        u0 = np.asarray(prob_params.get('u0', (2.0, 0.0)), dtype=float)
    elif u0_kind == 'randn':
        u0 = rng.standard_normal(P.init[0])
    else:
        u0 = np.asarray(P.u_exact(0.0)) + 1e-3 * rng.standard_normal(P.init[0])
    L.status.time = t0
    L.u[0] = P.dtype_u(P.init)
    L.u[0][:] = u0
    out = {'u0': np.array(u0)}
    if tau_seed is not None:
        trng = np.random.default_rng(tau_seed)
        for m in range(M):
            L.tau[m] = P.dtype_u(P.init)
            L.tau[m][:] = 1e-2 * trng.standard_normal(P.init[0])
        out['tau'] = np.stack([np.asarray(t) for t in L.tau])
    L.sweep.predict()

    def snap(tag):
        out[f'{tag}_u'] = np.stack([np.asarray(x) for x in L.u])
        out[f'{tag}_f'] = np.stack([fstack(x) for x in L.f])
        for rt in ('full_abs', 'last_abs', 'full_rel', 'last_rel'):
            L.params.residual_type = rt
            L.sweep.compute_residual()
            out[f'{tag}_res_{rt}'] = np.array(L.status.residual)
        L.params.residual_type = 'full_abs'
        for dcu in (False, True):
            L.sweep.params.do_coll_update = dcu
            L.sweep.compute_end_point()
            out[f'{tag}_uend_{int(dcu)}'] = np.asarray(L.uend).copy()
        L.sweep.params.do_coll_update = False

    snap('k0')
    wc = {k: [] for k in P.work_counters}
    for k in range(1, nsweeps + 1):
        L.sweep.updateVariableCoeffs(k)
        if getattr(L.sweep.genQI, 'isKDependent', lambda: False)():
            out[f'k{k}_QI'] = L.sweep.QI.copy()
        L.sweep.update_nodes()
        snap(f'k{k}')
        for key in wc:
            wc[key].append(P.work_counters[key].niter)
    for key in wc:
        out[f'work_{key}'] = np.array(wc[key])
    for k_, v in coll_dict(L.sweep).items():
        out[f'coll_{k_}'] = np.asarray(v)
    meta = dict(name=name, prob=prob, prob_params=prob_params, sweeper=sweeper, sweeper_params=sweeper_params,
                dt=dt, t0=t0, nsweeps=nsweeps, has_tau=tau_seed is not None)
    out['meta'] = np.array(json.dumps(meta))
    return out


def run_case(name, prob, prob_params, sweeper, sweeper_params, level_params, maxiter, t0, Tend, num_procs=1,
             controller_params=None, seed=None):
    desc = dict(problem_class=PROBS[prob], problem_params=dict(prob_params), sweeper_class=SWEEPERS[sweeper],
                sweeper_params=dict(sweeper_params), level_params=dict(level_params),
                step_params=dict(maxiter=maxiter))
    cp = dict(logger_level=40)
    cp.update(controller_params or {})
    C = controller_nonMPI(num_procs, cp, desc)
    P = C.MS[0].levels[0].prob
    u0 = P.u_exact(t0)
    if seed is not None:
        u0 = u0 + 1e-3 * np.random.default_rng(seed).standard_normal(np.shape(u0))
        u0 = P.dtype_u(P.init) + u0
    uend, stats = C.run(u0, t0, Tend)
    out = {'u0': np.asarray(u0).copy(), 'uend': np.asarray(uend).copy()}
    niter = get_sorted(stats, type='niter', sortby='time')
    out['niter_t'] = np.array([t for t, _ in niter])
    out['niter'] = np.array([v for _, v in niter])
    res = get_sorted(stats, type='residual_post_iteration', sortby='time')
    out['res_t'] = np.array([t for t, _ in res])
    out['res'] = np.array([v for _, v in res])
    for k_, v in coll_dict(C.MS[0].levels[0].sweep).items():
        out[f'coll_{k_}'] = np.asarray(v)
    if seed is None:
        out['err'] = np.array(abs(uend - P.u_exact(Tend)))
    meta = dict(name=name, prob=prob, prob_params=prob_params, sweeper=sweeper, sweeper_params=sweeper_params,
                level_params=level_params, maxiter=maxiter, t0=t0, Tend=Tend, num_procs=num_procs,
                controller_params=controller_params or {}, seed=seed)
    out['meta'] = np.array(json.dumps(meta))
    return out


def save(fname, cases):
    flat = {}
    for c in cases:
        name = json.loads(str(c['meta']))['name']
        for k, v in c.items():
            flat[f'{name}/{k}'] = v
    np.savez_compressed(os.path.join(OUT, fname), **flat)
    print(fname, len(cases), 'cases', os.path.getsize(os.path.join(OUT, fname)) // 1024, 'KiB')


def main():
    RR = dict(quad_type='RADAU-RIGHT')
    # ---- G2: single sweeps, heat, generic_implicit ----
    cases = []
    for nv, tag in ((64, '1d'), ((16, 16), '2d'), ((8, 8, 8), '3d')):
        for order in (2, 4):
            for M, QI, dt in ((3, 'IE', 1e-3), (5, 'LU', 1e-1), (5, 'IE', 1e-1)):
                cases.append(sweep_case(f'heat{tag}_o{order}_M{M}_{QI}_dt{dt:g}', 'heat_unforced',
                                        dict(nvars=nv, nu=0.1, freq=2, order=order, bc='periodic'),
                                        'generic_implicit', dict(num_nodes=M, QI=QI, **RR), dt))
    cases.append(sweep_case('heat3d_tau', 'heat_unforced', dict(nvars=(8, 8, 8), nu=0.1, freq=2, bc='periodic'),
                            'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 1e-2, tau_seed=7))
    cases.append(sweep_case('heat1d_tau_LU', 'heat_unforced', dict(nvars=64, nu=0.1, freq=2, bc='periodic'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 1e-2, tau_seed=8))
    cases.append(sweep_case('heat2d_gauss', 'heat_unforced', dict(nvars=(16, 16), nu=0.1, freq=2, bc='periodic'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', quad_type='GAUSS'), 1e-2))
    cases.append(sweep_case('heat2d_lobatto', 'heat_unforced', dict(nvars=(16, 16), nu=0.1, freq=2, bc='periodic'),
                            'generic_implicit', dict(num_nodes=4, QI='IE', quad_type='LOBATTO'), 1e-2))
    for QI in ('MIN-SR-S', 'MIN-SR-NS', 'IEpar', 'MIN-SR-FLEX', 'Qpar', 'PIC'):
        cases.append(sweep_case(f'heat3d_diag_{QI}', 'heat_unforced',
                                dict(nvars=(8, 8, 8), nu=0.1, freq=2, bc='periodic'),
                                'generic_implicit', dict(num_nodes=5, QI=QI, **RR), 1e-2, nsweeps=6 if 'FLEX' in QI else 3))
    cases.append(sweep_case('heat3d_cg', 'heat_unforced',
                            dict(nvars=(8, 8, 8), nu=0.1, freq=2, bc='periodic', solver_type='CG', lintol=1e-12),
                            'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 1e-2))
    cases.append(sweep_case('heat1d_dirichlet', 'heat_unforced',
                            dict(nvars=63, nu=0.1, freq=1, bc='dirichlet-zero'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 1e-2))
    for ig in ('zero', 'copy'):
        cases.append(sweep_case(f'heat1d_guess_{ig}', 'heat_unforced', dict(nvars=64, nu=0.1, freq=2, bc='periodic'),
                                'generic_implicit', dict(num_nodes=3, QI='IE', initial_guess=ig, **RR), 1e-2))
    save('sweeps_heat.npz', cases)

    # ---- G2 IMEX ----
    cases = []
    for nv, tag in ((64, '1d'), ((16, 16), '2d'), ((8, 8, 8), '3d')):
        cases.append(sweep_case(f'forced{tag}_M3', 'heat_forced', dict(nvars=nv, nu=0.1, freq=2, bc='periodic'),
                                'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 1e-2, u0_kind='exact'))
        cases.append(sweep_case(f'advdiff{tag}_M5', 'advdiff', dict(nvars=nv, nu=0.02, c=1.0, freq=2, order=2),
                                'imex_1st_order', dict(num_nodes=5, QI='IE', QE='EE', **RR), 1e-3))
        cases.append(sweep_case(f'advdiff{tag}_M3_LU_PIC', 'advdiff', dict(nvars=nv, nu=0.02, c=1.0, freq=2, order=4),
                                'imex_1st_order', dict(num_nodes=3, QI='LU', QE='PIC', **RR), 1e-2))
    cases.append(sweep_case('forced1d_dirichlet', 'heat_forced', dict(nvars=63, nu=0.1, freq=1, bc='dirichlet-zero'),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 1e-2, u0_kind='exact'))
    cases.append(sweep_case('advdiff3d_tau', 'advdiff', dict(nvars=(8, 8, 8), nu=0.02, c=1.0, freq=2, order=2),
                            'imex_1st_order', dict(num_nodes=5, QI='IE', QE='EE', **RR), 1e-3, tau_seed=3))
    cases.append(sweep_case('advdiff1d_upwind', 'advdiff',
                            dict(nvars=64, nu=0.02, c=1.0, freq=2, order=3, stencil_type='upwind'),
                            'imex_1st_order', dict(num_nodes=3, QI='IE', QE='EE', **RR), 1e-3))
    save('sweeps_imex.npz', cases)

    # ---- implicit advection (complex symbol) ----
    cases = []
    cases.append(sweep_case('adv1d_impl', 'advection', dict(nvars=64, c=1.0, freq=2, order=2, bc='periodic'),
                            'generic_implicit', dict(num_nodes=3, QI='IE', **RR), 1e-2))
    cases.append(sweep_case('adv2d_impl_up', 'advection',
                            dict(nvars=(16, 16), c=0.5, freq=2, order=3, stencil_type='upwind', bc='periodic'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 1e-2))
    save('sweeps_adv.npz', cases)

    # ---- G4 van der Pol ----
    cases = []
    cases.append(sweep_case('vdp_mu5', 'vanderpol', dict(u0=(2.0, 0.0), mu=5.0, newton_tol=1e-9),
                            'generic_implicit', dict(num_nodes=5, QI='LU', **RR), 0.05, t0=0.0, nsweeps=4))
    rng = np.random.default_rng(0)
    for i in range(8):
        u0 = tuple(float(x) for x in rng.uniform(-2, 2, 2))
        cases.append(sweep_case(f'vdp_rand{i}', 'vanderpol', dict(u0=u0, mu=5.0, newton_tol=1e-9),
                                'generic_implicit', dict(num_nodes=5, QI='LU', **RR), 0.05, t0=0.0, nsweeps=4))
    save('sweeps_vdp.npz', cases)

    # ---- G3 runs ----
    cases = []
    cfg1 = dict(prob='heat_unforced', prob_params=dict(nvars=1024, nu=0.1, freq=2, bc='periodic'),
                sweeper='generic_implicit', sweeper_params=dict(num_nodes=3, QI='IE', **RR),
                level_params=dict(dt=0.01, restol=1e-10), maxiter=50, t0=0.0, Tend=0.1)
    cases.append(run_case('config1', **cfg1))
    small = dict(prob='heat_unforced', prob_params=dict(nvars=(16, 16), nu=0.1, freq=2, bc='periodic'),
                 sweeper='generic_implicit', sweeper_params=dict(num_nodes=3, QI='LU', **RR),
                 level_params=dict(dt=0.02, restol=1e-9), maxiter=50, t0=0.0, Tend=0.16)
    for P_ in (1, 2, 4):
        for jac in (True, False):
            if P_ == 1 and not jac:
                continue
            cases.append(run_case(f'mssdc_P{P_}_{"jac" if jac else "gs"}', num_procs=P_,
                                  controller_params=dict(mssdc_jac=jac), seed=5, **small))
    fixedk = dict(prob='heat_unforced', prob_params=dict(nvars=(8, 8, 8), nu=0.1, freq=2, bc='periodic'),
                  sweeper='generic_implicit', sweeper_params=dict(num_nodes=5, QI='IE', **RR),
                  level_params=dict(dt=1e-3, restol=-1), maxiter=4, t0=0.0, Tend=3e-3)
    cases.append(run_case('fixedK_3d', seed=0, **fixedk))
    cases.append(run_case('fixedK_3d_P2', seed=0, num_procs=2, **fixedk))
    imex = dict(prob='heat_forced', prob_params=dict(nvars=(16, 16), nu=0.1, freq=2, bc='periodic'),
                sweeper='imex_1st_order', sweeper_params=dict(num_nodes=3, QI='LU', QE='EE', **RR),
                level_params=dict(dt=0.05, restol=1e-9, nsweeps=2), maxiter=50, t0=0.0, Tend=0.2)
    cases.append(run_case('forced2d_run', **imex))
    cases.append(run_case('forced2d_run_P2', num_procs=2, **imex))
    save('runs.npz', cases)


if __name__ == '__main__':
    main()


# ------------------------------------------------------------------------------------------------------
# multi-level goldens (space transfer, FAS restriction / prolongation, MLSDC and PFASST runs)
# ------------------------------------------------------------------------------------------------------
def transfer_cases():
    from pySDC.implementations.transfer_classes.TransferMesh import mesh_to_mesh

    out = {}
    for tag, nf, nc, io, ro in (('1d', 32, 16, 2, 2), ('2d', (16, 16), (8, 8), 6, 2), ('3d', (8, 8, 8), (4, 4, 4), 4, 2),
                                ('2d_inj', (16, 16), (8, 8), 2, 0), ('1d_66', 64, 32, 6, 6)):
        pf = heatNd_unforced(nvars=nf, nu=0.1, freq=2, bc='periodic')
        pc = heatNd_unforced(nvars=nc, nu=0.1, freq=2, bc='periodic')
        T = mesh_to_mesh(pf, pc, dict(iorder=io, rorder=ro, periodic=True))
        rng = np.random.default_rng(4)
        F = pf.dtype_u(pf.init)
        F[:] = rng.standard_normal(pf.init[0])
        G = pc.dtype_u(pc.init)
        G[:] = rng.standard_normal(pc.init[0])
        out[f'transfer_{tag}/fine'] = np.asarray(F).copy()
        out[f'transfer_{tag}/coarse'] = np.asarray(G).copy()
        out[f'transfer_{tag}/restricted'] = np.asarray(T.restrict(F)).copy()
        out[f'transfer_{tag}/prolonged'] = np.asarray(T.prolong(G)).copy()
        out[f'transfer_{tag}/meta'] = np.array(json.dumps(dict(name=f'transfer_{tag}', nf=nf, nc=nc, iorder=io, rorder=ro)))
    np.savez_compressed(os.path.join(OUT, 'transfer.npz'), **out)
    print('transfer.npz', os.path.getsize(os.path.join(OUT, 'transfer.npz')) // 1024, 'KiB')


def ml_description(prob, pp_lists, sweeper, sw, lp, maxiter, io=6, ro=2):
    from pySDC.implementations.transfer_classes.TransferMesh import mesh_to_mesh

    return dict(problem_class=PROBS[prob], problem_params=pp_lists, sweeper_class=SWEEPERS[sweeper],
                sweeper_params=sw, level_params=lp, step_params=dict(maxiter=maxiter),
                space_transfer_class=mesh_to_mesh, space_transfer_params=dict(rorder=ro, iorder=io, periodic=True))


def fas_case(name, prob, pp, sweeper, sw, dt, seed=0):
    """predict on the fine level, restrict, one coarse sweep, prolong: all node values / tau captured."""
    desc = ml_description(prob, pp, sweeper, sw, dict(dt=dt), 10)
    S = Step(desc)
    F, G = S.levels
    rng = np.random.default_rng(seed)
    for L in S.levels:
        L.status.time = 0.1
    u0 = np.asarray(F.prob.u_exact(0.0)) + 1e-2 * rng.standard_normal(F.prob.init[0])
    F.u[0] = F.prob.dtype_u(F.prob.init)
    F.u[0][:] = u0
    F.sweep.predict()
    F.sweep.update_nodes()
    out = {'u0': u0}

    def snap(tag):
        out[f'{tag}_fu'] = np.stack([np.asarray(x) for x in F.u])
        out[f'{tag}_ff'] = np.stack([fstack(x) for x in F.f])
        if G.u[0] is not None:
            out[f'{tag}_gu'] = np.stack([np.asarray(x) for x in G.u])
            out[f'{tag}_gf'] = np.stack([fstack(x) for x in G.f])
            out[f'{tag}_gtau'] = np.stack([np.asarray(x) for x in G.tau])

    snap('a')
    S.transfer(source=F, target=G)
    snap('b')
    G.sweep.update_nodes()
    G.sweep.compute_residual()
    out['c_gres'] = np.array(G.status.residual)
    snap('c')
    S.transfer(source=G, target=F)
    snap('d')
    out['coll_f_QI'] = F.sweep.QI
    out['coll_g_QI'] = G.sweep.QI
    if hasattr(F.sweep, 'QE'):
        out['coll_f_QE'] = F.sweep.QE
        out['coll_g_QE'] = G.sweep.QE
    out['meta'] = np.array(json.dumps(dict(name=name, prob=prob, prob_params=pp, sweeper=sweeper, sweeper_params=sw,
                                           dt=dt, t0=0.1)))
    return out


def ml_run_case(name, prob, pp, sweeper, sw, lp, maxiter, t0, Tend, num_procs, controller_params=None, seed=5,
                io=6, ro=2):
    desc = ml_description(prob, pp, sweeper, sw, lp, maxiter, io, ro)
    cp = dict(logger_level=40)
    cp.update(controller_params or {})
    C = controller_nonMPI(num_procs, cp, desc)
    P = C.MS[0].levels[0].prob
    u0 = P.u_exact(t0)
    if seed is not None:
        u0 = u0 + 1e-3 * np.random.default_rng(seed).standard_normal(P.init[0])
    u0 = P.dtype_u(P.init) + u0
    uend, stats = C.run(u0, t0, Tend)
    out = {'u0': np.asarray(u0).copy(), 'uend': np.asarray(uend).copy()}
    niter = get_sorted(stats, type='niter', sortby='time')
    out['niter_t'] = np.array([t for t, _ in niter])
    out['niter'] = np.array([v for _, v in niter])
    res = get_sorted(stats, type='residual_post_iteration', sortby='time')
    out['res'] = np.array([v for _, v in res])
    out['meta'] = np.array(json.dumps(dict(name=name, prob=prob, prob_params=pp, sweeper=sweeper, sweeper_params=sw,
                                           level_params=lp, maxiter=maxiter, t0=t0, Tend=Tend, num_procs=num_procs,
                                           controller_params=controller_params or {}, iorder=io, rorder=ro)))
    return out


def multilevel_main():
    transfer_cases()
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    heat2 = dict(nvars=[(16, 16), (8, 8)], nu=0.1, freq=2, bc='periodic')
    cases.append(fas_case('fas_heat2d_M3', 'heat_unforced', heat2, 'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 0.02))
    cases.append(fas_case('fas_heat2d_M53', 'heat_unforced', heat2, 'generic_implicit',
                          dict(num_nodes=[5, 3], QI='IE', **RR), 0.02))
    cases.append(fas_case('fas_heat1d', 'heat_unforced', dict(nvars=[64, 32], nu=0.1, freq=2, bc='periodic'),
                          'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 0.01))
    cases.append(fas_case('fas_forced2d', 'heat_forced', heat2, 'imex_1st_order',
                          dict(num_nodes=3, QI='LU', QE='EE', **RR), 0.02))
    save('fas.npz', cases)

    cases = []
    base = dict(prob='heat_unforced', pp=heat2, sweeper='generic_implicit', sw=dict(num_nodes=3, QI='LU', **RR),
                lp=dict(dt=0.02, restol=1e-9), maxiter=50, t0=0.0, Tend=0.16)
    cases.append(ml_run_case('mlsdc_heat2d', num_procs=1, **base))
    for P_ in (2, 4):
        cases.append(ml_run_case(f'pfasst_heat2d_P{P_}', num_procs=P_,
                                 controller_params=dict(predict_type='pfasst_burnin'), **base))
    cases.append(ml_run_case('pfasst_heat2d_P2_nopred', num_procs=2, **base))
    cases.append(ml_run_case('pfasst_heat2d_P4_all_to_done', num_procs=4,
                             controller_params=dict(predict_type='pfasst_burnin', all_to_done=True), **base))
    b53 = dict(base)
    b53['sw'] = dict(num_nodes=[5, 3], QI='IE', **RR)
    cases.append(ml_run_case('mlsdc_heat2d_M53', num_procs=1, **b53))
    cases.append(ml_run_case('pfasst_heat2d_M53_P2', num_procs=2, controller_params=dict(predict_type='pfasst_burnin'),
                             **b53))
    forced = dict(prob='heat_forced', pp=heat2, sweeper='imex_1st_order', sw=dict(num_nodes=3, QI='LU', QE='EE', **RR),
                  lp=dict(dt=0.05, restol=1e-9), maxiter=50, t0=0.0, Tend=0.2)
    cases.append(ml_run_case('mlsdc_forced2d', num_procs=1, **forced))
    cases.append(ml_run_case('pfasst_forced2d_P2', num_procs=2, controller_params=dict(predict_type='pfasst_burnin'),
                             **forced))
    h3 = dict(prob='heat_unforced', pp=dict(nvars=[(8, 8, 8), (4, 4, 4)], nu=0.1, freq=2, bc='periodic'),
              sweeper='generic_implicit', sw=dict(num_nodes=3, QI='LU', **RR), lp=dict(dt=0.01, restol=1e-9),
              maxiter=50, t0=0.0, Tend=0.04)
    cases.append(ml_run_case('pfasst_heat3d_P2', num_procs=2, controller_params=dict(predict_type='pfasst_burnin'),
                             io=4, **h3))
    save('runs_ml.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_ML', '1') == '1':
    multilevel_main()


def allencahn_main():
    """G6: allencahn2d_imex sweeps and two-level runs (mesh_to_mesh periodic transfer)."""
    from pySDC.implementations.problem_classes.AllenCahn_2D_FFT import allencahn2d_imex

    PROBS['allencahn2d'] = allencahn2d_imex
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    cases.append(sweep_case('ac2d_M3', 'allencahn2d', dict(nvars=(32, 32), nu=2, eps=0.04, radius=0.25),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 1e-3, t0=0.0, u0_kind='exact'))
    cases.append(sweep_case('ac2d_M5_IE', 'allencahn2d', dict(nvars=(64, 64), nu=2, eps=0.08, radius=0.25),
                            'imex_1st_order', dict(num_nodes=5, QI='IE', QE='EE', **RR), 5e-4, t0=0.0, u0_kind='exact'))
    save('sweeps_ac.npz', cases)
    cases = []
    pp = dict(nvars=[(32, 32), (16, 16)], nu=2, eps=0.04, radius=0.25)
    base = dict(prob='allencahn2d', pp=pp, sweeper='imex_1st_order', sw=dict(num_nodes=3, QI='LU', QE='EE', **RR),
                lp=dict(dt=1e-3, restol=1e-8), maxiter=50, t0=0.0, Tend=4e-3, seed=None)
    cases.append(ml_run_case('ac2d_mlsdc', num_procs=1, **base))
    cases.append(ml_run_case('ac2d_pfasst_P2', num_procs=2, controller_params=dict(predict_type='pfasst_burnin'), **base))
    cases.append(ml_run_case('ac2d_pfasst_P4', num_procs=4, controller_params=dict(predict_type='pfasst_burnin'), **base))
    sl = dict(base)
    sl['pp'] = dict(nvars=(32, 32), nu=2, eps=0.04, radius=0.25)
    save('runs_ac.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_ML', '1') == '1':
    allencahn_main()


def dirichlet_main():
    """tutorial-style 1-D Dirichlet runs (step_2/C, step_3: heat 1-D, dirichlet-zero)."""
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    cases.append(run_case('dirichlet_heat1d_1023', 'heat_unforced', dict(nvars=1023, nu=0.1, freq=4, bc='dirichlet-zero'),
                          'generic_implicit', dict(num_nodes=3, QI='LU', **RR), dict(dt=0.1, restol=1e-10), 50, 0.1, 0.5))
    cases.append(run_case('dirichlet_forced1d_127', 'heat_forced', dict(nvars=127, nu=0.1, freq=2, bc='dirichlet-zero'),
                          'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), dict(dt=0.05, restol=1e-10),
                          50, 0.0, 0.2))
    cases.append(run_case('dirichlet_heat1d_P2', 'heat_unforced', dict(nvars=255, nu=0.1, freq=2, bc='dirichlet-zero'),
                          'generic_implicit', dict(num_nodes=5, QI='IE', **RR), dict(dt=0.02, restol=1e-9), 50, 0.0,
                          0.08, num_procs=2, seed=3))
    save('runs_dirichlet.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_ML', '1') == '1':
    dirichlet_main()


def fft_transfer_main():
    """mesh_to_mesh_fft (1-D) / mesh_to_mesh_fft2d per-operator vectors and the G6 two-level Allen-Cahn run with
    the FFT transfer."""
    from types import SimpleNamespace
    from pySDC.implementations.datatype_classes.mesh import mesh
    from pySDC.implementations.problem_classes.AllenCahn_2D_FFT import allencahn2d_imex
    from pySDC.implementations.transfer_classes.TransferMesh_FFT import mesh_to_mesh_fft
    from pySDC.implementations.transfer_classes.TransferMesh_FFT2D import mesh_to_mesh_fft2d

    out = {}
    rng = np.random.default_rng(9)
    for tag, nf, nc in (('fft1d_32_16', 32, 16), ('fft1d_64_16', 64, 16), ('fft1d_256_128', 256, 128)):
        pf = SimpleNamespace(nvars=nf, init=(nf, None, np.dtype('float64')))
        pc = SimpleNamespace(nvars=nc, init=(nc, None, np.dtype('float64')))
        T = mesh_to_mesh_fft(pf, pc, {})
        F = mesh(pf.init)
        F[:] = rng.standard_normal(nf)
        G = mesh(pc.init)
        G[:] = rng.standard_normal(nc)
        out[f'{tag}/fine'], out[f'{tag}/coarse'] = np.asarray(F).copy(), np.asarray(G).copy()
        out[f'{tag}/restricted'] = np.asarray(T.restrict(F)).copy()
        out[f'{tag}/prolonged'] = np.asarray(T.prolong(G)).copy()
        out[f'{tag}/meta'] = np.array(json.dumps(dict(name=tag, kind='fft1d', nf=nf, nc=nc)))
    for tag, nf, nc in (('fft2d_32_16', 32, 16), ('fft2d_64_16', 64, 16), ('fft2d_128_64', 128, 64)):
        pf = SimpleNamespace(nvars=(nf, nf), init=((nf, nf), None, np.dtype('float64')))
        pc = SimpleNamespace(nvars=(nc, nc), init=((nc, nc), None, np.dtype('float64')))
        T = mesh_to_mesh_fft2d(pf, pc, {})
        F = mesh(pf.init)
        F[:] = rng.standard_normal((nf, nf))
        G = mesh(pc.init)
        G[:] = rng.standard_normal((nc, nc))
        out[f'{tag}/fine'], out[f'{tag}/coarse'] = np.asarray(F).copy(), np.asarray(G).copy()
        out[f'{tag}/restricted'] = np.asarray(T.restrict(F)).copy()
        out[f'{tag}/prolonged'] = np.asarray(T.prolong(G)).copy()
        out[f'{tag}/meta'] = np.array(json.dumps(dict(name=tag, kind='fft2d', nf=nf, nc=nc)))
    np.savez_compressed(os.path.join(OUT, 'transfer_fft.npz'), **out)
    print('transfer_fft.npz', os.path.getsize(os.path.join(OUT, 'transfer_fft.npz')) // 1024, 'KiB')

    PROBS['allencahn2d'] = allencahn2d_imex
    RR = dict(quad_type='RADAU-RIGHT')
    pp = dict(nvars=[(32, 32), (16, 16)], nu=2, eps=0.04, radius=0.25)
    cases = []
    for name, P_, cp in (('ac2d_fft2d_mlsdc', 1, None), ('ac2d_fft2d_pfasst_P2', 2, dict(predict_type='pfasst_burnin'))):
        desc = ml_description('allencahn2d', pp, 'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR),
                              dict(dt=1e-3, restol=1e-8), 50)
        desc['space_transfer_class'] = mesh_to_mesh_fft2d
        desc['space_transfer_params'] = {}
        cpar = dict(logger_level=40)
        cpar.update(cp or {})
        C = controller_nonMPI(P_, cpar, desc)
        P = C.MS[0].levels[0].prob
        u0 = P.dtype_u(P.init) + P.u_exact(0.0)
        uend, stats = C.run(u0, 0.0, 4e-3)
        o = {'u0': np.asarray(u0).copy(), 'uend': np.asarray(uend).copy()}
        niter = get_sorted(stats, type='niter', sortby='time')
        o['niter_t'] = np.array([t for t, _ in niter])
        o['niter'] = np.array([v for _, v in niter])
        o['res'] = np.array([v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')])
        o['meta'] = np.array(json.dumps(dict(name=name, prob='allencahn2d', prob_params=pp, sweeper='imex_1st_order',
                                             sweeper_params=dict(num_nodes=3, QI='LU', QE='EE', **RR),
                                             level_params=dict(dt=1e-3, restol=1e-8), maxiter=50, t0=0.0, Tend=4e-3,
                                             num_procs=P_, controller_params=cp or {}, transfer='mesh_to_mesh_fft2d')))
        cases.append(o)
    save('runs_ac_fft.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_FFT', '0') == '1':
    fft_transfer_main()


def relay_main():
    """multi-step SDC runs whose ranks all iterate the same number of times (fixed K / all_to_done): the setting
    in which controller_dist forwards end values over two hops (tests/test_dist_gloo.py, 3 and 4 ranks)."""
    RR = dict(quad_type='RADAU-RIGHT')
    h2 = dict(nvars=(16, 16), nu=0.1, freq=2, bc='periodic')
    sw = dict(num_nodes=3, QI='LU', **RR)
    cases = []
    cases.append(run_case('fixedK_2d_P4', 'heat_unforced', h2, 'generic_implicit', sw, dict(dt=0.02, restol=-1), 4, 0.0,
                          0.16, num_procs=4, seed=1))
    cases.append(run_case('fixedK_2d_P3', 'heat_unforced', h2, 'generic_implicit', sw, dict(dt=0.02, restol=-1), 3, 0.0,
                          0.12, num_procs=3, seed=2))
    cases.append(run_case('alltodone_2d_P4', 'heat_unforced', h2, 'generic_implicit', sw, dict(dt=0.02, restol=1e-9), 50,
                          0.0, 0.16, num_procs=4, controller_params=dict(all_to_done=True), seed=3))
    cases.append(run_case('fixedK_2d_P4_tail', 'heat_unforced', h2, 'generic_implicit', sw, dict(dt=0.02, restol=-1), 4,
                          0.0, 0.12, num_procs=4, seed=4))     # second block: 2 of 4 ranks active
    save('runs_relay.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_RELAY', '0') == '1':
    relay_main()


def cg_main():
    """solver_type='CG' (generic_ND_FD.py:252-260: scipy cg, rtol=lintol, atol=0, x0 = previous node value): sweeps
    with the accumulated work_counters['CG'] after every sweep."""
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    cases.append(sweep_case('cg_heat3d_16', 'heat_unforced', dict(nvars=(16, 16, 16), nu=0.1, freq=2, solver_type='CG',
                                                                  lintol=1e-12, liniter=1000),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 0.01, u0_kind='exact'))
    cases.append(sweep_case('cg_heat2d_32_IE', 'heat_unforced', dict(nvars=(32, 32), nu=0.1, freq=2, solver_type='CG',
                                                                     lintol=1e-10, liniter=1000),
                            'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 0.02, u0_kind='randn'))
    cases.append(sweep_case('cg_heat1d_64_o4', 'heat_unforced', dict(nvars=64, nu=0.1, freq=2, order=4, solver_type='CG',
                                                                     lintol=1e-12, liniter=1000),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 0.005, u0_kind='exact'))
    cases.append(sweep_case('cg_forced2d_16', 'heat_forced', dict(nvars=(16, 16), nu=0.1, freq=2, solver_type='CG',
                                                                  lintol=1e-12, liniter=1000),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 0.02, u0_kind='exact'))
    save('sweeps_cg.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_CG', '0') == '1':
    cg_main()


def vdp_jacobian_main():
    """vanderpol.solve_jacobian (Van_der_Pol_implicit.py:190-201) on random states and right-hand sides."""
    from pySDC.implementations.problem_classes.Van_der_Pol_implicit import vanderpol

    P = vanderpol(mu=5.0, u0=(2.0, 0.0), newton_tol=1e-9, newton_maxiter=100)
    rng = np.random.default_rng(21)
    U = rng.uniform(-2.0, 2.0, (2, 64))
    Rh = rng.standard_normal((2, 64))
    out = np.zeros((2, 64))
    for i in range(64):
        u = P.dtype_u(P.init)
        u[:] = U[:, i]
        r = P.dtype_u(P.init)
        r[:] = Rh[:, i]
        out[:, i] = np.asarray(P.solve_jacobian(r, 0.05, u))
    np.savez_compressed(os.path.join(OUT, 'vdp_jacobian.npz'), u=U, rhs=Rh, dt=0.05, mu=5.0, out=out)
    print('vdp_jacobian.npz')


if __name__ == '__main__' and os.environ.get('GOLDEN_VDPJ', '0') == '1':
    vdp_jacobian_main()


def dirichlet_ml_main():
    """non-periodic mesh_to_mesh (1-D dirichlet-zero, nested grids nf = 2 nc + 1) and the tutorials' multi-level
    runs on it: step_4/C-like SDC vs 3-level MLSDC, step_6/A-like two-level PFASST with 1/2/4 processes."""
    from pySDC.implementations.transfer_classes.TransferMesh import mesh_to_mesh

    out = {}
    rng = np.random.default_rng(8)
    for tag, nf, nc, io, ro in (('d1_63_31_o2', 63, 31, 2, 2), ('d1_127_63_o6', 127, 63, 6, 2), ('d1_31_15_o4', 31, 15, 4, 4),
                                ('d1_15_7_o6', 15, 7, 6, 2)):
        pf = heatNd_unforced(nvars=nf, nu=0.1, freq=2, bc='dirichlet-zero')
        pc = heatNd_unforced(nvars=nc, nu=0.1, freq=2, bc='dirichlet-zero')
        T = mesh_to_mesh(pf, pc, dict(iorder=io, rorder=ro, periodic=False))
        F = pf.dtype_u(pf.init)
        F[:] = rng.standard_normal(pf.init[0])
        G = pc.dtype_u(pc.init)
        G[:] = rng.standard_normal(pc.init[0])
        out[f'{tag}/fine'], out[f'{tag}/coarse'] = np.asarray(F).copy(), np.asarray(G).copy()
        out[f'{tag}/restricted'] = np.asarray(T.restrict(F)).copy()
        out[f'{tag}/prolonged'] = np.asarray(T.prolong(G)).copy()
        out[f'{tag}/meta'] = np.array(json.dumps(dict(name=tag, nf=nf, nc=nc, iorder=io, rorder=ro)))
    np.savez_compressed(os.path.join(OUT, 'transfer_dirichlet.npz'), **out)

    def ml_run(name, pp, sw, lp, maxiter, t0, Tend, num_procs, cp=None, io=6, ro=2, prob='heat_unforced',
               sweeper='generic_implicit'):
        desc = dict(problem_class=PROBS[prob], problem_params=pp, sweeper_class=SWEEPERS[sweeper], sweeper_params=sw,
                    level_params=lp, step_params=dict(maxiter=maxiter), space_transfer_class=mesh_to_mesh,
                    space_transfer_params=dict(rorder=ro, iorder=io, periodic=False))
        cpar = dict(logger_level=40)
        cpar.update(cp or {})
        C = controller_nonMPI(num_procs, cpar, desc)
        P = C.MS[0].levels[0].prob
        u0 = P.u_exact(t0)
        uend, stats = C.run(u0, t0, Tend)
        o = {'u0': np.asarray(u0).copy(), 'uend': np.asarray(uend).copy()}
        niter = get_sorted(stats, type='niter', sortby='time')
        o['niter_t'] = np.array([t for t, _ in niter])
        o['niter'] = np.array([v for _, v in niter])
        o['res'] = np.array([v for _, v in get_sorted(stats, type='residual_post_iteration', sortby='time')])
        o['meta'] = np.array(json.dumps(dict(name=name, prob=prob, prob_params=pp, sweeper=sweeper, sweeper_params=sw,
                                             level_params=lp, maxiter=maxiter, t0=t0, Tend=Tend, num_procs=num_procs,
                                             controller_params=cp or {}, iorder=io, rorder=ro, periodic=False)))
        return o

    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    # tutorial step_4/C: heat 1-D dirichlet, M=5 LU, dt=0.1, restol 1e-9, three levels
    cases.append(ml_run('t4c_mlsdc3', dict(nvars=[255, 127, 63], nu=0.1, freq=4, bc='dirichlet-zero'),
                        dict(num_nodes=5, QI='LU', **RR), dict(dt=0.1, restol=1e-9), 50, 0.1, 0.2, 1, io=6, ro=2))
    # tutorial step_6/A: two levels, M=3, dt=0.125, restol 5e-10, 8 steps
    base = dict(pp=dict(nvars=[63, 31], nu=0.1, freq=2, bc='dirichlet-zero'), sw=dict(num_nodes=3, QI='LU', **RR),
                lp=dict(dt=0.125, restol=5e-10), maxiter=50, t0=0.0, Tend=1.0, io=6, ro=2)
    cases.append(ml_run('t6a_mlsdc', num_procs=1, **base))
    for P_ in (2, 4):
        cases.append(ml_run(f't6a_pfasst_P{P_}', num_procs=P_, cp=dict(predict_type='pfasst_burnin', all_to_done=True),
                            **base))
    save('runs_ml_dirichlet.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_DML', '0') == '1':
    dirichlet_ml_main()


def relay8_main():
    """eight time ranks (the node size the benchmark scales to): fixed number of sweeps, two blocks, the second one with
    four active ranks only; and an all_to_done run."""
    RR = dict(quad_type='RADAU-RIGHT')
    h2 = dict(nvars=(16, 16), nu=0.1, freq=2, bc='periodic')
    sw = dict(num_nodes=3, QI='LU', **RR)
    cases = []
    cases.append(run_case('fixedK_2d_P8', 'heat_unforced', h2, 'generic_implicit', sw, dict(dt=0.01, restol=-1), 3, 0.0,
                          0.12, num_procs=8, seed=11))
    cases.append(run_case('alltodone_2d_P8', 'heat_unforced', h2, 'generic_implicit', sw, dict(dt=0.01, restol=1e-8), 50,
                          0.0, 0.08, num_procs=8, controller_params=dict(all_to_done=True), seed=12))
    save('runs_relay8.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_RELAY8', '0') == '1':
    relay8_main()


def skip_main():
    """runs with the sweeper parameter skip_residual_computation (core/sweeper.py:176-179): every stage (a fixed number
    of sweeps, the residual stays 0.0) and single stages (the residual of the other stage decides)."""
    RR = dict(quad_type='RADAU-RIGHT')
    ALL = ('IT_CHECK', 'IT_FINE', 'IT_DOWN', 'IT_UP', 'IT_COARSE')
    h2 = dict(nvars=(16, 16), nu=0.1, freq=2, bc='periodic')
    h3 = dict(nvars=(8, 8, 8), nu=0.1, freq=2, bc='periodic')
    ad = dict(nvars=64, nu=0.05, c=1.0, freq=2, bc='periodic')
    cases = []
    cases.append(run_case('skip_all_2d', 'heat_unforced', h2, 'generic_implicit',
                          dict(num_nodes=3, QI='LU', skip_residual_computation=ALL, **RR), dict(dt=0.02, restol=-1), 4, 0.0,
                          0.06, seed=1))
    cases.append(run_case('skip_all_3d_M5', 'heat_unforced', h3, 'generic_implicit',
                          dict(num_nodes=5, QI='IE', skip_residual_computation=ALL, **RR), dict(dt=0.01, restol=-1), 4, 0.0,
                          0.03, seed=2))
    cases.append(run_case('skip_all_imex', 'advdiff', ad, 'imex_1st_order',
                          dict(num_nodes=3, QI='LU', QE='EE', skip_residual_computation=ALL, **RR), dict(dt=0.01, restol=-1), 3,
                          0.0, 0.03, seed=3))
    cases.append(run_case('skip_all_2d_P3', 'heat_unforced', h2, 'generic_implicit',
                          dict(num_nodes=3, QI='LU', skip_residual_computation=ALL, **RR), dict(dt=0.02, restol=-1), 3, 0.0,
                          0.12, num_procs=3, seed=4))
    cases.append(run_case('skip_fine_2d', 'heat_unforced', h2, 'generic_implicit',
                          dict(num_nodes=3, QI='LU', skip_residual_computation=('IT_FINE',), **RR), dict(dt=0.02, restol=1e-9), 50,
                          0.0, 0.06, seed=5))
    cases.append(run_case('skip_check_2d_P2', 'heat_unforced', h2, 'generic_implicit',
                          dict(num_nodes=3, QI='LU', skip_residual_computation=('IT_CHECK',), **RR), dict(dt=0.02, restol=1e-9),
                          50, 0.0, 0.08, num_procs=2, seed=6))
    save('runs_skip.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_SKIP', '0') == '1':
    skip_main()


def guess_main():
    """predictor variants on IMEX problems with a time-dependent forcing: 'copy' leaves f(t0) at every node, 'zero'
    zeros - the first sweep then integrates exactly those stored values (core/sweeper.py:140-158)."""
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    for guess in ('copy', 'zero', 'spread'):
        cases.append(sweep_case(f'forced2d_guess_{guess}', 'heat_forced', dict(nvars=(16, 16), nu=0.1, freq=2),
                                'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', initial_guess=guess, **RR), 0.05,
                                u0_kind='exact'))
    cases.append(sweep_case('forced3d_guess_copy', 'heat_forced', dict(nvars=(8, 8, 8), nu=0.1, freq=2),
                            'imex_1st_order', dict(num_nodes=5, QI='IE', QE='EE', initial_guess='copy', **RR), 0.02,
                            u0_kind='randn'))
    cases.append(sweep_case('forced1d_guess_zero', 'heat_forced', dict(nvars=64, nu=0.1, freq=2),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='PIC', initial_guess='zero', **RR), 0.01,
                            u0_kind='exact'))
    for nm, prob, pp, sweeper, sw in (
            ('heat2d_guess_random', 'heat_unforced', dict(nvars=(16, 16), nu=0.1, freq=2), 'generic_implicit',
             dict(num_nodes=3, QI='LU', initial_guess='random', **RR)),
            ('forced2d_guess_random', 'heat_forced', dict(nvars=(16, 16), nu=0.1, freq=2), 'imex_1st_order',
             dict(num_nodes=3, QI='LU', QE='EE', initial_guess='random', **RR)),
            ('advdiff1d_guess_random', 'advdiff', dict(nvars=64, nu=0.02, c=1.0, freq=2), 'imex_1st_order',
             dict(num_nodes=3, QI='IE', QE='EE', initial_guess='random', **RR)),
            ('advdiff2d_guess_copy', 'advdiff', dict(nvars=(16, 16), nu=0.02, c=1.0, freq=2), 'imex_1st_order',
             dict(num_nodes=3, QI='IE', QE='EE', initial_guess='copy', **RR))):
        cases.append(sweep_case(nm, prob, pp, sweeper, sw, 0.02, u0_kind='exact'))
    save('sweeps_guess.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_GUESS', '0') == '1':
    guess_main()


def _subsample(a):
    """every 4th point per axis at staggered offsets: 16^3 of a 64^3 field (keeps a big-grid fixture small)"""
    a = np.asarray(a)
    return a[..., 1::4, 2::4, 3::4].copy()


def big3d_main():
    """the bench's data flow pinned to the reference at a size where the fused kernels run (n >= 64): heat 3-D 64^3,
    M=5, IE, solver_type='CG' (generic_ND_FD.py:252-260; the reference's SuperLU solve is not feasible at this size),
    stiffness dt*nu*12/dx^2 = 315 like the headline workload.  Full fields would be 20 MB per snapshot, so node values
    are stored on a 16^3 subsample together with the max / l2 norms of the full fields; u0 and the end values are full.
    Plus a two-step run to restol (iteration counts) with QI=LU."""
    RR = dict(quad_type='RADAU-RIGHT')
    nv = (64, 64, 64)
    pp = dict(nvars=nv, nu=0.1, freq=2, order=2, bc='periodic', solver_type='CG', lintol=1e-12, liniter=1000)
    dt = 1e-3 * (512 / 64) ** 2
    full = sweep_case('cg64_heat3d_M5_IE', 'heat_unforced', pp, 'generic_implicit', dict(num_nodes=5, QI='IE', **RR),
                      dt, t0=0.0, nsweeps=3, seed=0, u0_kind='exact')
    out = {}
    for k, v in full.items():
        if k.endswith('_u') or k.endswith('_f'):
            out[k + '_sub'] = _subsample(v)
            out[k + '_max'] = np.max(np.abs(v.reshape(v.shape[0], -1)), axis=1)
            out[k + '_l2'] = np.sqrt(np.sum(v.reshape(v.shape[0], -1) ** 2, axis=1))
        elif '_uend_' in k and k != 'k3_uend_0':
            out[k + '_sub'] = _subsample(v)
        else:
            out[k] = v
    cases = [out]
    run = run_case('cg64_heat3d_run_LU', 'heat_unforced', pp, 'generic_implicit', dict(num_nodes=5, QI='LU', **RR),
                   dict(dt=dt, restol=1e-9), 50, 0.0, 2 * dt, seed=0)
    assert np.array_equal(run['u0'], full['u0'])
    del run['u0']  # same input as the sweep case
    cases.append(run)
    save('sweeps_big3d.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_BIG3D', '0') == '1':
    big3d_main()


def cfg5_main():
    """BASELINE config 5 at its stated resolution, pinned through the reference's 2-D problem: allencahn2d_imex
    256^2 / 128^2, two levels, M=3 on both, mesh_to_mesh iorder 6 / rorder 2, dt=1e-3 - MLSDC (1 process) and PFASST with 8
    processes (controller_nonMPI, pfasst_burnin).  A 3-D run whose initial value does not depend on z must reproduce
    these planes (the 3-D spectral Laplacian and the tensor-product transfer act as their 2-D counterparts on it)."""
    from pySDC.implementations.problem_classes.AllenCahn_2D_FFT import allencahn2d_imex

    PROBS['allencahn2d'] = allencahn2d_imex
    RR = dict(quad_type='RADAU-RIGHT')
    pp = dict(nvars=[(256, 256), (128, 128)], nu=2, eps=0.04, radius=0.25)
    base = dict(prob='allencahn2d', pp=pp, sweeper='imex_1st_order', sw=dict(num_nodes=3, QI='LU', QE='EE', **RR),
                lp=dict(dt=1e-3, restol=1e-8, nsweeps=1), maxiter=50, t0=0.0, seed=5)
    cases = [ml_run_case('cfg5_ac2d_mlsdc', num_procs=1, Tend=2e-3, **base),
             ml_run_case('cfg5_ac2d_pfasst_P8', num_procs=8, Tend=8e-3,
                         controller_params=dict(predict_type='pfasst_burnin'), **base)]
    save('runs_cfg5.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_CFG5', '0') == '1':
    cfg5_main()


def ml8_main():
    """two-level PFASST with EIGHT processes (one node's worth of time ranks, BASELINE config 5's layout) through the
    reference's serial controller: heat 2-D 16^2 / 8^2 with burn-in predictor, 16 steps = two blocks; and IMEX forced heat."""
    RR = dict(quad_type='RADAU-RIGHT')
    heat2 = dict(nvars=[(16, 16), (8, 8)], nu=0.1, freq=2, bc='periodic')
    cases = []
    base = dict(prob='heat_unforced', pp=heat2, sweeper='generic_implicit', sw=dict(num_nodes=3, QI='LU', **RR),
                lp=dict(dt=0.02, restol=1e-9), maxiter=50, t0=0.0, Tend=0.32)
    cases.append(ml_run_case('pfasst_heat2d_P8', num_procs=8, controller_params=dict(predict_type='pfasst_burnin'), **base))
    forced = dict(prob='heat_forced', pp=heat2, sweeper='imex_1st_order', sw=dict(num_nodes=3, QI='LU', QE='EE', **RR),
                  lp=dict(dt=0.05, restol=1e-9), maxiter=50, t0=0.0, Tend=0.4)
    cases.append(ml_run_case('pfasst_forced2d_P8', num_procs=8, controller_params=dict(predict_type='pfasst_burnin'),
                             **forced))
    save('runs_ml8.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_ML8', '0') == '1':
    ml8_main()


def dirichlet_nd_main():
    """dirichlet-zero in 2-D / 3-D (generic_ND_FD.py:99-133, helpers/problem_helper.py:143-224; order 2): sweeps of the
    implicit and the IMEX sweeper and a run to restol, by the reference."""
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    cases.append(sweep_case('heat2d_dirichlet', 'heat_unforced', dict(nvars=(15, 15), nu=0.1, freq=(1, 2), bc='dirichlet-zero'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 2e-2))
    cases.append(sweep_case('heat3d_dirichlet', 'heat_unforced', dict(nvars=(7, 7, 7), nu=0.1, freq=(1, 1, 2), bc='dirichlet-zero'),
                            'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 1e-2))
    cases.append(sweep_case('forced2d_dirichlet', 'heat_forced', dict(nvars=(15, 15), nu=0.1, freq=(1, 3), bc='dirichlet-zero'),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 2e-2, u0_kind='exact'))
    cases.append(sweep_case('forced3d_dirichlet', 'heat_forced', dict(nvars=(7, 7, 7), nu=0.1, freq=(1, 2, 1), bc='dirichlet-zero'),
                            'imex_1st_order', dict(num_nodes=3, QI='IE', QE='EE', **RR), 1e-2, u0_kind='exact'))
    save('sweeps_dirichlet_nd.npz', cases)
    cases = []
    cases.append(run_case('heat2d_dirichlet_run', prob='heat_unforced', prob_params=dict(nvars=(31, 31), nu=0.1, freq=(1, 1), bc='dirichlet-zero'),
                          sweeper='generic_implicit', sweeper_params=dict(num_nodes=3, QI='LU', **RR),
                          level_params=dict(dt=0.05, restol=1e-9), maxiter=50, t0=0.0, Tend=0.15))
    cases.append(run_case('forced3d_dirichlet_run_P2', prob='heat_forced', prob_params=dict(nvars=(7, 7, 7), nu=0.1, freq=(1, 1, 1), bc='dirichlet-zero'),
                          sweeper='imex_1st_order', sweeper_params=dict(num_nodes=3, QI='LU', QE='EE', **RR),
                          level_params=dict(dt=0.05, restol=1e-9), maxiter=50, t0=0.0, Tend=0.2, num_procs=2))
    save('runs_dirichlet_nd.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_DND', '0') == '1':
    dirichlet_nd_main()


def nsweeps2_main():
    """lock-step multi-step SDC with TWO sweeps per iteration (level_params nsweeps=2): between the sweeps of an iteration
    it_fine sends again (controller_MPI.py:736-768) - the setting in which a send left in flight must not meet an early
    rewrite of the end value (controller_dist: side-stream posting only with one sweep per iteration)."""
    RR = dict(quad_type='RADAU-RIGHT')
    h2 = dict(nvars=(16, 16), nu=0.1, freq=2, bc='periodic')
    sw = dict(num_nodes=3, QI='LU', **RR)
    cases = [run_case('fixedK_2d_P3_nsweeps2', 'heat_unforced', h2, 'generic_implicit', sw, dict(dt=0.02, restol=-1, nsweeps=2),
                      3, 0.0, 0.12, num_procs=3, seed=6),
             run_case('fixedK_2d_P2_nsweeps2', 'heat_unforced', h2, 'generic_implicit', sw, dict(dt=0.02, restol=-1, nsweeps=2),
                      3, 0.0, 0.08, num_procs=2, seed=7)]
    save('runs_nsweeps2.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_NSW2', '0') == '1':
    nsweeps2_main()


def _thin(v, step):
    """every step-th point per spatial axis at staggered offsets (leading axes - nodes, components - are kept)"""
    v = np.asarray(v)
    return v[..., 1::step].copy() if v.ndim <= 2 else v[..., 1::step, (step // 2 + 1)::step].copy()


def _reduce_fields(full, step, keep_full=()):
    """a sweep case with its node fields replaced by a subsample plus max / l2 norms of the full fields (a 1024^2 case
    would otherwise be 50 MB per snapshot); u0 is rebuilt by the test from the seed and checked against `u0_sub`"""
    out = {}
    for k, v in full.items():
        if k in keep_full:
            out[k] = v
        elif k.endswith('_u') or k.endswith('_f'):
            flat = v.reshape(v.shape[0], -1)
            out[k + '_sub'] = _thin(v, step)
            out[k + '_max'] = np.max(np.abs(flat), axis=1)
            out[k + '_l2'] = np.sqrt(np.sum(flat ** 2, axis=1))
        elif '_uend_' in k or k == 'u0':
            out[k + '_sub'] = _thin(v, step)
            out[k + '_max'] = np.array(np.max(np.abs(v)))
            out[k + '_l2'] = np.array(np.sqrt(np.sum(np.asarray(v) ** 2)))
        else:
            out[k] = v
    return out


def pin1024_main():
    """the template instances the bench launches (line length 1024, M=5 nodes: k_spec_z<1024,5,...>, the strided passes
    <1024,8>) pinned to the reference on FULL spectra: heat 1-D N=1024 and 2-D 1024^2, M=5 RADAU-RIGHT, IE, the stiffness
    of the headline workload (dt*nu/dx^2 = 26.2: dt = 2.5e-4 at nu = 0.1), input = sine mode + 1e-3 seeded noise, three
    sweeps with the reference's SuperLU solve (76 s per solve at 1024^2: 20 minutes), plus a 1-D run to restol.  A 3-D
    1024^3 field that does not depend on one axis reduces to the 2-D problem exactly (the stencil of that axis sums to
    zero), so the 2-D case also pins the 3-D kernels at the bench's own size, one pair of axes at a time
    (tests/test_gpu_fullsize.py)."""
    RR = dict(quad_type='RADAU-RIGHT')
    dt = 2.5e-4
    sw = dict(num_nodes=5, QI='IE', **RR)
    h1 = dict(nvars=1024, nu=0.1, freq=2, order=2, bc='periodic')
    c1 = sweep_case('pin_heat1d_1024_M5_IE', 'heat_unforced', h1, 'generic_implicit', sw, dt, t0=0.0, nsweeps=3, seed=0,
                    u0_kind='exact')
    r1 = run_case('pin_heat1d_1024_run', 'heat_unforced', h1, 'generic_implicit', sw, dict(dt=dt, restol=1e-10), 50, 0.0,
                  3 * dt, seed=0)
    h2 = dict(nvars=(1024, 1024), nu=0.1, freq=2, order=2, bc='periodic', solver_type='direct')
    full = sweep_case('pin_heat2d_1024_M5_IE', 'heat_unforced', h2, 'generic_implicit', sw, dt, t0=0.0, nsweeps=3, seed=0,
                      u0_kind='exact')
    c2 = _reduce_fields(full, 16)
    c2['k3_uend_0_q'] = np.asarray(full['k3_uend_0'])[1::4, 3::4].copy()     # a denser look at the end value
    save('sweeps_pin1024.npz', [c1, r1, c2])


if __name__ == '__main__' and os.environ.get('GOLDEN_PIN1024', '0') == '1':
    pin1024_main()


def gmres_main():
    """solver_type='GMRES' (generic_ND_FD.py:241-250: scipy gmres, restart 20, rtol=lintol, atol=0, x0 = previous node
    value, callback_type='legacy'): sweeps with the accumulated work_counters['GMRES'] (one count per inner iteration)
    after every sweep - the non-symmetric advection operators, for which CG is not an option, and a heat case."""
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    cases.append(sweep_case('gmres_adv1d_64', 'advection', dict(nvars=64, c=1.0, freq=2, order=2, stencil_type='center',
                                                                solver_type='GMRES', lintol=1e-12, liniter=1000),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 0.01, u0_kind='exact'))
    cases.append(sweep_case('gmres_adv2d_32_upwind', 'advection', dict(nvars=(32, 32), c=1.0, freq=2, order=1, stencil_type='upwind',
                                                                       solver_type='GMRES', lintol=1e-10, liniter=1000),
                            'generic_implicit', dict(num_nodes=3, QI='IE', **RR), 0.02, u0_kind='randn'))
    cases.append(sweep_case('gmres_heat3d_16', 'heat_unforced', dict(nvars=(16, 16, 16), nu=0.1, freq=2, solver_type='GMRES',
                                                                    lintol=1e-12, liniter=1000),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 0.01, u0_kind='exact'))
    cases.append(sweep_case('gmres_heat2d_32_stiff', 'heat_unforced', dict(nvars=(32, 32), nu=0.1, freq=2, solver_type='GMRES',
                                                                          lintol=1e-10, liniter=1000),
                            'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 0.05, u0_kind='randn'))
    save('sweeps_gmres.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_GMRES', '0') == '1':
    gmres_main()


def dirichlet_ho_main():
    """dirichlet-zero with stencils of order 4 / 6 / 8: the reference shifts the stencil in the rows next to the boundary
    (helpers/problem_helper.py:143-224), which makes the operator non-symmetric; its solves are SuperLU ('direct') or GMRES.
    Sweeps in 1-D / 2-D / 3-D, a run to restol, and the 1-D operator matrices themselves (dense) for the construction test."""
    from pySDC.helpers import problem_helper

    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    cases.append(sweep_case('heat1d_dirichlet_o4', 'heat_unforced', dict(nvars=63, nu=0.1, freq=3, order=4, bc='dirichlet-zero'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 5e-3))
    cases.append(sweep_case('heat1d_dirichlet_o6', 'heat_unforced', dict(nvars=31, nu=0.1, freq=2, order=6, bc='dirichlet-zero'),
                            'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 1e-2))
    cases.append(sweep_case('heat2d_dirichlet_o4', 'heat_unforced', dict(nvars=(15, 15), nu=0.1, freq=(1, 2), order=4, bc='dirichlet-zero'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 2e-2))
    cases.append(sweep_case('heat3d_dirichlet_o4_gmres', 'heat_unforced',
                            dict(nvars=(7, 7, 7), nu=0.1, freq=(1, 1, 2), order=4, bc='dirichlet-zero', solver_type='GMRES',
                                 lintol=1e-12, liniter=1000),
                            'generic_implicit', dict(num_nodes=3, QI='IE', **RR), 1e-2))
    cases.append(sweep_case('forced2d_dirichlet_o8', 'heat_forced', dict(nvars=(15, 15), nu=0.1, freq=(1, 3), order=8, bc='dirichlet-zero'),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 2e-2, u0_kind='exact'))
    save('sweeps_dirichlet_ho.npz', cases)
    cases = [run_case('heat2d_dirichlet_o4_run', prob='heat_unforced',
                      prob_params=dict(nvars=(31, 31), nu=0.1, freq=(1, 1), order=4, bc='dirichlet-zero'), sweeper='generic_implicit',
                      sweeper_params=dict(num_nodes=3, QI='LU', **RR), level_params=dict(dt=0.05, restol=1e-9), maxiter=50, t0=0.0,
                      Tend=0.15)]
    save('runs_dirichlet_ho.npz', cases)
    mats = {}
    for order in (4, 6, 8):
        for size in (15, 31):
            A, _ = problem_helper.get_finite_difference_matrix(derivative=2, order=order, stencil_type='center', dx=1.0 / (size + 1),
                                                              size=size, dim=1, bc='dirichlet-zero')
            mats[f'A_o{order}_n{size}'] = A.toarray()
    np.savez_compressed(os.path.join(OUT, 'dirichlet_ho_matrices.npz'), **mats)


if __name__ == '__main__' and os.environ.get('GOLDEN_DHO', '0') == '1':
    dirichlet_ho_main()


def boundary_main():
    """Neumann / mixed / inhomogeneous boundaries (helpers/problem_helper.py:143-224; generic_ND_FD.py:50-70).
    boundary_matrices.npz: A (dense) and b of get_finite_difference_matrix for every branch - both kinds of ends and their
    mixes, val != 0, reduce, neumann_bc_order, centred and one-sided interior stencils, dim 1 and (for b's indexing) dim 2 / 3.
    sweeps_neumann.npz / runs_neumann.npz: the problem classes with such boundaries (the constructor of GenericNDimFinDiff
    passes neither bcParams nor b on, generic_ND_FD.py:140-148: what the classes compute is the homogeneous operator)."""
    from pySDC.helpers import problem_helper

    combos = [(2, 2, 'center'), (2, 4, 'center'), (2, 6, 'center'), (1, 2, 'center'), (1, 4, 'center'), (1, 1, 'upwind'),
              (1, 3, 'upwind'), (1, 5, 'upwind'), (1, 2, 'forward'), (1, 3, 'backward')]
    bcs = ['dirichlet', 'neumann', ('dirichlet', 'neumann'), ('neumann', 'dirichlet'), 'neumann-zero']
    pars = [None, {'val': 1.5}, {'reduce': True}, {'neumann_bc_order': 2},
            [{'val': -0.5, 'reduce': True}, {'val': 2.0, 'neumann_bc_order': 3}]]
    cases, arrays = [], {}

    def add(der, order, kind, bc, par, size, dim):
        dx = 1.0 / (size + 1)
        given = None if par is None else ([dict(p) for p in par] if isinstance(par, list) else dict(par))
        A, b = problem_helper.get_finite_difference_matrix(derivative=der, order=order, stencil_type=kind, dx=dx, size=size,
                                                           dim=dim, bc=bc, bc_params=given)
        k = len(cases)
        cases.append(dict(derivative=der, order=order, stencil_type=kind, bc=bc, bc_params=par, size=size, dim=dim, dx=dx))
        arrays[f'A{k}'] = A.toarray() if dim == 1 else np.zeros((1, 1))
        arrays[f'b{k}'] = b

    for der, order, kind in combos:
        for bc in bcs:
            for par in pars:
                add(der, order, kind, bc, par, 17, 1)
    for dim in (2, 3):
        add(2, 4, 'center', ('neumann', 'dirichlet'), [{'val': 1.0}, {'val': 2.0}], 9, dim)
        add(1, 3, 'upwind', 'dirichlet', {'val': -3.0}, 9, dim)
    np.savez_compressed(os.path.join(OUT, 'boundary_matrices.npz'), cases=np.array(json.dumps(cases)), **arrays)

    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    cases.append(sweep_case('heat1d_neumann_o2', 'heat_unforced', dict(nvars=63, nu=0.1, freq=3, order=2, bc='neumann'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 5e-3))
    cases.append(sweep_case('heat1d_neumann_zero_o4', 'heat_unforced', dict(nvars=32, nu=0.1, freq=2, order=4, bc='neumann-zero'),
                            'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 1e-2))
    cases.append(sweep_case('heat1d_mixed_o2', 'heat_unforced', dict(nvars=40, nu=0.1, freq=1, order=2, bc=('dirichlet', 'neumann')),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 1e-2))
    cases.append(sweep_case('heat1d_dirichlet_even_grid', 'heat_unforced', dict(nvars=48, nu=0.1, freq=2, order=2, bc='dirichlet'),
                            'generic_implicit', dict(num_nodes=3, QI='IE', **RR), 1e-2))
    cases.append(sweep_case('heat2d_neumann_o2', 'heat_unforced', dict(nvars=(16, 16), nu=0.1, freq=(1, 2), order=2, bc='neumann'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 2e-2))
    cases.append(sweep_case('heat2d_mixed_o4', 'heat_unforced', dict(nvars=(15, 15), nu=0.1, freq=(1, 1), order=4, bc=('neumann', 'dirichlet')),
                            'generic_implicit', dict(num_nodes=3, QI='IE', **RR), 2e-2))
    cases.append(sweep_case('heat3d_neumann_o2_gmres', 'heat_unforced',
                            dict(nvars=(8, 8, 8), nu=0.1, freq=(1, 1, 2), order=2, bc='neumann', solver_type='GMRES', lintol=1e-12,
                                 liniter=1000),
                            'generic_implicit', dict(num_nodes=3, QI='IE', **RR), 1e-2))
    cases.append(sweep_case('forced1d_neumann_o2', 'heat_forced', dict(nvars=31, nu=0.1, freq=2, order=2, bc='neumann'),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 1e-2, u0_kind='exact'))
    cases.append(sweep_case('forced2d_mixed_o4', 'heat_forced', dict(nvars=(15, 15), nu=0.1, freq=(1, 3), order=4, bc=('dirichlet', 'neumann')),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 2e-2, u0_kind='exact'))
    cases.append(sweep_case('advection1d_dirichlet_upwind3', 'advection', dict(nvars=40, c=1.0, freq=2, order=3, stencil_type='upwind', bc='dirichlet'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 1e-2))
    save('sweeps_neumann.npz', cases)
    cases = [run_case('heat2d_neumann_run', prob='heat_unforced',
                      prob_params=dict(nvars=(24, 24), nu=0.1, freq=(1, 1), order=2, bc='neumann'), sweeper='generic_implicit',
                      sweeper_params=dict(num_nodes=3, QI='LU', **RR), level_params=dict(dt=0.05, restol=1e-9), maxiter=50, t0=0.0,
                      Tend=0.15),
             run_case('forced1d_mixed_run', prob='heat_forced',
                      prob_params=dict(nvars=63, nu=0.1, freq=1, order=2, bc=('neumann', 'dirichlet')), sweeper='imex_1st_order',
                      sweeper_params=dict(num_nodes=3, QI='LU', QE='EE', **RR), level_params=dict(dt=0.02, restol=1e-9), maxiter=50,
                      t0=0.0, Tend=0.06)]
    save('runs_neumann.npz', cases)


if __name__ == '__main__' and os.environ.get('GOLDEN_BC', '0') == '1':
    boundary_main()


def pin512_main():
    """config 3's kernel instances (line length 512, M = 5, complex symbol: k_spec_z<512,5,1,1>, the strided passes <512,8>)
    pinned to the reference on a FULL spectrum: advection-diffusion IMEX 2-D 512^2 (implicit diffusion nu = 0.02, explicit
    centred advection c = 1, order 2), imex_1st_order M=5 RADAU-RIGHT, QI=IE, QE=EE, dt = 1e-3 - the bench's config-3
    parameters - input = sine mode + 1e-3 seeded noise, three sweeps with the reference's SuperLU solve.  A 512^3 field
    that does not depend on one axis reduces to this 2-D problem exactly (both stencils sum to zero along that axis), so
    the case also pins the 3-D launches of config 3 at their own size (tests/test_gpu_pin512.py)."""
    RR = dict(quad_type='RADAU-RIGHT')
    sw = dict(num_nodes=5, QI='IE', QE='EE', **RR)
    pp = dict(nvars=(512, 512), nu=0.02, c=1.0, freq=2, order=2, bc='periodic')
    full = sweep_case('pin_advdiff2d_512_M5_IE', 'advdiff', pp, 'imex_1st_order', sw, 1e-3, t0=0.0, nsweeps=3, seed=0,
                      u0_kind='exact')
    c = _reduce_fields(full, 8)
    c['k3_uend_0_q'] = np.asarray(full['k3_uend_0'])[1::4, 3::4].copy()
    save('sweeps_pin512.npz', [c])


if __name__ == '__main__' and os.environ.get('GOLDEN_PIN512', '0') == '1':
    pin512_main()


def radix3_main():
    """even periodic grids that are not powers of two (generic_ND_FD.py:126-129 only asks for even nvars): 96 (1-D), 48^2,
    24^3 - and 192 in 1-D - with the reference's direct solver (SuperLU).  The engine's line transforms of length 3 * 2^p
    (csrc/fft.hpp) give these grids the exact Fourier solve: sweeps of heat / advection / IMEX advection-diffusion / forced
    heat, and runs to a tolerance."""
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    for nv, tag in ((96, '1d_96'), (192, '1d_192'), ((48, 48), '2d_48'), ((24, 24, 24), '3d_24')):
        for order, M, QI, dt in ((2, 5, 'IE', 1e-2), (4, 3, 'LU', 5e-2)):
            if tag == '3d_24' and order == 4:
                continue   # (one 24^3 case with five nodes, one with a FAS correction below: the file stays small)
            cases.append(sweep_case(f'r3_heat{tag}_o{order}_M{M}_{QI}', 'heat_unforced',
                                    dict(nvars=nv, nu=0.1, freq=2, order=order, bc='periodic'),
                                    'generic_implicit', dict(num_nodes=M, QI=QI, **RR), dt, nsweeps=2 if tag == '3d_24' else 3))
    cases.append(sweep_case('r3_adv1d_96', 'advection', dict(nvars=96, c=1.0, freq=2, order=2, stencil_type='center', bc='periodic'),
                            'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 1e-2))
    cases.append(sweep_case('r3_advdiff2d_48', 'advdiff', dict(nvars=(48, 48), nu=0.02, c=1.0, freq=2, order=2),
                            'imex_1st_order', dict(num_nodes=3, QI='IE', QE='EE', **RR), 1e-2))
    cases.append(sweep_case('r3_forced2d_48', 'heat_forced', dict(nvars=(48, 48), nu=0.1, freq=2),
                            'imex_1st_order', dict(num_nodes=3, QI='LU', QE='EE', **RR), 2e-2, u0_kind='exact'))
    cases.append(sweep_case('r3_heat3d_24_tau', 'heat_unforced', dict(nvars=(24, 24, 24), nu=0.1, freq=2),
                            'generic_implicit', dict(num_nodes=3, QI='IE', **RR), 2e-2, nsweeps=2, tau_seed=7))
    save('sweeps_radix3.npz', cases)
    runs = [run_case('r3_run_heat3d_24', 'heat_unforced', dict(nvars=(24, 24, 24), nu=0.1, freq=2), 'generic_implicit',
                     dict(num_nodes=3, QI='IE', **RR), dict(dt=0.02, restol=1e-9), 20, 0.0, 0.06, seed=3),
            run_case('r3_run_heat2d_48_P2', 'heat_unforced', dict(nvars=(48, 48), nu=0.1, freq=2), 'generic_implicit',
                     dict(num_nodes=5, QI='LU', **RR), dict(dt=0.02, restol=1e-10), 30, 0.0, 0.08, num_procs=2, seed=4),
            run_case('r3_run_forced1d_96', 'heat_forced', dict(nvars=96, nu=0.1, freq=2), 'imex_1st_order',
                     dict(num_nodes=3, QI='IE', QE='EE', **RR), dict(dt=0.01, restol=1e-9), 30, 0.0, 0.03)]
    save('runs_radix3.npz', runs)


if __name__ == '__main__' and os.environ.get('GOLDEN_R3', '0') == '1':
    radix3_main()


def radix5_main():
    """the same for grids of 5 * 2^p points (line transforms with a radix-5 stage): 80 and 160 (1-D), 40^2; in 3-D the
    shortest line the engine's kernels take is 40, and a sweep case at 40^3 is 13 MB of node values - the 3-D grid is pinned
    by the oracle (tests/test_gpu_edges.py), whose Fourier solve the 1-D / 2-D cases of this file pin at these line lengths: the
    reference's SuperLU needs minutes per solve at 40^3."""
    RR = dict(quad_type='RADAU-RIGHT')
    cases = [sweep_case('r5_heat1d_80_o2_M5_IE', 'heat_unforced', dict(nvars=80, nu=0.1, freq=2, order=2, bc='periodic'),
                        'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 1e-2, nsweeps=3),
             sweep_case('r5_heat1d_160_o4_M3_LU', 'heat_unforced', dict(nvars=160, nu=0.1, freq=2, order=4, bc='periodic'),
                        'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 5e-2, nsweeps=3),
             sweep_case('r5_heat2d_40_o2_M5_IE', 'heat_unforced', dict(nvars=(40, 40), nu=0.1, freq=2, order=2, bc='periodic'),
                        'generic_implicit', dict(num_nodes=5, QI='IE', **RR), 1e-2, nsweeps=3),
             sweep_case('r5_adv1d_80', 'advection', dict(nvars=80, c=1.0, freq=2, order=2, stencil_type='center', bc='periodic'),
                        'generic_implicit', dict(num_nodes=3, QI='LU', **RR), 1e-2),
             sweep_case('r5_advdiff2d_40', 'advdiff', dict(nvars=(40, 40), nu=0.02, c=1.0, freq=2, order=2),
                        'imex_1st_order', dict(num_nodes=3, QI='IE', QE='EE', **RR), 1e-2)]
    save('sweeps_radix5.npz', cases)
    runs = [run_case('r5_run_heat2d_40', 'heat_unforced', dict(nvars=(40, 40), nu=0.1, freq=2), 'generic_implicit',
                     dict(num_nodes=3, QI='IE', **RR), dict(dt=0.02, restol=1e-9), 20, 0.0, 0.06, seed=5),
            run_case('r5_run_forced1d_80', 'heat_forced', dict(nvars=80, nu=0.1, freq=2), 'imex_1st_order',
                     dict(num_nodes=3, QI='IE', QE='EE', **RR), dict(dt=0.01, restol=1e-9), 30, 0.0, 0.03)]
    save('runs_radix5.npz', runs)


if __name__ == '__main__' and os.environ.get('GOLDEN_R5', '0') == '1':
    radix5_main()


def transfer3d_main():
    """mesh_to_mesh in 3-D on grids the one-launch transfers of round 6 take (sdc_transfer_apply_nested: restriction with
    full weighting streamed once, prolongation through an LDS tile): 64^3 <-> 32^3 with iorder 6 / rorder 2 (BASELINE config
    5's orders) and iorder 4.  The fields hold small integers so that the file stays small."""
    from pySDC.implementations.transfer_classes.TransferMesh import mesh_to_mesh

    out = {}
    for tag, n, io, ro in (('64_io6', 64, 6, 2), ('64_io4', 64, 4, 2), ('64_io2', 64, 2, 2), ('48_io8', 48, 8, 2)):
        nf, nc = (n,) * 3, (n // 2,) * 3
        pf = heatNd_unforced(nvars=nf, nu=0.1, freq=2, bc='periodic')
        pc = heatNd_unforced(nvars=nc, nu=0.1, freq=2, bc='periodic')
        T = mesh_to_mesh(pf, pc, dict(iorder=io, rorder=ro, periodic=True))
        rng = np.random.default_rng(11)
        F = pf.dtype_u(pf.init)
        F[:] = rng.integers(-8, 9, size=nf).astype(float)
        G = pc.dtype_u(pc.init)
        G[:] = rng.integers(-8, 9, size=nc).astype(float)
        out[f'transfer3d_{tag}/fine'] = np.asarray(F).astype(np.int8)
        out[f'transfer3d_{tag}/coarse'] = np.asarray(G).astype(np.int8)
        out[f'transfer3d_{tag}/restricted'] = np.asarray(T.restrict(F)).copy()
        out[f'transfer3d_{tag}/prolonged'] = np.asarray(T.prolong(G)).copy()
        out[f'transfer3d_{tag}/meta'] = np.array(json.dumps(dict(name=f'transfer3d_{tag}', nf=list(nf), nc=list(nc), iorder=io, rorder=ro)))
    np.savez_compressed(os.path.join(OUT, 'transfer3d.npz'), **out)
    print('transfer3d.npz', os.path.getsize(os.path.join(OUT, 'transfer3d.npz')) // 1024, 'KiB')


if __name__ == '__main__' and os.environ.get('GOLDEN_T3D', '0') == '1':
    transfer3d_main()


def ad1d_main():
    """AdvectionDiffusionEquation_1D_FFT.py: advectiondiffusion1d_imex (the IMEX parity anchor, 8c G2) and
    advectiondiffusion1d_implicit - sweeps from random start values and runs to a tolerance at N = 64 / 256."""
    from pySDC.implementations.problem_classes.AdvectionDiffusionEquation_1D_FFT import (advectiondiffusion1d_imex,
                                                                                         advectiondiffusion1d_implicit)

    PROBS['ad1d_imex'] = advectiondiffusion1d_imex
    PROBS['ad1d_implicit'] = advectiondiffusion1d_implicit
    RR = dict(quad_type='RADAU-RIGHT')
    cases = []
    for n in (64, 256):
        for M, QI, dt in ((3, 'LU', 1e-2), (5, 'IE', 5e-3)):
            cases.append(sweep_case(f'ad1d_imex_{n}_M{M}_{QI}', 'ad1d_imex', dict(nvars=n, c=1.0, freq=2, nu=0.02),
                                    'imex_1st_order', dict(num_nodes=M, QI=QI, QE='EE', **RR), dt))
            cases.append(sweep_case(f'ad1d_impl_{n}_M{M}_{QI}', 'ad1d_implicit', dict(nvars=n, c=1.0, freq=2, nu=0.02),
                                    'generic_implicit', dict(num_nodes=M, QI=QI, **RR), dt))
    cases.append(sweep_case('ad1d_imex_96_M3_tau', 'ad1d_imex', dict(nvars=96, c=0.5, freq=-1, nu=0.02), 'imex_1st_order',
                            dict(num_nodes=3, QI='LU', QE='EE', **RR), 1e-2, tau_seed=5, u0_kind='exact'))
    cases.append(sweep_case('ad1d_imex_64_M2_gauss', 'ad1d_imex', dict(nvars=64, c=1.0, freq=-1, nu=0.02, L=2.0), 'imex_1st_order',
                            dict(num_nodes=2, QI='IE', QE='EE', quad_type='GAUSS'), 1e-2, u0_kind='exact'))
    save('sweeps_ad1d.npz', cases)
    runs = [run_case('ad1d_imex_run_256', 'ad1d_imex', dict(nvars=256, c=1.0, freq=-1, nu=0.02), 'imex_1st_order',
                     dict(num_nodes=3, QI='LU', QE='EE', **RR), dict(dt=0.02, restol=1e-10), 50, 0.0, 0.1),
            run_case('ad1d_imex_run_64_P2', 'ad1d_imex', dict(nvars=64, c=1.0, freq=2, nu=0.02), 'imex_1st_order',
                     dict(num_nodes=3, QI='LU', QE='EE', **RR), dict(dt=0.01, restol=1e-9), 50, 0.0, 0.04, num_procs=2),
            run_case('ad1d_impl_run_256', 'ad1d_implicit', dict(nvars=256, c=1.0, freq=4, nu=0.02), 'generic_implicit',
                     dict(num_nodes=5, QI='LU', **RR), dict(dt=0.02, restol=1e-10), 50, 0.0, 0.1),
            run_case('ad1d_impl_run_64_rand', 'ad1d_implicit', dict(nvars=64, c=1.0, freq=0, nu=0.02), 'generic_implicit',
                     dict(num_nodes=3, QI='IE', **RR), dict(dt=0.005, restol=1e-9), 50, 0.0, 0.02)]
    save('runs_ad1d.npz', runs)


if __name__ == '__main__' and os.environ.get('GOLDEN_AD1D', '0') == '1':
    ad1d_main()
