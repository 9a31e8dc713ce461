"""First-contact GPU probe: parity of the HIP kernels against the oracle + first timings."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import oracle_lib as O
import scpp_amd

def main():
    res = {}
    m = scpp_amd.RocketQuat().loadParameters()
    K = 50
    # ---- discretize parity ----
    B = 8
    x0 = m.randomized_initial_states(B)
    alg = scpp_amd.SCAlgorithm(m, K=K, batch_max=8192).initialize()
    ctx = alg.ctx
    print(ctx.lib.scpp_hip_version())
    ctx.sc_setup(m.p, alg.opts, x0)
    ctx.discretize()
    A, Bm, C, S, Z = ctx.download_dd()
    worst = 0.
    for b in range(B):
        sc = O.SC(0, K=K); sc.randomize(20260927, b); sc.set_solver(1); sc.solve()
        X, U, t = sc.iterate(0)
        Ao, Bo, Co, So, Zo = O.discretize(0, m.flow_params(x0[b]), X, U, t)
        for a, o in ((A[b], Ao), (Bm[b], Bo), (C[b], Co), (S[b], So), (Z[b], Zo)):
            worst = max(worst, np.abs(a - o).max() / np.abs(o).max())
    res['discretize_rel_err_vs_oracle'] = worst
    print('discretize rel err', worst)
    # ---- SC parity, mfma on/off ----
    r = O.sc_batch(K, 20260927, 0, B, nthreads=8, solver=1)
    for mf in (0, 1):
        ctx.set_socp_opts(use_mfma=mf)
        t0 = time.time(); nconv = alg.solve(x0); dt = time.time() - t0
        out = alg.getSolution()
        sx = np.abs(r['X']).max(axis=1, keepdims=True); su = np.abs(r['U']).max(axis=1, keepdims=True)
        relX = (np.abs(out['X'] - r['X']) / np.maximum(sx, 1e-9)).max(); relU = (np.abs(out['U'] - r['U']) / np.maximum(su, 1e-9)).max()
        print('mfma', mf, 'B=8 time', dt, 'relX', relX, 'relU', relU, 'iters', out['sc_iters'], r['iters'], 'ipm', out['ipm_iters'], r['ipm_iters'], 'status', out['status'])
        res[f'sc_parity_mfma{mf}'] = dict(relX=relX, relU=relU, iters_equal=bool((out['sc_iters'] == r['iters']).all()), ipm_equal=bool((out['ipm_iters'] == r['ipm_iters']).all()))
    # ---- timings ----
    for Bt in (256, 2048, 8192):
        x0 = m.randomized_initial_states(Bt)
        ctx.timing(reset=True)
        t0 = time.time(); nconv = alg.solve(x0); dt = time.time() - t0
        tm = ctx.timing(reset=True)
        out = alg.getSolution()
        print('B', Bt, 'wall', dt, 'traj/s', Bt / dt, 'nconv', nconv, 'mean sc iters', out['sc_iters'].mean(), 'mean ipm', out['ipm_iters'].mean(), 'status!=0', int((out['status'] != 0).sum()), tm)
        res[f'B{Bt}'] = dict(wall=dt, traj_per_s=Bt / dt, timing=tm, mean_ipm=float(out['ipm_iters'].mean()), fails=int((out['status'] != 0).sum()))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'probe.json'), 'w'), indent=1)

if __name__ == '__main__':
    main()
