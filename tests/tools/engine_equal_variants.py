"""Streaming engines of ONE library on the GPU for every instantiation of the persistent kernel (round 6: both models, both input holds): the persistent
kernel (one launch, a wavefront per slot) against the pool engine (rounds of launches) -- result rows bitwise, and the time of each.
usage: engine_equal_variants.py [N] [slots] [out.json]"""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
from scpp_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
out = sys.argv[3] if len(sys.argv) > 3 else None


def config(foh):
    d = tempfile.mkdtemp()
    cfg = os.path.join(d, "config")
    shutil.copytree(os.path.join(os.path.dirname(scpp_amd.__file__), "config"), cfg)
    for mdl in ("RocketQuat", "Rocket2D"):
        p = os.path.join(cfg, mdl, "SCvx.info")
        t = open(p).read()
        if not foh:
            t = t.replace("interpolate_input                   true", "interpolate_input                   false")
        if mdl == "Rocket2D":  # the nondimensionalised Rocket2D configuration converges
            t = t.replace("nondimensionalize                   false", "nondimensionalize                   true")
        open(p, "w").write(t)
    return cfg


res, ok = [], True
for model, K in (("RocketQuat", 50), ("Rocket2D", 30)):
    for foh in (True, False):
        m = (scpp_amd.RocketQuat if model == "RocketQuat" else scpp_amd.Rocket2D)(config(foh)).loadParameters()
        x0 = m.randomized_initial_states(N, first=5000)
        rows, rate = {}, {}
        for name, eng in (("pools", _lib.STREAM_POOLS), ("persistent", _lib.STREAM_PERSISTENT)):
            v = scpp_amd.SCvxAlgorithm(m, K=K, batch_max=S).initialize()
            assert v.opts.interpolate_input == int(foh)
            v.ctx.set_stream_engine(eng)
            v.solveStream(x0[:64], slots=min(S, 64))  # warm-up
            t0 = time.time(); n = v.solveStream(x0, slots=S); dt = time.time() - t0
            assert (v.ctx.stream_rounds()["pools"] == 0) == (name == "persistent")
            rows[name] = v.ctx.stream_download_rows()
            g = scpp_amd.Context.unpack_stream_rows(rows[name], K, *((14, 4) if model == "RocketQuat" else (6, 2)))
            rate[name] = n / dt
            print(f"{model:10s} {'FOH' if foh else 'ZOH'} K={K} {name:10s} N={N} slots={S}: {dt:.3f} s, {n} converged ({n / dt:.0f}/s), status!=0: {int((g['status'] != 0).sum())}, "
                  f"solves {g['solves'].mean():.2f}, ipm iterations {g['ipm_iters'].mean():.1f}", flush=True)
            v.ctx.close()
        a, b = rows["pools"], rows["persistent"]
        eq = a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))
        ok = ok and eq
        res.append({"model": model, "hold": "FOH" if foh else "ZOH", "K": K, "instances": N, "slots": S, "converged_per_s": rate, "rows_bitwise_equal": bool(eq),
                    "persistent_over_pools": rate["persistent"] / rate["pools"]})
        print("   rows bitwise equal:", eq, " persistent / pools = %.3f" % (rate["persistent"] / rate["pools"]), flush=True)
if out:
    json.dump({"what": "persistent kernel against the pool engine, one library, per instantiation (tests/tools/engine_equal_variants.py)", "rows": res}, open(out, "w"), indent=1)
print("ENGINE_EQUAL_OK" if ok else "ENGINE_EQUAL_FAILED")
sys.exit(0 if ok else 1)
