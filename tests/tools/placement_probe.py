"""Are the two throughput regimes of DESIGN.md section 5 (3.6 % apart, same library, same box) a property of an ALLOCATION?  N contexts of the
bench's shape alive at once in one process, the same streaming job on each in turn, several rounds: a context that is fast stays fast <=> the
regime is decided by the physical placement its workspace got.  usage: placement_probe.py [contexts] [rounds] [slots]   (GPU box)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, scpp_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
m = scpp_amd.RocketQuat().loadParameters()
x = m.randomized_initial_states(2 * B, seed=20260927, first=0)
algs = [scpp_amd.SCvxAlgorithm(m, K=50, batch_max=B, device=0).initialize() for _ in range(N)]
algs[0].solveStream(x[:1024], slots=1024)
probe = {}
for i, a in enumerate(algs):  # does a SHORT warm probe predict the ranking of the long jobs?  (what a placement selection would run)
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        nc = a.solveStream(x[:4096], slots=B)
        a.ctx.stream_download()
        ts.append(nc / (time.perf_counter() - t0))
    probe[i] = ts
    print("probe context %d  %s converged/s (4096 instances, three times)" % (i, ["%.0f" % t for t in ts]), flush=True)
rows = []
for r in range(R):
    for i, a in enumerate(algs):
        t0 = time.perf_counter()
        nc = a.solveStream(x, slots=B)
        a.ctx.stream_download()
        dt = time.perf_counter() - t0
        prof = a.ctx.stream_profile()
        rows.append({"round": r, "context": i, "converged_per_s": nc / dt})
        print("round %d context %d  %.1f converged/s" % (r, i, nc / dt), flush=True)
per = {i: [q["converged_per_s"] for q in rows if q["context"] == i] for i in range(N)}
print(json.dumps({"what": "N contexts alive at once, same job on each in turn", "slots": B, "per_context": per, "probe_4096": probe,
                  "spread_between_contexts": max(np.mean(v) for v in per.values()) / min(np.mean(v) for v in per.values()) - 1.,
                  "largest_spread_within_a_context": max(max(v) / min(v) - 1. for v in per.values())}))
