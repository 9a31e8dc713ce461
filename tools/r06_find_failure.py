"""Which instance of the one-million-trajectory soak job fails?  Runs the soak's job (bench.py --steps 128 --warmup 1: instance ids 8192 .. 129 x 8192 - 1) through the
shipped library and writes the ids, initial states and result scalars of every instance whose status is not 0 to gpurun_out/r06_failure.json (round 6: one of
1 048 576 with split step lengths, none with the common one).  usage (GPU box): python tools/r06_find_failure.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import scpp_amd

m = scpp_amd.RocketQuat().loadParameters()
B, steps = 8192, 128
alg = scpp_amd.SCvxAlgorithm(m, K=50, batch_max=B).initialize()
x0 = m.randomized_initial_states(B * steps, first=B)
n = alg.solveStream(x0, slots=B)
rows = alg.getStreamSolution()
bad = np.nonzero(rows["status"] != 0)[0]
out = {"instances": int(B * steps), "converged": int(n), "failures": []}
for i in bad:
    out["failures"].append({"instance_id": int(B + i), "row": int(i), "status": int(rows["status"][i]), "sc_iters": int(rows["sc_iters"][i]), "solves": int(rows["solves"][i]),
                            "ipm_iters": int(rows["ipm_iters"][i]), "nu_norm": float(rows["nu_norm"][i]), "trust_region": float(rows["trust_region"][i]),
                            "x_init": [float(v) for v in x0[i]]})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_failure.json"), "w"), indent=1)
print(json.dumps(out)[:2000])
