#!/usr/bin/env python
"""Per-field traffic table of the interior-point solve: every buffer access of the kernel SOURCES, counted by the CPU wave emulator's
tracer (tests/emu/hip_emu.h, SCPP_EMU_TRAFFIC) over a whole SCvx run of one RocketQuat instance at K = 50 -- per phase, record block and field, loads and
stores, in bytes per interior-point iteration.  Lane bytes (8 per lane and access, lanes outside a record's range are not counted -- the hardware drops
them too); the PMC figure (2.08 MB per instance-iteration) is the same traffic at 128-byte-line granularity as the L2 passes it on.

    python tools/traffic_table.py [K] [instance] [out.json]        (K = 50: about two minutes on one core)
"""
import json, os, subprocess, sys, tempfile, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 50
INST = int(sys.argv[2]) if len(sys.argv) > 2 else 0
OUT = sys.argv[3] if len(sys.argv) > 3 else None

RUN = r'''
import json, sys, numpy as np
sys.path.insert(0, %r)
import scpp_amd, __graft_entry__ as g
m = scpp_amd.RocketQuat().loadParameters()
alg = scpp_amd.SCvxAlgorithm(m, K=%d, batch_max=1, library=g.EMU_LIB).initialize()
alg.ctx.set_stream_engine(scpp_amd._lib.STREAM_POOLS)   # (the persistent kernel calls the same phase functions; the pool engine's launches keep the phases' names apart)
n = alg.solve(m.randomized_initial_states(1, first=%d))
o = alg.getSolution()
print("RESULT " + json.dumps(dict(converged=int(n), ipm_iters=int(o["ipm_iters"][0]), solves=int(o["solves"][0]), sc_iters=int(o["sc_iters"][0]), status=int(o["status"][0]))))
''' % (ROOT, K, INST)

# names of the fields of the records (csrc/ipm_kernel.h: Lay<RocketQuatSC>; NV = 16, NS = 36, NCONES = 6, LP0 = 34, HS_N, NL = 14)
def stage_fields():
    NV, NS, NC, LP0 = 16, 36, 6, 34
    f, o = [], 0
    for name, n in (("W", NV), ("delta", 1), ("DW", NV), ("ddelta", 1), ("WBAR", NV), ("UHAT", 3), ("HDD", 1), ("HDW", NV), ("RXW", NV), ("RXD", 1), ("S", NS), ("Z", NS),
                    ("DS", NS), ("DZ", NS), ("RZ", NS), ("TZ", NS), ("LS (lambda)", NS), ("DSS", NS), ("ETA", NC), ("WB", LP0), ("BXW", NV), ("BXD", 1), ("WBK (fall-back)", NV + 1)):
        f.append((name, o, n)); o += n
    return f, o
SEG = ["NU", "NUB", "S1", "Z1", "S2", "Z2", "DNU", "DNUB", "DS1 (p1)", "DZ1", "DS2 (p2)", "DZ2", "LAM", "DLAM", "RY", "RXNU", "RXNUB", "RZ1", "RZ2", "TZ1", "TZ2", "QV", "BTN", "BNB",
       "DINV", "CV", "BXNU", "BXNUB", "BY"]
def exchange_fields():
    NV = 16
    f = [("X_BETA", 0, NV), ("X_BCW", 16, NV), ("X_VW", 32, NV), ("X_HS + X_HC", 48, 32), ("X_WBT", 80, NV), ("X_RHO", 96, 16), ("X_BCL", 112, 16), ("X_VL", 128, 16), ("X_EINV", 144, 16),
         ("X_S", 160, 16)]
    return f
def factor_fields():
    return [("Li (lower triangle 16)", 0, 136), ("Yt 16 x 14", 136, 224), ("Ti (lower triangle 14)", 360, 105), ("pad", 465, 7)]
def field_name(region, idx):
    if region.startswith("stage"):
        for name, o, n in stage_fields()[0]:
            if o <= idx < o + n:
                return name
    if region.startswith("segment"):
        return SEG[idx // 14] if idx // 14 < len(SEG) else "?"
    if region.startswith("exchange"):
        for name, o, n in exchange_fields():
            if o <= idx < o + n:
                return name
    if region.startswith("factor"):
        for name, o, n in factor_fields():
            if o <= idx < o + n:
                return name
    if region.startswith("saved"):
        return "a (forward column, 16)" if idx % 48 < 16 or (idx < 48) else "c (forward column, 14)"
    if region.startswith("dd copy"):
        return "A" if idx < 196 else "B" if idx < 252 else "C" if idx < 308 else "S" if idx < 322 else "Z"
    return "(all)"

with tempfile.TemporaryDirectory() as d:
    tf = os.path.join(d, "traffic.json")
    env = dict(os.environ, SCPP_EMU_TRAFFIC=tf)
    r = subprocess.run([sys.executable, "-c", RUN], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    cnt = json.load(open(tf))
iters, solves = res["ipm_iters"], res["solves"]
INIT = ("phSetup", "phInitPrimalRhs", "phInitPrimalFinish", "phInitDualRhs", "phInitDualFinish", "phWarmInit", "phDataNorms", "phSegLdsCopy", "solve: outside the phases", "solve: outputs")
def ours(phase):
    return phase.startswith("ph") or phase.startswith("solve") or "Sweep" in phase  # (other kernels: set-up, integration, cost)
# 1. distinct 128-byte lines per invocation of a phase ("U|phase|region|rw"): what leaves the L1 / L2 once the lanes of an instruction and the
#    instructions of one phase that touch a line again have coalesced -- the quantity the PMC counters see (at 53 % L2 hit rate of the lane requests)
uline = collections.defaultdict(int); lane = collections.defaultdict(int); lane_f = collections.defaultdict(int)
for key, n in cnt.items():
    if key.startswith("U|"):
        _, phase, region, rw = key.split("|")
        if ours(phase):
            uline[(phase, region, rw)] += 128 * n
    else:
        phase, region, idx, rw = key.split("|")
        if ours(phase):
            lane[(phase, region, rw)] += 8 * n
            lane_f[(phase, region, field_name(region, int(idx)), rw)] += 8 * n
by_region = collections.defaultdict(lambda: [0., 0.]); by_field = collections.defaultdict(lambda: [0., 0.]); by_phase = collections.defaultdict(lambda: [0., 0.])
lane_region = collections.defaultdict(lambda: [0., 0.])
init_bytes = [0., 0.]
for (phase, region, rw), b in uline.items():
    w = 1 if rw == "S" else 0
    if phase in INIT:
        init_bytes[w] += b
        by_phase["(per solve) " + phase][w] += b
        continue
    by_region[region][w] += b
    by_phase[phase][w] += b
    lane_region[region][w] += lane.get((phase, region, rw), 0)
# 2. per field: the lane bytes of the field, scaled by lines / lane bytes of its (phase, record) -- a field's share of the lines its record's accesses touch
for (phase, region, fname, rw), b in lane_f.items():
    if phase in INIT:
        continue
    w = 1 if rw == "S" else 0
    lb = lane.get((phase, region, rw), 0)
    if lb:
        by_field[(region, fname)][w] += b * uline.get((phase, region, rw), 0) / lb
tot = sum(v[0] + v[1] for v in by_region.values())
lane_tot = sum(v for (ph, _, _), v in lane.items() if ph not in INIT)
S = {"what": "traffic of the interior-point solve per iteration of one instance, counted by the emulator's tracer over a whole SCvx run: distinct 128-byte lines per "
             "invocation of a phase (x 128 B); `lane_bytes` = 8 B per lane and access before any coalescing", "K": K, "instance": INST, **res,
     "bytes_per_iteration_main_loop": tot / iters, "bytes_per_solve_outside_the_main_loop": (init_bytes[0] + init_bytes[1]) / solves,
     "bytes_per_iteration_all": (tot + init_bytes[0] + init_bytes[1]) / iters, "lane_bytes_per_iteration_main_loop": lane_tot / iters,
     "records": {k: {"read": v[0] / iters, "written": v[1] / iters, "share": (v[0] + v[1]) / tot, "lane_bytes_read": lane_region[k][0] / iters, "lane_bytes_written": lane_region[k][1] / iters}
                 for k, v in sorted(by_region.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))},
     "fields": [{"record": k[0], "field": k[1], "read": v[0] / iters, "written": v[1] / iters, "share": (v[0] + v[1]) / tot}
                for k, v in sorted(by_field.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))],
     "phases": {k: {"read": v[0] / (solves if k.startswith("(per solve)") else iters), "written": v[1] / (solves if k.startswith("(per solve)") else iters)}
                for k, v in sorted(by_phase.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))}}
if OUT:
    json.dump(S, open(OUT, "w"), indent=1)
print("K = %d, instance %d: %d SCvx iterations, %d solves, %d interior-point iterations" % (K, INST, res["sc_iters"], solves, iters))
print("main loop: %.0f B per iteration in distinct lines (%.0f lane bytes); outside the main loop (set-up, initialisation / warm start, norms, outputs): %.0f B per solve; together %.0f B per iteration"
      % (S["bytes_per_iteration_main_loop"], S["lane_bytes_per_iteration_main_loop"], S["bytes_per_solve_outside_the_main_loop"], S["bytes_per_iteration_all"]))
print("\n| record | read B/iter | written B/iter | share | lane bytes read / written |\n|---|---|---|---|---|")
for k, v in S["records"].items():
    print("| %s | %.0f | %.0f | %.1f %% | %.0f / %.0f |" % (k, v["read"], v["written"], 100 * v["share"], v["lane_bytes_read"], v["lane_bytes_written"]))
print("\n| record | field | read B/iter | written B/iter | share |\n|---|---|---|---|---|")
for f in S["fields"][:45]:
    print("| %s | %s | %.0f | %.0f | %.1f %% |" % (f["record"].split(" [")[0], f["field"], f["read"], f["written"], 100 * f["share"]))
print("\n| phase | read | written | (B per iteration; per solve where marked) |\n|---|---|---|---|")
for k, v in S["phases"].items():
    print("| %s | %.0f | %.0f | |" % (k, v["read"], v["written"]))
