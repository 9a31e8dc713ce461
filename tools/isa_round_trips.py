#!/usr/bin/env python3
"""Static ISA statistics of the ipm_kernel phases and sweeps (gfx950 assembly of scpp_hip.cpp).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -ffp-contract=on --cuda-device-only -S -o /tmp/scpp.s scpp_amd/csrc/scpp_hip.cpp
  python tools/isa_round_trips.py /tmp/scpp.s [name-filter]

Per function: instructions, scratch dwords stored / loaded (callee-saved-register saves + spills), vector-memory operations, and
EXPOSED MEMORY ROUND TRIPS: an s_waitcnt vmcnt(N) counts when it covers an operation issued after the previous counted wait, i.e.
when the wavefront has to sit out a fresh memory latency (the hardware counter is in order, so vmcnt(N) waits for everything but
the N most recent operations).  All paths are counted (both sides of a branch), so the number is an upper bound per call.
This is the measurement behind the round-2 restructuring of the phases: load groups closed by a scheduling
barrier, unconditional buffer accesses in the sweeps, chunk functions.  With `--skeleton name` it prints the order of load /
store / wait / MFMA groups of one function.
"""
import re
import sys

W = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4, "b32": 1, "b64": 2, "b96": 3, "b128": 4}


def functions(path):
    funcs, cur = {}, None
    for l in open(path).read().split("\n"):
        m = re.match(r"^(_Z[\w]+):", l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if cur is not None:
            funcs[cur].append(l)
    return funcs


def instructions(body, labels=False):
    pat = r"^\s+[vsbdg]\w*_" if not labels else r"^(\s+[vsbdg]\w*_|\.LBB)"
    return [l.split(";")[0].strip() for l in body if re.match(pat, l)]


def short(name):
    n = re.sub(r"_ZN4scpp3ipm(L?)\d+", "", name)
    return n[:44]


def stats(path, flt):
    for name, body in functions(path).items():
        if ("3ipm" not in name and "--all" not in sys.argv) or (flt and flt not in name):
            continue
        ins = instructions(body)
        st = ld = issued = done = last_rt = rts = 0
        for t in ins:
            m = re.search(r"scratch_(store|load)_(\w+)", t)
            if m:
                if m.group(1) == "store":
                    st += W.get(m.group(2), 1)
                else:
                    ld += W.get(m.group(2), 1)
            if re.search(r"(buffer|global|flat|scratch)_(load|store|atomic)", t):
                issued += 1
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
            if m:
                cover = issued - int(m.group(1))
                if cover > done:
                    if cover > last_rt:
                        rts += 1
                        last_rt = issued
                    done = cover
        print(f"{short(name):46s} ins {len(ins):5d}  scratch st/ld dwords {st:4d}/{ld:4d}  vm ops {issued:4d}  exposed round trips {rts:4d}"
              f"  mfma {sum('v_mfma' in t for t in ins):3d}  lds {sum(t.startswith('ds_') for t in ins):4d}")


def skeleton(path, flt):
    for name, body in functions(path).items():
        if flt not in name:
            continue
        print(name)
        prev, n = None, 0
        for i, t in enumerate(instructions(body, labels=True)):
            kind = None
            if t.startswith("s_waitcnt") and "vmcnt" in t:
                kind = "WAIT " + t
            elif re.search(r"(global|buffer|scratch)_store", t):
                kind = "store"
            elif re.search(r"(global|buffer|scratch|flat)_load", t):
                kind = "load"
            elif "v_mfma" in t:
                kind = "mfma"
            elif t.startswith(".LBB"):
                kind = "LABEL " + t
            elif "s_cbranch" in t or "s_branch" in t or "s_swappc" in t:
                kind = "BR " + t
            if kind:
                if kind == prev:
                    n += 1
                else:
                    if prev and n > 1:
                        print("        x", n)
                    print(f"{i:6d} {kind}")
                    n = 1
                prev = kind
        return


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--skeleton":
        skeleton(sys.argv[1], sys.argv[3])
    else:
        stats(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "RocketQuat")
