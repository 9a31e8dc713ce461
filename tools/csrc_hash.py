#!/usr/bin/env python
"""Content hash of the kernel sources (scpp_amd/csrc/*, include/scpp_hip.h): the identity of the library a profile was taken from.
The GPU box has no .git, so a commit id cannot be checked there; tools/pmc_hbm.sh stores this hash in its summary (`csrc_sha`) and
bench.py marks an imported PMC summary `"stale": true` when the hash of the sources it runs on differs."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "scpp_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".cpp")))
    files.append(os.path.join(ROOT, "include", "scpp_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    # the floating-point flags the library is built with are part of its identity (round 5: -ffp-contract=on changes every result at rounding level)
    for line in open(os.path.join(ROOT, "__graft_entry__.py")):
        if line.startswith("HIP_FP_FLAGS ="):
            h.update(line.strip().encode())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
