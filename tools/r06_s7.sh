#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for n in 2 3 4 6; do echo "== chunks $n"; SCPP_SC_CHUNKS=$n python tests/tools/sc_sim_prof.py 4096 8 | tail -1; done
