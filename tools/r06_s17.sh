#!/bin/bash
# round 6, session 17: closing session of the shipped library (final6) followed by the one-million-trajectory soak job
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/r06_final.sh final6
bash tools/r06_soak.sh
