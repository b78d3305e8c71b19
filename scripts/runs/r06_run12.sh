#!/bin/bash
# round 6, run 12: THE profile collection of the round on the frozen kernel sources (scripts/profile_r06.sh)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
timeout 3300 bash scripts/profile_r06.sh 2>&1 | tail -60
