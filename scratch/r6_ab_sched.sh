#!/bin/bash
# round 6: A/B of the schedule switches of the training step (captured, bf16x6), alternating inside one session; arguments = extra specs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
AB_ARITH=bf16x6 AB_REPS=${AB_REPS:-3} timeout 1200 python scratch/ab_engine.py "" "$@" 2>&1 | tail -24
