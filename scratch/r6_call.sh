#!/bin/bash
# usage: scratch/r6_call.sh <tag> <commands...>   - one gpurun call of round 6: logs under gpurun_out/r6/<tag>_*.log (replaces the 50 one-off r5_call*.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r6
tag=$1; shift
i=0
for cmd in "$@"; do
    i=$((i+1))
    echo "=== [$tag/$i] $cmd"
    bash -c "$cmd" > gpurun_out/r6/${tag}_$i.log 2>&1
    echo "rc=$?"; tail -${TAIL:-25} gpurun_out/r6/${tag}_$i.log
done
