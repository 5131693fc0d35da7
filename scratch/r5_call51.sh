#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag17; rm -rf $out; mkdir -p $out
run() { name=$1; shift; env DIAG_NOSAFE=1 "$@" timeout 1500 python scratch/r5_bursts_diag.py f32 1500 eager > $out/$name.log 2>&1; echo "$name: $(grep 'repetitions differ' $out/$name.log)"; }
run wholerf_a FN_LIB=lib_wholerf.so
run wholerf_b FN_LIB=lib_wholerf.so
