#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag18; rm -rf $out; mkdir -p $out
run() { name=$1; shift; env DIAG_NOSAFE=1 "$@" timeout 1500 python scratch/r5_bursts_diag.py f32 1500 eager > $out/$name.log 2>&1; echo "$name: $(grep 'repetitions differ' $out/$name.log)"; }
run endnop3_a FN_LIB=lib_endnop3.so
run endnop3_b FN_LIB=lib_endnop3.so
