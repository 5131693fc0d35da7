#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag9; rm -rf $out; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 1500 python scratch/r5_bursts_diag.py f32 1500 eager > $out/$name.log 2>&1; echo "$name: $(grep 'repetitions differ' $out/$name.log)"; }
run fence1 DIAG_VARIANT=0x100
run fence2 DIAG_VARIANT=0x100
run plain X=1
