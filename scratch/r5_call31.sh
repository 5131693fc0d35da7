#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag8; rm -rf $out; mkdir -p $out
run() { name=$1; shift; env "$@" timeout 1500 python scratch/r5_bursts_diag.py f32 1500 eager > $out/$name.log 2>&1; echo "$name: $(grep 'repetitions differ' $out/$name.log)"; grep -A1 DIFFERS $out/$name.log | grep -v "^--\|DIFFERS" | cut -c1-60 | sort | uniq -c; }
run serialize DIAG_SERIALIZE=1
run older_loops DIAG_VARIANT=0x2000
run nofill DIAG_NOFILL=1
