#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/call35; rm -rf $out; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "placement or pipeline or live_at or backward_scan or forward_scan" -p no:cacheprovider > $out/tests.log 2>&1; tail -3 $out/tests.log
timeout 1500 python scratch/r5_bursts_diag.py f32 1500 eager > $out/eager_mitigated.log 2>&1; grep "repetitions differ" $out/eager_mitigated.log
timeout 1500 python scratch/r5_bursts_diag.py bf16x6 1500 none > $out/dp_x6_mitigated.log 2>&1; grep "repetitions differ" $out/dp_x6_mitigated.log
