#!/bin/bash
# full gpu suite + bench line + contention numbers + a first short DP soak
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/c2_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/c2_tests.log
tail -15 gpurun_out/c2_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/c2_bench.json 2> gpurun_out/c2_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/c2_bench.err; cut -c1-1500 gpurun_out/c2_bench.json
FN_FORCE_DIST=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode > gpurun_out/c2_bench_dist1.json 2> gpurun_out/c2_bench_dist1.err; echo "bench dist rc=$?"; tail -3 gpurun_out/c2_bench_dist1.err; cut -c1-600 gpurun_out/c2_bench_dist1.json
timeout 600 python scratch/contention.py > gpurun_out/c2_contention.txt 2>&1; cat gpurun_out/c2_contention.txt
timeout 1500 python scratch/soak_dp.py 20 > gpurun_out/c2_soak_dp.txt 2>&1; tail -5 gpurun_out/c2_soak_dp.txt
