#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/c3_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/c3_tests.log; tail -4 gpurun_out/c3_tests.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode 2>>gpurun_out/c3_bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], {k:(v['frac'],v['us_per_step']) for k,v in d['roofline_all'].items() if 'gemm' in k})"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c3_prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode > $GRAFT_REPO_ROOT/gpurun_out/c3_prof.log 2>&1
python $GRAFT_REPO_ROOT/scratch/prof_summary.py $GRAFT_REPO_ROOT/gpurun_out/c3_prof/bench_results.db 45 > $GRAFT_REPO_ROOT/gpurun_out/c3_kernel_stats.txt
python $GRAFT_REPO_ROOT/scratch/prof_timeline.py $GRAFT_REPO_ROOT/gpurun_out/c3_prof/bench_results.db 100 3 > $GRAFT_REPO_ROOT/gpurun_out/c3_timeline.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/c3_prof
head -1 $GRAFT_REPO_ROOT/gpurun_out/c3_kernel_stats.txt; tail -14 $GRAFT_REPO_ROOT/gpurun_out/c3_timeline.txt
