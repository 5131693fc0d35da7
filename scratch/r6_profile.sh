#!/bin/bash
# round 6: bench line (full JSON), rocprofv3 kernel statistics + timeline of 3 captured steps -> gpurun_out/r6/prof_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6/prof_${1:-x}
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python bench.py ${BENCH_ARGS:---no-cpu-baseline --no-decode} > $O/bench_train.json 2> $O/bench_train.err; tail -3 $O/bench_train.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --sustain 0 --no-other-arith --no-cpu-baseline --no-decode > $O/prof_bench.log 2>&1
python $R/scratch/prof_summary.py $O/prof/bench_results.db 45 > $O/kernel_stats.txt
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 100 3 > $O/timeline.txt
rm -rf $O/prof
python - <<PY
import json
d=json.loads(open("$O/bench_train.json").read().strip().split("\n")[-1])
print("ms_per_step", d["ms_per_step"], "f32 leg", d.get("fp32_mfma_ms_per_step"), "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
for k,v in d.get("roofline_by_symbol",{}).items(): print("  SYM %-40s n=%5.1f avg %7.1f us  %8.1f us/step  frac %.3f %s" % (k, v["launches_per_step"], v["avg_launch_us"], v["us_per_step"], v["frac"], v.get("arith","")))
for k,v in d.get("roofline_all",{}).items(): print("  %-26s %8.1f us  frac %.3f  us/step %s %s" % (k, v["avg_launch_us"], v["frac"], v.get("us_per_step"), v.get("arith","")))
PY
