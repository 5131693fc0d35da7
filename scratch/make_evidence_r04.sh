#!/bin/bash
# round 4 evidence set: full gpu suite, bench line (20 steps + 200 sustained), rocprofv3 kernel stats + timeline of the same command, PMC passes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
if [ "$1" != "notests" ]; then
  timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
fi
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --sustain 0 --no-x6 --no-cpu-baseline --no-decode > $O/prof_bench.log 2>&1
python $R/scratch/prof_summary.py $O/prof/bench_results.db 45 > $O/kernel_stats.txt
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 100 3 > $O/timeline.txt
rm -rf $O/prof
bash $R/scratch/pmc_step_r04.sh > $O/pmc.log 2>&1
cp -r $R/gpurun_out/pmc_step $O/
tail -25 $O/tests.log; tail -3 $O/bench_train.err; python - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04/bench_train.json").read().strip().split("\n")[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "sustained", d.get("sustained_ms_per_step"), "step_frac", d["roofline"].get("step_frac"))
print("roofline", {k:v for k,v in d["roofline"].items() if k not in ("note",)})
print("scans", d.get("roofline_scans"))
for k,v in d.get("roofline_by_symbol",{}).items(): print("  SYM %-40s %8.1f us/step  frac %.3f" % (k, v["us_per_step"], v["frac"]))
for k,v in d.get("roofline_all",{}).items(): print("  %-24s %8.1f us  frac %.3f  us/step %s" % (k, v["avg_launch_us"], v["frac"], v.get("us_per_step")))
print("cpu", d.get("cpu_baseline")); print("decode", {k:v for k,v in d.get("decode",{}).items() if k!="roofline"}, d.get("decode",{}).get("roofline",{}).get("frac"))
PY
cat $O/pmc_step/derived.txt
