#!/bin/bash
# round 4, final state: full gpu suite (default arithmetic and --x6), bench line (20 steps + 200 sustained, decode leg, bf16x6 leg, cpu baselines)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu --durations=10 > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 1800 python -m pytest tests -x -q -m gpu --x6 > $O/tests_x6.log 2>&1; echo "tests --x6 rc=$?" >> $O/tests_x6.log
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
tail -4 $O/tests.log; tail -3 $O/tests_x6.log; tail -3 $O/bench_train.err
python - <<'PY'
import json,os
d=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r04/bench_train.json").read().strip().split("\n")[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "sustained", d.get("sustained_ms_per_step"))
print("x6", {k:v for k,v in d.get("bf16x6_opt_in",{}).items() if k in ("ms_per_step","value")})
print("decode", {k:v for k,v in d.get("decode",{}).items() if k in ("value","ms_per_pass")}, d.get("decode",{}).get("roofline",{}).get("frac"), d.get("decode",{}).get("roofline",{}).get("avg_launch_us"))
PY
