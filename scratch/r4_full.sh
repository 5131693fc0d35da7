#!/bin/bash
# full gpu suite + the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests.log 2>&1; tail -4 gpurun_out/r4_gpu_tests.log
timeout 600 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_bench.json").read().strip().split("\n")[-1])
print("ms_per_step", d["ms_per_step"], "value", d["value"], "step_frac", d["roofline"].get("step_frac"))
for k,v in d.get("roofline_all",{}).items(): print("  %-24s %8.1f us  frac %.3f  us/step %s" % (k, v["avg_launch_us"], v["frac"], v.get("us_per_step")))
PY
