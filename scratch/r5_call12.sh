#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r5/call12_full.log 2>&1
echo "full rc=$?"; tail -4 gpurun_out/r5/call12_full.log
timeout 1500 python -m pytest tests -x -q -m gpu --x6 > gpurun_out/r5/call12_full_x6.log 2>&1
echo "full --x6 rc=$?"; tail -4 gpurun_out/r5/call12_full_x6.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --arith bf16x6 --no-decode > gpurun_out/r5/call12_bench_x6.json 2> gpurun_out/r5/call12_bench_x6.err
echo "bench rc=$?"; grep -E "bench" gpurun_out/r5/call12_bench_x6.err | tail -6; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5/call12_bench_x6.json"))
print(d["ms_per_step"], d.get("fp32_mfma_ms_per_step"))
for k,v in d.get("roofline_by_symbol",{}).items(): print("  SYM %-34s %8.1f us/step  frac %.3f  arith %s" % (k, v["us_per_step"], v["frac"], v.get("arith")))
for k,v in d.get("roofline_all",{}).items(): print("  %-22s %8.1f us  frac %.3f  us/step %s %s" % (k, v["avg_launch_us"], v["frac"], v.get("us_per_step"), v.get("arith")))
PY
