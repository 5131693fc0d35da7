#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests -q -m gpu -s -k "adversarial" 2>&1 | grep -E "adversarial 2|passed|failed" > gpurun_out/r5/call6_adv.log; cat gpurun_out/r5/call6_adv.log
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r5/call6_full.log 2>&1
echo "full rc=$?"; tail -6 gpurun_out/r5/call6_full.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain 0 --arith bf16x6 > gpurun_out/r5/call6_bench_x6.json 2> gpurun_out/r5/call6_bench_x6.err
echo "bench rc=$?"; grep -E "bench" gpurun_out/r5/call6_bench_x6.err | tail -12; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5/call6_bench_x6.json"))
print(d["ms_per_step"], d["arith"], d.get("fp32_mfma_ms_per_step"), d.get("fp32_mfma_leg"))
print("roofline", {k:v for k,v in d["roofline"].items() if k not in ("note","step_frac_note")})
print("scans", d.get("roofline_scans"))
for k,v in d.get("roofline_by_symbol",{}).items(): print("  SYM %-34s %8.1f us/step  frac %.3f  arith %s" % (k, v["us_per_step"], v["frac"], v.get("arith")))
for k,v in d.get("roofline_all",{}).items(): print("  %-22s %8.1f us  frac %.3f  us/step %s %s" % (k, v["avg_launch_us"], v["frac"], v.get("us_per_step"), v.get("arith")))
print("decode", d.get("decode",{}).get("value"), d.get("decode",{}).get("roofline",{}).get("frac"))
PY
