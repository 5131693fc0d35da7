#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/r5/call5_full.log 2>&1
echo "full rc=$?"; tail -6 gpurun_out/r5/call5_full.log
timeout 1200 python -m pytest tests -x -q -m gpu --x6 > gpurun_out/r5/call5_full_x6.log 2>&1
echo "full --x6 rc=$?"; tail -6 gpurun_out/r5/call5_full_x6.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode --sustain 0 > gpurun_out/r5/call5_bench.json 2> gpurun_out/r5/call5_bench.err
echo "bench rc=$?"; grep -E "bench" gpurun_out/r5/call5_bench.err | tail -8; python - <<'PY'
import json
d=json.load(open("gpurun_out/r5/call5_bench.json"))
print(d["ms_per_step"], d.get("bf16x6_opt_in"))
PY
