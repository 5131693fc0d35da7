#!/bin/bash
# round 4: data-parallel step through real RCCL with one rank - eager launches (the default with peers since round 4) against the captured step; burst test output
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
for g in 1 0; do
  FN_FORCE_DIST=1 FN_DP_GRAPH=$g timeout 300 python bench.py --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-decode > $O/bench_rccl1_graph$g.json 2> $O/bench_rccl1_graph$g.err
  python - <<PY
import json
d=json.loads(open("$O/bench_rccl1_graph$g.json").read().strip().split("\n")[-1])
print("FN_FORCE_DIST=1 FN_DP_GRAPH=$g: %.3f ms/step  comm %s" % (d["ms_per_step"], {k: v for k, v in d.get("comm", {}).items() if k != "note"}))
PY
  grep "host enqueue" $O/bench_rccl1_graph$g.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-decode 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('no collectives: %.3f ms/step' % d['ms_per_step'])"
timeout 300 python -m pytest tests/test_parallel_rccl.py -q -s -m gpu -k "bursts" 2>&1 | grep -E "bursts|passed|failed"
