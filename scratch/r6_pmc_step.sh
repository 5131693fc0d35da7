#!/bin/bash
# round 6: PMC counters of the training step's kernels (benchmark shape, arithmetic of the package default unless ARITH is set), one rocprofv3 pass per
# counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; no tracing domain besides --kernel-trace).  Output: gpurun_out/pmc_r06/<group>.csv (raw
# per-dispatch rows), then scratch/r6_pmc_json.py turns them into r06_pmc_traffic.json (what bench.py reads) and r06_pmc_training_step.txt.
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r06
rm -rf $O; mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/$tag -o p --output-format csv -- python $R/scratch/pmc_step.py 2 ${ARITH:-} > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then cp $f $O/$tag.csv; rm -rf $O/$tag; else echo "no csv for $grp"; tail -3 $O/$tag.log; fi
done
python $R/scratch/r6_pmc_json.py $O $O/r06_pmc_traffic.json > $O/r06_pmc_training_step.txt; cat $O/r06_pmc_training_step.txt
rm -f $O/*.csv
