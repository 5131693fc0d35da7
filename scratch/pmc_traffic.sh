#!/bin/bash
# HBM traffic of the dominant kernel (encoder forward step, 4 scans): FETCH_SIZE and WRITE_SIZE in SEPARATE passes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc2
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout 90 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc2/$tag -o p --output-format csv -- python $R/scratch/bench_scan.py fwd4only > $R/gpurun_out/pmc2/$tag.log 2>&1
  f=$(find $R/gpurun_out/pmc2/$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/scratch/pmc_avg.py $f gru_fwd_step; else echo "no csv for $grp"; tail -2 $R/gpurun_out/pmc2/$tag.log; fi
done
