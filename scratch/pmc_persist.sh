#!/bin/bash
# HBM traffic + MFMA counters of the dominant kernel (weight-stationary encoder forward scan): one rocprofv3 pass per counter group
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc3
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc3/$tag -o p --output-format csv -- python $R/scratch/bench_enc_scan.py > $R/gpurun_out/pmc3/$tag.log 2>&1
  f=$(find $R/gpurun_out/pmc3/$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/scratch/pmc_avg.py $f gru_fwd_persist; else echo "no csv for $grp"; tail -2 $R/gpurun_out/pmc3/$tag.log; fi
done
