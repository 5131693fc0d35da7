#!/bin/bash
# round 4: PMC counters of the tokens-only decode kernels at 2048 rows (one rocprofv3 pass per counter group, no other tracing domain); per-kernel averages + derived figures
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_decode
rm -rf $O; mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/$tag -o p --output-format csv -- python $R/scratch/prof_decode_cells.py 2048 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    for k in "gru_cell_wlds_ovl_kernel<4, 2, true, true>" "gru_cell_wlds_ovl_kernel<4, 2, false, false>" "out_argmax_lds8_kernel"; do
      echo "== $k"; python $R/scratch/pmc_avg.py $f "$k"
    done > $O/$tag.txt
    rm -rf $O/$tag
  else echo "no csv for $grp"; tail -3 $O/$tag.log; fi
done
python $R/scratch/pmc_derive.py $O > $O/derived.txt; cat $O/derived.txt
