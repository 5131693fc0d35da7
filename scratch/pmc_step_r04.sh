#!/bin/bash
# round 4: PMC counters of the training step's kernels (benchmark shape), one rocprofv3 pass per counter group (FETCH_SIZE and WRITE_SIZE do
# not fit one pass; no tracing domains besides --kernel-trace).  Output: gpurun_out/pmc_step/<group>.txt = per-kernel averages, per launch
# shape (template instance) AND per symbol (all instances merged: what bench.py's roofline_by_symbol / `roofline.traffic` use).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_step
rm -rf $O; mkdir -p $O
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/$tag -o p --output-format csv -- python $R/scratch/pmc_step.py 2 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    for k in "gru_fwd_pp_kernel<1>" "gru_fwd_pp_kernel<2>" "gru_bwd_rs_kernel<2>" "gru_bwd_rs_kernel<1>" "gru_fwd_pp_kernel" "gru_bwd_rs_kernel" "gemm_tn_kernel" "gemm_kernel<128" "gemm_nt_direct_kernel<1" "gemm_nt_direct_kernel<4" "out_head_kernel" "eg_piece_kernel"; do
      echo "== $k"; python $R/scratch/pmc_avg.py $f "$k"
    done > $O/$tag.txt
    echo "== gemm_tn_kernel grid 196608 = 48 tiles x 16 K ranges (dW_hh / dW_ih2 at K = 65280-65536 rows and, since the 16-range split, the attribute decoders at K = 16384)" >> $O/$tag.txt; python $R/scratch/pmc_avg.py $f "gemm_tn_kernel" 196608 >> $O/$tag.txt
    rm -rf $O/$tag
  else echo "no csv for $grp"; tail -3 $O/$tag.log; fi
done
python $R/scratch/pmc_derive.py $O > $O/derived.txt; cat $O/derived.txt
