#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s4c10; mkdir -p $O
for grp in "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-24)
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/$tag -o p --output-format csv -- python $R/scratch/pmc_step.py 2 > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*counter_collection.csv" | head -1)
  for k in "out_head_kernel" "gemm_kernel<128, 128, 32, 2, 2, true, true>" "eg_piece_kernel"; do echo "== $k"; python $R/scratch/pmc_avg.py $f "$k"; done > $O/$tag.txt
  rm -rf $O/$tag
done
cat $O/*.txt
