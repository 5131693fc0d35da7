#!/bin/bash
# PMC passes over the forward 4-scan step kernel (separate runs per counter group, kernel-trace only)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|name)\s*:\s*\S+|^[A-Z][A-Za-z0-9_]+" | head -0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  FN_FWD_CFG=${CFG:-2,0,3} rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc/$tag -o p --output-format csv -- python $R/scratch/bench_scan.py fwd4only > $R/gpurun_out/pmc/$tag.log 2>&1
  f=$(find $R/gpurun_out/pmc/$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/scratch/pmc_avg.py $f gru_fwd_step; else echo "no csv for $grp"; tail -3 $R/gpurun_out/pmc/$tag.log; fi
done
