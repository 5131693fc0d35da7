#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s4c4
bash $R/scratch/pmc_step.sh > $R/gpurun_out/s4c4/pmc.log 2>&1
grep -A3 "grid 196608" $R/gpurun_out/pmc_step/*.txt
