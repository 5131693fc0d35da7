#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/soak_x6
scratch/r5_soak_x6.sh 1 30
scratch/r5_soak_x6.sh 31 31 AMD_SERIALIZE_KERNEL=3
