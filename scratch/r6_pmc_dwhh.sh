#!/bin/bash
# PMC counters of the dW_hh product (1536 x 512 x 65536, 16 K ranges): round-5 per-wave kernel vs producer / consumer kernel.  Separate passes per counter group.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6/pmc_dwhh
mkdir -p $O
cat > /tmp/dwhh_one.py <<'PY'
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev); H = 512; rows = 65536
torch.manual_seed(0)
dgx, dghn, hp = torch.randn(rows, 3 * H, device=dev), torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev) * 0.3
dW = torch.zeros(3 * H, H, device=dev)
for pw in (True, False):
    ops.dw_x6, ops.x6_perwave = True, pw
    for _ in range(4):
        ops.gru_dwhh(dgx, dghn, hp, dW, splitk=16)
torch.cuda.synchronize()
PY
i=0
for grp in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o p --output-format csv -- python /tmp/dwhh_one.py > $O/p$i.log 2>&1
    f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
    for k in gemm_tn_x6_kernel gemm_tn_x6w_kernel; do echo "== $k"; python $R/scratch/pmc_avg.py "$f" $k; done
    rm -rf $O/p$i
done
