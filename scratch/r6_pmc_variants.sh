#!/bin/bash
# cycles (not time) of the x6w GEMM with parts switched off: is the "loads + MFMAs are additive" effect clock (DVFS) or stall cycles?
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6/pmc_variants
mkdir -p $O
cat > /tmp/dwhh_v.py <<'PY'
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd import _lib
if os.environ.get("FN_LIB"):
    _lib.LIB_PATH = os.path.join(R, "scratch", os.environ["FN_LIB"])
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev); H = 512; rows = 65536
torch.manual_seed(0)
dgx, dghn, hp = torch.randn(rows, 3 * H, device=dev), torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev) * 0.3
dW = torch.zeros(3 * H, H, device=dev)
ops.dw_x6, ops.x6_wide = True, False
for _ in range(6):
    ops.gru_dwhh(dgx, dghn, hp, dW, splitk=16)
torch.cuda.synchronize()
PY
for lib in "" lib_noloadcut.so lib_noload.so lib_nocut.so lib_nomma.so lib_loadsonly.so; do
    echo "=== ${lib:-product}"
    FN_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/p -o p --output-format csv -- python /tmp/dwhh_v.py > $O/p.log 2>&1
    f=$(find $O/p -name "*counter_collection.csv" | head -1)
    python $R/scratch/pmc_avg.py "$f" gemm_tn_x6w_kernel
    FN_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d $O/q -o q --output-format csv -- python /tmp/dwhh_v.py > $O/q.log 2>&1
    f=$(find $O/q -name "*kernel_stats.csv" | head -1); grep x6w "$f" | cut -c1-200
    rm -rf $O/p $O/q
done
