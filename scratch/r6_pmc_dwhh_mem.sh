#!/bin/bash
# memory-side PMC counters of the dW_hh product (1536 x 512 x 65536, 16 K ranges): per-wave kernel vs producer / consumer kernel
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6/pmc_dwhh_mem
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
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/p$i -o p --output-format csv -- python /tmp/dwhh_one.py > $O/p$i.log 2>&1
    f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
    if [ -z "$f" ]; then echo "group '$grp' failed:"; tail -3 $O/p$i.log; continue; fi
    for k in gemm_tn_x6_kernel gemm_tn_x6w_kernel; do echo "== $k"; python $R/scratch/pmc_avg.py "$f" $k; done
    rm -rf $O/p$i
done
