#!/bin/bash
# per-dispatch MFMA utilisation of the weight-gradient GEMMs (one rocprofv3 PMC pass over two eager training-step passes)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_gemm
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/sq -o p --output-format csv -- python $R/scratch/pmc_step.py 1 > $O/sq.log 2>&1
f=$(find $O/sq -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if "gemm_tn_kernel" in r["Kernel_Name"] or "gemm_kernel<128" in r["Kernel_Name"] or "gru_fwd_persist_kernel<4, 1, 2" in r["Kernel_Name"]:
            key = (r["Dispatch_Id"], r["Kernel_Name"][:60], r["Grid_Size"])
            rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
out = []
for (did, name, grid), c in rows.items():
    if "GRBM_GUI_ACTIVE" not in c or c["GRBM_GUI_ACTIVE"] == 0: continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    out.append((cyc, name, grid, c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, c["SQ_WAIT_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), c["SQ_WAIT_INST_ANY"] / max(1, c["SQ_WAVE_CYCLES"]), c["SQ_INSTS_MFMA"]))
for cyc, name, grid, busy, wa, wi, nm in sorted(out, reverse=True)[:30]:
    print("%10.0f cyc  grid %8s  MFMA busy %5.1f%%  wait_any %5.1f%%  wait_inst %5.1f%%  mfma_insts %12.0f  %s" % (cyc, grid, 100 * busy, 100 * wa, 100 * wi, nm, name))
PY
rm -rf $O/sq
