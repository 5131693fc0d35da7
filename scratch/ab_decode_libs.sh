#!/bin/bash
# A/B of two library builds on the large-batch greedy decode (2048 rows): per-kernel averages (rocprofv3 --stats, eager launches) and
# us per token of the graph replay.  usage: bash scratch/ab_decode_libs.sh scratch/lib_a.so scratch/lib_b.so
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cp $R/music-fader-nets_amd/libfadernets_hip.so /tmp/lib_ship.so
for lib in "$@"; do
  tag=$(basename $lib .so)
  cp $R/$lib $R/music-fader-nets_amd/libfadernets_hip.so
  cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o d -- python $R/scratch/prof_decode.py > /dev/null 2>&1
  echo "== $tag: per-kernel averages (eager, 2048 rows)"; python $R/scratch/prof_summary.py $O/prof_$tag/d_results.db 8
  rm -rf $O/prof_$tag
  cd $R
  timeout 300 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from mfn_import import load_package
pkg = load_package()
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev); m.eval()
for Bi in (800, 1024, 2048):
    z = torch.randn(Bi, 280, device=dev)
    eng = m.engine(); eng.single_launch_decode = False; eng.cell_decode_rows = 512
    pkg.greedy_decode(m, z, 300, want_logp=False); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        t0 = time.perf_counter(); _, tk = pkg.greedy_decode(m, z, 300, want_logp=False); torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print("   rows %4d cells path: %.1f us per token  (token checksum %d)" % (Bi, min(ms) / 300 * 1e3, int(tk.long().sum())))
PY
done
cp /tmp/lib_ship.so $R/music-fader-nets_amd/libfadernets_hip.so
