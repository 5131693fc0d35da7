"""per-kernel picture of the staged-GEMM-cell greedy decode at Bi rows (rocprofv3 --kernel-trace --stats -- python scratch/prof_decode_cells.py Bi)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
pkg = load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
Bi = int(sys.argv[1]); variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=512, z_dims=128, n_step=256, n_component=2).to(dev)
m.eval()
eng = m.engine()
eng.single_launch_decode, eng.cell_decode_rows = False, 1
eng.ops.cell_variant = variant
z = torch.randn(Bi, 280, device=dev)
steps = 100
eng.fused_argmax = os.environ.get("FUSED", "1") == "1"
for _ in range(2): pkg.greedy_decode(m, z, steps, want_logp=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): pkg.greedy_decode(m, z, steps, want_logp=False)
torch.cuda.synchronize()
print("Bi=%d cell variant %d: %.1f us/token" % (Bi, variant, (time.perf_counter() - t0) / 3 / steps * 1e6), flush=True)
