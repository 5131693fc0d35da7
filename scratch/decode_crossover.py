"""one-launch block pipeline (fn_decode_greedy) against the per-token cells (fn_gru_cell_f32, automatic kernel choice) around their crossover: us per token"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
pkg = load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=512, z_dims=128, n_step=256, n_component=2).to(dev)
m.eval()
eng = m.engine()
eng.single_launch_skip = (0, -1)
steps = 100
def t(z):
    for _ in range(2): pkg.greedy_decode(m, z, steps, want_logp=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): pkg.greedy_decode(m, z, steps, want_logp=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3 / steps * 1e6
for Bi in (512, 640, 768, 800, 896, 1024, 1152, 1280, 1408, 1536, 2048):
    z = torch.randn(Bi, 280, device=dev)
    eng.single_launch_decode, eng.single_launch_rows, eng.cell_decode_rows = True, 4096, 1 << 30
    a = t(z)
    eng.single_launch_decode, eng.cell_decode_rows = False, 1
    c = t(z)
    print("Bi=%5d: one launch %.1f us/token | cells %.1f   sync error %s" % (Bi, a, c, eng.ops.gru_sync_error()), flush=True)
