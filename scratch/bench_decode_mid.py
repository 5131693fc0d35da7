"""greedy decode of Bi sequences x 100 steps: ONE launch (fn_decode_greedy, block pipeline above 32 rows) vs the per-token kernels (graph replay)
vs the staged-GEMM cells - us per token.  -> profiles/r03_decode_block_pipeline.txt"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import shutil
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
if libs:          # A/B of builds inside one gpurun call
    shutil.copy(os.path.join(R, libs[0]), os.path.join(R, "music-fader-nets_amd/libfadernets_hip.so"))
only_one = "--one" in sys.argv
import torch
from mfn_import import load_package
pkg = load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=512, z_dims=128, n_step=256, n_component=2).to(dev)
m.eval()
eng = m.engine()
eng.single_launch_skip = (0, -1)
eng.single_launch_rows = 4096          # measure the one-launch pipeline over its whole range, whatever the default threshold
steps = 100
def t(z):
    for _ in range(2): pkg.greedy_decode(m, z, steps, want_logp=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): pkg.greedy_decode(m, z, steps, want_logp=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 3 / steps * 1e6
for Bi in (8, 32, 64, 128, 192, 256, 320, 384, 448, 512, 640, 704, 800, 1024, 1536, 2048):
    z = torch.randn(Bi, 280, device=dev)
    eng.single_launch_decode, eng.cell_decode_rows = True, 1 << 30
    a = t(z)
    b = c = float("nan")
    if not only_one:
        eng.single_launch_decode = False
        b = t(z)
        eng.cell_decode_rows = 1
        c = t(z)
    print("Bi=%5d: one launch %.1f us/token | per-token kernels %.1f | staged-GEMM cells %.1f   sync error %s" % (Bi, a, b, c, eng.ops.gru_sync_error()), flush=True)
