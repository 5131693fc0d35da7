"""greedy decode of Bi sequences x 100 steps (hipGraph replay): per-token scan-step kernels vs fn_gru_cell_f32 cells, to place Engine.cell_decode_rows"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
pkg = load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=512, z_dims=128, n_step=256, n_component=2).to(dev)
m.eval()
steps = 100
for Bi in (48, 64, 128, 256, 512, 768, 1024):
    z = torch.randn(Bi, 280, device=dev)
    res = []
    for thr in (1 << 30, 1):
        m.engine().cell_decode_rows = thr
        for _ in range(2):
            pkg.greedy_decode(m, z, steps, want_logp=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            pkg.greedy_decode(m, z, steps, want_logp=False)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 3 / steps * 1e6)
    print("Bi=%5d: scan-step path %.1f us/token, cell path %.1f us/token" % (Bi, res[0], res[1]), flush=True)
