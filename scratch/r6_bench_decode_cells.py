"""round 6: tokens-only greedy decode of Bi sequences x 100 steps (hipGraph replay) with the per-token cells on the fp32 MFMA and on the bf16 x 6 cell kernel;
tokens of the two paths compared (first differing position per row)"""
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
for Bi in (768, 1024, 1536, 2048):
    z = torch.randn(Bi, 280, device=dev)
    toks = {}
    for x6 in (False, True):
        m.engine().ops.cell_x6, m.engine().ops.cell_x6_rows = x6, 0
        for _ in range(2):
            out = pkg.greedy_decode(m, z, steps, want_logp=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            out = pkg.greedy_decode(m, z, steps, want_logp=False)
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 5 / steps * 1e6
        toks[x6] = (out[1] if isinstance(out, tuple) else out).clone()
        print("Bi=%5d cells %-7s %.1f us/token" % (Bi, "bf16x6" if x6 else "fp32", us), flush=True)
    same = (toks[True] == toks[False])
    first = torch.where(same.all(1), torch.full((Bi,), steps, device=dev), (~same).float().argmax(1))
    print("      rows with identical tokens: %d of %d; earliest differing position %d" % (int(same.all(1).sum()), Bi, int(first.min())), flush=True)
m.engine().ops.cell_x6, m.engine().ops.cell_x6_rows = True, 2048
