import os, sys, faulthandler
faulthandler.enable()
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, numpy as np
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd import synth
dev = torch.device("cuda:0")
H, Z, B, T, Tr = 64, 32, 6, 20, 8
torch.manual_seed(0)
m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=H, z_dims=Z, n_step=T, n_component=2).to(dev)
tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
tr.use_graph = os.environ.get("GRAPH", "1") == "1"
d = torch.randint(0, 342, (B, T), device=dev); r = torch.randint(0, 3, (B, Tr), device=dev); n = torch.randint(0, 16, (B, Tr), device=dev)
c = torch.rand(B, 24, device=dev); rd = torch.rand(B, device=dev); nd = torch.rand(B, device=dev)
step = 100
for it in range(4):
    print("step", it, flush=True)
    step, tup = tr.train(step, None, None, None, d, r, n, c, rd, nd)
    torch.cuda.synchronize()
    print("  ->", tup[0], flush=True)
