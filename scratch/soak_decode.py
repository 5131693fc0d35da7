"""tokens-only greedy decode (300 steps) repeated on warm buffers at several row counts: every replay must give the same tokens (hipGraph replay of the per-token cells / the one-launch pipeline)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
pkg = load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=512, z_dims=128, n_step=256, n_component=2).to(dev)
m.eval()
eng = m.engine()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for Bi in (600, 705, 800, 1000, 1024, 1025, 1280, 1536, 1537, 2000, 2048):
    z = torch.randn(Bi, 280, device=dev)
    _, t0 = pkg.greedy_decode(m, z, 300, want_logp=False)
    t0 = t0.clone()
    bad = 0
    for _ in range(reps):
        _, t = pkg.greedy_decode(m, z, 300, want_logp=False)
        bad += not torch.equal(t, t0)
    print("rows %4d: %d of %d replays differ, tokens in [%d, %d], sync error %s" % (Bi, bad, reps, int(t0.min()), int(t0.max()), eng.ops.gru_sync_error()), flush=True)
