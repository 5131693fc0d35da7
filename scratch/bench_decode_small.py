"""single-launch decode vs the per-token-kernel path: same tokens? time per step?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
pkg = load_package()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=512, z_dims=128, n_step=256, n_component=2).to(dev)
m.eval()
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for Bi, steps in ((1, 100), (8, 100), (16, 300), (24, 100), (32, 300)):
    z = torch.randn(Bi, 280, device=dev)
    m.engine().single_launch_decode = False
    lp0, tk0 = pkg.greedy_decode(m, z, steps)
    t0 = t(lambda: pkg.greedy_decode(m, z, steps))
    m.engine().single_launch_decode = True
    lp1, tk1 = pkg.greedy_decode(m, z, steps)
    t1 = t(lambda: pkg.greedy_decode(m, z, steps))
    same = bool((tk0 == tk1).all())
    first_diff = int((tk0 != tk1).any(0).nonzero()[0]) if not same else -1
    print("Bi=%3d steps=%3d: tokens equal=%s (first differing step %d), max|dlogp| %.2e | per-token kernels %.1f us/step, single launch %.1f us/step, sync_err=%s" %
          (Bi, steps, same, first_diff, float((lp0 - lp1).abs().max()), t0 * 1e3 / steps, t1 * 1e3 / steps, m.engine().ops.gru_sync_error()), flush=True)
