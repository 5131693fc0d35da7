import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
B, H, V = 2048, 512, 342
torch.manual_seed(0)
hp = torch.randn(B, H, device=dev) * 0.5; x = torch.randn(B, H, device=dev) * 0.5
whh = torch.randn(3*H, H, device=dev) / 22; wih = torch.randn(3*H, H, device=dev) / 22
bhh = torch.randn(3*H, device=dev) * 0.1; bih = torch.randn(3*H, device=dev) * 0.1
tab = torch.randn(V, 3*H, device=dev) * 0.3; rb = torch.randn(B, 3*H, device=dev) * 0.3
toks = torch.randint(0, V, (B, 8), dtype=torch.int32, device=dev)
out = torch.zeros(B, H, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best * 1e3
for v, name in ((1, "128 rows, waves 4x1"), (2, "128 rows, waves 2x2"), (3, "64 rows, waves 4x1"), (0, "64 rows, waves 2x2")):
    t1 = t(lambda: ops.gru_cell(hp, whh, bhh, out, b_ih=bih, gx_table=tab, idx=toks[:, 3], gx_rowbias=rb, variant=v))
    t2 = t(lambda: ops.gru_cell(hp, whh, bhh, out, x=x, w_ih=wih, b_ih=bih, variant=v))
    print("%-22s layer 1 cell (token row + h W_hh) %.1f us = %.1f TFLOP/s | layer 2 cell (x W_ih + h W_hh) %.1f us = %.1f TFLOP/s"
          % (name, t1, 2.0 * B * 3 * H * H / t1 / 1e6, t2, 4.0 * B * 3 * H * H / t2 / 1e6), flush=True)
