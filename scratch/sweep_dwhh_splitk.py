"""fn_gru_dwhh_f32 at the attribute decoders' shape (rows = Tr * B = 16384, H = 512: [1536 x 16384] x [16384 x 512]) and the full-length shape (65536 rows): us per launch by K-split depth"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H = 512
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for rows, sks in ((16384, (4, 5, 8, 10, 11, 16, 21, 32)), (65536, (16, 21, 32))):
    dgx = torch.randn(rows, 3 * H, device=dev); dghn = torch.randn(rows, H, device=dev); hp = torch.randn(rows, H, device=dev); dW = torch.zeros(3 * H, H, device=dev)
    flop = 2.0 * rows * 3 * H * H
    ref = None
    for sk in sks:
        us = t(lambda: ops.gru_dwhh(dgx, dghn, hp, dW, splitk=sk))
        if ref is None: ref = dW.clone()
        err = float((dW - ref).abs().max() / ref.abs().max())
        print("rows %6d splitk %3d (%4d workgroups): %7.1f us = %.3f of the fp32 MFMA peak   rel diff to the first %.1e" % (rows, sk, 48 * sk, us, flop / (us * 1e-6) / 157.3e12, err), flush=True)
