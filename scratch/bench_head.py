import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
B, T, H, V, ld = 256, 256, 512, 342, 344
torch.manual_seed(0)
h = torch.randn(T * B, H, device=dev); W = torch.randn(V, H, device=dev) * 0.1; bias = torch.randn(V, device=dev)
tgt = torch.randint(0, V, (B, T), dtype=torch.int32, device=dev)
logits = torch.zeros(T * B, ld, device=dev); nll = torch.zeros(T * B, device=dev); dl = torch.zeros(T * B, ld, device=dev)
def t(fn, reps=8):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best * 1e3
tg = t(lambda: ops.gemm(h, W, logits[:, :V], bias=bias))
ts = t(lambda: ops.vocab_logsoftmax(logits, B, T, V, target=tgt, nll_rows=nll, grad_scale=0.1, dlogits=logits))
ops.gemm(h, W, logits[:, :V], bias=bias)
tf = t(lambda: ops.out_head(h, W, bias, B, T, tgt, nll_rows=nll, grad_scale=0.1, dlogits=dl))
tn = t(lambda: ops.out_head(h, W, bias, B, T, tgt, nll_rows=nll))
print("unfused: gemm %.1f us + log-softmax/NLL/seed %.1f us = %.1f us | fused head %.1f us (%.1f TFLOP/s on the projection) | fused, NLL only %.1f us"
      % (tg, ts, tg + ts, tf, 2.0 * T * B * V * H / tf / 1e6, tn))
