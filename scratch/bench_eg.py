"""token-segment sums (embed.hip): time of fn_token_sort and fn_embed_grad_sorted at the C1 shape, for the benchmark's token
distribution, for all-equal tokens (rows of one segment are consecutive in memory) and for tokens sorted by time"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
from music_fader_nets_amd.synth import synth_batch
dev = torch.device("cuda:0"); ops = HipOps(dev)
B, T, N3, V = 256, 256, 1536, 342
def t(fn, reps=10):
    fn(); torch.cuda.synchronize(); ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return float(np.mean(ms))
dg = [torch.randn(T, B, N3, device=dev) for _ in range(4)]
outs = [torch.zeros(N3, V, device=dev) for _ in range(4)]
cases = {"bench batch": torch.from_numpy(synth_batch(np.random.RandomState(0), B, T, 64)["d"]).to(dev).to(torch.int32),
         "all token 7": torch.full((B, T), 7, dtype=torch.int32, device=dev),
         "token = t": torch.arange(T, dtype=torch.int32, device=dev).view(1, T).expand(B, T).contiguous(),
         "uniform random": torch.randint(0, V, (B, T), dtype=torch.int32, device=dev)}
for name, idx in cases.items():
    ms_sort = t(lambda: ops.token_sort(idx, V))
    h = ops.token_sort(idx, V)
    for nj in (1, 4):
        jobs = [dict(dgx=dg[i], out=outs[i], transposed=True, reverse=i & 1) for i in range(nj)]
        ms = t(lambda: ops.embed_grad_sorted(h, jobs))
        print("%-16s sort %6.1f us | %d job(s): %7.1f us  %6.2f TB/s" % (name, ms_sort * 1e3, nj, ms * 1e3, nj * T * B * N3 * 4 / ms / 1e9), flush=True)
