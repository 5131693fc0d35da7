import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd.synth import synth_batch
from music_fader_nets_amd.decode import greedy_decode
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev).eval()
for Bi, steps in ((8, 100), (256, 300), (2048, 300)):
    z = torch.randn(Bi, 280, device=dev)
    greedy_decode(model, z, steps, want_logp=False); torch.cuda.synchronize()
    t0 = time.perf_counter(); lp, tok = greedy_decode(model, z, steps, want_logp=False); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("Bi=%5d steps=%d: %.1f ms total (host enqueue %.1f ms) -> %.2f M tokens/s, %.1f us/step" % (Bi, steps, (t2-t0)*1e3, (t1-t0)*1e3, Bi*steps/(t2-t0)/1e6, (t2-t0)/steps*1e6), flush=True)
