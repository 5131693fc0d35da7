import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd.synth import synth_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
for use_graph in (False, True):
    tr = pkg.GMVAETrainer(model, lr=1e-3, beta=0.2) if use_graph is False else tr
    tr.use_graph = use_graph
    b = synth_batch(np.random.RandomState(0), 256, 256, 64)
    batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
    eps = (torch.randn(256, 128, device=dev), torch.randn(256, 128, device=dev))
    step = 20000
    for _ in range(3):
        tr.step_device(step, batch, eps); step += 1
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); tr.step_device(step, batch, eps); step += 1
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("graph=%s  host enqueue from idle %.2f ms, total %.2f ms" % (use_graph, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
