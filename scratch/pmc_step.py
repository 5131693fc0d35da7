"""eager (no hipGraph) forward+backward passes of the training step at the benchmark shape: the process rocprofv3 --pmc profiles
(scratch/pmc_step.sh).  PMC collection serialises dispatches, so the per-kernel counters are those of the kernel running alone."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd.synth import synth_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
if len(sys.argv) > 2:                      # arithmetic of the deep MFMA products: f32 / bf16x6 (default: the package default)
    m.set_arith(sys.argv[2])
tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
tr.use_graph = False
b = synth_batch(np.random.RandomState(0), 256, 256, 64)
batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
torch.manual_seed(99); eps = tr.draw_eps(256, 256)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    tr.loss_and_grads(20000, batch, eps)
torch.cuda.synchronize()
print("done", tr._tuple8(0.2, 256, False)[0])
