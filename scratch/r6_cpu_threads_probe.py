"""round 6: the CPU baseline's thread count: ONE train step of the dense-one-hot torch.nn restatement at B = 32, T = 256 with 32 / 64 / 96 / 128
threads, each in its own process under a 150 s limit (256 threads took 2814 s in round 4), plain and under `numactl`-style pinning through the scheduler affinity
(first N cores = the first socket's)."""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
n = int(sys.argv[1]); pin = sys.argv[2] == "pin"
if pin:
    os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:n]))
import numpy as np, torch
torch.set_num_threads(n)
from mfn_import import load_package
load_package()
from oracle import cpu_baseline as cb, gmvae_oracle as orc
from importlib import import_module
H, Z, B, T, Tr = 512, 128, 32, 256, 64
sd = orc.init_state_dict(H, Z)
model = cb.build(sd, H, Z)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
synth = import_module("music_fader_nets_amd.synth")
b = synth.synth_batch(np.random.RandomState(1), B, T, Tr)
er, en = torch.randn(B, Z), torch.randn(B, Z)
cb.train_step(model, opt, synth.synth_batch(np.random.RandomState(2), 4, 16, 4), torch.randn(4, Z), torch.randn(4, Z), 20000)
t0 = time.perf_counter(); cb.train_step(model, opt, b, er, en, 20000); t1 = time.perf_counter()
cb.train_step(model, opt, b, er, en, 20001); t2 = time.perf_counter()
print("threads %%3d %%-4s: first step %%.2f s, second %%.2f s -> %%.0f tokens/s" %% (n, "pin" if pin else "", t1 - t0, t2 - t1, B * T / (t2 - t1)), flush=True)
''' % R
print("usable cores:", len(os.sched_getaffinity(0)), flush=True)
for n in (32, 64, 96, 128):
    for pin in ("", "pin"):
        try:
            r = subprocess.run([sys.executable, "-c", CHILD, str(n), pin], capture_output=True, text=True, timeout=150)
            print((r.stdout.strip().splitlines() or ["(no output) " + r.stderr.strip()[-200:]])[-1], flush=True)
        except subprocess.TimeoutExpired:
            print("threads %3d %-4s: > 150 s" % (n, pin), flush=True)
