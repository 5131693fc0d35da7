"""Soak: N graph replays of the benchmark training step + M fader-sweep decodes; the sticky sync-error word must stay clear, losses finite."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd.synth import synth_batch
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
b = synth_batch(np.random.RandomState(0), 256, 256, 64)
batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
torch.manual_seed(99); eps = tr.draw_eps(256, 256)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
step = 20000
t0 = time.perf_counter()
for i in range(N):
    tr.step_device(step, batch, eps); step += 1
    if (i + 1) % 500 == 0:
        torch.cuda.synchronize()
        t8 = tr._tuple8(0.2, 256, False)
        assert np.isfinite(t8[0]), t8
        assert not m.engine().ops.gru_sync_error(), "sync error after %d steps" % (i + 1)
        print("step %d: loss %.4f, %.3f ms/step" % (i + 1, t8[0], (time.perf_counter() - t0) / (i + 1) * 1e3), flush=True)
m.eval()
z = torch.randn(2048, 280, device=dev)
for i in range(20):
    lp, tok = pkg.greedy_decode(m, z, 300, want_logp=False)
torch.cuda.synchronize()
assert not m.engine().ops.gru_sync_error()
for rows in (8, 256, 800, 1536):          # one launch: one block | 32-row blocks | 64-row blocks (7 / 12 per replica)
    zs = torch.randn(rows, 280, device=dev)
    for i in range(50 if rows == 8 else 15):
        lp, tok = pkg.greedy_decode(m, zs, 300, want_logp=False)
    torch.cuda.synchronize()
    assert not m.engine().ops.gru_sync_error(), rows
print("soak ok: %d steps, 20 large-batch decodes (staged-GEMM cells), 50 + 3 x 15 single-launch decodes of 8 / 256 / 800 / 1536 rows x 300 steps" % N)
