"""Benchmark-shape training step (eager, loss_and_grads) while another stream HOLDS compute units (fn_occupy_cus: N workgroups with 64 KB LDS,
relaunched back to back for the whole step): bit-identity of the gradients, sync-error word, slow-down.  -> profiles/r03_cu_contention.txt"""
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
ops = m.engine().ops
side = torch.cuda.Stream()
def run(blocks, hold_ms):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if blocks:
        with torch.cuda.stream(side):
            for _ in range(int(hold_ms / 0.5)):
                ops.occupy_cus(blocks, 64 * 1024, 1_000_000)
    t = tr.loss_and_grads(20000, batch, eps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t, tr.flat.grad.clone(), dt
run(0, 0); t_ref, g_ref, dt_ref = run(0, 0)
print("alone: %.2f ms (eager fwd+bwd, host-timed)" % (dt_ref * 1e3))
for blocks in (8, 32):
    for rep in range(2):
        t, g, dt = run(blocks, 40)
        print("%3d CUs held CONTINUOUSLY for ~40 ms: step %.2f ms (x%.2f), bit-identical %s, sync error %s" % (blocks, dt * 1e3, dt / dt_ref, bool(t == t_ref and torch.equal(g, g_ref)), ops.gru_sync_error()), flush=True)
# bursts: what a gradient bucket looks like - ~0.3 ms of residency every ~3 ms (a 1-thread spin kernel in between holds no CU resources)
def run_bursts(blocks, n_bursts, hold_cycles, gap_cycles):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(n_bursts):
            ops.occupy_cus(blocks, 64 * 1024, hold_cycles)
            torch.cuda._sleep(gap_cycles)
    t = tr.loss_and_grads(20000, batch, eps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return t, tr.flat.grad.clone(), dt
for blocks in (8, 16, 32):
    for rep in range(3):
        t, g, dt = run_bursts(blocks, 9, 600_000, 6_000_000)
        print("%3d CUs held in 9 bursts of ~0.3 ms, ~3 ms apart: step %.2f ms (x%.3f), bit-identical %s, sync error %s" % (blocks, dt * 1e3, dt / dt_ref, bool(t == t_ref and torch.equal(g, g_ref)), ops.gru_sync_error()), flush=True)
