"""Single-rank data-parallel soak: N fresh processes, each builds the communicator (RCCL through fn_comm_*), captures the benchmark
training step WITH its collectives into one hipGraph and replays it; every run must capture (no fallback to eager launches), stay finite and
leave the sync-error word clear.  usage: python scratch/soak_dp.py [runs] -> profiles/r04_dp_graph_soak.txt
(round 2: the same through torch.distributed's process group failed 6 of 150 captures and took the watchdog thread down)"""
import json, os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
CHILD = r"""
import os, sys, warnings
sys.path.insert(0, %r)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd import parallel
from music_fader_nets_amd.synth import synth_batch
ctx, local = parallel.init_from_env()
dev = torch.device("cuda", local)
torch.manual_seed(1234)
m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2, dist_ctx=ctx)
b = synth_batch(np.random.RandomState(0), 256, 256, 64)
batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
torch.manual_seed(99); eps = tr.draw_eps(256, 256)
step = 20000
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    for i in range(6):
        tr.step_device(step, batch, eps); step += 1
torch.cuda.synchronize()
t8 = tr._tuple8(0.2, 256, False)
ok = bool(np.isfinite(t8[0])) and tr.use_graph and len(tr._graphs) == 1 and ctx.rccl is not None and not m.engine().ops.gru_sync_error()
print("RESULT", int(ok), round(t8[0], 4), [str(x.message)[:80] for x in w])
ctx.rccl.close(); torch.distributed.destroy_process_group()
""" % R
env = dict(os.environ, FN_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
bad, t0 = 0, time.time()
for i in range(N):
    env["MASTER_PORT"] = str(29600 + i % 200)
    p = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    ok = p.returncode == 0 and line and line[0].split()[1] == "1"
    if not ok:
        bad += 1
        print("run %d FAILED rc=%d %s %s" % (i, p.returncode, line, p.stderr[-400:]), flush=True)
    if (i + 1) % 10 == 0:
        print("%d runs, %d failures, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
print("soak_dp: %d single-rank runs (graph-captured step with RCCL collectives through fn_comm_*), %d failures" % (N, bad))
