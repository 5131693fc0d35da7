"""One single-rank RCCL run of the captured training step (FN_FORCE_DIST=1 FN_DP_GRAPH=1): exit code 0 = the capture worked and 3 replays ran.
argv: flags 'noside' (loss terms / regulariser gather on the main lane), 'syncbuckets' (gradient buckets as blocking collectives)"""
import os, sys
os.environ.update(FN_FORCE_DIST="1", FN_DP_GRAPH="1")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd import parallel
from music_fader_nets_amd.synth import synth_batch
ctx, local = parallel.init_from_env("nccl")
dev = torch.device("cuda:%d" % local)
if "syncbuckets" in sys.argv:
    import torch.distributed as dist
    def start_bucket(self, flat_view):
        if flat_view.numel():
            dist.all_reduce(flat_view, op=dist.ReduceOp.SUM, group=self.group)
    parallel.DataParallelContext.start_bucket = start_bucket
torch.manual_seed(1234)
m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2, dist_ctx=ctx)
if "noside" in sys.argv:
    m.engine().losses_on_side = False
b = synth_batch(np.random.RandomState(0), 256, 256, 64)
batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
torch.manual_seed(99); eps = tr.draw_eps(256, 256)
step = 20000
import warnings
with warnings.catch_warnings():
    warnings.simplefilter("error")                 # the trainer warns when the capture fails
    for _ in range(5):
        tr.step_device(step, batch, eps); step += 1
torch.cuda.synchronize()
assert tr.use_graph and len(tr._graphs) == 1
torch.distributed.barrier(); torch.distributed.destroy_process_group()
