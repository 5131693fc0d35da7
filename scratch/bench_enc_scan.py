"""The encoder forward scan exactly as the training step runs it: 4 scans x B=256 x T=256, H=512, one weight-stationary launch."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
B, T, H, V = 256, 256, 512, 342
torch.manual_seed(0)
fw = []
for s in range(4):
    w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
    wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
    fw.append(dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                   gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                   h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev)))
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gru_seq_fwd(fw); e1.record(); torch.cuda.synchronize()
    print("encoder forward scan: %.3f ms per launch, %.2f us per time step" % (e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / T), flush=True)
