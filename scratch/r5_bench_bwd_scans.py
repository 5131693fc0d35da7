"""round 5: backward scans at the two launch shapes of the training step - encoder (4 scans x 256 rows x 256 steps, 64-row groups) and one decoder-pipeline
launch (2 scans x 256 rows x 32 steps, 32-row groups, external gradient + dh0) - on the fp32 MFMA (register-stationary ping-pong) and on the bf16 x 6
kernel; HIP events on the launch stream, each launch alone; max difference of the gate gradients."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd import _lib
if os.environ.get("FN_LIB"):
    _lib.LIB_PATH = os.path.join(R, "scratch", os.environ["FN_LIB"])
    print("library:", _lib.LIB_PATH)
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H = 512
torch.manual_seed(0)


def mk(n, B, T, enc):
    fw, bw = [], []
    for s in range(n):
        w = (torch.randn(3 * H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3 * H, H), device=dev); ops.frag_pack(w, wf)
        wt = torch.zeros(ops.frag_floats(H, 3 * H), device=dev)
        wt3 = torch.zeros(ops.frag_floats(H, 3 * H) * 3 // 2, device=dev)
        ops.weight_images([("frag_t", w, wt), ("frag3_t", w, wt3)])
        h0 = None if enc else torch.randn(B, H, device=dev) * 0.1
        f = dict(B=B, T=T, H=H, w_hh_frag=wf, b_hh=torch.zeros(3 * H, device=dev), b_ih=torch.zeros(3 * H, device=dev), h0=h0,
                 gx_dense=torch.randn(T, B, 3 * H, device=dev) * 0.3, h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
        fw.append(f)
        bw.append(dict(B=B, T=T, H=H, w_hh_t_frag=wt, w_hh_t_frag3=wt3, h0=h0, h_all=f["h_all"], gates=f["gates"],
                       dh_last=torch.randn(B, H, device=dev) if enc else None, dh_ext=None if enc else torch.randn(T, B, H, device=dev) * 0.1,
                       dgx_all=torch.zeros(T, B, 3 * H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), dh0=None if enc else torch.zeros(B, H, device=dev),
                       dgx_rowsum=torch.zeros(B, 3 * H, device=dev), dghn_rowsum=torch.zeros(B, H, device=dev), scratch=torch.zeros(B, H, device=dev)))
    ops.gru_seq_fwd(fw)
    return bw


def timeit(scans, x6, reps=6):
    ops.dw_x6 = x6
    ms = []
    for rep in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gru_seq_bwd(scans); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ops.dw_x6 = False
    assert not ops.gru_sync_error()
    return min(ms[1:]), sum(ms[1:]) / (reps - 1)


for name, scans, T in (("encoder 4 x 256 rows x 256 steps", mk(4, 256, 256, True), 256), ("decoder launch 2 x 256 rows x 32 steps", mk(2, 256, 32, False), 32),
                       ("decoder launch 2 x 256 rows x 64 steps", mk(2, 256, 64, False), 64)):
    ref = None
    for tag, x6 in (("fp32 register-stationary", False), ("bf16x6", True)):
        best, mean = timeit(scans, x6)
        g = scans[0]["dgx_all"].clone()
        if ref is None:
            ref = g
        print("%-42s %-26s best %.3f ms  mean %.3f ms  = %.2f us per step   max |dgx - dgx_fp32| / max |dgx| = %.2e" % (name, tag, best, mean, best * 1e3 / T,
              float((g - ref).abs().max()) / float(ref.abs().max())), flush=True)
