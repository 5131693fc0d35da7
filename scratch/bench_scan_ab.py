"""Encoder-shaped (4 scans x 256 rows -> 128-row groups) and decoder-shaped (2 scans x 256 rows -> 64-row groups) scans, forward and backward,
for an A/B of two library builds in one session: python scratch/bench_scan_ab.py scratch/lib_x.so"""
import os, sys, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    shutil.copy(os.path.join(R, sys.argv[1]), os.path.join(R, "music-fader-nets_amd/libfadernets_hip.so"))
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
ops.variant = int(os.environ.get("FN_VARIANT", "0"))          # FnGruFwd.variant of every scan launch (tiling experiments)
H, V = 512, 342
def mk(n, B, T, dense):
    torch.manual_seed(0)
    fw, bw = [], []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
        d = dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.randn(3*H, device=dev) * 0.1, b_ih=torch.randn(3*H, device=dev) * 0.1,
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
        if dense and s == 1: d["gx_dense"] = torch.randn(T, B, 3*H, device=dev) * 0.3
        else: d.update(gx_table=torch.randn(V, 3*H, device=dev) * 0.3, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev))
        fw.append(d)
        bw.append(dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=None, h_all=d["h_all"], gates=d["gates"], dh_ext=torch.randn(T, B, H, device=dev) * 0.01,
                       dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), scratch=torch.zeros(B, H, device=dev),
                       dgx_rowsum=torch.zeros(B, 3*H, device=dev), dghn_rowsum=torch.zeros(B, H, device=dev)))
    return fw, bw
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for name, (n, B, T, dense) in (("encoder shape 4x256 rows T=256", (4, 256, 256, False)), ("decoder shape 2x256 rows T=64", (2, 256, 64, True))):
    fw, bw = mk(n, B, T, dense)
    tf = t(lambda: ops.gru_seq_fwd(fw)); tb = t(lambda: ops.gru_seq_bwd(bw))
    chk = sum(float(d["h_all"].double().sum()) for d in fw), sum(float(d["dgx_all"].double().sum()) for d in bw)
    print("%-34s fwd %.3f ms (%.2f us/step)  bwd %.3f ms (%.2f us/step)  checksums %.6f %.6f  err %d" % (name, tf, tf * 1e3 / T, tb, tb * 1e3 / T, chk[0], chk[1], ops.gru_sync_error(False)), flush=True)
