import os, sys, shutil, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
shutil.copy(os.path.join(R, "scratch/lib_timing.so"), os.path.join(R, "music-fader-nets_amd/libfadernets_hip.so"))
import torch, numpy as np
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
from music_fader_nets_amd import _lib
dev = torch.device("cuda:0"); ops = HipOps(dev)
lib = _lib.load()
B, T, H, V = 256, 8, 512, 342
def mk(n):
    fw = []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        fw.append(dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                 h0=torch.randn(B, H, device=dev) * 0.1, gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev)))
    return fw
buf = (ctypes.c_ulonglong * 512)()
for n in (1, 4):
    fw = mk(n)
    for rep in range(3):
        ops.gru_seq_fwd(fw); torch.cuda.synchronize()
    lib.fn_dbg_read.argtypes = [ctypes.c_void_p]; lib.fn_dbg_read(buf)
    a = np.array(list(buf), dtype=np.int64).reshape(64, 8)[:, :5]
    d = a - a[:, :1]
    print("scans=%d  per-WG cycles since kernel entry of wave0 (median over 64 WGs): prologue_issued %d, loop_done %d, reduced %d, barrier %d, end %d ; WG start spread %d cycles" %
          (n, *np.median(d[:, 1:], axis=0).astype(int), 0 if False else int(np.median(d[:,4])), int(a[:, 0].max() - a[:, 0].min())))
    print("   (counter ticks; 100 MHz or shader clock?) first rows:", d[:3].tolist())
