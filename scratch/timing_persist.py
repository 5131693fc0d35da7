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
B, T, H, V = 256, 24, 512, 342
def mk(n):
    fw = []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        fw.append(dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                 h0=torch.randn(B, H, device=dev) * 0.1, gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev)))
    return fw
buf = (ctypes.c_ulonglong * 64)()
lib.fn_pdbg_read.argtypes = [ctypes.c_void_p]
for n in (1, 2, 4):
    fw = mk(n)
    for rep in range(3):
        ops.gru_seq_fwd(fw); torch.cuda.synchronize()
        lib.fn_pdbg_read(buf)
        a = np.array(list(buf), dtype=np.int64).reshape(8, 8)[:, :7]
        d = a - a[:, :1]
        print("scans=%d rep %d  step-10 stamps (ticks since step start) [gx issued, polled, kloop, red, epi, drained, arrived]:" % (n, rep))
        for r in d[:3]: print("    ", r.tolist())
