"""round 4: in-kernel cycle stamps of the ping-pong scans (FN_TIMING build, scratch/build_all.sh): step 10 of workgroups 0-7, per half
[statement start, K loop done, barrier passed, epilogue issued], and how often workgroup 0 found the other half's counter short."""
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
T, H, V = 24, 512, 342
buf = (ctypes.c_ulonglong * 64)(); cb = (ctypes.c_ulonglong * 8)()
lib.fn_pdbg_read.argtypes = [ctypes.c_void_p]; lib.fn_pcnt_read.argtypes = [ctypes.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for n, Bn in ((4, 256), (2, 256)):
    fw, bw = [], []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
        d = dict(B=Bn, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                 gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (Bn, T), dtype=torch.int32, device=dev),
                 h_all=torch.zeros(T, Bn, H, device=dev), gates=torch.zeros(T, ops.gates_floats(Bn, H), device=dev))
        fw.append(d)
        bw.append(dict(B=Bn, T=T, H=H, w_hh_t_frag=wtf, h0=None, h_all=d["h_all"], gates=d["gates"], dh_ext=torch.randn(T, Bn, H, device=dev) * 0.01,
                       dgx_all=torch.zeros(T, Bn, 3*H, device=dev), dghn_all=torch.zeros(T, Bn, H, device=dev), scratch=torch.zeros(Bn, H, device=dev),
                       dgx_rowsum=torch.zeros(Bn, 3*H, device=dev), dghn_rowsum=torch.zeros(Bn, H, device=dev)))
    for rep in range(3):
        lib.fn_pcnt_read(cb); c0 = cb[0]
        (ops.gru_seq_fwd(fw) if which == "fwd" else ops.gru_seq_bwd(bw)); torch.cuda.synchronize()
        lib.fn_pdbg_read(buf); lib.fn_pcnt_read(cb)
        a = np.array(list(buf), dtype=np.int64).reshape(8, 8)
    print("%s scans=%d B=%d step 10, cycles since the start of half A's statement: [A: start, K loop done, barrier, epilogue issued | B: the same]; short polls of workgroup 0 in the last launch: %d of %d phases"
          % (which, n, Bn, cb[0] - c0, 2 * T))
    for r in (a - a[:, :1])[:4]: print("    ", r.tolist())
    print("     workgroup 0 deltas", np.diff(a[0]).tolist())
