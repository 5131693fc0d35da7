"""In-kernel cycle stamps of one time step (step 10) of the weight-stationary scans with the flag-in-data hand-over (FN_TIMING build,
scratch/build_all.sh): ticks since the top of the step for the first lane of three workgroups, plus how often the ring was
(re)requested and how many probe polls the step took."""
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
buf = (ctypes.c_ulonglong * 128)()
lib.fn_pdbg_read.argtypes = [ctypes.c_void_p]
ORDER = [8, 0, 1, 7, 9, 2, 3, 4, 5, 6]
NAMES = "top, (a) issued, k-entry, ring arrived(last), probe done(last), kloop done, barrier, published, outputs, refill"
def show(tag):
    lib.fn_pdbg_read(buf)
    a = np.array(list(buf), dtype=np.int64).reshape(8, 16)
    d = a[:, ORDER] - a[:, 8:9]
    print("%s  [%s] | ring requests, polls" % (tag, NAMES))
    for r, full in zip(d[:4], a[:4]):
        print("    ", r.tolist(), "|", int(full[10]), int(full[11]))
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
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gru_seq_fwd(fw); e1.record(); torch.cuda.synchronize()
        show("fwd scans=%d B=%d rep %d: %.2f us/step" % (n, Bn, rep, e0.elapsed_time(e1) * 1e3 / T))
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gru_seq_bwd(bw); e1.record(); torch.cuda.synchronize()
        show("bwd scans=%d B=%d rep %d: %.2f us/step" % (n, Bn, rep, e0.elapsed_time(e1) * 1e3 / T))
    assert not ops.gru_sync_error()
