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
B, T, H, V = 256, 64, 512, 342
torch.manual_seed(0)
fw, bw = [], []
for s in range(4):
    w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
    wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
    wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
    d = dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.randn(3*H, device=dev) * 0.1, b_ih=torch.randn(3*H, device=dev) * 0.1,
             gx_table=torch.randn(V, 3*H, device=dev) * 0.3, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
             h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
    fw.append(d)
    bw.append(dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=None, h_all=d["h_all"], gates=d["gates"], dh_last=torch.randn(B, H, device=dev) * 0.1,
                   dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), scratch=torch.zeros(B, H, device=dev),
                   dgx_rowsum=torch.zeros(B, 3*H, device=dev), dghn_rowsum=torch.zeros(B, H, device=dev)))
buf = (ctypes.c_ulonglong * 64)()
lib.fn_pdbg_read.argtypes = [ctypes.c_void_p]
ops.gru_seq_fwd(fw, variant=0x400); ops.gru_seq_bwd(bw, variant=0x400); torch.cuda.synchronize()
names = ["dgx_all", "dghn_all", "dgx_rowsum", "dghn_rowsum"]
ref = [d[k].clone() for d in bw for k in names]
for v in [int(x, 16) for x in sys.argv[1:]] or [0, 0x4000]:
    for kind in ("fwd", "bwd"):
        for rep in range(2):
            for b in bw: b["dgx_rowsum"].zero_(); b["dghn_rowsum"].zero_()
            (ops.gru_seq_fwd(fw, variant=v) if kind == "fwd" else ops.gru_seq_bwd(bw, variant=v)); torch.cuda.synchronize()
            lib.fn_pdbg_read(buf)
            a = np.array(list(buf), dtype=np.int64).reshape(8, 8)[:, [0, 1, 7, 2, 3, 4, 5, 6]] // T
            print("%s variant %x rep %d: per wave, mean cycles per step in [top of step, counter wait, token wait, K loop, transposition, epilogue, store drain, arrive]; sum" % (kind, v, rep))
            for r in a: print("    ", r.tolist(), int(r.sum()))
    out = [d[k] for d in bw for k in names]
    for i, (x, y) in enumerate(zip(ref, out)):
        df = (x - y).abs()
        if df.max().item() > 0:
            print("   bwd output %s of scan %d: max abs diff %.3e (max abs %.3e), differing %d of %d" % (names[i % 4], i // 4, df.max().item(), x.abs().max().item(), int((df > 0).sum()), df.numel()))
    print("   bwd outputs compared")
