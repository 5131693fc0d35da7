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
torch.manual_seed(0)
fw = []
for s in range(4):
    w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
    wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
    fw.append(dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.randn(3*H, device=dev) * 0.1, b_ih=torch.randn(3*H, device=dev) * 0.1,
             gx_table=torch.randn(V, 3*H, device=dev) * 0.3, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
             h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev)))
buf = (ctypes.c_ulonglong * 64)()
lib.fn_pdbg_read.argtypes = [ctypes.c_void_p]
ops.gru_seq_fwd(fw, variant=0x400); torch.cuda.synchronize()
ref = [d["h_all"].clone() for d in fw] + [d["gates"].clone() for d in fw]
for sg in (255,):
  for tokbit in (0x1000, 0x1800):
    for rep in range(2):
        ops.gru_seq_fwd(fw, variant=(sg << 16) | tokbit); torch.cuda.synchronize()
        lib.fn_pdbg_read(buf)
        a = np.array(list(buf), dtype=np.int64).reshape(8, 8)[:, [0, 1, 7, 2, 3, 4, 5, 6]]
        base = a[:, 0].min()
        print("variant bits %x" % tokbit, "stagger %d rep %d: per wave, cycles since the earliest step start [start, polled, first chunk, kloop, transposed, epilogue, drained, arrived]" % (sg, rep))
        for r in a: print("    ", (r - base).tolist())
    out = [d["h_all"] for d in fw] + [d["gates"] for d in fw]
    for i, (x, y) in enumerate(zip(ref, out)):
        df = (x - y).abs()
        print("   output %d: max abs diff %.3e, differing elements %d of %d, first differing step %s" % (i, df.max().item(), int((df > 0).sum()), df.numel(),
              (df.view(T, -1).amax(1) > 0).nonzero().flatten()[:3].tolist()))
