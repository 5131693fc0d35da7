"""In-kernel cycle stamps of the role workgroups of fn_decode_greedy (FN_TIMING build, scratch/build_all.sh) at step 10 of block 0:
where a 32-row block's time goes in each role (L1, P2, L2, OUT, ARG).  Ticks relative to the layer-1 role's first stamp."""
import os, sys, shutil, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
shutil.copy(os.path.join(R, "scratch/lib_timing.so"), os.path.join(R, "music-fader-nets_amd/libfadernets_hip.so"))
import torch, numpy as np
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=512, z_dims=128, n_step=256, n_component=2).to(dev)
m.eval(); m.engine().single_launch_rows = 2048
buf = (ctypes.c_ulonglong * 80)()
lib.fn_ddbg_read.argtypes = [ctypes.c_void_p]
NAMES = ["L1 : top, C1 ok, operands+MFMA, reduced, token (C4 / C5) ok, stored, arrived | next block's top",
         "P2 : top, C1 ok, arrived | next top", "L2 : top, C3 ok, MFMA+reduced, C2 ok, arrived | next top", "OUT: top, C3 ok, arrived | next top", "ARG: top, C4 ok, arrived | next top"]
IDX = [[0, 1, 2, 3, 5, 6, 7, 8], [0, 1, 7, 8], [0, 1, 3, 4, 7, 8], [0, 1, 7, 8], [0, 1, 7, 8]]
for Bi in (256, 800, 1536):
    z = torch.randn(Bi, 280, device=dev)
    for _ in range(2):
        pkg.greedy_decode(m, z, 40, want_logp=False)
    torch.cuda.synchronize()
    lib.fn_ddbg_read(buf)
    a = np.array(list(buf), dtype=np.int64).reshape(5, 16)
    print("Bi=%d" % Bi)
    for role in range(5):                 # ticks relative to the role's own first stamp of block 0
        print("   %-100s %s" % (NAMES[role], (a[role, IDX[role]] - a[role, 0]).tolist()))
