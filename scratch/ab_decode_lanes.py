"""round 4: large-batch greedy decode (staged-GEMM cells) with the batch cut into 1-4 row ranges on their own streams - us per token"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
pkg = load_package()
dev = torch.device("cuda:0")
torch.manual_seed(1234)
m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
m.eval()
eng = m.engine()
eng.single_launch_decode = False
eng.cell_decode_rows = 512
steps = 300
for Bi in (1024, 1536, 2048, 4096):
    z = torch.randn(Bi, 280, device=dev)
    ref = None
    for lanes in (1, 2, 3, 4, 1, 2):
        eng.decode_lanes = lanes
        _, tk = pkg.greedy_decode(m, z, steps, want_logp=False)
        torch.cuda.synchronize()
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            _, tk = pkg.greedy_decode(m, z, steps, want_logp=False)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        if ref is None:
            ref = tk.clone()
        print("rows %5d lanes %d: %7.2f ms per decode = %6.1f us per token   tokens identical to 1 lane: %s" % (Bi, lanes, min(ms), min(ms) / steps * 1e3, bool(torch.equal(ref, tk))), flush=True)
