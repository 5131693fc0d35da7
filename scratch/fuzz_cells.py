"""round 4: random row counts through the large-batch decode kernels - fn_gru_cell_f32 automatic choice (LDS-free / LDS-resident-slice / fills under the K loops) against its staged
form (variant 8) and fp64, fn_out_argmax_f32 against fp64 logits; H = 512 (and a few other sizes), ragged row blocks, both layers' argument sets, repeated launches on warm buffers"""
import os, sys, random
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0); torch.manual_seed(1)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
worst = 0.0; bad = 0; nt = 0
V = 342
for case in range(n):
    H = 512 if case % 5 else random.choice([64, 128, 256, 1024])
    B = random.choice([513, 514, 777, 1023, 1024, 1025, 1300, 1535, 1536, 1537, 2047, 2048, 2049, 3000]) if case % 3 == 0 else random.randint(513, 2100)
    layer2 = bool(case & 1)
    hp = torch.randn(B, H, device=dev) * 0.5
    whh = torch.randn(3 * H, H, device=dev) / H ** 0.5; bhh = torch.randn(3 * H, device=dev) * 0.1; bih = torch.randn(3 * H, device=dev) * 0.1
    kw = dict(b_ih=bih)
    if layer2:
        kw.update(x=torch.randn(B, H, device=dev), w_ih=torch.randn(3 * H, H, device=dev) / H ** 0.5)
    else:
        best = torch.zeros(B, dtype=torch.int64, device=dev)
        hh = torch.randn(B, 64, device=dev); Wo = torch.randn(V, 64, device=dev) / 8; bo = torch.randn(V, device=dev) * 0.1
        ops.out_argmax(hh, Wo, bo, best)
        ref_tok = (hh.double() @ Wo.double().t() + bo.double())
        top2 = ref_tok.topk(2, dim=1)
        clear = (top2.values[:, 0] - top2.values[:, 1]) > 1e-4
        tok = torch.zeros(B, 1, dtype=torch.int32, device=dev); ops.best_tokens(best.view(1, B), V, tok)
        if not torch.equal(tok[:, 0].long()[clear], top2.indices[:, 0][clear]):
            bad += 1; print("ARGMAX MISMATCH", B)
        nt += 1
        kw.update(gx_table=torch.randn(V, 3 * H, device=dev) * 0.3, gx_rowbias=torch.randn(B, 3 * H, device=dev) * 0.3, idx_best=best, best_v=V)
    o0 = torch.zeros(B, H, device=dev); o1 = torch.zeros(B, H, device=dev); o2 = torch.zeros(B, H, device=dev)
    ops.gru_cell(hp, whh, bhh, o0, variant=0, **kw)
    ops.gru_cell(hp, whh, bhh, o2, variant=0, **kw)
    ops.gru_cell(hp, whh, bhh, o1, variant=8, **kw)
    e = float((o0 - o1).abs().max())
    worst = max(worst, e)
    if e > 2e-5 or not torch.equal(o0, o2):
        bad += 1; print("MISMATCH B=%d H=%d layer2=%d err %.3g repeat-equal %s" % (B, H, layer2, e, torch.equal(o0, o2)))
torch.cuda.synchronize()
print("%d cases (%d with the packed-argmax token path), %d mismatches, worst |automatic - staged| %.3g" % (n, nt, bad, worst))
