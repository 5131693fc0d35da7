import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import *
from fake_ops import FakeOps
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = "cuda:0"; ops = HipOps(torch.device(dev)); fake = FakeOps()
torch.manual_seed(0)
B, Z, K = 3, 32, 2
pre = torch.randn(B, 2*Z)*0.1; eps = torch.randn(B, Z)
mu_lk = torch.randn(K, Z)*0.01; lv = torch.full((K, Z), -4.0)
names = ("sigma","z","ll","qy","y","terms")
o = dict(sigma=torch.zeros(B,Z), z=torch.zeros(B,Z), ll=torch.zeros(B,K), qy=torch.zeros(B,K), y=torch.zeros(B,dtype=torch.int32), terms=torch.zeros(B,4))
fake.latent_fwd(pre, eps, mu_lk, lv, None, *[o[k] for k in names])
print("qy", o["qy"])
g = lambda t: None if t is None else t.to(dev)
for tag, ups, w in (("w_lat", [None]*5, (0.03, 0.0, 0.0)), ("w_cls", [None]*5, (0.0, 0.03, 0.0)),
                    ("g_qy", [None, None, None, None, torch.randn(B, K)], (0.0, 0.0, 0.0)),
                    ("g_ll", [None, None, None, torch.randn(B, K), None], (0.0, 0.0, 0.0)),
                    ("g_z", [torch.randn(B, Z), None, None, None, None], (0.0, 0.0, 0.0))):
    dc, mc = torch.zeros(B, 2*Z), torch.zeros(B, K*Z)
    dd, md = g(dc.clone()), g(mc.clone())
    fake.latent_bwd(pre, eps, mu_lk, lv, None, o["z"], o["qy"], *ups, *w, dc, mc)
    ops.latent_bwd(g(pre), g(eps), g(mu_lk), g(lv), None, g(o["z"]), g(o["qy"]), *[g(u) for u in ups], *w, dd, md)
    print(tag, "dpre", relerr(dd.cpu().numpy(), dc.numpy()), "dmu", relerr(md.cpu().numpy(), mc.numpy()), "| max", float(dc.abs().max()), float(mc.abs().max()))
    if tag in ("w_cls", "g_qy"):
        print("   row0 kernel", dd[0, :4].cpu().numpy(), "fake", dc[0, :4].numpy())
