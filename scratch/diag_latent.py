import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import *
from fake_ops import FakeOps
pkg = load_package()
gold = load_golden("small")
dev = "cuda:0"
m = make_model(64, 32, sd_from(gold, "w0/"), device=dev)
tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
b = batch_of(gold)
batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"], None)
eps = (torch.from_numpy(gold["eps_r"]).to(dev), torch.from_numpy(gold["eps_n"]).to(dev))
dl_sd, lat_up, w, beta0, Bg = tr._forward_losses(20000, batch, eps, want_grads=True)
eng = m.engine()
S = eng.saved
# f64 forward
torch.set_default_dtype(torch.float64)
sd64 = {k: v.double() for k, v in sd_from(gold, "w0/").items()}
fw64 = orc.forward(sd64, torch.from_numpy(b["d"]), torch.from_numpy(b["r"]), torch.from_numpy(b["n"]), torch.from_numpy(b["c"]).double(), torch.from_numpy(gold["eps_r"]).double(), torch.from_numpy(gold["eps_n"]).double())
torch.set_default_dtype(torch.float32)
for e in "rn":
    for nm, key in (("ll", "ll_"+e), ("qy", "qy_"+e), ("z", "z_"+e), ("mu","mu_"+e), ("sigma","sigma_"+e)):
        gpu = S["lat"][e][nm].cpu().double().numpy() if nm not in ("mu",) else S["pre"][e][:, :32].cpu().double().numpy()
        ref = gold["fw_"+key].astype(np.float64); ex = fw64[key].numpy()
        print("%-8s gpu-f64 %.3e  ref-f64 %.3e   (abs, max|x| %.3e)" % (key, np.abs(gpu-ex).max(), np.abs(ref-ex).max(), np.abs(ex).max()))
print("qy_n f64:\n", fw64["qy_n"].numpy())
# backward kernel vs fake on identical inputs
fake = FakeOps()
for e in "rn":
    pre = S["pre"][e]; z = S["lat"][e]["z"]; qy = S["lat"][e]["qy"]
    gz = torch.randn(6, 32)
    P = eng.p
    dd, md = torch.zeros(6, 64, device=dev), torch.zeros(6, 64, device=dev)
    eng.ops.latent_bwd(pre, S["eps"][e], P["mu_%s_lookup.weight"%e], P["logvar_%s_lookup.weight"%e], None, z, qy, gz.to(dev), None, None, None, None, w[0], w[1], w[2], dd, md)
    dc, mc = torch.zeros(6, 64), torch.zeros(6, 64)
    fake.latent_bwd(pre.cpu(), S["eps"][e].cpu(), P["mu_%s_lookup.weight"%e].cpu(), P["logvar_%s_lookup.weight"%e].cpu(), None, z.cpu(), qy.cpu(), gz, None, None, None, None, w[0], w[1], w[2], dc, mc)
    print(e, "latent_bwd kernel vs fake(autograd): dpre", relerr(dd.cpu().numpy(), dc.numpy()), "dmu_rows", relerr(md.cpu().numpy(), mc.numpy()))
    # fake in float64
    dc64, mc64 = torch.zeros(6, 64, dtype=torch.float64), torch.zeros(6, 64, dtype=torch.float64)
    torch.set_default_dtype(torch.float64)
    fake.latent_bwd(pre.cpu().double(), S["eps"][e].cpu().double(), P["mu_%s_lookup.weight"%e].cpu().double(), P["logvar_%s_lookup.weight"%e].cpu().double(), None, z.cpu().double(), qy.cpu().double(), gz.double(), None, None, None, None, w[0], w[1], w[2], dc64, mc64)
    torch.set_default_dtype(torch.float32)
    print(e, "   kernel vs fake64: dpre", relerr(dd.cpu().numpy(), dc64.numpy()), " fake32 vs fake64", relerr(dc.numpy(), dc64.numpy()))
