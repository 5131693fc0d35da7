"""Driver of tests/test_host_twins.py: runs in a subprocess with the AddressSanitizer runtime preloaded (libfadernets_host.so is an ASAN
build) and pushes the `small` golden fixture + torch-autograd references through the HOST twins of the C ABI via ctypes."""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from mfn_import import load_package  # noqa: E402

load_package()
from music_fader_nets_amd import _lib as L  # noqa: E402  (structure definitions only: the HIP library is NOT loaded)

lib = C.CDLL(os.path.join(ROOT, "music-fader-nets_amd", "libfadernets_host.so"))
lib.fn_frag_floats_host.restype = C.c_size_t
lib.fn_gru_gates_floats_host.restype = C.c_size_t
vp = C.c_void_p


def P(a):
    return None if a is None else vp(a.ctypes.data)


def f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def frag(mat):
    mat = f32(mat)
    out = np.zeros(lib.fn_frag_floats_host(*mat.shape), np.float32)
    assert lib.fn_frag_pack_host(P(mat), mat.shape[0], mat.shape[1], mat.shape[1], P(out), None) == 0
    return out


def close(a, b, tol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    e = np.abs(a - b).max() / max(1e-12, np.abs(b).max())
    assert e < tol, "%s: rel err %.3e" % (what, e)


def scan_fwd(B, T, H, w_hh, b_hh, b_ih, table, idx, reverse, gates=True, h0=None):
    d = L.FnGruFwd()
    keep = dict(wf=frag(w_hh), b_hh=f32(b_hh), b_ih=f32(b_ih), table=f32(table), idx=np.ascontiguousarray(idx, np.int32),
                h_all=np.zeros((T, B, H), np.float32), gates=np.zeros((T, lib.fn_gru_gates_floats_host(B, H)), np.float32) if gates else None,
                ws=np.zeros(2 * lib.fn_frag_floats_host(B, H), np.float32), h0=None if h0 is None else f32(h0))
    d.B, d.T, d.H, d.reverse = B, T, H, reverse
    d.w_hh_frag, d.b_hh, d.b_ih, d.gx_table, d.idx, d.idx_ld = P(keep["wf"]), P(keep["b_hh"]), P(keep["b_ih"]), P(keep["table"]), P(keep["idx"]), idx.shape[1]
    d.h_all, d.gates, d.frag_ws, d.h0 = P(keep["h_all"]), P(keep["gates"]), P(keep["ws"]), P(keep["h0"])
    assert lib.fn_gru_seq_fwd_host(C.byref(d), 1, None) == 0
    return keep


def test_small_golden_encoder_and_latent():
    g = np.load(os.path.join(HERE, "golden", "small.npz"))
    H, Z, K, B, T, Tr = (int(x) for x in g["meta_dims"])
    for e in ("r", "n"):
        hs = []
        for sfx, rev in (("_l0", 0), ("_l0_reverse", 1)):
            w_ih = g["w0/gru_%s.weight_ih%s" % (e, sfx)]                      # [3H][342]: the one-hot projection is a row gather of W_ih^T
            k = scan_fwd(B, T, H, g["w0/gru_%s.weight_hh%s" % (e, sfx)], g["w0/gru_%s.bias_hh%s" % (e, sfx)], g["w0/gru_%s.bias_ih%s" % (e, sfx)],
                         w_ih.T, g["d"], rev)
            hs.append(k["h_all"][T - 1])
        x = np.concatenate(hs, 1)                                             # [h_fwd | h_rev] (gmm_model.py:85)
        mu = x @ g["w0/mu_%s.weight" % e].T + g["w0/mu_%s.bias" % e]
        v = x @ g["w0/var_%s.weight" % e].T + g["w0/var_%s.bias" % e]
        close(mu, g["fw_mu_" + e], 2e-5, "mu_" + e)
        close(np.exp(v), g["fw_sigma_" + e], 2e-5, "sigma_" + e)
        pre = f32(np.concatenate([mu, v], 1))
        eps = f32(g["eps_" + e])
        mk, lv = f32(g["w0/mu_%s_lookup.weight" % e]), f32(g["w0/logvar_%s_lookup.weight" % e])
        sigma, z = np.zeros((B, Z), np.float32), np.zeros((B, Z), np.float32)
        ll, qy, y, terms = np.zeros((B, K), np.float32), np.zeros((B, K), np.float32), np.zeros(B, np.int32), np.zeros((B, 4), np.float32)
        assert lib.fn_latent_fwd_host(P(pre), P(eps), P(mk), P(lv), B, Z, K, None, P(sigma), P(z), P(ll), P(qy), P(y), P(terms), None) == 0
        close(z, g["fw_z_" + e], 2e-5, "z_" + e)
        close(ll, g["fw_ll_" + e], 2e-5, "ll_" + e)
        close(qy, g["fw_qy_" + e], 5e-3, "qy_" + e)                            # softmax of ~1e3 log-likelihoods: fp32-ill-conditioned in the reference itself
        assert np.array_equal(y, g["fw_y_" + e])
    print("small golden: encoder scans + heads + latent block through the host twins OK")


def test_gru_backward_vs_autograd():
    torch.manual_seed(0)
    B, T, H, V = 5, 6, 32, 11
    w_hh, b_hh, b_ih = torch.randn(3 * H, H) / math.sqrt(H), torch.randn(3 * H) * 0.1, torch.randn(3 * H) * 0.1
    table, idx = torch.randn(V, 3 * H) * 0.5, torch.randint(0, V, (B, T))
    h0 = (torch.randn(B, H) * 0.3).requires_grad_(True)
    gx = [(table[idx[:, t]] + b_ih).requires_grad_(True) for t in range(T)]
    h, hs = h0, []
    for t in range(T):
        gh = h @ w_hh.t() + b_hh
        r, z = torch.sigmoid(gx[t][:, :H] + gh[:, :H]), torch.sigmoid(gx[t][:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gx[t][:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        hs.append(h)
    dh_ext, dh_last = torch.randn(T, B, H), torch.randn(B, H)
    loss = sum((hs[t] * dh_ext[t]).sum() for t in range(T)) + (hs[-1] * dh_last).sum()
    grads = torch.autograd.grad(loss, [h0] + gx)
    k = scan_fwd(B, T, H, w_hh.numpy(), b_hh.numpy(), b_ih.numpy(), table.numpy(), idx.numpy(), 0, h0=h0.detach().numpy())
    close(k["h_all"], torch.stack(hs).detach().numpy(), 2e-6, "h_all")
    d = L.FnGruBwd()
    keep = dict(wt=frag(w_hh.t().contiguous().numpy()), dh_last=f32(dh_last), dh_ext=f32(dh_ext), dgx=np.zeros((T, B, 3 * H), np.float32),
                dghn=np.zeros((T, B, H), np.float32), dh0=np.zeros((B, H), np.float32), rs=np.zeros((B, 3 * H), np.float32),
                scr=np.zeros((B, H), np.float32), ws=np.zeros(2 * lib.fn_frag_floats_host(B, 3 * H), np.float32))
    d.B, d.T, d.H = B, T, H
    d.w_hh_t_frag, d.h0, d.h_all, d.gates = P(keep["wt"]), P(k["h0"]), P(k["h_all"]), P(k["gates"])
    d.dh_last, d.dh_ext, d.dgx_all, d.dghn_all, d.dh0 = P(keep["dh_last"]), P(keep["dh_ext"]), P(keep["dgx"]), P(keep["dghn"]), P(keep["dh0"])
    d.dgx_rowsum, d.scratch, d.frag_ws = P(keep["rs"]), P(keep["scr"]), P(keep["ws"])
    assert lib.fn_gru_seq_bwd_host(C.byref(d), 1, None) == 0
    close(keep["dh0"], grads[0].numpy(), 5e-6, "dh0")
    close(keep["dgx"], torch.stack(grads[1:]).numpy(), 5e-6, "dgx_all")
    close(keep["rs"], torch.stack(grads[1:]).sum(0).numpy(), 5e-6, "dgx_rowsum")
    print("GRU backward twin vs torch autograd OK")


def test_latent_backward_vs_autograd():
    from fake_ops import FakeOps
    torch.manual_seed(1)
    B, Z, K = 7, 16, 2
    pre, eps = torch.randn(B, 2 * Z) * 0.3, torch.randn(B, Z)
    mk, lv = torch.randn(K, Z) * 0.5, torch.full((K, Z), -1.0) + torch.randn(K, Z) * 0.1
    g_z = torch.randn(B, Z)
    fake = FakeOps()
    for labels, w3 in ((None, torch.tensor([0.03, 0.02, 0.0])), (torch.randint(0, K, (B,), dtype=torch.int32), torch.tensor([0.03, 0.0, 0.05]))):
        sig, z, ll, qy = torch.zeros(B, Z), torch.zeros(B, Z), torch.zeros(B, K), torch.zeros(B, K)
        y, terms = torch.zeros(B, dtype=torch.int32), torch.zeros(B, 4)
        fake.latent_fwd(pre, eps, mk, lv, labels, sig, z, ll, qy, y, terms)
        o = [np.zeros((B, Z), np.float32), np.zeros((B, Z), np.float32), np.zeros((B, K), np.float32), np.zeros((B, K), np.float32),
             np.zeros(B, np.int32), np.zeros((B, 4), np.float32)]
        a = [f32(pre), f32(eps), f32(mk), f32(lv)]
        lab = None if labels is None else np.ascontiguousarray(labels.numpy())
        assert lib.fn_latent_fwd_host(P(a[0]), P(a[1]), P(a[2]), P(a[3]), B, Z, K, P(lab), *[P(x) for x in o], None) == 0
        close(o[5], terms.numpy(), 1e-5, "terms")
        dpre_ref, dmu_ref = torch.zeros(B, 2 * Z), torch.zeros(B, K * Z)
        fake.latent_bwd(pre, eps, mk, lv, labels, z, qy, g_z, None, None, None, None, w3, dpre_ref, dmu_ref)
        dpre, dmu = np.zeros((B, 2 * Z), np.float32), np.zeros((B, K * Z), np.float32)
        gz, w = f32(g_z), f32(w3)
        assert lib.fn_latent_bwd_host(P(a[0]), P(a[1]), P(a[2]), P(a[3]), B, Z, K, P(lab), P(o[1]), P(o[3]), P(gz), None, None, None, None, P(w),
                                      P(dpre), P(dmu), None) == 0
        close(dpre, dpre_ref.numpy(), 2e-4, "dpre")
        close(dmu, dmu_ref.numpy(), 2e-4, "dmu_lk_rows")
    print("latent forward / backward twins vs autograd OK")


def test_out_head_and_adam():
    torch.manual_seed(2)
    B, T, H, V, ld = 3, 4, 32, 21, 24
    h, W, bias = torch.randn(T * B, H), (torch.randn(V, H) * 0.3).requires_grad_(True), torch.randn(V) * 0.1
    hh = h.clone().requires_grad_(True)
    target = torch.randint(0, V, (B, T), dtype=torch.int32)
    logits = hh @ W.t() + bias
    lp = torch.log_softmax(logits, -1)
    tg = target.t().reshape(-1).long()                                        # row = t * B + b
    nll = -lp[torch.arange(T * B), tg]
    gs = 5.0 / (B * T)
    dlog_ref = torch.autograd.grad((nll * gs).sum(), logits)[0]
    nl, dl = np.zeros(T * B, np.float32), np.full((T * B, ld), 7.0, np.float32)
    a = [f32(h), f32(W.detach()), f32(bias), np.ascontiguousarray(target.numpy())]
    assert lib.fn_out_head_f32_host(P(a[0]), H, P(a[1]), H, P(a[2]), B, T, V, H, P(a[3]), C.c_float(gs), P(nl), P(dl), ld, None) == 0
    close(nl, nll.detach().numpy(), 2e-6, "nll_rows")
    close(dl[:, :V], dlog_ref.numpy(), 2e-5, "dlogits")
    assert not dl[:, V:].any()
    assert lib.fn_out_head_f32_host(P(a[0]), H, P(a[1]), H, P(a[2]), B, T, 400, H, P(a[3]), C.c_float(gs), P(nl), None, 0, None) == L.FN_E_UNSUPPORTED
    # clip_grad_norm_(1) + Adam vs torch.optim.Adam, three steps
    n = 1000
    p = torch.nn.Parameter(torch.randn(n))
    opt = torch.optim.Adam([p], lr=1e-3)
    mine, m, v = f32(p.detach()), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for t in range(1, 4):
        g = torch.randn(n) * (3.0 if t == 1 else 0.01)                       # first step clipped, the others not
        p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([p], 1.0)
        opt.step()
        gg, ss = f32(g), np.zeros(1, np.float32)
        assert lib.fn_sumsq_f32_host(P(gg), C.c_int64(n), P(ss), None, 0, None) == 0
        hyper = f32([1e-3 / (1 - 0.9 ** t), 1.0 / math.sqrt(1 - 0.999 ** t)])
        assert lib.fn_clip_adam_host(P(mine), P(gg), P(m), P(v), C.c_int64(n), P(ss), C.c_float(1.0), P(hyper), C.c_float(0.9), C.c_float(0.999),
                                     C.c_float(1e-8), None) == 0
        close(mine, p.detach().numpy(), 2e-6, "Adam step %d" % t)
    print("fused head + clip/Adam twins OK")


def test_argument_errors():
    arr = (L.FnGruFwd * 9)()
    assert lib.fn_gru_seq_fwd_host(None, 1, None) == -1 and lib.fn_gru_seq_fwd_host(arr, 9, None) == -5
    assert lib.fn_gru_seq_fwd_host(arr, 1, None) == -1                        # NULL members
    assert lib.fn_latent_fwd_host(None, None, None, None, 1, 1, 1, None, None, None, None, None, None, None, None) == -1
    assert lib.fn_clip_adam_host(None, None, None, None, C.c_int64(4), None, C.c_float(1), None, C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8), None) == -1
    print("argument errors OK")


if __name__ == "__main__":
    test_argument_errors()
    test_small_golden_encoder_and_latent()
    test_gru_backward_vs_autograd()
    test_latent_backward_vs_autograd()
    test_out_head_and_adam()
    print("HOST TWINS OK")
