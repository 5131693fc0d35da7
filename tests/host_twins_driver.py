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
for _name, (_res, _args) in L.SIGNATURES.items():          # the twins carry the signatures of their device counterparts (include/fadernets_host.h)
    if _name in ("fn_frag_floats", "fn_gru_gates_floats") or not hasattr(lib, _name + "_host"):
        continue
    _fn = getattr(lib, _name + "_host")
    _fn.restype, _fn.argtypes = _res, _args


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


# ---------------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[0] (tests/golden/c0.npz: hidden 512, z 128, B=8, T=64, Tr=16 - the reference's own forward / loss / greedy decode on seeded
# weights) through the twins of the dense products, the sub-decoder head, the regulariser and the eval-mode decode
# ---------------------------------------------------------------------------------------------------------------------------------
def c0_model():
    g = np.load(os.path.join(HERE, "golden", "c0.npz"))
    H, Z, K, B, T, Tr = (int(x) for x in g["meta_dims"])
    pkg = load_package()
    torch.manual_seed(1234)                                                  # the seeded construction c0.npz was generated from
    m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=H, z_dims=Z, n_step=32, n_component=K)
    sd = {k: f32(v.detach().numpy()) for k, v in m.state_dict().items()}
    for k in ("gru_r.weight_hh_l0", "linear_out_g.weight", "mu_r_lookup.weight"):
        v = sd[k].astype(np.float64)
        np.testing.assert_allclose([v.sum(), np.abs(v).sum()], g["w0sum/" + k][:2], rtol=1e-6)
    return g, sd, (H, Z, K, B, T, Tr)


def gemm(a_k, b_k, M, N, K, A, Bm, bias=None, beta=0.0, Cin=None, alpha=1.0):
    out = np.zeros((M, N), np.float32) if Cin is None else f32(Cin).copy()
    A, Bm = f32(A), f32(Bm)
    bias = None if bias is None else f32(bias)
    assert lib.fn_gemm_f32_host(a_k, b_k, M, N, K, alpha, P(A), A.shape[1], P(Bm), Bm.shape[1], beta, P(out), N, P(bias), 1, None, 0, None) == 0
    return out


def test_c0_subdecoder_and_regulariser(g, sd, dims):
    """gmm_model.py:100-117 (rhythm sub-decoder incl. the TIME-axis log_softmax) and trainer_gmm.py:199-217 on the reference's z of c0"""
    H, Z, K, B, T, Tr = dims
    z = f32(g["fw_z_r"])
    W = sd["gru_d_r.weight_ih_l0"]                                            # [3H][3 + Z]: attribute one-hot | z (tiled over time, :103-104)
    rowbias = gemm(1, 1, B, 3 * H, Z, z, np.ascontiguousarray(W[:, 3:]))
    h0 = gemm(1, 1, B, H, Z, z, sd["linear_init_r.weight"], bias=sd["linear_init_r.bias"])
    d = L.FnGruFwd()
    keep = dict(wf=frag(sd["gru_d_r.weight_hh_l0"]), b_hh=sd["gru_d_r.bias_hh_l0"], b_ih=sd["gru_d_r.bias_ih_l0"],
                table=f32(np.ascontiguousarray(W[:, :3].T)), idx=np.ascontiguousarray(g["r"], np.int32), rb=rowbias, h0=h0,
                h_all=np.zeros((Tr, B, H), np.float32), ws=np.zeros(2 * lib.fn_frag_floats_host(B, H), np.float32))
    d.B, d.T, d.H, d.reverse = B, Tr, H, 0
    d.w_hh_frag, d.b_hh, d.b_ih, d.gx_table, d.idx, d.idx_ld = P(keep["wf"]), P(keep["b_hh"]), P(keep["b_ih"]), P(keep["table"]), P(keep["idx"]), Tr
    d.gx_rowbias, d.h0, d.h_all, d.frag_ws = P(keep["rb"]), P(keep["h0"]), P(keep["h_all"]), P(keep["ws"])
    assert lib.fn_gru_seq_fwd_host(C.byref(d), 1, None) == 0
    logits = gemm(1, 1, Tr * B, 3, H, keep["h_all"].reshape(Tr * B, H), sd["linear_out_r.weight"], bias=sd["linear_out_r.bias"])     # [Tr][B][3]
    logp, nll, dl = np.zeros((B, Tr, 3), np.float32), np.zeros((B, 3), np.float32), np.zeros((Tr, B, 3), np.float32)
    tgt = np.ascontiguousarray(g["r"], np.int32)
    gs = 1.0 / (B * Tr)
    assert lib.fn_time_logsoftmax_host(P(logits), B, Tr, 3, P(logp), P(tgt), P(nll), gs, P(dl), None) == 0
    close(logp, g["fw_r_out"], 2e-5, "r_out (time-axis log_softmax)")
    close(nll.sum() / (B * Tr), g["loss_unsup_20000"][2], 2e-5, "CE_R")
    lt = torch.from_numpy(logits).view(Tr, B, 3).requires_grad_(True)        # autograd of the same NLL through torch's dim=time softmax
    lp = torch.log_softmax(lt, 0)
    ce = -lp.gather(2, torch.from_numpy(g["r"]).t().unsqueeze(-1)).sum() * gs
    close(dl, torch.autograd.grad(ce, lt)[0].numpy(), 2e-5, "dlogits (time axis)")
    gout = torch.randn(B, Tr, 3)
    dl2 = np.zeros((Tr, B, 3), np.float32)
    go = f32(gout)
    assert lib.fn_time_logsoftmax_bwd_host(P(logp), P(go), B, Tr, 3, P(dl2), None) == 0
    close(dl2, torch.autograd.grad((torch.log_softmax(lt, 0).permute(1, 0, 2) * gout).sum(), lt)[0].numpy(), 2e-5, "time_logsoftmax_bwd")
    for e, dens, want in (("r", g["r_density"], g["loss_reg"][0]), ("n", g["n_density"], g["loss_reg"][1])):
        z0, a = f32(g["fw_z_" + e][:, 0]), np.ascontiguousarray(dens, np.float64)
        rows, dz = np.zeros(B, np.float32), np.zeros(B, np.float32)
        assert lib.fn_pairwise_reg_host(P(z0), P(a), B, 0, B, P(rows), 1.0 / (B * B), P(dz), None) == 0
        close(rows.sum() / (B * B), want, 2e-5, "l_" + e)
        zt = torch.from_numpy(z0).requires_grad_(True)
        D = zt[:, None] - zt[None, :]
        S = torch.sign(torch.from_numpy(np.subtract.outer(a, a))).float()
        close(dz, torch.autograd.grad(((torch.tanh(D) - S) ** 2).mean(), zt)[0].numpy(), 2e-5, "d l_%s / d z0" % e)
        half = np.zeros(B // 2, np.float32)                                   # data-parallel form: rows [4, 8) of the global batch
        assert lib.fn_pairwise_reg_host(P(z0), P(a), B, B // 2, B // 2, P(half), 1.0 / (B * B), None, None) == 0
        close(half, rows[B // 2:], 1e-6, "row-offset form")
    print("c0: sub-decoder (time-axis head) + regulariser through the host twins OK")


def test_c0_global_decoder_cells_and_greedy_decode(g, sd, dims):
    """gmm_model.py:119-149: teacher-forced steps through fn_gru_cell_f32_host + fn_gemm_f32_host vs the reference's `out`; eval-mode greedy
    decode through fn_decode_greedy_host vs the reference's token indices (bit-exact up to its first near-tie)"""
    H, Z, K, B, T, Tr = dims
    zc = f32(np.concatenate([g["fw_z_r"], g["fw_z_n"], g["c"]], 1))          # [B][280]
    Wih = sd["grucell_g.weight_ih"]                                           # [3H][342 + 280]
    table = f32(np.ascontiguousarray(Wih[:, :342].T))
    NS = 6                                                                    # teacher-forced steps checked (plain loops under ASAN)
    rowbias = gemm(1, 1, B, 3 * H, 280, zc, np.ascontiguousarray(Wih[:, 342:]))
    hx0 = gemm(1, 1, B, H, 280, zc, sd["linear_init_global.weight"], bias=sd["linear_init_global.bias"])
    hx1 = None
    toks = np.ascontiguousarray(np.concatenate([np.full((B, 1), 341), g["d"][:, :NS - 1]], 1), np.int32)      # input of step i = d[:, i-1] (:120-121,142)
    for i in range(NS):
        c = L.FnGruCell()
        o1 = np.zeros((B, H), np.float32)
        col = np.ascontiguousarray(toks[:, i])
        c.B, c.H, c.gx_table, c.idx, c.idx_ld, c.start_token = B, H, P(table), P(col), 1, 341
        c.gx_rowbias, c.h_prev, c.ldh, c.w_hh, c.ldw_hh = P(rowbias), P(hx0), H, P(sd["grucell_g.weight_hh"]), H
        c.b_ih, c.b_hh, c.h_out, c.ldo = P(sd["grucell_g.bias_ih"]), P(sd["grucell_g.bias_hh"]), P(o1), H
        assert lib.fn_gru_cell_f32_host(C.byref(c), None) == 0
        hx0 = o1
        if i == 0:
            hx1 = hx0.copy()                                                  # :134-135
        c2 = L.FnGruCell()
        o2 = np.zeros((B, H), np.float32)
        c2.B, c2.H, c2.x, c2.ldx, c2.K1, c2.w_ih, c2.ldw_ih = B, H, P(hx0), H, H, P(sd["grucell_g_2.weight_ih"]), H
        c2.h_prev, c2.ldh, c2.w_hh, c2.ldw_hh = P(hx1), H, P(sd["grucell_g_2.weight_hh"]), H
        c2.b_ih, c2.b_hh, c2.h_out, c2.ldo = P(sd["grucell_g_2.bias_ih"]), P(sd["grucell_g_2.bias_hh"]), P(o2), H
        assert lib.fn_gru_cell_f32_host(C.byref(c2), None) == 0
        hx1 = o2
        logits = gemm(1, 1, B, 342, H, hx1, sd["linear_out_g.weight"], bias=sd["linear_out_g.bias"])
        close(torch.log_softmax(torch.from_numpy(logits), -1).numpy(), g["fw_out"][:, i], 5e-5, "out[:, %d]" % i)
    assert lib.fn_gru_cell_f32_host(None, None) == L.FN_E_NULL
    # eval-mode greedy decode of the reference's dec_z
    nb, steps = 3, 24
    zd = f32(g["dec_z"][:nb])
    d = L.FnDecode()
    keep = dict(w1=frag(sd["grucell_g.weight_hh"]), w2i=frag(sd["grucell_g_2.weight_ih"]), w2h=frag(sd["grucell_g_2.weight_hh"]),
                wo=frag(sd["linear_out_g.weight"]), rb=gemm(1, 1, nb, 3 * H, 280, zd, np.ascontiguousarray(Wih[:, 342:])),
                h0=gemm(1, 1, nb, H, 280, zd, sd["linear_init_global.weight"], bias=sd["linear_init_global.bias"]),
                tok=np.zeros((nb, steps), np.int32), logp=np.zeros((nb, steps, 342), np.float32),
                ws=np.zeros(lib.fn_decode_ws_bytes_host(nb, H, 342) // 4 + 4, np.float32), sync=np.zeros(8, np.int32))
    d.B, d.steps, d.H, d.V, d.start_token = nb, steps, H, 342, 341
    d.w_hh1_frag, d.b_hh1, d.b_ih1, d.table1, d.rowbias1, d.h0 = P(keep["w1"]), P(sd["grucell_g.bias_hh"]), P(sd["grucell_g.bias_ih"]), P(table), P(keep["rb"]), P(keep["h0"])
    d.w_ih2_frag, d.b_ih2, d.w_hh2_frag, d.b_hh2 = P(keep["w2i"]), P(sd["grucell_g_2.bias_ih"]), P(keep["w2h"]), P(sd["grucell_g_2.bias_hh"])
    d.w_out_frag, d.b_out, d.tokens, d.tok_ld, d.logp, d.ws, d.sync_ws = P(keep["wo"]), P(sd["linear_out_g.bias"]), P(keep["tok"]), steps, P(keep["logp"]), P(keep["ws"]), P(keep["sync"])
    assert lib.fn_decode_greedy_host(C.byref(d), None) == 0
    close(keep["logp"][:, 0], g["dec_logp_first"][:nb], 1e-4, "log-probabilities of the first decode step")
    checked = 0
    for i in range(nb):
        tight = np.nonzero(g["dec_gap"][i, :steps] < 1e-4)[0]
        upto = int(tight[0]) if len(tight) else steps
        assert np.array_equal(keep["tok"][i, :upto], g["dec_tokens"][i, :upto]), (i, upto, keep["tok"][i], g["dec_tokens"][i, :steps])
        checked += upto
    assert checked >= 0.9 * nb * steps
    d.steps = 0
    assert lib.fn_decode_greedy_host(C.byref(d), None) == L.FN_E_SHAPE
    # the same decode on the per-token cells with the argmax packed by the output layer (fn_out_argmax_f32 -> FnGruCell.idx_best -> fn_best_tokens)
    best = np.zeros((steps, nb), np.uint64)
    s1, s2 = keep["h0"].copy(), None
    for i in range(steps):
        c = L.FnGruCell()
        o1 = np.zeros((nb, H), np.float32)
        c.B, c.H, c.gx_table, c.start_token = nb, H, P(table), 341
        if i > 0:
            c.idx_best, c.best_v = best[i - 1].ctypes.data, 342
        c.gx_rowbias, c.h_prev, c.ldh, c.w_hh, c.ldw_hh = P(keep["rb"]), P(s1), H, P(sd["grucell_g.weight_hh"]), H
        c.b_ih, c.b_hh, c.h_out, c.ldo = P(sd["grucell_g.bias_ih"]), P(sd["grucell_g.bias_hh"]), P(o1), H
        assert lib.fn_gru_cell_f32_host(C.byref(c), None) == 0
        s1 = o1
        if i == 0:
            s2 = s1.copy()
        c2 = L.FnGruCell()
        o2 = np.zeros((nb, H), np.float32)
        c2.B, c2.H, c2.x, c2.ldx, c2.K1, c2.w_ih, c2.ldw_ih = nb, H, P(s1), H, H, P(sd["grucell_g_2.weight_ih"]), H
        c2.h_prev, c2.ldh, c2.w_hh, c2.ldw_hh = P(s2), H, P(sd["grucell_g_2.weight_hh"]), H
        c2.b_ih, c2.b_hh, c2.h_out, c2.ldo = P(sd["grucell_g_2.bias_ih"]), P(sd["grucell_g_2.bias_hh"]), P(o2), H
        assert lib.fn_gru_cell_f32_host(C.byref(c2), None) == 0
        s2 = o2
        assert lib.fn_out_argmax_f32_host(P(s2), H, P(sd["linear_out_g.weight"]), H, P(sd["linear_out_g.bias"]), nb, 342, H, best[i].ctypes.data, None) == 0
    tok2 = np.zeros((nb, steps), np.int32)
    assert lib.fn_best_tokens_host(best.ctypes.data, steps, nb, 342, P(tok2), steps, None) == 0
    for i in range(nb):
        tight = np.nonzero(g["dec_gap"][i, :steps] < 1e-4)[0]
        upto = int(tight[0]) if len(tight) else steps
        assert np.array_equal(tok2[i, :upto], g["dec_tokens"][i, :upto]), (i, upto, tok2[i], g["dec_tokens"][i, :steps])
    assert lib.fn_out_argmax_f32_host(None, H, None, H, None, nb, 342, H, None, None) == L.FN_E_NULL
    assert lib.fn_best_tokens_host(best.ctypes.data, steps, nb, 342, P(tok2), steps - 1, None) == L.FN_E_SHAPE
    print("c0: teacher-forced decoder cells (%d steps) + greedy decode (%d x %d tokens, %d compared bit for bit) through the host twins OK" % (NS, nb, steps, checked))


def test_c0_weight_gradient_products_and_segment_sums(g, sd, dims):
    """dW = dY^T X forms (fn_gemm_f32 a_k = 0, b_k = 0; fn_gru_dwhh_f32) and the token-segment sums behind dW_ih[:, :V] on c0's token matrix"""
    H, Z, K, B, T, Tr = dims
    rng = np.random.RandomState(3)
    M, N, Kk = 24, 20, 40
    A, Bm, bias, C0 = rng.randn(M, Kk), rng.randn(N, Kk), rng.randn(N), rng.randn(M, N)
    close(gemm(1, 1, M, N, Kk, A, Bm, bias=bias), A @ Bm.T + bias, 2e-6, "gemm NT + bias")
    close(gemm(1, 0, M, N, Kk, A, Bm.T.copy(), beta=0.5, Cin=C0), A @ Bm.T + 0.5 * C0, 2e-6, "gemm NN + beta")
    close(gemm(0, 0, M, N, Kk, A.T.copy(), Bm.T.copy(), alpha=2.0), 2.0 * (A @ Bm.T), 2e-6, "gemm TN (weight-gradient form)")
    assert lib.fn_gemm_f32_host(1, 1, M, N, Kk, 1.0, None, Kk, None, Kk, 0.0, None, N, None, 1, None, 0, None) == L.FN_E_NULL
    h = 32
    rows = T * B
    dgx, dghn, hp, dW0 = f32(rng.randn(rows, 3 * h)), f32(rng.randn(rows, h)), f32(rng.randn(rows, h)), f32(rng.randn(3 * h, h))
    dW = dW0.copy()
    assert lib.fn_gru_dwhh_f32_host(P(dgx), P(dghn), P(hp), rows, h, 1.0, P(dW), 1, None, 0, None) == 0
    lhs = np.concatenate([dgx[:, :2 * h], dghn], 1).astype(np.float64)
    close(dW, lhs.T @ hp.astype(np.float64) + dW0, 2e-6, "dW_hh = [dr' | dz' | dn' r]^T h_prev")
    # token sort of c0's event tokens + the segment sums of three scans that read them: forward, reverse, input shifted by one (decoder layer 1)
    V, N3 = 342, 48
    idx = np.ascontiguousarray(g["d"], np.int32)
    img = np.zeros(lib.fn_token_sort_ints_host(rows, V), np.int32)
    ws = np.zeros(lib.fn_token_sort_ws_bytes_host(rows, V) // 4 + 4, np.int32)
    assert lib.fn_token_sort_host(P(idx), B, T, T, V, P(img), P(ws), ws.nbytes, None) == 0
    seg, order = img[:V + 1], img[2 * (V + 1) + 2:]
    pos_tok = idx.T.reshape(-1)                                               # token of position r = tau * B + b
    assert seg[0] == 0 and seg[V] == rows and np.array_equal(np.sort(order), np.arange(rows))
    for v in (0, 1, 57, 341):
        run = order[seg[v]:seg[v + 1]]
        assert (pos_tok[run] == v).all() and (np.diff(run) > 0).all()         # grouped by token, stable
    jobs = (L.FnEmbedGrad * 3)()
    dg = [f32(rng.randn(T, B, N3)) for _ in range(3)]
    outs = [np.zeros((V, N3), np.float32), np.zeros((V, N3), np.float32), np.zeros((N3, V + 2), np.float32)]
    for j, (rev, sh, tr_) in enumerate(((0, 0, 0), (1, 0, 0), (0, -1, 1))):
        jobs[j].dgx_all, jobs[j].out, jobs[j].out_ld, jobs[j].transposed = P(dg[j]), P(outs[j]), outs[j].shape[1], tr_
        jobs[j].reverse, jobs[j].idx_shift, jobs[j].start_token = rev, sh, 341
    ws2 = np.zeros(lib.fn_embed_grad_sorted_ws_bytes_host(rows, B, V, N3, 3) // 4 + 4, np.float32)
    assert lib.fn_embed_grad_sorted_host(jobs, 3, B, T, N3, V, P(img), P(ws2), ws2.nbytes, None) == 0
    for j, (rev, sh) in enumerate(((0, 0), (1, 0), (0, -1))):
        want = np.zeros((V, N3), np.float64)
        for p in range(T):
            tau = (T - 1 - p if rev else p) + sh
            tok = idx[:, tau] if tau >= 0 else np.full(B, 341)
            np.add.at(want, tok, dg[j][p].astype(np.float64))
        got = outs[j] if j < 2 else outs[j][:, :V].T
        close(got, want, 2e-6, "segment sums of scan %d" % j)
    assert not outs[2][:, V:].any()
    jobs[0].idx_shift = 1
    assert lib.fn_embed_grad_sorted_host(jobs, 3, B, T, N3, V, P(img), P(ws2), ws2.nbytes, None) == L.FN_E_SHAPE
    print("c0: weight-gradient products + token sort / segment sums through the host twins OK")


if __name__ == "__main__":
    test_argument_errors()
    test_small_golden_encoder_and_latent()
    test_gru_backward_vs_autograd()
    test_latent_backward_vs_autograd()
    test_out_head_and_adam()
    c0 = c0_model()
    test_c0_subdecoder_and_regulariser(*c0)
    test_c0_global_decoder_cells_and_greedy_decode(*c0)
    test_c0_weight_gradient_products_and_segment_sums(*c0)
    print("HOST TWINS OK")
