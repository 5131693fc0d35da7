"""Schedule of the SINGLE-ENCODER siblings of the GM-VAE (reference ``model_v2.py``: MusicAttrSingleVAE :174-285, MusicAttrCVAE
:288-423, MusicAttrFaderNets :438-586) on the same HIP kernels.

Topology: ONE bidirectional GRU encoder over the one-hot event tokens (CVAE: + two per-sequence density columns, which enter the
scan as a per-row constant ``W_ih[:, 342:344] [r_density, n_density]``) -> mu | exp(var) heads of width ZL -> z = mu + sigma * eps ->
teacher-forced two-cell global decoder conditioned on ``[z | cond]`` (cond = chroma (24) or the two densities).  No sub-decoders, no
mixture prior: the latent block reuses the mixture kernels with one zero-mean, unit-scale dummy component, whose "KL to the
component" IS the KL to N(0, 1) the siblings' losses use (trainer_singlevae.py:100-102, trainer_cvae.py:100-103, trainer_fader.py:98-101).

Everything heavy is inherited from ``Engine``: weight-stationary scans, chunk-pipelined decoder (forward and backward), batched weight
gradients, token-segment sums.
"""
import torch

from .engine import E_VOCAB, Engine, ops_sort


class SingleEncEngine(Engine):
    def __init__(self, ops, params, hidden, zlat, device, gru="gru", enc_extra=0):
        super().__init__(ops, params, hidden, zlat, 1, device)
        self.gru = gru + "."
        self.enc_extra = enc_extra          # extra dense input columns of the encoder GRU after the 342 one-hot columns (CVAE: 2)
        dev = self.dev
        self.mu_lk = torch.zeros(1, zlat, device=dev)          # the dummy component: mean 0, "logvar" 0 -> scale exp(0) = 1
        self.lv_lk = torch.zeros(1, zlat, device=dev)

    def _gru_sets(self):
        return {"e": (self.gru, "_l0", E_VOCAB), "e_reverse": (self.gru, "_l0_reverse", E_VOCAB),
                "g": ("grucell_g.", "", E_VOCAB), "g2": ("grucell_g_2.", "", 0)}

    # ------------------------------------------------------------------------------------------
    def encode(self, d, extra=None, save=True):
        """2 concurrent scans (forward / reverse direction) + mu | var heads -> pre [B][2 ZL] (mu | log-sigma)"""
        ops, P, H, Z = self.ops, self.p, self.H, self.Z
        B, T = d.shape
        scans, hall, gates = [], {}, {}
        for key, sfx, rev in (("e", "_l0", 0), ("e_reverse", "_l0_reverse", 1)):
            hall[key] = self.buf("enc_h_" + key, (T, B, H))
            gates[key] = self.buf("enc_g_" + key, (T, ops.gates_floats(B, H))) if save else None
            rb = None
            if self.enc_extra:
                rb = self.buf("enc_rb_" + key, (B, 3 * H))
                ops.gemm(extra, P[self.gru + "weight_ih" + sfx][:, E_VOCAB:], rb)
            scans.append(dict(B=B, T=T, H=H, reverse=rev, w_hh_frag=self.whh_f[key], w_hh_frag3=self.whh_f3.get(key), b_hh=P[self.gru + "bias_hh" + sfx],
                              b_ih=P[self.gru + "bias_ih" + sfx], gx_table=self.tab[key], idx=d, idx_shift=0, gx_rowbias=rb,
                              h_all=hall[key], gates=gates[key]))
        ops.gru_seq_fwd(scans)
        pre = self.buf("pre_e", (B, 2 * Z))
        hf, hb = hall["e"][T - 1], hall["e_reverse"][T - 1]
        ops.gemm_multi([dict(C=pre[:, c0:c0 + Z], segs=[(hf, P[head + ".weight"][:, :H]), (hb, P[head + ".weight"][:, H:])],
                             bias=P[head + ".bias"]) for head, c0 in (("mu", 0), ("var", Z))])
        return dict(pre=pre, h_all=hall, gates=gates)

    def latent1(self, pre, eps):
        B, Z = eps.shape
        o = dict(sigma=self.buf("sigma_e", (B, Z)), z=self.buf("z_e", (B, Z)), ll=self.buf("ll_e", (B, 1)), qy=self.buf("qy_e", (B, 1)),
                 y=self.buf("y_e", (B,), torch.int32), terms=self.buf("terms_e", (B, 4)))
        self.ops.latent_fwd(pre, eps, self.mu_lk, self.lv_lk, None, o["sigma"], o["z"], o["ll"], o["qy"], o["y"], o["terms"])
        return o

    def pack_zc(self, z, cond):
        """[z | cond]: model_v2.py:280 (chroma), :420 / :578 (the two densities)"""
        Z = self.Z
        zc = self.buf("zc", (z.shape[0], Z + cond.shape[1]))
        zc[:, :Z].copy_(z)
        zc[:, Z:].copy_(cond)
        return zc

    def forward(self, d, cond, eps, extra=None, save=True, head=True):
        sort = None
        if save:
            self.side_wait_main()
            with self.on_side():
                sort = {"d": ops_sort(self, "d", d, E_VOCAB)}
        enc = self.encode(d, extra, save)
        lat = self.latent1(enc["pre"], eps)
        dec = self.global_decoder_tf(d, self.pack_zc(lat["z"], cond), save, head)
        self.main_wait_side()
        S = dict(d=d, cond=cond, extra=extra, eps=eps, enc=enc, lat=lat, dec=dec, sort=sort)
        self.saved = S if save else None
        return S

    # ------------------------------------------------------------------------------------------
    def backward(self, G, g_z, w3, after_decoders=None, g_mu=None, g_sigma=None):
        """Backward of forward(); dlogits of the decoder must already sit in saved['dec']['logits'].
        g_z   [B][ZL] upstream gradient wrt z (zero-filled, or holding the regulariser / adversarial parts); the decoder's is added
        w3    device {w_lat, w_cls, w_clf}: w_lat * sum_b mean_Z KL(q || N(0,1)) is the fused KL term (fn_latent_bwd)
        g_mu, g_sigma   optional upstream gradients wrt the heads' outputs (the drop-in forward's autograd node: a reference-style loss
              computes its KL from Normal(mu, sigma) in torch)"""
        ops, P, H, Z = self.ops, self.p, self.H, self.Z
        S = self.saved
        d, enc, lat = S["d"], S["enc"], S["lat"]
        B, T = d.shape
        sk_T = self._splitk(T * B)
        gd = self._bwd_global_decoder_scans(S)
        drb_g, dh0_g = gd["drb_g"], gd["dh0_g"]
        ops.gemm_multi([dict(C=g_z, beta=1.0, segs=[(drb_g, P["grucell_g.weight_ih"][:, E_VOCAB:E_VOCAB + Z]),
                                                    (dh0_g, P["linear_init_global.weight"][:, :Z])])], a_k=True, b_k=False)
        self.side_wait_main()
        with self.on_side():
            self._bwd_global_decoder_params(G, S, gd)
            if after_decoders is not None:
                after_decoders()
        dpre = self.buf("dpre_e", (B, 2 * Z))
        dmu_rows = self.buf("dmulk_rows_e", (B, Z))
        ops.latent_bwd(enc["pre"], S["eps"], self.mu_lk, self.lv_lk, None, lat["z"], lat["qy"], g_z, g_mu, g_sigma, None, None, w3, dpre, dmu_rows)
        hf, hb = enc["h_all"]["e"][T - 1], enc["h_all"]["e_reverse"][T - 1]
        dhf, dhb = self.buf("enc_dhf", (B, H)), self.buf("enc_dhb", (B, H))
        for i, (head, c0) in enumerate((("mu", 0), ("var", Z))):
            W = P[head + ".weight"]                          # [ZL][2H]
            dp = dpre[:, c0:c0 + Z]
            ops.gemm(dp, W[:, :H], dhf, a_k=True, b_k=False, beta=float(i))
            ops.gemm(dp, W[:, H:], dhb, a_k=True, b_k=False, beta=float(i))
            dW = G[head + ".weight"]
            ops.gemm(dp, hf, dW[:, :H], a_k=False, b_k=False)
            ops.gemm(dp, hb, dW[:, H:], a_k=False, b_k=False)
            self.colsum(dp, G[head + ".bias"])
        scans, encb = [], {}
        for key, dh in (("e", dhf), ("e_reverse", dhb)):
            encb[key] = dict(dgx=self.buf("enc_dgx_" + key, (T, B, 3 * H)), dghn=self.buf("enc_dghn_" + key, (T, B, H)),
                             rs=self.zbuf("enc_rs_" + key, (B, 3 * H)), rsn=self.zbuf("enc_rsn_" + key, (B, H)))
            scans.append(dict(B=B, T=T, H=H, w_hh_t_frag=self.whh_t[key], w_hh_t_frag3=self.whh_t3.get(key), h0=None, h_all=enc["h_all"][key], gates=enc["gates"][key],
                              dh_last=dh, dgx_all=encb[key]["dgx"], dghn_all=encb[key]["dghn"], scratch=self.buf("enc_scr_" + key, (B, H)),
                              dgx_rowsum=encb[key]["rs"], dghn_rowsum=encb[key]["rsn"]))
        ops.gru_seq_bwd(scans)          # the side lane's parameter-gradient GEMMs run beside it
        keys = (("e", "_l0", 0), ("e_reverse", "_l0_reverse", 1))
        ops.embed_grad_sorted(S["sort"]["d"], [dict(dgx=encb[key]["dgx"], out=G[self.gru + "weight_ih" + sfx][:, :E_VOCAB], transposed=True,
                                                    reverse=rev) for key, sfx, rev in keys])
        for key, sfx, rev in keys:
            self._gru_weight_grads(key, self.gru, sfx, T, B, encb[key]["dgx"], encb[key]["dghn"], enc["h_all"][key], None, G, sk_T,
                                   encb[key]["rs"], encb[key]["rsn"])
            self.colsum(encb[key]["rs"], G[self.gru + "bias_ih" + sfx])
            if self.enc_extra:                               # the dense density columns: (sum over time of the gate gradients)^T extra
                ops.gemm(encb[key]["rs"], S["extra"], G[self.gru + "weight_ih" + sfx][:, E_VOCAB:], a_k=False, b_k=False)
        self.flush_colsums()
        self.main_wait_side()
