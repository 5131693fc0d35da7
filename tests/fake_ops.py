"""TEST-ONLY stand-in for music-fader-nets_amd/hipops.HipOps.

Implements the *documented semantics* of every C-ABI entry point (include/fadernets.h) with plain torch on
the CPU, so that the host-side schedule (engine.py / trainer.py / parallel.py: which op runs on which buffer
in which order, gradient routing, data-parallel reductions) can be checked against the oracle without a GPU.
It is never imported by the product; the product path fails loudly without the HIP library.
"""
import math

import torch

LOG_2PI = math.log(2 * math.pi)


class FakeOps:
    name = "fake-cpu"

    def __init__(self):
        self.calls = []
        self.lane = ""

    # -- dense --------------------------------------------------------------------------------------
    def gemm(self, A, B, Cm, a_k=True, b_k=True, alpha=1.0, beta=0.0, bias=None, splitk=1, lean=False, nt_x6=True):
        self.calls.append("gemm")
        a = A if a_k else A.t()
        b = B.t() if b_k else B
        assert a.shape[0] == Cm.shape[0] and b.shape[1] == Cm.shape[1] and a.shape[1] == b.shape[0], "gemm shapes"
        res = alpha * (a @ b)
        if bias is not None:
            res = res + bias
        if beta != 0.0:
            res = res + beta * Cm
        Cm.copy_(res)

    def transpose(self, src, dst):
        assert tuple(dst.shape) == (src.shape[1], src.shape[0])
        dst.copy_(src.t())

    def colsum(self, X, out, beta=0.0):
        assert out.numel() == X.shape[1] and out.is_contiguous()
        s = X.sum(dim=0)
        out.copy_((beta * out if beta != 0.0 else 0) + s.view_as(out))

    def colsum_multi(self, jobs):
        for j in jobs:
            self.colsum(j[0], j[1], j[2] if len(j) > 2 else 0.0)

    def axpy(self, alpha, x, y):
        assert x.is_contiguous() and y.is_contiguous()
        y.add_(alpha * x)

    def sum(self, x, out, scale=1.0):
        out.copy_((x.sum() * scale).view_as(out))

    # -- GRU ----------------------------------------------------------------------------------------
    @staticmethod
    def frag_floats(rows, K):
        return (rows + 15) // 16 * 16 * K

    @staticmethod
    def frag_pack(src, dst):
        """the operand image is opaque to callers; the fake keeps the plain row-major matrix in its head"""
        dst[: src.numel()].copy_(src.reshape(-1))

    @staticmethod
    def gates_floats(B, H):
        return 4 * H * ((B + 15) // 16 * 16)

    @staticmethod
    def _gview(gates, p, B, H):
        """the gate buffer is opaque to callers; the fake keeps plain [B][4][H] in the head of each step's slab"""
        return gates[p].reshape(-1)[: B * 4 * H].view(B, 4, H)

    @staticmethod
    def _tok(s, p):
        tau = (s["T"] - 1 - p if s.get("reverse", 0) else p) + s.get("idx_shift", 0)
        if tau < 0:
            return torch.full((s["B"],), s.get("start_token", 0), dtype=torch.long)
        return s["idx"][:, tau].long()

    def gru_seq_fwd(self, scans, persistent=True, cu_budget=0):
        self.calls.append("gru_seq_fwd")
        if not scans or len(scans) > 8:
            raise RuntimeError("fn_gru_seq_fwd failed: too many scans in one call (code -5)")      # FN_E_COUNT, as the library
        for s in scans:
            B, T, H = s["B"], s["T"], s["H"]
            h = s["h0"] if s.get("h0") is not None else torch.zeros(B, H)
            if s.get("h0_frag") is not None:          # the operand image and the row-major state must describe the same h0
                assert torch.equal(s["h0_frag"][: B * H].view(B, H), h)
            for p in range(T):
                gx = torch.zeros(B, 3 * H)
                if s.get("b_ih") is not None:
                    gx = gx + s["b_ih"]
                if s.get("gx_dense") is not None:
                    gx = gx + s["gx_dense"][p]
                if s.get("gx_table") is not None:
                    gx = gx + s["gx_table"][self._tok(s, p)]
                if s.get("gx_rowbias") is not None:
                    gx = gx + s["gx_rowbias"]
                gh = h @ s["w_hh_frag"][: 3 * H * H].view(3 * H, H).t() + s["b_hh"]
                r = torch.sigmoid(gx[:, :H] + gh[:, :H])
                z = torch.sigmoid(gx[:, H:2 * H] + gh[:, H:2 * H])
                n = torch.tanh(gx[:, 2 * H:] + r * gh[:, 2 * H:])
                h = (1 - z) * n + z * h
                s["h_all"][p].copy_(h)
                if s.get("gates") is not None:
                    g = self._gview(s["gates"], p, B, H)
                    g[:, 0].copy_(r), g[:, 1].copy_(z), g[:, 2].copy_(n), g[:, 3].copy_(gh[:, 2 * H:])
            if s.get("h_last_frag") is not None:
                self.frag_pack(h, s["h_last_frag"])

    def gru_seq_bwd(self, scans, persistent=True, cu_budget=0):
        self.calls.append("gru_seq_bwd")
        for s in scans:
            B, T, H = s["B"], s["T"], s["H"]
            carry = torch.zeros(B, H)
            if s.get("dh_last") is not None:
                carry = carry + s["dh_last"]
            for q in range(T - 1, -1, -1):
                dh = carry + (s["dh_ext"][q] if s.get("dh_ext") is not None else 0)
                g = self._gview(s["gates"], q, B, H)
                r, z, n, hn = g[:, 0], g[:, 1], g[:, 2], g[:, 3]
                hp = (s["h0"] if s.get("h0") is not None else torch.zeros(B, H)) if q == 0 else s["h_all"][q - 1]
                dn = dh * (1 - z)
                dz = dh * (hp - n)
                dnp = dn * (1 - n * n)
                dr = dnp * hn
                dzp = dz * z * (1 - z)
                drp = dr * r * (1 - r)
                dgx = torch.cat([drp, dzp, dnp], dim=1)
                s["dgx_all"][q].copy_(dgx)
                s["dghn_all"][q].copy_(dnp * r)
                if s.get("dgx_rowsum") is not None:
                    s["dgx_rowsum"].add_(dgx)
                if s.get("dghn_rowsum") is not None:
                    s["dghn_rowsum"].add_(dnp * r)
                dgh = torch.cat([drp, dzp, dnp * r], dim=1)
                carry = dh * z + dgh @ s["w_hh_t_frag"][: 3 * H * H].view(H, 3 * H).t()
            if s.get("dh0") is not None:
                s["dh0"].copy_(carry)

    def gru_cell(self, h_prev, w_hh, b_hh, h_out, x=None, w_ih=None, b_ih=None, gx_table=None, idx=None, start_token=0, gx_rowbias=None, variant=None,
                 idx_best=None, best_v=0):
        B, H = h_prev.shape
        if idx_best is not None:
            idx = (best_v - 1 - (idx_best & 0xffffffff)).to(torch.int32)
        gi = torch.zeros(B, 3 * H)
        if x is not None:
            gi = gi + x @ w_ih.t()
        if b_ih is not None:
            gi = gi + b_ih
        if gx_table is not None:
            tok = idx.long() if idx is not None else torch.full((B,), int(start_token), dtype=torch.long)
            gi = gi + gx_table[tok]
        if gx_rowbias is not None:
            gi = gi + gx_rowbias
        gh = h_prev @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h_out.copy_((1.0 - z) * n + z * h_prev)

    def gru_dwhh(self, dgx, dghn, hprev, dW, beta=0.0, splitk=1, lean=False):
        H = hprev.shape[1]
        g = torch.cat([dgx[:, : 2 * H], dghn], dim=1)
        dW.copy_(beta * dW + g.t() @ hprev)

    def embed_grad(self, dgx_all, idx, idx_shift, start_token, reverse, V, out):
        T, B, N3 = dgx_all.shape
        out.zero_()
        fake = dict(T=T, B=B, reverse=reverse, idx_shift=idx_shift, start_token=start_token, idx=idx)
        for p in range(T):
            out.index_add_(0, self._tok(fake, p), dgx_all[p])

    def gemm_multi(self, jobs, a_k=True, b_k=True):
        for j in jobs:
            acc = 0
            for A, B in j["segs"]:
                acc = acc + (A if a_k else A.t()) @ (B.t() if b_k else B)
            beta = float(j.get("beta", 0.0))
            out = acc + beta * j["C"] if beta != 0.0 else acc          # C may be uninitialised memory when beta == 0
            if j.get("bias") is not None:
                out = out + j["bias"]
            j["C"].copy_(out)

    def weight_images(self, jobs):
        for kind, src, dst in jobs:
            if kind == "transpose":
                dst.view(src.shape[1], src.shape[0]).copy_(src.t())
            elif kind == "copy":
                dst.view(src.shape).copy_(src)
            elif kind == "frag":
                self.frag_pack(src, dst)
            else:
                self.frag_pack(src.t().contiguous(), dst)

    def token_sort(self, idx, V, img=None):
        """handle of a sorted token matrix (the HIP backend keeps the sort image of fn_token_sort)"""
        return dict(idx=idx.clone(), V=V)

    def embed_grad_sorted(self, handle, jobs):
        """jobs: dict(dgx, out, reverse, idx_shift, start_token, transposed): out = table [V][N3], or its transpose"""
        V = handle["V"]
        for j in jobs:
            tab = torch.zeros(V, j["dgx"].shape[2])
            self.embed_grad(j["dgx"], handle["idx"], j.get("idx_shift", 0), j.get("start_token", 0), j.get("reverse", 0), V, tab)
            j["out"].copy_(tab.t() if j.get("transposed") else tab)

    def time_sum(self, X, out):
        out.copy_(X.sum(0).view_as(out))

    # -- heads --------------------------------------------------------------------------------------
    def vocab_logsoftmax(self, logits, B, T, E, logp_bt=None, target=None, nll_rows=None, grad_scale=0.0, dlogits=None):
        x = logits[:, :E].view(T, B, E)
        lp = torch.log_softmax(x, dim=-1)
        if logp_bt is not None:
            logp_bt.copy_(lp.permute(1, 0, 2))
        if target is not None:
            tg = target.long().t().contiguous()            # [T][B]
            if nll_rows is not None:
                nll_rows.copy_(-lp.gather(2, tg.unsqueeze(-1)).reshape(-1))
            if dlogits is not None:
                g = lp.exp()
                g.scatter_add_(2, tg.unsqueeze(-1), -torch.ones(T, B, 1))
                dlogits[:, :E].copy_((grad_scale * g).view(T * B, E))

    def out_head(self, h, W, bias, B, T, target, nll_rows=None, grad_scale=0.0, dlogits=None):
        V = W.shape[0]
        logits = h @ W.t() + bias
        if dlogits is not None:
            dlogits.zero_()
        self.vocab_logsoftmax(logits, B, T, V, target=target, nll_rows=nll_rows, grad_scale=grad_scale, dlogits=dlogits)

    def vocab_logsoftmax_bwd(self, logp_bt, gout_bt, dlogits):
        B, T, E = logp_bt.shape
        g = gout_bt - logp_bt.exp() * gout_bt.sum(-1, keepdim=True)
        dlogits[:, :E].copy_(g.permute(1, 0, 2).reshape(T * B, E))

    def out_argmax(self, h, W, bias, best):
        """fn_out_argmax_f32 semantics: packed (order-preserving key of the logit) << 32 | (V - 1 - column), max-combined into best"""
        V = W.shape[0]
        logits = h @ W.t() + bias
        bits = logits.contiguous().view(torch.int32).to(torch.int64) & 0xffffffff
        key = torch.where(bits >> 31 != 0, bits ^ 0xffffffff, bits ^ 0x80000000)
        words = (key << 32) | (V - 1 - torch.arange(V, dtype=torch.int64)).view(1, -1)
        # unsigned comparison of 64-bit words held in int64: flip the sign bit
        flip = torch.tensor(-0x8000000000000000, dtype=torch.int64)
        m = ((words ^ flip).max(1)[0]) ^ flip
        best.copy_(torch.where((best ^ flip) > (m ^ flip), best, m))

    def best_tokens(self, best, V, tokens):
        steps = best.shape[0]
        tokens[:, :steps] = (V - 1 - (best & 0xffffffff)).t().to(torch.int32)

    def vocab_argmax(self, logits, E, logp_out, tok_out):
        lp = torch.log_softmax(logits[:, :E], dim=-1)
        if logp_out is not None:
            logp_out.copy_(lp)
        tok_out.copy_(logits[:, :E].max(1)[1].to(torch.int32))

    def time_logsoftmax(self, logits, logp_bt=None, target=None, nll_bc=None, grad_scale=0.0, dlogits=None):
        Tr, B, Cc = logits.shape
        lp = torch.log_softmax(logits, dim=0)              # over time
        if logp_bt is not None:
            logp_bt.copy_(lp.permute(1, 0, 2))
        if target is not None:
            oh = torch.zeros(Tr, B, Cc).scatter_(2, target.long().t().unsqueeze(-1), 1.0)
            if nll_bc is not None:
                nll_bc.copy_(-(lp * oh).sum(0))
            if dlogits is not None:
                dlogits.copy_(grad_scale * (lp.exp() * oh.sum(0, keepdim=True) - oh))

    def time_logsoftmax_bwd(self, logp_bt, gout_bt, dlogits):
        g = gout_bt - logp_bt.exp() * gout_bt.sum(1, keepdim=True)
        dlogits.copy_(g.permute(1, 0, 2))

    # -- latent -------------------------------------------------------------------------------------
    @staticmethod
    def _latent_math(pre, eps, mu_lk, lv_lk):
        Z = eps.shape[1]
        K = mu_lk.shape[0]
        mu, s = pre[:, :Z], torch.exp(pre[:, Z:])
        z = mu + s * eps
        ll = torch.stack([(-0.5 * ((z - mu_lk[k]) ** 2 / torch.exp(lv_lk[k]) + lv_lk[k] + LOG_2PI)).sum(1) + math.log(1.0 / K)
                          for k in range(K)], dim=1)
        qy = torch.softmax(ll, dim=1)
        sp = torch.exp(lv_lk)                              # [K][Z] used as STD
        vr = (s.unsqueeze(1) / sp) ** 2
        t1 = ((mu.unsqueeze(1) - mu_lk) / sp) ** 2
        klm = (0.5 * (vr + t1 - 1 - torch.log(vr))).mean(-1)      # [B][K]
        return mu, s, z, ll, qy, klm

    def latent_fwd(self, pre, eps, mu_lk, lv_lk, labels, sigma, z, ll, qy, y, terms):
        mu, s, zz, l, q, klm = self._latent_math(pre, eps, mu_lk, lv_lk)
        sigma.copy_(s), z.copy_(zz), ll.copy_(l), qy.copy_(q)
        y.copy_(q.max(1)[1].to(torch.int32))
        terms.zero_()
        terms[:, 0] = (q * klm).sum(1)
        terms[:, 1] = (q * torch.log_softmax(l, dim=1)).mean(1)
        if labels is not None:
            lb = labels.long().view(-1, 1)
            terms[:, 2] = klm.gather(1, lb).view(-1)
            terms[:, 3] = -torch.log_softmax(q, dim=1).gather(1, lb).view(-1)

    def latent_bwd(self, pre, eps, mu_lk, lv_lk, labels, z, qy, g_z, g_mu, g_sigma, g_ll, g_qy, w3, dpre, dmu_lk_rows):
        w_lat, w_cls, w_clf = (0.0, 0.0, 0.0) if w3 is None else (float(w3[0]), float(w3[1]), float(w3[2]))
        # gradients by autograd of the documented forward + fused loss: checks the schedule AND documents the maths
        with torch.enable_grad():
            self._latent_bwd(pre, eps, mu_lk, lv_lk, labels, g_z, g_mu, g_sigma, g_ll, g_qy, w_lat, w_cls, w_clf, dpre, dmu_lk_rows)

    def _latent_bwd(self, pre, eps, mu_lk, lv_lk, labels, g_z, g_mu, g_sigma, g_ll, g_qy, w_lat, w_cls, w_clf, dpre, dmu_lk_rows):
        B, Z = eps.shape
        K = mu_lk.shape[0]
        pre_ = pre.clone().requires_grad_(True)
        mlk = mu_lk.unsqueeze(0).repeat(B, 1, 1).clone().requires_grad_(True)   # per-row copy -> per-row gradient
        tot = torch.zeros(())
        rows = []
        for b in range(B):
            mu, s, zz, l, q, klm = self._latent_math(pre_[b:b + 1], eps[b:b + 1], mlk[b], lv_lk)
            rows.append((mu, s, zz, l, q, klm))
        mu = torch.cat([r[0] for r in rows]); s = torch.cat([r[1] for r in rows]); zz = torch.cat([r[2] for r in rows])
        l = torch.cat([r[3] for r in rows]); q = torch.cat([r[4] for r in rows]); klm = torch.cat([r[5] for r in rows])
        for g, t in ((g_z, zz), (g_mu, mu), (g_sigma, s), (g_ll, l), (g_qy, q)):
            if g is not None:
                tot = tot + (g * t).sum()
        if labels is None:
            tot = tot + w_lat * (q * klm).sum() + w_cls * (q * torch.log_softmax(l, dim=1)).mean(1).sum()
        else:
            lb = labels.long().view(-1, 1)
            tot = tot + w_lat * klm.gather(1, lb).sum() + w_clf * (-torch.log_softmax(q, dim=1).gather(1, lb)).sum()
        gp, gm = torch.autograd.grad(tot, [pre_, mlk], allow_unused=True)
        dpre.copy_(gp if gp is not None else torch.zeros_like(pre))
        if dmu_lk_rows is not None:
            dmu_lk_rows.copy_((gm if gm is not None else torch.zeros_like(mlk)).reshape(B, K * Z))

    def pairwise_reg(self, z0_all, attr_all, row0, nrows, loss_rows, grad_scale=0.0, dz0=None):
        assert attr_all.dtype == torch.float64
        zi = z0_all[row0:row0 + nrows].view(-1, 1)
        ai = attr_all[row0:row0 + nrows].view(-1, 1)
        th = torch.tanh(zi - z0_all.view(1, -1))
        sg = torch.sign(ai - attr_all.view(1, -1)).float()
        loss_rows.copy_(((th - sg) ** 2).sum(1))
        if dz0 is not None:
            dz0.copy_(grad_scale * 4.0 * ((th - sg) * (1 - th * th)).sum(1))

    # -- optimiser ----------------------------------------------------------------------------------
    def sumsq(self, g, out):
        out.copy_((g.double() ** 2).sum().float().view_as(out))

    def step_params(self, counters, beta, lr, beta1, beta2, supervised, inv_global_batch, advance, out):
        step, t = int(counters[0]), int(counters[1]) + (1 if advance else 0)
        beta0 = 0.0 if step < 1000 else min((step - 10000) / 10000 * beta, beta)
        tt = max(t, 1)
        out[0] = beta0 * inv_global_batch
        out[1] = 0.0 if supervised else beta0 * inv_global_batch
        out[2] = inv_global_batch if supervised else 0.0
        out[3] = lr / (1 - beta1 ** tt)
        out[4] = 1.0 / math.sqrt(1 - beta2 ** tt)
        out[5] = beta0
        out[6] = min(step / 2000 * 1e-4, 1e-4)
        out[7] = beta * inv_global_batch
        if advance:
            counters[0] = step + 1
            counters[1] = t

    def masked_prob(self, logits, E, ranges, sums=None, w=None, dlogits=None):
        p = torch.softmax(logits[:, :E].double(), dim=1)
        (lo0, hi0), (lo1, hi1) = ranges
        s0, s1 = p[:, lo0:hi0].sum(1), p[:, lo1:hi1].sum(1)
        if sums is not None:
            sums.copy_(torch.stack([s0, s1], dim=1).float())
        if dlogits is not None:
            a = torch.zeros_like(p)
            a[:, lo0:hi0] += w[:, 0:1].double()
            a[:, lo1:hi1] += w[:, 1:2].double()
            dot = w[:, 0].double() * s0 + w[:, 1].double() * s1
            dlogits[:, :E].copy_((p * (a - dot.unsqueeze(1))).float())

    def adv_head(self, z, w_r, w_n, b_r, b_n, mask, dens, lam_dev, inv_global_batch, o, loss_rows, da=None, g_z=None):
        Z = w_r.numel()
        zz = z[:, :Z]
        pre = torch.stack([zz @ w_r.view(-1) + b_r.view(()), zz @ w_n.view(-1) + b_n.view(())], dim=1)
        ov = torch.relu(pre) * mask
        df = ov - dens
        o.copy_(ov)
        loss_rows.copy_(df * df)
        lam = float(lam_dev[0]) if lam_dev is not None else 0.0
        dav = lam * 2.0 * df * inv_global_batch * mask * (pre > 0).float()
        if da is not None:
            da.copy_(dav)
        if g_z is not None:
            g_z[:, :Z].sub_(dav[:, 0:1] * w_r.view(1, -1) + dav[:, 1:2] * w_n.view(1, -1))

    def clip_adam(self, p, g, m, v, sumsq, max_norm, hyper, beta1, beta2, eps):
        coef = min(1.0, max_norm / (math.sqrt(float(sumsq[0])) + 1e-6))
        gg = g * coef
        m.mul_(beta1).add_(gg, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
        p.sub_(float(hyper[0]) * m / (v.sqrt() * float(hyper[1]) + eps))

    def onehot_to_index(self, oh, idx):
        idx.copy_(oh.max(-1)[1].to(torch.int32))
