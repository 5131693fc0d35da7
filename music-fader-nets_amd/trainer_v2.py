"""Fused training / evaluation steps of the single-encoder siblings: ``train`` / ``evaluate`` of the reference's
``trainer_singlevae.py:84-160``, ``trainer_cvae.py:84-135`` and ``trainer_fader.py:84-146`` (same argument lists and returned tuples),
on ``SingleEncEngine`` + the fused clip / Adam kernel of ``GMVAETrainer``.

    SingleVAETrainer   loss = 5 CE_X + beta * KL(q || N(0,1)).mean() + l_r + l_n       (the reference adds its CONSTANT beta, not the
                       annealed beta0 it computes, trainer_singlevae.py:90-104); the pairwise regulariser acts on z[:, 0] (rhythm
                       density) and z[:, 1] (note density) (:107-120);        -> (loss, CE_X, l_r, l_n)
    CVAETrainer        loss = CE_X + beta0(step) * KL.mean()  (trainer_cvae.py:84-103)  -> (loss, CE_X);  ``evaluate`` takes no step, re-derives
                       the two densities from the rhythm / note tokens (:122-126) and reads the module-level ``step`` (= 0 for the whole run)
    FaderTrainer       loss = CE_X + beta0 * KL.mean() + l_adv_r + l_adv_n,  l_adv = min(step / 2000 * 1e-4, 1e-4) * MSE(head(reverse(z)), density)
                       (trainer_fader.py:84-110)                              -> (loss, CE_X, l_adv_r, l_adv_n)
"""
import math

import numpy as np
import torch

from .engine import E_VOCAB
from .trainer import GMVAETrainer, S_CE_X, S_L_N, S_L_R, S_TERMS_R, beta_schedule

S_ADV = 9                     # stats[9:11] = sum_b (o - density)^2 of the two adversarial heads


class _SingleEncTrainer(GMVAETrainer):
    CE_WEIGHT = 1.0

    def prepare_batch(self, d, r, n, c, r_density, n_density, y_label=None):
        b = super().prepare_batch(d, r, n, c, r_density, n_density, None)
        return b

    def draw_eps(self, B, T):
        dev = self.flat.param.device
        if self.dist is None or self.dist.world == 1:
            eps, extra = self.model._draw(B, T)
        else:                                      # data parallel: global draw, this rank's rows (GMVAETrainer.draw_eps)
            eps, extra = self.model._draw(B * self.dist.world, T)
            lo = self.dist.rank * B
            eps, extra = eps[lo:lo + B].contiguous(), (None if extra is None else extra[lo:lo + B].contiguous())
        return (eps.to(dev), None if extra is None else extra.to(dev))

    # hooks --------------------------------------------------------------------------------------
    def _cond(self, batch):
        raise NotImplementedError

    def _enc_extra(self, batch):
        return None

    def _kl_weight(self):
        return self.sp[0:3]           # {beta0 / Bg, ., .}: only w_lat acts (one component: the class term is constant)

    def _extra_terms(self, eng, batch, eps, lat, g_z, Bg, want_grads):
        pass

    # ---------------------------------------------------------------------------------------------
    def _forward_losses(self, step, batch, eps, want_grads):
        m = self.model
        eng = m.engine()
        ops = eng.ops
        d = batch[0]
        B, T = d.shape
        Bg = B if self.dist is None else self.dist.global_batch(B)
        fused = eng.fused_head
        if want_grads:
            eng.begin_step()                       # one fill for every zero-initialised accumulator of the step
        S = eng.forward(d, self._cond(batch), eps[0], self._enc_extra(batch), save=True, head=not fused)
        dec, lat = S["dec"], S["lat"]
        st = self.stats
        nll = eng.buf("nll_rows", (T * B,))
        gs = self.CE_WEIGHT / (Bg * T) if want_grads else 0.0
        if fused:
            ops.out_head(dec["hx1"].view(T * B, eng.H), eng.p["linear_out_g.weight"], eng.p["linear_out_g.bias"], B, T, d, nll_rows=nll,
                         grad_scale=gs, dlogits=dec["logits"] if want_grads else None)
        else:
            ops.vocab_logsoftmax(dec["logits"], B, T, E_VOCAB, target=d, nll_rows=nll, grad_scale=gs, dlogits=dec["logits"] if want_grads else None)
        ops.sum(nll, st[S_CE_X:S_CE_X + 1], 1.0 / (Bg * T))
        ops.colsum(lat["terms"], st[S_TERMS_R:S_TERMS_R + 4])
        g_z = eng.zbuf("g_z_e", (B, eng.Z)) if want_grads else None
        self._extra_terms(eng, batch, eps, lat, g_z, Bg, want_grads)
        return g_z, self._kl_weight(), None, beta_schedule(step, self.beta), Bg

    def _run_backward(self, fw, hook):
        self.model.engine().backward(self.flat.G, fw[0], fw[1], after_decoders=hook)

    def _pairwise(self, eng, lat, g_z, col, attr, slot, Bg, want_grads, tag):
        """the regulariser of GMVAETrainer._forward_losses on latent column `col`"""
        ops = eng.ops
        B = attr.shape[0]
        z0 = eng.buf("reg_z0_" + tag, (B,))
        z0.copy_(lat["z"][:, col])
        if self.dist is not None:
            z0_all, a_all, row0 = self.dist.gather_rows(z0, attr)
        else:
            z0_all, a_all, row0 = z0, attr, 0
        lrow = eng.buf("reg_rows_" + tag, (B,))
        dz0 = eng.buf("reg_dz0_" + tag, (B,)) if want_grads else None
        ops.pairwise_reg(z0_all, a_all, row0, B, lrow, 1.0 / (Bg * Bg), dz0)
        ops.sum(lrow, self.stats[slot:slot + 1], 1.0 / (Bg * Bg))
        if want_grads:
            g_z[:, col].copy_(dz0)

    def _stats(self):
        if self.dist is not None:
            self.dist.all_reduce_sum(self.stats)
        s = self.stats.tolist()
        check = getattr(self.model.engine().ops, "gru_sync_error", None)
        if check is not None and check(clear=True):
            raise RuntimeError("a weight-stationary GRU launch timed out waiting for its row group (sync error word set, now cleared): "
                               "the results of this step are invalid")
        return s

    def train(self, step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=None):
        batch = self.prepare_batch(d if d is not None else d_oh, r if r is not None else r_oh, n if n is not None else n_oh, c, r_density, n_density)
        if eps is None:
            eps = self.draw_eps(*batch[0].shape)
        beta0, Bg = self.step_device(step, batch, eps)
        return step + 1, self._tuple8(beta0, Bg, False, step)

    def _evaluate(self, step, batch, eps):
        if eps is None:
            eps = self.draw_eps(*batch[0].shape)
        self._sync_step_counter(step)
        Bg = batch[0].shape[0] if self.dist is None else self.dist.global_batch(batch[0].shape[0])
        self.model.engine().ops.step_params(self.counters, self.beta, self.lr, 0.9, 0.999, False, 1.0 / Bg, False, self.sp)
        fw = self._forward_losses(step, batch, eps, want_grads=False)
        return self._tuple8(fw[3], fw[4], False, step)

    def loss_and_grads(self, step, batch, eps):
        """forward + losses + backward WITHOUT the optimiser update (parity tests): fills flat.grad, returns the model's tuple"""
        eng = self.model.engine()
        self._sync_step_counter(step)
        Bg = batch[0].shape[0] if self.dist is None else self.dist.global_batch(batch[0].shape[0])
        eng.ops.step_params(self.counters, self.beta, self.lr, 0.9, 0.999, False, 1.0 / Bg, False, self.sp)
        fw = self._forward_losses(step, batch, eps, want_grads=True)
        self._run_backward(fw, None)
        eng.ops.sumsq(self.flat.grad, self.sumsq)
        return self._tuple8(fw[3], fw[4], False, step)


class SingleVAETrainer(_SingleEncTrainer):
    CE_WEIGHT = 5.0

    def _cond(self, batch):
        return batch[3]

    def _kl_weight(self):
        w = self.__dict__.get("_w3")
        if w is None:
            w = self._w3 = torch.zeros(3, device=self.flat.param.device)
        w[0:1].copy_(self.sp[7:8])            # beta / Bg, device-side copy (graph safe)
        return w

    def _extra_terms(self, eng, batch, eps, lat, g_z, Bg, want_grads):
        self._pairwise(eng, lat, g_z, 0, batch[4], S_L_R, Bg, want_grads, "r")
        self._pairwise(eng, lat, g_z, 1, batch[5], S_L_N, Bg, want_grads, "n")

    def _tuple8(self, beta0, Bg, supervised=False, step=0, cached=None):
        s = self._stats()
        ce_x, l_r, l_n, kld = s[S_CE_X], s[S_L_R], s[S_L_N], s[S_TERMS_R] / Bg
        return (5 * ce_x + self.beta * kld + l_r + l_n, ce_x, l_r, l_n)

    @torch.no_grad()
    def evaluate(self, step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=None):
        batch = self.prepare_batch(d if d is not None else d_oh, r if r is not None else r_oh, n if n is not None else n_oh, c, r_density, n_density)
        return self._evaluate(step, batch, eps)


class CVAETrainer(_SingleEncTrainer):
    def prepare_batch(self, d, r, n, c, r_density, n_density, y_label=None):
        b = list(super().prepare_batch(d, r, n, c, r_density, n_density))
        dev = self.flat.param.device
        # the densities enter the MODEL here (float32 columns of the encoder / decoder inputs, trainer_cvae.py:199-200)
        # (the reference's loops hand them over as (B, 1) float tensors, trainer_cvae.py:171 / trainer_fader.py:180; (B,) arrays work too)
        b[4], b[5] = b[4].reshape(-1), b[5].reshape(-1)
        b.append(torch.stack([b[4].float(), b[5].float()], dim=1).to(dev).contiguous())
        return tuple(b)

    def _cond(self, batch):
        return batch[7]

    def _enc_extra(self, batch):
        return batch[7]

    def _tuple8(self, beta0, Bg, supervised=False, step=0, cached=None):
        s = self._stats()
        ce_x, kld = s[S_CE_X], s[S_TERMS_R] / Bg
        return (ce_x + beta0 * kld, ce_x)

    @torch.no_grad()
    def evaluate(self, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=None):
        """trainer_cvae.py:120-135: densities re-derived from the tokens (share of 1s in r; mean of n), float32; beta0 of step 0"""
        rt = torch.as_tensor(r if r is not None else r_oh)
        nt = torch.as_tensor(n if n is not None else n_oh)
        if rt.dim() == 3:
            rt, nt = rt.argmax(-1), nt.argmax(-1)
        rd = torch.tensor([float((k == 1).sum()) / len(k) for k in rt.cpu().numpy()])
        nd = torch.tensor([float(k.sum()) / len(k) for k in nt.cpu().numpy()])
        batch = self.prepare_batch(d if d is not None else d_oh, rt, nt, c, rd, nd)
        return self._evaluate(0, batch, eps)


class FaderTrainer(CVAETrainer):
    def _enc_extra(self, batch):
        return None

    def _extra_terms(self, eng, batch, eps, lat, g_z, Bg, want_grads):
        ops, P = eng.ops, eng.p
        B = batch[0].shape[0]
        o, lrow = eng.buf("adv_o", (B, 2)), eng.buf("adv_rows", (B, 2))
        da = eng.buf("adv_da", (B, 2)) if want_grads else None
        mask = eps[1] if eps[1] is not None else torch.ones(B, 2, device=o.device)
        ops.adv_head(lat["z"], P["discriminator_r.weight"], P["discriminator_n.weight"], P["discriminator_r.bias"], P["discriminator_n.bias"],
                     mask.contiguous(), batch[7], self.sp[6:7], 1.0 / Bg, o, lrow, da, g_z)
        ops.colsum(lrow, self.stats[S_ADV:S_ADV + 2])
        self._adv = (da, lat["z"])

    def _run_backward(self, fw, hook):
        super()._run_backward(fw, hook)
        ops, G = self.model.engine().ops, self.flat.G
        da, z = self._adv
        for a, e in enumerate(("r", "n")):                         # dW = da^T z, db = column sum of da
            ops.gemm(da[:, a:a + 1], z, G["discriminator_%s.weight" % e], a_k=False, b_k=False)
            ops.colsum(da[:, a:a + 1], G["discriminator_%s.bias" % e])

    def _tuple8(self, beta0, Bg, supervised=False, step=0, cached=None):
        s = self._stats()
        lam = min(step / 2000 * 1e-4, 1e-4)
        ce_x, kld = s[S_CE_X], s[S_TERMS_R] / Bg
        l_adv_r, l_adv_n = lam * s[S_ADV] / Bg, lam * s[S_ADV + 1] / Bg
        return (ce_x + beta0 * kld + l_adv_r + l_adv_n, ce_x, l_adv_r, l_adv_n)

    @torch.no_grad()
    def evaluate(self, step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=None):
        batch = self.prepare_batch(d if d is not None else d_oh, r if r is not None else r_oh, n if n is not None else n_oh, c, r_density, n_density)
        return self._evaluate(step, batch, eps)


# ---------------------------------------------------------------------------------------------------------------------------------
class GLSRTrainer(GMVAETrainer):
    """``train`` / ``evaluate`` of the reference's ``trainer_glsr.py`` (:267-321) for ``MusicAttrRegVAE``:
    loss = 5 CE_X + CE_R + CE_N + beta0(step) * (KL_r + KL_n)  (:87-115; here the step argument IS used)  and, once step > 20, the GLSR
    regulariser of Hadjeres et al. as the reference implements it (:118-264): for the rhythm and then the note latent, z[:, 0] is moved by
    +-delta (delta = (1 + U(0,1)) * 1e-2 per sample), the TRAIN-mode (teacher-forced) global decoder is run for 100 steps on both, an
    attribute is read off the output probabilities, and  -log N(0,1)( (attr+ - attr-) / (2 delta) )  is averaged over the batch.
      note density   = sum_t P_t(note-on tokens 2..89)                                                   (:135-137)
      rhythm density = a host-side walk over the 100 steps (:139-165): between "time separators" (steps whose time-shift mass
                       P_t(180..277) >= 0.9) the note-on mass OF SAMPLE 0 (``played_notes[0][i]`` - reproduced) is accumulated; a flush
                       adds 1 (no gradient) if the accumulated mass exceeds 1e-2, else the mass itself; divided by sum_t P_t(180..277).
    The four extra decodes and their backward passes run on the HIP kernels (teacher-forced decoder scans, fn_masked_prob); the walk
    over the 2 x B x 100 masses is host arithmetic exactly as in the reference (it branches on ``.item()`` per step), so the step syncs
    with the host four times and is not graph-captured.  -> (loss, CE_X, CE_R, CE_N, l_r, l_n).  Not data parallel (sample 0 is global)."""
    NOTES, SEPS, EPSILON, STEPS = (2, 90), (180, 278), 1e-2, 100

    def __init__(self, model, lr=1e-3, beta=0.1, max_norm=1.0, dist_ctx=None):
        if dist_ctx is not None:
            raise ValueError("GLSRTrainer is single-process: the reference's rhythm-density walk reads sample 0 of the batch for every row")
        super().__init__(model, lr=lr, beta=beta, max_norm=max_norm)
        self.use_graph = False
        self.flat2 = torch.zeros_like(self.flat.grad)          # parameter gradients of one extra decoder pass
        self.acc = torch.zeros_like(self.flat.grad)            # ... summed over the four passes
        self.G2 = {k: self.flat2[o:o + self.flat.G[k].numel()].view_as(self.flat.G[k]) for k, o in self.flat.offsets.items()}
        self._glsr_active = False

    def draw_eps(self, B, T, step=None):
        """forward draws (randn x2, T x rand(1)), then - if the regulariser is active - per latent: rand(B) for the deltas and the
        2 x 100 rand(1) of the two train-mode decodes (trainer_glsr.py:175-177,183-186, model_v2.py:129-131)"""
        dev = self.flat.param.device
        eps = list(self.model._draw_eps(B, T, dev))
        deltas = []
        if step is not None and step > 20:
            for _ in range(2):
                deltas.append(((1 + torch.rand(B)) * self.EPSILON).to(dev))
                for _ in range(2 * self.STEPS):
                    torch.rand(1)
        return tuple(eps + [deltas])

    # ---------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _rhythm_density(PN, PS):
        """host walk of trainer_glsr.py:139-165 on float32 masses PN / PS [steps][B] -> (r [B], dPN0 [steps][B]: d r_b / d PN[i][0] per
        (i, b), dPS [steps][B]: d r_b / d PS[i][b])"""
        steps, B = PS.shape
        r = np.zeros(B, np.float32)
        dPN0, dPS = np.zeros((steps, B), np.float32), np.zeros((steps, B), np.float32)
        for b in range(B):
            total, total_grad_idx, cur, cur_idx, started = np.float32(0), [], np.float32(0), [], False
            for i in range(steps):
                if PS[i, b] < 0.9:
                    cur = np.float32(cur + PN[i, 0])
                    cur_idx.append(i)
                    started = True
                else:
                    if not started or cur == 0:
                        continue
                    if cur > 1e-2:
                        total = np.float32(total + 1)
                    else:
                        total = np.float32(total + cur)
                        total_grad_idx += cur_idx
                    cur, cur_idx, started = np.float32(0), [], False
            S = np.float32(PS[:, b].sum(dtype=np.float32))
            rb = np.float32(total / S)
            if rb != 0:
                r[b] = rb
                for i in total_grad_idx:
                    dPN0[i, b] += 1.0 / S
                dPS[:, b] = -total / (S * S)
        return r, dPN0, dPS

    def _regulariser(self, eng, S, batch, Bg, want_grads):
        ops, Z = eng.ops, eng.Z
        d, r, n, c, rd, nd, labels = batch
        B, T = d.shape
        lat = S["lat"]
        dev = d.device
        gz = {e: eng.zbuf("g_z_" + e, (B, Z)) for e in ("r", "n")}
        lat_up = {e: dict(g_z=gz[e]) for e in ("r", "n")}
        self.stats[S_L_R:S_L_N + 1].zero_()
        self._glsr_active = False
        deltas = self._cur_eps[2] if len(self._cur_eps) > 2 else []
        if self._step_now <= 20 or not deltas:
            return lat_up
        if T < self.STEPS:
            raise IndexError("GLSR decodes %d teacher-forced steps: the batch needs T >= %d (trainer_glsr.py:183)" % (self.STEPS, self.STEPS))
        st_ = self.STEPS
        d100 = eng.buf("glsr_d", (B, st_), torch.int32)
        d100.copy_(d[:, :st_])
        base = eng.pack_zc(lat["r"]["z"], lat["n"]["z"], c).clone()
        ranges = (self.NOTES, self.SEPS)

        def decode(attr, sign, save):
            zc = eng.buf("glsr_zc", base.shape)
            zc.copy_(base)
            zc[:, 0 if attr == 0 else Z] += sign * deltas[attr]
            eng.buf_ns = "glsr/"             # own buffers: the main pass's saved decoder state (same shapes for B x H tensors) stays intact
            try:
                return eng.global_decoder_tf(d100, zc, save=save)
            finally:
                eng.buf_ns = ""

        sums = eng.buf("glsr_sums", (st_ * B, 2))
        mass = {}
        for attr in (0, 1):
            for sign in (1.0, -1.0):
                dec = decode(attr, sign, False)
                ops.masked_prob(dec["logits"], E_VOCAB, ranges, sums=sums)
                mass[(attr, sign)] = sums.view(st_, B, 2).cpu().numpy().copy()            # host sync (the reference calls .item() per step)
        # ---- host: attributes, losses, and d loss / d masses ------------------------------------------------------------------
        half_log_2pi = 0.5 * math.log(2 * math.pi)
        w = {}
        dl = [deltas[a].cpu().numpy() for a in (0, 1)]
        rp = self._rhythm_density(mass[(0, 1.0)][:, :, 0], mass[(0, 1.0)][:, :, 1])
        rm = self._rhythm_density(mass[(0, -1.0)][:, :, 0], mass[(0, -1.0)][:, :, 1])
        g_r = ((rp[0] - rm[0]) / (2 * dl[0])).astype(np.float32)
        l_r = float(np.mean(0.5 * g_r * g_r + half_log_2pi, dtype=np.float32))
        for sign, (rv, dPN0, dPS) in ((1.0, rp), (-1.0, rm)):
            coef = sign * g_r / (2 * dl[0]) / Bg                                          # d l_r / d r(sign)_b
            ww = np.zeros((st_, B, 2), np.float32)
            ww[:, 0, 0] = (dPN0 * coef[None, :]).sum(1)                                   # every row's walk reads the notes of SAMPLE 0
            ww[:, :, 1] = dPS * coef[None, :]
            w[(0, sign)] = ww
        n_p, n_m = mass[(1, 1.0)][:, :, 0].sum(0, dtype=np.float32), mass[(1, -1.0)][:, :, 0].sum(0, dtype=np.float32)
        g_n = ((n_p - n_m) / (2 * dl[1])).astype(np.float32)
        l_n = float(np.mean(0.5 * g_n * g_n + half_log_2pi, dtype=np.float32))
        for sign in (1.0, -1.0):
            ww = np.zeros((st_, B, 2), np.float32)
            ww[:, :, 0] = (sign * g_n / (2 * dl[1]) / Bg)[None, :]
            w[(1, sign)] = ww
        self.stats[S_L_R:S_L_N + 1].copy_(torch.tensor([l_r, l_n], dtype=torch.float32))
        if not want_grads:
            return lat_up
        # ---- backward of the four decodes: recompute forward (activations saved), seed dlogits, reverse scans -----------------------
        from .engine import ops_sort
        self.acc.zero_()
        gzc = eng.zbuf("glsr_gzc", base.shape)
        S2 = dict(d=d100, sort={"d": ops_sort(eng, "d100", d100, E_VOCAB)})
        P = eng.p
        wdev = eng.buf("glsr_w", (st_ * B, 2))
        for attr in (0, 1):
            for sign in (1.0, -1.0):
                if not np.any(w[(attr, sign)]):
                    continue                                                              # constant attribute: no gradient
                S2["dec"] = decode(attr, sign, True)
                wdev.copy_(torch.from_numpy(w[(attr, sign)]).view(st_ * B, 2))
                ops.masked_prob(S2["dec"]["logits"], E_VOCAB, ranges, w=wdev, dlogits=S2["dec"]["logits"])
                eng.buf_ns = "glsr/"
                try:
                    gd = eng._bwd_global_decoder_scans(S2)
                finally:
                    eng.buf_ns = ""
                ops.gemm(gd["drb_g"], P["grucell_g.weight_ih"][:, E_VOCAB:], gzc, a_k=True, b_k=False, beta=1.0)
                ops.gemm(gd["dh0_g"], P["linear_init_global.weight"], gzc, a_k=True, b_k=False, beta=1.0)
                self.flat2.zero_()
                eng._bwd_global_decoder_params(self.G2, S2, gd)
                ops.axpy(1.0, self.flat2, self.acc)
        gz["r"].copy_(gzc[:, :Z])
        gz["n"].copy_(gzc[:, Z:2 * Z])
        self._glsr_active = True
        return lat_up

    def _forward_losses(self, step, batch, eps, want_grads):
        self._cur_eps = eps
        # The regulariser runs four more decoder passes (weight-stationary, whole-chip launches that spin on each other's progress).  On the
        # side lane they would be in flight together with the main pass's decoder backward on the main stream: two such grids of 256
        # workgroups cannot be resident at once at hidden 512 / batch 256 and would starve each other into the bounded-spin error.
        self.model.engine().losses_on_side = False
        return super()._forward_losses(step, batch, eps, want_grads)

    def _run_backward(self, fw, hook):
        super()._run_backward(fw, hook)
        if self._glsr_active:
            self.model.engine().ops.axpy(1.0, self.acc, self.flat.grad)

    def train(self, step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=None):
        batch = self.prepare_batch(d if d is not None else d_oh, r if r is not None else r_oh, n if n is not None else n_oh, c, r_density, n_density)
        if eps is None:
            eps = self.draw_eps(*batch[0].shape, step=step)
        beta0, Bg = self.step_device(step, batch, eps)
        return step + 1, self._tuple8(beta0, Bg, False)[:6]

    @torch.no_grad()
    def evaluate(self, step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=None):
        batch = self.prepare_batch(d if d is not None else d_oh, r if r is not None else r_oh, n if n is not None else n_oh, c, r_density, n_density)
        if eps is None:
            eps = self.draw_eps(*batch[0].shape, step=step)
        self.model.engine()
        fw = self._forward_losses(step, batch, eps, want_grads=False)
        return self._tuple8(fw[3], fw[4], False)[:6]
