"""The reference's ``model_v2.py`` model families on the MI355X HIP kernels: same class names, constructor keywords, ``state_dict``
keys (parameter containers built in the reference's order -> bit-identical seeded initial weights), ``forward`` argument lists and
return nesting.

    MusicAttrRegVAE     model_v2.py:9-171     (vae_model.py: two encoders + sub-decoders, autograd-connected drop-in forward)
    MusicAttrSingleVAE  model_v2.py:174-285   one encoder, latent of width 2 Z, decoder conditioned on [z | chroma]
    MusicAttrCVAE       model_v2.py:288-423   encoder input = [one-hot | r_density | n_density], decoder conditioned on [z | densities]
    MusicAttrFaderNets  model_v2.py:438-586   CVAE-style decoder + two adversarial density regressors behind a gradient reversal

The three single-encoder classes run on ``SingleEncEngine``.  In train mode with autograd enabled their ``forward`` hangs on ONE autograd
node (``_SingleEncFunction``; the Fader heads on ``_AdvHeadFunction`` behind the gradient reversal), so the reference's
``loss.backward(); optimizer.step()`` loops work (call ``weights_changed()`` after the step); the fast path are the fused trainers of
``trainer_v2.py`` (``SingleVAETrainer``, ``CVAETrainer``, ``FaderTrainer``), which reproduce ``train`` / ``evaluate`` of the reference's
``trainer_singlevae.py`` / ``trainer_cvae.py`` / ``trainer_fader.py``.
Random draws follow the reference's order on torch's global CPU generator: ``randn(B, ZL)`` for the reparameterisation, (Fader only:
the two dropout masks of model_v2.py:574-575), then - in train mode - one ``torch.rand(1)`` per decoder step (:260, :400, :552).
"""
import torch
from torch import nn
from torch.distributions import Normal

from .engine import E_VOCAB
from .engine_single import SingleEncEngine
from .gmm_model import MusicAttrRegGMVAE
from .vae_model import MusicAttrRegVAE  # noqa: F401  (re-exported: the fourth class of the reference module)


class _SingleEncModel(MusicAttrRegGMVAE):
    GRU = "gru"                 # attribute name of the encoder
    ENC_EXTRA = 0               # dense encoder input columns after the one-hot ones
    UNUSED = ()                 # parameter prefixes that exist in state_dict but never take part in forward

    def _common(self, roll_dims, rhythm_dims, hidden_dims, z_dims, n_step, k, zlat):
        self.n_step = n_step
        self.roll_dims = roll_dims
        self.hidden_dims = hidden_dims
        self.eps = 100
        self.rhythm_dims = rhythm_dims
        self.sample = None
        self.iteration = 0
        self.z_dims = z_dims
        self.latent_dim = zlat          # width of the latent the heads produce
        self.k = torch.FloatTensor([k])
        self.n_component = 1
        self._engine = None
        self._engine_key = None
        self._weights_version = None
        self._version = 0

    def _make_engine(self, ops, dev):
        return SingleEncEngine(ops, self._engine_params(), self.hidden_dims, self.latent_dim, dev, gru=self.GRU, enc_extra=self.ENC_EXTRA)

    def used_parameters(self):
        return [(k, p) for k, p in self.named_parameters() if not k.startswith(self.UNUSED)]

    def _device(self):
        return self.mu.weight.device

    def _draw(self, B, T):
        """the reference's draws for one forward: eps, (subclass extras), T x rand(1) in train mode"""
        eps = torch.randn(B, self.latent_dim)
        extra = self._draw_extra(B)
        if self.training:
            for _ in range(T):
                torch.rand(1)
        return eps, extra

    def _draw_extra(self, B):
        return None

    def approx_qy_x(self, *a, **k):
        raise AttributeError("%s has no mixture posterior" % type(self).__name__)

    def sub_decoders(self, *a, **k):
        raise AttributeError("%s has no sub-decoders (the reference's method refers to modules that do not exist, model_v2.py:349-368)"
                             % type(self).__name__)

    def encode(self, *a, **k):
        return self.encoder(*a, **k)

    @torch.no_grad()
    def _encode_dis(self, x, extra=None):
        eng = self.engine()
        d = self._indices(x, self.roll_dims)
        enc = eng.encode(d, extra, save=False)
        Z = self.latent_dim
        lat = eng.latent1(enc["pre"], torch.zeros(d.shape[0], Z, device=d.device))
        return Normal(enc["pre"][:, :Z].clone(), lat["sigma"].clone())

    def _forward_core(self, x, cond, extra, eps):
        """encode -> z -> global decoder (teacher forced in train mode, greedy in eval mode) -> (out, dis, z_lat).  Train mode with autograd
        enabled: the outputs hang on ONE autograd node whose backward runs the HIP backward kernels (a reference-style
        ``loss.backward(); optimizer.step()`` loop works; call ``model.weights_changed()`` after the optimiser step)."""
        if self.training and torch.is_grad_enabled():
            eng = self.engine()
            d = self._indices(x, self.roll_dims)
            names = [k for k, _ in self.used_parameters() if not k.startswith("discriminator_")]
            plist = [p for k, p in self.used_parameters() if not k.startswith("discriminator_")]
            out, mu, sigma, z = _SingleEncFunction.apply(self, names, d, cond, extra, eps, *plist)
            return out, Normal(mu, sigma), z
        return self._forward_core_no_grad(x, cond, extra, eps)

    @torch.no_grad()
    def _forward_core_no_grad(self, x, cond, extra, eps):
        eng = self.engine()
        d = self._indices(x, self.roll_dims)
        B, T = d.shape
        Z = self.latent_dim
        S = eng.forward(d, cond, eps, extra, save=False) if self.training else None
        if self.training:
            out = torch.empty(B, T, E_VOCAB, device=d.device)
            eng.ops.vocab_logsoftmax(S["dec"]["logits"], B, T, E_VOCAB, logp_bt=out)
            pre, lat = S["enc"]["pre"], S["lat"]
        else:
            enc = eng.encode(d, extra, save=False)
            lat = eng.latent1(enc["pre"], eps)
            pre = enc["pre"]
            from .decode import greedy_decode
            out, _ = greedy_decode(self, eng.pack_zc(lat["z"], cond).clone(), T)
        return out, Normal(pre[:, :Z].clone(), lat["sigma"].clone()), lat["z"].clone()


class MusicAttrSingleVAE(_SingleEncModel):
    """model_v2.py:174-285."""
    GRU = "gru"

    def __init__(self, roll_dims, rhythm_dims, note_dims, chroma_dims, hidden_dims, z_dims, n_step, k=1000):
        nn.Module.__init__(self)
        if (roll_dims, chroma_dims) != (342, 24):
            raise ValueError("the HIP path is built for roll / chroma dims 342 / 24")
        self.gru = nn.GRU(roll_dims, hidden_dims, batch_first=True, bidirectional=True)
        self.e_dropout = nn.Dropout(p=0.3)                       # constructed by the reference, never applied (:191, :225-231)
        self.mu, self.var = nn.Linear(hidden_dims * 2, z_dims * 2), nn.Linear(hidden_dims * 2, z_dims * 2)
        self.linear_init_global = nn.Linear(z_dims * 2 + 24, hidden_dims)
        self.grucell_g = nn.GRUCell(z_dims * 2 + 24 + roll_dims, hidden_dims)
        self.grucell_g_2 = nn.GRUCell(hidden_dims, hidden_dims)
        self.linear_out_g = nn.Linear(hidden_dims, roll_dims)
        self._common(roll_dims, rhythm_dims, hidden_dims, z_dims, n_step, k, zlat=2 * z_dims)

    def encoder(self, x):
        return self._encode_dis(x)

    def forward(self, x, chroma, eps=None):
        """-> (out, dis, z) with z = cat([z, chroma]) (model_v2.py:263-285)"""
        if self.training:
            self.sample = x
            self.iteration += 1
        dev = self._device()
        B, T = x.shape[0], x.shape[1]
        if eps is None:
            eps, _ = self._draw(B, T)
        c = chroma.float().contiguous().to(dev)
        out, dis, z = self._forward_core(x, c, None, eps.float().contiguous().to(dev))
        return out, dis, torch.cat([z, c], dim=1)


class MusicAttrCVAE(_SingleEncModel):
    """model_v2.py:288-423."""
    GRU = "gru_e"
    ENC_EXTRA = 2
    UNUSED = ("c_r.", "c_n.")

    def __init__(self, roll_dims, rhythm_dims, note_dims, chroma_dims, hidden_dims, z_dims, n_step, k=1000):
        nn.Module.__init__(self)
        if roll_dims != 342:
            raise ValueError("the HIP path is built for roll dims 342")
        self.gru_e = nn.GRU(roll_dims + 2, hidden_dims, batch_first=True, bidirectional=True)
        self.c_r = nn.Linear(z_dims, 3)
        self.c_n = nn.Linear(z_dims, 3)
        self.mu, self.var = nn.Linear(hidden_dims * 2, z_dims), nn.Linear(hidden_dims * 2, z_dims)
        self.linear_init_global = nn.Linear(z_dims + 2, hidden_dims)
        self.grucell_g = nn.GRUCell(z_dims + 2 + roll_dims, hidden_dims)
        self.grucell_g_2 = nn.GRUCell(hidden_dims, hidden_dims)
        self.linear_out_g = nn.Linear(hidden_dims, roll_dims)
        self._common(roll_dims, rhythm_dims, hidden_dims, z_dims, n_step, k, zlat=z_dims)

    def _dens(self, r_density, n_density):
        dev = self._device()
        return torch.cat([torch.as_tensor(r_density).float().view(-1, 1), torch.as_tensor(n_density).float().view(-1, 1)], dim=1).to(dev).contiguous()

    def encoder(self, x, r_density, n_density, chroma=None):
        return self._encode_dis(x, self._dens(r_density, n_density))

    def forward(self, x, rhythm, note, chroma, r_density, n_density, eps=None):
        """-> (out, dis, z) with z = cat([z, r_density, n_density]) (model_v2.py:398-423); rhythm / note / chroma are accepted and
        unused, as in the reference"""
        if self.training:
            self.sample = x
            self.iteration += 1
        dev = self._device()
        if eps is None:
            eps, _ = self._draw(x.shape[0], x.shape[1])
        dens = self._dens(r_density, n_density)
        out, dis, z = self._forward_core(x, dens, dens, eps.float().contiguous().to(dev))
        return out, dis, torch.cat([z, dens], dim=-1)


class MusicAttrFaderNets(_SingleEncModel):
    """model_v2.py:438-586 (``ReverseLayerF`` :426-435 lives in the fused backward: the encoder receives the negated gradient of the
    adversarial heads, csrc/loss.hip adv_head_kernel)."""
    GRU = "gru_e"
    UNUSED = ("c_r.", "c_n.")
    P_DROP = 0.3

    def __init__(self, roll_dims, rhythm_dims, note_dims, chroma_dims, hidden_dims, z_dims, n_step, k=1000):
        nn.Module.__init__(self)
        if roll_dims != 342:
            raise ValueError("the HIP path is built for roll dims 342")
        self.gru_e = nn.GRU(roll_dims, hidden_dims, batch_first=True, bidirectional=True)
        self.c_r = nn.Linear(z_dims, 3)
        self.c_n = nn.Linear(z_dims, 3)
        self.mu, self.var = nn.Linear(hidden_dims * 2, z_dims), nn.Linear(hidden_dims * 2, z_dims)
        self.discriminator_r = nn.Linear(z_dims, 1)
        self.discriminator_n = nn.Linear(z_dims, 1)
        self.dropout = nn.Dropout(p=self.P_DROP)
        self.linear_init_global = nn.Linear(z_dims + 2, hidden_dims)
        self.grucell_g = nn.GRUCell(z_dims + 2 + roll_dims, hidden_dims)
        self.grucell_g_2 = nn.GRUCell(hidden_dims, hidden_dims)
        self.linear_out_g = nn.Linear(hidden_dims, roll_dims)
        self._common(roll_dims, rhythm_dims, hidden_dims, z_dims, n_step, k, zlat=z_dims)

    _dens = MusicAttrCVAE._dens

    def _draw_extra(self, B):
        """the two dropout keep-masks, scaled by 1/(1-p), drawn as the reference's nn.Dropout draws them on the CPU generator
        (rhythm head first, model_v2.py:574-575); all ones in eval mode.

        RNG contract: the parity pin is the reference's CPU path on identical seeds (tests/golden/siblings.npz), where nn.Dropout
        consumes the CPU generator between randn(B, Z) and the T x rand(1) draws - reproduced here.  The reference run on a GPU draws
        its masks from the CUDA generator instead (whose Philox stream cannot be reproduced by another kernel anyway), so after the
        first Fader forward its CPU generator is 2 x B draws behind ours; pass eps / mask explicitly to be independent of that."""
        if not self.training:
            return torch.ones(B, 2)
        one = torch.ones(B, 1)
        return torch.cat([nn.functional.dropout(one, self.P_DROP, True), nn.functional.dropout(one, self.P_DROP, True)], dim=1)

    def encoder(self, x):
        return self._encode_dis(x)

    def adversarial_heads(self, z, mask, dens=None):
        """(r_out, n_out) = dropout(relu(discriminator(reverse(z)))) -> two (B, 1) tensors (model_v2.py:572-575); in train mode with
        autograd enabled connected to the discriminators and - through the gradient reversal - to z"""
        if self.training and torch.is_grad_enabled():
            self.engine()
            o = _AdvHeadFunction.apply(self, z, mask.float().contiguous().to(z.device), self.discriminator_r.weight, self.discriminator_r.bias,
                                       self.discriminator_n.weight, self.discriminator_n.bias)
            return o[:, 0:1], o[:, 1:2]
        return self._adversarial_heads_no_grad(z, mask, dens)

    @torch.no_grad()
    def _adversarial_heads_no_grad(self, z, mask, dens=None):
        eng = self.engine()
        B = z.shape[0]
        o, lr_ = torch.empty(B, 2, device=z.device), torch.empty(B, 2, device=z.device)
        dens = torch.zeros(B, 2, device=z.device) if dens is None else dens
        P = eng.p
        eng.ops.adv_head(z.contiguous(), P["discriminator_r.weight"], P["discriminator_n.weight"], P["discriminator_r.bias"],
                         P["discriminator_n.bias"], mask.float().contiguous().to(z.device), dens, None, 1.0, o, lr_)
        return o[:, 0:1].clone(), o[:, 1:2].clone()

    def forward(self, x, rhythm, note, chroma, r_density, n_density, eps=None, mask=None):
        """-> ((out, r_out, n_out), dis, z) with z = cat([z, r_density, n_density]) (model_v2.py:556-586)"""
        if self.training:
            self.sample = x
            self.iteration += 1
        dev = self._device()
        if eps is None:
            eps, mask = self._draw(x.shape[0], x.shape[1])
        elif mask is None:
            mask = torch.ones(x.shape[0], 2)
        dens = self._dens(r_density, n_density)
        out, dis, z = self._forward_core(x, dens, None, eps.float().contiguous().to(dev))
        r_out, n_out = self.adversarial_heads(z, mask, dens)
        return (out, r_out, n_out), dis, torch.cat([z, dens], dim=-1)


# ------------------------------------------------------------------------------------------------------------------------------
# autograd nodes of the drop-in forward of the single-encoder families
# ------------------------------------------------------------------------------------------------------------------------------
class _SingleEncFunction(torch.autograd.Function):
    """forward = SingleEncEngine.forward + vocabulary log-softmax, backward = SingleEncEngine.backward (upstream gradients wrt out, mu,
    sigma and z; the reference-style losses compute their KL from Normal(mu, sigma) in torch)"""

    @staticmethod
    def forward(ctx, model, names, d, cond, extra, eps, *params):
        eng = model._engine
        B, T = d.shape
        Z = eng.Z
        S = eng.forward(d, cond, eps, extra, save=True)
        out = torch.empty(B, T, E_VOCAB, device=d.device)
        eng.ops.vocab_logsoftmax(S["dec"]["logits"], B, T, E_VOCAB, logp_bt=out)
        ctx.model, ctx.names, ctx.fw_id = model, names, id(S)
        ctx.save_for_backward(out)
        return out, S["enc"]["pre"][:, :Z].clone(), S["lat"]["sigma"].clone(), S["lat"]["z"].clone()

    @staticmethod
    def backward(ctx, g_out, g_mu, g_sigma, g_z):
        model = ctx.model
        eng = model._engine
        S = eng.saved
        if S is None or id(S) != ctx.fw_id:
            raise RuntimeError("backward() must follow the forward() that produced these outputs (the engine keeps one set of saved activations)")
        (out,) = ctx.saved_tensors
        dense = lambda g, ref: torch.zeros_like(ref) if g is None else g.float().contiguous()
        eng.ops.vocab_logsoftmax_bwd(out, dense(g_out, out), S["dec"]["logits"])
        gz = dense(g_z, S["lat"]["z"]).clone()
        G = {k: torch.zeros_like(p) for k, p in model.used_parameters() if k in ctx.names}
        eng.backward(G, gz, None, g_mu=None if g_mu is None else g_mu.float().contiguous(),
                     g_sigma=None if g_sigma is None else g_sigma.float().contiguous())
        return (None,) * 6 + tuple(G[k] for k in ctx.names)


class _AdvHeadFunction(torch.autograd.Function):
    """o[b][a] = relu(w_a . z[b] + b_a) * mask[b][a] behind the gradient reversal of model_v2.py:426-435 (z receives MINUS the gradient)"""

    @staticmethod
    def forward(ctx, model, z, mask, w_r, b_r, w_n, b_n):
        eng = model._engine
        B = z.shape[0]
        o, lrow = torch.empty(B, 2, device=z.device), torch.empty(B, 2, device=z.device)
        zc = z.detach().float().contiguous()
        eng.ops.adv_head(zc, w_r.data, w_n.data, b_r.data, b_n.data, mask, torch.zeros(B, 2, device=z.device), None, 1.0, o, lrow)
        ctx.model = model
        ctx.save_for_backward(zc, mask, o, w_r, w_n)
        return o

    @staticmethod
    def backward(ctx, g_o):
        eng = ctx.model._engine
        ops = eng.ops
        z, mask, o, w_r, w_n = ctx.saved_tensors
        da = (g_o.float() * mask * (o > 0).float()).contiguous()          # d o / d pre: the keep-mask where the ReLU is open
        Z = w_r.numel()
        W = torch.cat([w_r.data.view(1, Z), w_n.data.view(1, Z)], 0).contiguous()       # [2][Z]
        dz = torch.empty(z.shape[0], Z, device=z.device)
        ops.gemm(da, W, dz, a_k=True, b_k=False, alpha=-1.0)                              # gradient reversal
        dW = torch.empty(2, Z, device=z.device)
        ops.gemm(da, z[:, :Z].contiguous() if z.shape[1] != Z else z, dW, a_k=False, b_k=False)
        db = torch.empty(2, device=z.device)
        ops.colsum(da, db)
        dz_full = dz if z.shape[1] == Z else torch.cat([dz, torch.zeros(z.shape[0], z.shape[1] - Z, device=z.device)], 1)
        return None, dz_full, None, dW[0:1].view_as(w_r), db[0:1].view(-1), dW[1:2].view_as(w_n), db[1:2].view(-1)
