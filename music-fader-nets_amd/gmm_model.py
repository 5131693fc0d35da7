"""MusicAttrRegGMVAE with the reference's class surface, executed by the MI355X HIP kernels.

Drop-in for ``/root/reference/gmm_model.py:10-259``: same constructor keywords, same ``state_dict`` keys
and shapes (the parameter containers are the same torch modules, built in the same order, so a seeded
construction yields bit-identical initial weights), same methods (``forward, encode, sub_decoders,
global_decoder, approx_qy_x, _sampling``), same nested return tuple of ``forward`` (gmm_model.py:251-259).
None of the torch modules is ever *called*: all arithmetic runs in libfadernets_hip.so.  On a CPU
tensor, or without the library, every method raises - there is no eager fallback.
"""
import numpy as np
import torch
from torch import nn
from torch.distributions import Normal

from .engine import E_VOCAB, LOGIT_LD, Engine

UNUSED_PREFIXES = ("gru_c.", "gru_d_c.", "mu_c.", "var_c.", "c_r.", "c_n.", "linear_init_c.", "linear_out_c.")
FROZEN = ("logvar_r_lookup.weight", "logvar_n_lookup.weight")


class MusicAttrRegGMVAE(nn.Module):
    def __init__(self, roll_dims, rhythm_dims, note_dims, chroma_dims, hidden_dims, z_dims, n_step, n_component=4):
        super().__init__()
        if (roll_dims, rhythm_dims, note_dims, chroma_dims) != (342, 3, 16, 24):
            raise ValueError("the HIP path is built for roll/rhythm/note/chroma dims 342/3/16/24 (trainer_gmm.py:35-38)")
        self.n_component = n_component
        self.latent_dim = z_dims
        self.roll_dims = roll_dims
        self.hidden_dims = hidden_dims
        self.eps = 100                   # teacher-forcing threshold: torch.rand(1) < 100 always (gmm_model.py:30,141)
        self.sample = None
        # ---- parameter containers, in the construction order of gmm_model.py:33-71 -------------------
        self.gru_r = nn.GRU(roll_dims, hidden_dims, batch_first=True, bidirectional=True)
        self.gru_n = nn.GRU(roll_dims, hidden_dims, batch_first=True, bidirectional=True)
        self.gru_c = nn.GRU(roll_dims, hidden_dims, batch_first=True, bidirectional=True)
        self.c_r = nn.Linear(z_dims, 3)
        self.c_n = nn.Linear(z_dims, 3)
        self.gru_d_r = nn.GRU(z_dims + rhythm_dims, hidden_dims, batch_first=True)
        self.gru_d_n = nn.GRU(z_dims + note_dims, hidden_dims, batch_first=True)
        self.gru_d_c = nn.GRU(z_dims + chroma_dims, hidden_dims, batch_first=True)
        self.mu_r, self.var_r = nn.Linear(hidden_dims * 2, z_dims), nn.Linear(hidden_dims * 2, z_dims)
        self.mu_n, self.var_n = nn.Linear(hidden_dims * 2, z_dims), nn.Linear(hidden_dims * 2, z_dims)
        self.mu_c, self.var_c = nn.Linear(hidden_dims * 2, z_dims), nn.Linear(hidden_dims * 2, z_dims)
        self.linear_init_global = nn.Linear(z_dims * 2 + 24, hidden_dims)
        self.grucell_g = nn.GRUCell(z_dims * 2 + 24 + roll_dims, hidden_dims)
        self.grucell_g_2 = nn.GRUCell(hidden_dims, hidden_dims)
        self.linear_init_r = nn.Linear(z_dims, hidden_dims)
        self.linear_init_n = nn.Linear(z_dims, hidden_dims)
        self.linear_init_c = nn.Linear(z_dims, hidden_dims)
        self.linear_out_r = nn.Linear(hidden_dims, rhythm_dims)
        self.linear_out_n = nn.Linear(hidden_dims, note_dims)
        self.linear_out_c = nn.Linear(z_dims, chroma_dims)
        self.linear_out_g = nn.Linear(hidden_dims, roll_dims)
        for name in ("mu_r_lookup", "mu_n_lookup"):               # gmm_model.py:151-165
            emb = nn.Embedding(n_component, z_dims)
            nn.init.xavier_uniform_(emb.weight)
            setattr(self, name, emb)
        for name in ("logvar_r_lookup", "logvar_n_lookup"):       # gmm_model.py:167-183 with pow_exp=-2, frozen
            emb = nn.Embedding(n_component, z_dims)
            nn.init.constant_(emb.weight, float(np.log(np.exp(-2.0) ** 2)))
            emb.weight.requires_grad = False
            setattr(self, name, emb)
        self._engine = None
        self._engine_key = None
        self._weights_version = -1
        self._version = 0

    # ------------------------------------------------------------------------------------------
    # engine plumbing
    # ------------------------------------------------------------------------------------------
    def used_parameters(self):
        """(name, parameter) pairs that take part in forward and are trainable: exactly the parameters that
        receive a gradient in the reference."""
        return [(k, p) for k, p in self.named_parameters() if not k.startswith(UNUSED_PREFIXES) and k not in FROZEN]

    def engine(self):
        dev = self.mu_r.weight.device
        # _ops_override is a TEST hook (tests/fake_ops.py checks the host-side schedule without a GPU);
        # the product never sets it, and without it a CPU model raises here.
        if dev.type != "cuda" and getattr(self, "_ops_override", None) is None:
            raise RuntimeError("MusicAttrRegGMVAE runs on the MI355X HIP kernels only; call .cuda() first "
                               "(there is deliberately no CPU fallback)")
        key = (dev, tuple(p.data_ptr() for _, p in self.named_parameters()))
        if self._engine is None or self._engine_key != key:
            from .hipops import HipOps
            ops = self._ops_override if getattr(self, "_ops_override", None) is not None else HipOps(dev)
            self._engine = self._make_engine(ops, dev)
            self._engine_key = key
            self._weights_version = -1
        if self._weights_version != self._version:
            self._engine.refresh_weights()
            self._weights_version = self._version
        return self._engine

    def _make_engine(self, ops, dev):
        """the schedule object of this model family (subclasses with another topology return their own)"""
        return Engine(ops, self._engine_params(), self.hidden_dims, self.latent_dim, self.n_component, dev)

    def _engine_params(self):
        """name -> tensor table the engine works on (subclasses may add tensors that are not parameters)"""
        return {k: p.data for k, p in self.named_parameters()}

    def weights_changed(self):
        """Tell the engine that parameter values changed (optimizer.step(), load_state_dict): the transposed
        weight images are re-derived before the next forward."""
        self._version += 1

    def load_state_dict(self, *a, **k):
        res = super().load_state_dict(*a, **k)
        self.weights_changed()
        return res

    def _indices(self, x, V, ids_ndim=None):
        """Accept the reference's one-hot tensors (trainer_gmm.py:296-303) or integer ids -> int32 ids.

        Integer dtypes are ids.  A floating tensor is a one-hot tensor (last dim V) - unless the caller states the rank of an id
        tensor (`ids_ndim`): the data loaders yield token ids as float32 [B][T] (ptb_v2.py:459-470 pads into float arrays) which the
        reference casts with ``.cuda().long()`` (trainer_gmm.py:323,397); a float tensor of that rank is cast the same way."""
        if x.dtype in (torch.int32, torch.int64, torch.int16, torch.int8, torch.uint8):
            return x.to(torch.int32).contiguous()
        if ids_ndim is not None and x.dim() == ids_ndim:
            return x.long().to(torch.int32).contiguous()
        if x.shape[-1] != V:
            raise ValueError("expected a one-hot tensor with last dim %d, got %s" % (V, tuple(x.shape)))
        eng = self.engine()
        idx = torch.empty(x.shape[:-1], dtype=torch.int32, device=x.device)
        eng.ops.onehot_to_index(x.contiguous().float(), idx)
        return idx

    def _draw_eps(self, B, T, device):
        """eps exactly as the reference consumes the CPU generator: randn(B,Z) for z_r, randn(B,Z) for z_n
        (gmm_model.py:230,234-235) and, in train mode, T draws of torch.rand(1) in the decoder loop (:140)."""
        eps_r = torch.randn(B, self.latent_dim)
        eps_n = torch.randn(B, self.latent_dim)
        if self.training:
            for _ in range(T):
                torch.rand(1)
        return eps_r.to(device), eps_n.to(device)

    # ------------------------------------------------------------------------------------------
    # reference API
    # ------------------------------------------------------------------------------------------
    def _sampling(self, x):
        """one-hot of the row-wise argmax, first index on ties (gmm_model.py:73-80)."""
        idx = self._indices(x, x.shape[-1]).long()
        return torch.zeros_like(x).scatter_(1, idx.view(-1, 1), 1.0)

    @torch.no_grad()
    def encode(self, x):
        """gmm_model.py:82-98 -> (Normal(mu_r, sigma_r), Normal(mu_n, sigma_n)); no autograd (eval-side API)."""
        eng = self.engine()
        d = self._indices(x, self.roll_dims)
        pre = eng.encode(d, save=False)
        Z = self.latent_dim
        zero = torch.zeros(d.shape[0], Z, device=d.device)
        lat = eng.latent(pre, {"r": zero, "n": zero})
        return (Normal(pre["r"][:, :Z].clone(), lat["r"]["sigma"].clone()),
                Normal(pre["n"][:, :Z].clone(), lat["n"]["sigma"].clone()))

    @torch.no_grad()
    def approx_qy_x(self, z, mu_lookup, logvar_lookup, n_component):
        """gmm_model.py:194-218 -> (logLogit_qy_x, qy_x); mu/logvar lookups are nn.Embedding modules."""
        eng = self.engine()
        B, Z = z.shape
        pre = torch.cat([z.float(), torch.zeros_like(z, dtype=torch.float32)], dim=1).contiguous()   # mu = z, sigma = 1
        eps = torch.zeros(B, Z, device=z.device)
        K = n_component
        out = [torch.empty(B, Z, device=z.device) for _ in range(2)]
        ll, qy = torch.empty(B, K, device=z.device), torch.empty(B, K, device=z.device)
        y = torch.empty(B, dtype=torch.int32, device=z.device)
        terms = torch.empty(B, 4, device=z.device)
        eng.ops.latent_fwd(pre, eps, mu_lookup.weight.data[:K].contiguous(), logvar_lookup.weight.data[:K].contiguous(), None,
                           out[0], out[1], ll, qy, y, terms)
        return ll, qy

    @torch.no_grad()
    def global_decoder(self, z, steps):
        """gmm_model.py:119-149 -> (B, steps, 342) log-probabilities.

        eval mode: greedy argmax feedback (:147-148), the call of test_class.py:253 / notebook cell 15.
        train mode: teacher forced with ``self.sample`` (the x of the last train-mode forward, :141-142), one ``torch.rand(1)`` draw
        per step as the reference makes (:140).  No autograd through a direct call (training goes through forward())."""
        if not self.training:
            from .decode import greedy_decode
            logp, _ = greedy_decode(self, z, steps)
            return logp
        if self.sample is None:
            raise RuntimeError("global_decoder() in train mode reads self.sample (gmm_model.py:142): run forward() first or call model.eval()")
        eng = self.engine()
        d = self._indices(self.sample, self.roll_dims, ids_ndim=2)
        if d.shape[1] < steps or d.shape[0] != z.shape[0]:
            raise IndexError("teacher forcing needs self.sample of shape (%d, >=%d, ...), got %s" % (z.shape[0], steps, tuple(self.sample.shape)))
        for _ in range(steps):
            torch.rand(1)
        d = d[:, :steps].contiguous()
        dec = eng.global_decoder_tf(d, z.float().contiguous(), save=False)
        out = torch.empty(z.shape[0], steps, E_VOCAB, device=z.device)
        eng.ops.vocab_logsoftmax(dec["logits"], z.shape[0], steps, E_VOCAB, logp_bt=out)
        return out

    @torch.no_grad()
    def sub_decoders(self, rhythm, z_r, note, z_n):
        """gmm_model.py:100-117 -> (rhythm_out, note_out, 0, 0), log_softmax over the TIME axis; no autograd."""
        eng = self.engine()
        r = self._indices(rhythm, 3)
        n = self._indices(note, 16)
        B, Tr = r.shape
        d = torch.zeros(B, 1, dtype=torch.int32, device=r.device)
        c = torch.zeros(B, 24, device=r.device)
        dec = eng.decoders(d, r, n, c, z_r.float().contiguous(), z_n.float().contiguous())
        outs = []
        for e, Ce in (("r", 3), ("n", 16)):
            lp = torch.empty(B, Tr, Ce, device=r.device)
            eng.ops.time_logsoftmax(dec["sd"][e]["logits"], logp_bt=lp)
            outs.append(lp)
        return outs[0], outs[1], 0, 0

    def forward(self, x, rhythm, note, chroma, eps=None):
        """gmm_model.py:220-259.  Returns the reference's nested tuple.

        train mode, autograd on: outputs are connected to the parameters through one fused autograd node whose backward runs the
        HIP backward kernels.  eval mode (``model.eval()``: what the evaluators are in after their first ``shift``,
        test_class.py:238,250): ``out`` is the GREEDY decode - the decoder feeds back ``_sampling(out)`` (:146-148) - and no
        ``torch.rand(1)`` is drawn; forward only.  Under ``torch.no_grad()`` nothing is saved for a backward either."""
        if self.training:
            self.sample = x
        self.engine()
        d = self._indices(x, self.roll_dims)
        r = self._indices(rhythm, 3)
        n = self._indices(note, 16)
        c = chroma.float().contiguous()
        B, T = d.shape
        if eps is None:
            eps = self._draw_eps(B, T, d.device)
        eps_r, eps_n = (e.float().contiguous() for e in eps)
        if not self.training or not torch.is_grad_enabled():
            res = self._forward_only(d, r, n, c, eps_r, eps_n)
        else:
            names = [k for k, _ in self.used_parameters()]
            plist = [p for _, p in self.used_parameters()]
            res = _GMVAEFunction.apply(self, names, d, r, n, c, eps_r, eps_n, *plist)
        out, r_out, n_out, mu_r, sg_r, mu_n, sg_n, z_r, z_n, ll_r, ll_n, qy_r, qy_n, y_r, y_n = res
        dis_r, dis_n = Normal(mu_r, sg_r), Normal(mu_n, sg_n)
        return ((out, r_out, n_out, 0, 0), (dis_r, dis_n), (z_r, z_n), (ll_r, ll_n), (qy_r, qy_n), (y_r, y_n))

    @torch.no_grad()
    def _forward_only(self, d, r, n, c, eps_r, eps_n):
        """forward without a backward: eval mode (greedy global decoder) or train mode under no_grad (teacher forced)."""
        eng = self._engine
        ops = eng.ops
        B, T = d.shape
        Tr = r.shape[1]
        Z = eng.Z
        pre = eng.encode(d, save=False)
        lat = eng.latent(pre, {"r": eps_r, "n": eps_n})
        z_r, z_n = lat["r"]["z"], lat["n"]["z"]
        sd = eng.sub_decoders_fwd(r, n, z_r, z_n, save=False)
        r_out = torch.empty(B, Tr, 3, device=d.device)
        n_out = torch.empty(B, Tr, 16, device=d.device)
        ops.time_logsoftmax(sd["r"]["logits"], logp_bt=r_out)
        ops.time_logsoftmax(sd["n"]["logits"], logp_bt=n_out)
        zc = eng.pack_zc(z_r, z_n, c)
        if self.training:
            dec = eng.global_decoder_tf(d, zc, save=False)
            out = torch.empty(B, T, E_VOCAB, device=d.device)
            ops.vocab_logsoftmax(dec["logits"], B, T, E_VOCAB, logp_bt=out)
        else:
            from .decode import greedy_decode
            out, _ = greedy_decode(self, zc.clone(), T)
        return (out, r_out, n_out,
                pre["r"][:, :Z].clone(), lat["r"]["sigma"].clone(), pre["n"][:, :Z].clone(), lat["n"]["sigma"].clone(),
                z_r.clone(), z_n.clone(), lat["r"]["ll"].clone(), lat["n"]["ll"].clone(),
                lat["r"]["qy"].clone(), lat["n"]["qy"].clone(), lat["r"]["y"].long(), lat["n"]["y"].long())


class _GMVAEFunction(torch.autograd.Function):
    """One autograd node for the whole model: forward = Engine.forward + log-softmax heads,
    backward = Engine.backward.  Used by the drop-in ``model(...)`` call; the fused trainer step
    (trainer.py) bypasses autograd altogether."""

    @staticmethod
    def forward(ctx, model, names, d, r, n, c, eps_r, eps_n, *params):
        eng = model._engine
        ops = eng.ops
        S = eng.forward(d, r, n, c, eps_r, eps_n)
        B, T = d.shape
        Tr = r.shape[1]
        dec, lat, pre = S["dec"], S["lat"], S["pre"]
        Z = eng.Z
        out = torch.empty(B, T, E_VOCAB, device=d.device)
        ops.vocab_logsoftmax(dec["logits"], B, T, E_VOCAB, logp_bt=out)
        r_out = torch.empty(B, Tr, 3, device=d.device)
        n_out = torch.empty(B, Tr, 16, device=d.device)
        ops.time_logsoftmax(dec["sd"]["r"]["logits"], logp_bt=r_out)
        ops.time_logsoftmax(dec["sd"]["n"]["logits"], logp_bt=n_out)
        ctx.model, ctx.names = model, names
        ctx.save_for_backward(out, r_out, n_out)
        ctx.fw_id = id(S)
        outs = (out, r_out, n_out,
                pre["r"][:, :Z].clone(), lat["r"]["sigma"].clone(), pre["n"][:, :Z].clone(), lat["n"]["sigma"].clone(),
                lat["r"]["z"].clone(), lat["n"]["z"].clone(), lat["r"]["ll"].clone(), lat["n"]["ll"].clone(),
                lat["r"]["qy"].clone(), lat["n"]["qy"].clone(), lat["r"]["y"].long(), lat["n"]["y"].long())
        ctx.mark_non_differentiable(outs[13], outs[14])
        return outs

    @staticmethod
    def backward(ctx, g_out, g_rout, g_nout, g_mu_r, g_sg_r, g_mu_n, g_sg_n, g_z_r, g_z_n, g_ll_r, g_ll_n, g_qy_r, g_qy_n, *_):
        model = ctx.model
        eng = model._engine
        ops = eng.ops
        S = eng.saved
        if S is None or id(S) != ctx.fw_id:
            raise RuntimeError("backward() must follow the forward() that produced these outputs (the engine keeps one "
                               "set of saved activations)")
        out, r_out, n_out = ctx.saved_tensors
        B, T, _ = out.shape
        dev = out.device
        dec = S["dec"]

        def dense(g, ref):
            return torch.zeros_like(ref) if g is None else g.float().contiguous()

        ops.vocab_logsoftmax_bwd(out, dense(g_out, out), dec["logits"])
        dl_sd = {}
        for e, lp, g in (("r", r_out, g_rout), ("n", n_out, g_nout)):
            dl_sd[e] = eng.buf("sd_dlogits_" + e, dec["sd"][e]["logits"].shape)
            ops.time_logsoftmax_bwd(lp, dense(g, lp), dl_sd[e])
        Z = eng.Z

        def opt(g):
            return None if g is None else g.float().contiguous()

        lat_up = {
            "r": dict(g_z=dense(g_z_r, S["lat"]["r"]["z"]).clone(), g_mu=opt(g_mu_r), g_sigma=opt(g_sg_r), g_ll=opt(g_ll_r), g_qy=opt(g_qy_r)),
            "n": dict(g_z=dense(g_z_n, S["lat"]["n"]["z"]).clone(), g_mu=opt(g_mu_n), g_sigma=opt(g_sg_n), g_ll=opt(g_ll_n), g_qy=opt(g_qy_n)),
        }
        G = {k: torch.empty_like(p) for k, p in model.used_parameters()}
        eng.backward(G, dl_sd, lat_up, None)
        return (None,) * 8 + tuple(G[k] for k in ctx.names)
