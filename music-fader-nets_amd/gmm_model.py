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
        self._weights_version = None
        self._version = 0
        self.arith = None                # arithmetic of the deep MFMA products (arith.py): None = the package default, "f32" or "bf16x6"

    # ------------------------------------------------------------------------------------------
    # engine plumbing
    # ------------------------------------------------------------------------------------------
    def used_parameters(self):
        """(name, parameter) pairs that take part in forward and are trainable: exactly the parameters that
        receive a gradient in the reference."""
        return [(k, p) for k, p in self.named_parameters() if not k.startswith(UNUSED_PREFIXES) and k not in FROZEN]

    def _device(self):
        return self.mu_r.weight.device

    def _make_ops(self, dev):
        """the kernel backend of this model: the HIP library on an MI355X, nothing else (there is deliberately no CPU fallback)"""
        if dev.type != "cuda":
            raise RuntimeError("%s runs on the MI355X HIP kernels only; call .cuda() first (there is deliberately no CPU fallback)" % type(self).__name__)
        from .hipops import HipOps
        return HipOps(dev)

    def set_arith(self, name):
        """choose the arithmetic of the deep MFMA products (arith.py: "f32", "bf16x6" or None = package default); takes effect with the
        next ``engine()`` call: kernel table switched, weight images (bf16 triple images of the recurrent matrices) re-derived.  Captured
        training steps are keyed by it (trainer.py)."""
        from . import arith
        self.arith = None if name is None else arith.check(name)
        return self

    def _param_versions(self):
        """torch's in-place counters of the parameters: an ``optimizer.step()`` / ``copy_`` of an unmodified reference loop moves them, the
        fused trainer (which writes through the flat buffer and refreshes the images itself) does not"""
        return sum(p._version for p in self.parameters())

    def engine(self):
        dev = self._device()
        key = (dev, tuple(p.data_ptr() for _, p in self.named_parameters()))
        if self._engine is None or self._engine_key != key:
            self._engine = self._make_engine(self._make_ops(dev), dev)
            self._engine_key = key
            self._weights_version = None
        from . import arith
        want = arith.resolve(getattr(self, "arith", None)) == arith.BF16X6
        if hasattr(self._engine.ops, "dw_x6") and bool(self._engine.ops.dw_x6) != want:
            self._engine.ops.dw_x6 = want           # the engine's kernel table follows the model's choice, whichever engine is current
            self._weights_version = None
        # the engine keeps transposed / fragment-order images of the recurrent weights: re-derive them when the values changed -
        # announced (weights_changed(), load_state_dict) or detected (in-place updates by a torch optimiser)
        ver = (self._version, self._param_versions())
        if self._weights_version != ver:
            self._engine.refresh_weights()
            self._weights_version = ver
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

    # Direct calls of the sub-modules (the reference allows them anywhere, gmm_model.py:82-218): in train mode with autograd enabled each is ONE
    # autograd node (classes _EncodeFunction ... below) whose backward runs the matching part of the HIP backward; in eval mode / under
    # no_grad they are forward only.  The activations a node saves live in their own buffer namespace, so a direct call never disturbs
    # the saved state of a full forward() that still waits for its backward().
    def _autograd_on(self):
        return self.training and torch.is_grad_enabled() and type(self) is MusicAttrRegGMVAE

    def _named(self, prefixes):
        kp = [(k, p) for k, p in self.used_parameters() if k.startswith(prefixes)]
        return [k for k, _ in kp], [p for _, p in kp]

    def encode(self, x):
        """gmm_model.py:82-98 -> (Normal(mu_r, sigma_r), Normal(mu_n, sigma_n))."""
        if self._autograd_on():
            self.engine()
            names, plist = self._named(_EncodeFunction.PREFIXES)
            mu_r, sg_r, mu_n, sg_n = _EncodeFunction.apply(self, names, self._indices(x, self.roll_dims), *plist)
            return Normal(mu_r, sg_r), Normal(mu_n, sg_n)
        return self._encode_forward_only(x)

    @torch.no_grad()
    def _encode_forward_only(self, x):
        eng = self.engine()
        d = self._indices(x, self.roll_dims)
        pre = eng.encode(d, save=False)
        Z = self.latent_dim
        zero = torch.zeros(d.shape[0], Z, device=d.device)
        lat = eng.latent(pre, {"r": zero, "n": zero})
        return (Normal(pre["r"][:, :Z].clone(), lat["r"]["sigma"].clone()),
                Normal(pre["n"][:, :Z].clone(), lat["n"]["sigma"].clone()))

    def approx_qy_x(self, z, mu_lookup, logvar_lookup, n_component):
        """gmm_model.py:194-218 -> (logLogit_qy_x, qy_x); mu/logvar lookups are nn.Embedding modules."""
        if self._autograd_on() and (z.requires_grad or mu_lookup.weight.requires_grad):
            self.engine()
            return _QyFunction.apply(self, n_component, z.float().contiguous(), mu_lookup.weight, logvar_lookup.weight)
        return self._approx_qy_x_forward_only(z, mu_lookup, logvar_lookup, n_component)

    @torch.no_grad()
    def _approx_qy_x_forward_only(self, z, mu_lookup, logvar_lookup, n_component):
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

    def global_decoder(self, z, steps):
        """gmm_model.py:119-149 -> (B, steps, 342) log-probabilities.

        eval mode: greedy argmax feedback (:147-148), the call of test_class.py:253 / notebook cell 15.
        train mode: teacher forced with ``self.sample`` (the x of the last train-mode forward, :141-142), one ``torch.rand(1)`` draw
        per step as the reference makes (:140); with autograd enabled the result is connected to z and the decoder's parameters."""
        if self._autograd_on():
            if self.sample is None:
                raise RuntimeError("global_decoder() in train mode reads self.sample (gmm_model.py:142): run forward() first or call model.eval()")
            self.engine()
            d = self._indices(self.sample, self.roll_dims, ids_ndim=2)
            if d.shape[1] < steps or d.shape[0] != z.shape[0]:
                raise IndexError("teacher forcing needs self.sample of shape (%d, >=%d, ...), got %s" % (z.shape[0], steps, tuple(self.sample.shape)))
            for _ in range(steps):
                torch.rand(1)
            names, plist = self._named(_DecoderFunction.PREFIXES)
            return _DecoderFunction.apply(self, names, d[:, :steps].contiguous(), z.float().contiguous(), *plist)
        return self._global_decoder_forward_only(z, steps)

    @torch.no_grad()
    def _global_decoder_forward_only(self, z, steps):
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

    def sub_decoders(self, rhythm, z_r, note, z_n):
        """gmm_model.py:100-117 -> (rhythm_out, note_out, 0, 0), log_softmax over the TIME axis."""
        if self._autograd_on():
            self.engine()
            names, plist = self._named(_SubDecFunction.PREFIXES)
            r_out, n_out = _SubDecFunction.apply(self, names, self._indices(rhythm, 3), self._indices(note, 16), z_r.float().contiguous(),
                                                 z_n.float().contiguous(), *plist)
            return r_out, n_out, 0, 0
        return self._sub_decoders_forward_only(rhythm, z_r, note, z_n)

    @torch.no_grad()
    def _sub_decoders_forward_only(self, rhythm, z_r, note, z_n):
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


# ------------------------------------------------------------------------------------------------------------------------------
# autograd nodes of the DIRECT sub-module calls (encode / sub_decoders / global_decoder / approx_qy_x in train mode)
# ------------------------------------------------------------------------------------------------------------------------------
class _Namespace:
    """run engine work inside a private buffer namespace"""

    def __init__(self, eng, ns):
        self.eng, self.ns = eng, ns

    def __enter__(self):
        self.prev = self.eng.buf_ns
        self.eng.buf_ns = self.ns
        gen = self.eng.__dict__.setdefault("_direct_gen", {})
        return gen

    def __exit__(self, *a):
        self.eng.buf_ns = self.prev


def _check_generation(eng, ns, mine, what):
    if eng.__dict__.get("_direct_gen", {}).get(ns) != mine:
        raise RuntimeError("backward() of a direct %s() call must run before the next %s() call (the engine keeps one set of saved "
                           "activations per sub-module)" % (what, what))


def _dense_or_zero(g, ref):
    return torch.zeros_like(ref) if g is None else g.float().contiguous()


class _EncodeFunction(torch.autograd.Function):
    """model.encode(x) with a backward: encoder scans + mu / var heads (gmm_model.py:82-98)"""
    PREFIXES = ("gru_r.", "gru_n.", "mu_r.", "var_r.", "mu_n.", "var_n.")
    NS = "direct_enc/"

    @staticmethod
    def forward(ctx, model, names, d, *params):
        from .engine import ops_sort
        eng = model._engine
        Z = eng.Z
        with _Namespace(eng, _EncodeFunction.NS) as gen:
            gen[_EncodeFunction.NS] = ctx.gen = gen.get(_EncodeFunction.NS, 0) + 1
            sort = ops_sort(eng, "d", d, E_VOCAB)
            pre = eng.encode(d, save=True)
            zero = eng.buf("zero_eps", (d.shape[0], Z), zero_init=True)
            lat = eng.latent(pre, {"r": zero, "n": zero})
        ctx.S = dict(d=d, pre=pre, lat=lat, eps={"r": zero, "n": zero}, labels=None, sort={"d": sort})
        ctx.model, ctx.names = model, names
        return (pre["r"][:, :Z].clone(), lat["r"]["sigma"].clone(), pre["n"][:, :Z].clone(), lat["n"]["sigma"].clone())

    @staticmethod
    def backward(ctx, g_mu_r, g_sg_r, g_mu_n, g_sg_n):
        model = ctx.model
        eng = model._engine
        _check_generation(eng, _EncodeFunction.NS, ctx.gen, "encode")
        S = ctx.S
        B, Z = S["d"].shape[0], eng.Z
        G = {k: torch.zeros_like(p) for k, p in model.used_parameters() if k in ctx.names}
        with _Namespace(eng, _EncodeFunction.NS):
            lat_up = {}
            for e, gm, gs in (("r", g_mu_r, g_sg_r), ("n", g_mu_n, g_sg_n)):
                lat_up[e] = dict(g_z=eng.zbuf("g_z_" + e, (B, Z)), g_mu=None if gm is None else gm.float().contiguous(),
                                 g_sigma=None if gs is None else gs.float().contiguous())
            eng.backward_encoder(G, S, lat_up, None)
        return (None, None, None) + tuple(G[k] for k in ctx.names)


class _DecoderFunction(torch.autograd.Function):
    """model.global_decoder(z, steps) in train mode (teacher forced, gmm_model.py:119-149) with a backward wrt z and the decoder"""
    PREFIXES = ("linear_out_g.", "grucell_g_2.", "grucell_g.", "linear_init_global.")
    NS = "direct_dec/"

    @staticmethod
    def forward(ctx, model, names, d, zc, *params):
        from .engine import ops_sort
        eng = model._engine
        B, T = d.shape
        with _Namespace(eng, _DecoderFunction.NS) as gen:
            gen[_DecoderFunction.NS] = ctx.gen = gen.get(_DecoderFunction.NS, 0) + 1
            sort = ops_sort(eng, "d", d, E_VOCAB)
            zcs = eng.buf("zc_in", tuple(zc.shape))
            zcs.copy_(zc)
            dec = eng.global_decoder_tf(d, zcs, save=True)
            out = torch.empty(B, T, E_VOCAB, device=d.device)
            eng.ops.vocab_logsoftmax(dec["logits"], B, T, E_VOCAB, logp_bt=out)
        ctx.S = dict(d=d, dec=dec, sort={"d": sort})
        ctx.model, ctx.names = model, names
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g_out):
        model = ctx.model
        eng = model._engine
        ops, P = eng.ops, eng.p
        _check_generation(eng, _DecoderFunction.NS, ctx.gen, "global_decoder")
        (out,) = ctx.saved_tensors
        S = ctx.S
        G = {k: torch.zeros_like(p) for k, p in model.used_parameters() if k in ctx.names}
        with _Namespace(eng, _DecoderFunction.NS):
            ops.vocab_logsoftmax_bwd(out, _dense_or_zero(g_out, out), S["dec"]["logits"])
            gd = eng._bwd_global_decoder_scans(S)
            eng._bwd_global_decoder_params(G, S, gd)
            dzc = torch.empty_like(S["dec"]["zc"])
            ops.gemm_multi([dict(C=dzc, segs=[(gd["drb_g"], P["grucell_g.weight_ih"][:, E_VOCAB:]), (gd["dh0_g"], P["linear_init_global.weight"])])],
                           a_k=True, b_k=False)
            eng.main_wait_side()
        return (None, None, None, dzc) + tuple(G[k] for k in ctx.names)


class _SubDecFunction(torch.autograd.Function):
    """model.sub_decoders(rhythm, z_r, note, z_n) (gmm_model.py:100-117, TIME-axis log_softmax) with a backward wrt z_r, z_n and both decoders"""
    PREFIXES = ("gru_d_r.", "gru_d_n.", "linear_out_r.", "linear_out_n.", "linear_init_r.", "linear_init_n.")
    NS = "direct_sd/"

    @staticmethod
    def forward(ctx, model, names, r, n, z_r, z_n, *params):
        from .engine import ops_sort
        eng = model._engine
        B, Tr = r.shape
        with _Namespace(eng, _SubDecFunction.NS) as gen:
            gen[_SubDecFunction.NS] = ctx.gen = gen.get(_SubDecFunction.NS, 0) + 1
            sorts = {"r": ops_sort(eng, "r", r, 3), "n": ops_sort(eng, "n", n, 16)}
            zs = {}
            for e, z in (("r", z_r), ("n", z_n)):
                zs[e] = eng.buf("z_in_" + e, tuple(z.shape))
                zs[e].copy_(z)
            sd = eng.sub_decoders_fwd(r, n, zs["r"], zs["n"], save=True)
            outs = []
            for e, Ce in (("r", 3), ("n", 16)):
                lp = torch.empty(B, Tr, Ce, device=r.device)
                eng.ops.time_logsoftmax(sd[e]["logits"], logp_bt=lp)
                outs.append(lp)
        ctx.S = dict(sd=sd, sorts=sorts, z=zs, B=B, Tr=Tr)
        ctx.model, ctx.names = model, names
        ctx.save_for_backward(*outs)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_r, g_n):
        model = ctx.model
        eng = model._engine
        ops, P = eng.ops, eng.p
        _check_generation(eng, _SubDecFunction.NS, ctx.gen, "sub_decoders")
        r_out, n_out = ctx.saved_tensors
        S = ctx.S
        B, Tr = S["B"], S["Tr"]
        G = {k: torch.zeros_like(p) for k, p in model.used_parameters() if k in ctx.names}
        with _Namespace(eng, _SubDecFunction.NS):
            dl_sd = {}
            for e, lp, g in (("r", r_out, g_r), ("n", n_out, g_n)):
                dl_sd[e] = eng.buf("sd_dlogits_" + e, S["sd"][e]["logits"].shape)
                ops.time_logsoftmax_bwd(lp, _dense_or_zero(g, lp), dl_sd[e])
            sdb = eng._bwd_sub_decoder_scans(S["sd"], dl_sd, B, Tr)
            eng._bwd_sub_decoder_params(G, S["sd"], sdb, dl_sd, S["sorts"], S["z"], B, Tr)
            eng.flush_colsums()
            dz = {}
            for e, Ce in (("r", 3), ("n", 16)):
                dz[e] = torch.empty_like(S["z"][e])
                ops.gemm_multi([dict(C=dz[e], segs=[(sdb[e]["drb"], P["gru_d_%s.weight_ih_l0" % e][:, Ce:]), (sdb[e]["dh0"], P["linear_init_%s.weight" % e])])],
                               a_k=True, b_k=False)
        return (None, None, None, None, dz["r"], dz["n"]) + tuple(G[k] for k in ctx.names)


class _QyFunction(torch.autograd.Function):
    """model.approx_qy_x(z, mu_lookup, logvar_lookup, K) (gmm_model.py:194-218) with a backward wrt z and the component means"""

    @staticmethod
    def forward(ctx, model, K, z, mu_w, lv_w):
        eng = model._engine
        B, Z = z.shape
        pre = torch.cat([z, torch.zeros_like(z)], dim=1).contiguous()        # mu = z, sigma = exp(0) = 1, eps = 0  ->  the sample IS z
        eps = torch.zeros(B, Z, device=z.device)
        mu, lv = mu_w.data[:K].contiguous(), lv_w.data[:K].contiguous()
        sig, zz = torch.empty(B, Z, device=z.device), torch.empty(B, Z, device=z.device)
        ll, qy = torch.empty(B, K, device=z.device), torch.empty(B, K, device=z.device)
        y, terms = torch.empty(B, dtype=torch.int32, device=z.device), torch.empty(B, 4, device=z.device)
        eng.ops.latent_fwd(pre, eps, mu, lv, None, sig, zz, ll, qy, y, terms)
        ctx.model, ctx.K = model, K
        ctx.save_for_backward(pre, eps, mu, lv, zz, qy, mu_w)
        return ll, qy

    @staticmethod
    def backward(ctx, g_ll, g_qy):
        eng = ctx.model._engine
        pre, eps, mu, lv, zz, qy, mu_w = ctx.saved_tensors
        B, Z = eps.shape
        K = ctx.K
        dpre = torch.empty(B, 2 * Z, device=pre.device)
        dmu_rows = torch.empty(B, K * Z, device=pre.device)
        gz = torch.zeros(B, Z, device=pre.device)
        eng.ops.latent_bwd(pre, eps, mu, lv, None, zz, qy, gz, None, None, None if g_ll is None else g_ll.float().contiguous(),
                           None if g_qy is None else g_qy.float().contiguous(), None, dpre, dmu_rows)
        dmu = torch.zeros_like(mu_w)
        eng.ops.colsum(dmu_rows, dmu[:K].view(-1))
        return None, None, dpre[:, :Z].contiguous(), dmu, None
