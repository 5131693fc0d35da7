"""MusicAttrRegVAE - the vanilla-VAE sibling of the GM-VAE (reference ``model_v2.py:9-171``) on the same HIP kernels.

Same encoder / sub-decoder / global-decoder blocks as ``MusicAttrRegGMVAE`` (compare model_v2.py:81-143 with gmm_model.py:82-149),
without the Gaussian-mixture prior: ``forward`` returns ``((out, r_out, n_out), (dis_r, dis_n), (z_r, z_n))`` (model_v2.py:165-171).
The parameter containers are built in the reference's order (model_v2.py:26-60), so a seeded construction gives the reference's
initial weights and ``state_dict`` has the reference's keys.  The latent block reuses the mixture kernels with ONE zero-mean dummy
component that is neither a parameter nor part of ``state_dict``; its posterior terms are simply not used.
"""
import torch
from torch import nn
from torch.distributions import Normal

from .gmm_model import MusicAttrRegGMVAE, _GMVAEFunction


class MusicAttrRegVAE(MusicAttrRegGMVAE):
    def __init__(self, roll_dims, rhythm_dims, note_dims, chroma_dims, hidden_dims, z_dims, n_step, k=1000):
        nn.Module.__init__(self)
        if (roll_dims, rhythm_dims, note_dims, chroma_dims) != (342, 3, 16, 24):
            raise ValueError("the HIP path is built for roll/rhythm/note/chroma dims 342/3/16/24 (trainer.py:30-33)")
        # ---- parameter containers, in the construction order of model_v2.py:26-60 ---------------------------------
        self.gru_r = nn.GRU(roll_dims, hidden_dims, batch_first=True, bidirectional=True)
        self.gru_n = nn.GRU(roll_dims, hidden_dims, batch_first=True, bidirectional=True)
        self.gru_c = nn.GRU(roll_dims, hidden_dims, batch_first=True, bidirectional=True)
        self.gru_d_r = nn.GRU(z_dims + rhythm_dims, hidden_dims, batch_first=True)
        self.gru_d_n = nn.GRU(z_dims + note_dims, hidden_dims, batch_first=True)
        self.gru_d_c = nn.GRU(z_dims + chroma_dims, hidden_dims, batch_first=True)
        self.c_r = nn.Linear(z_dims, 3)
        self.c_n = nn.Linear(z_dims, 3)
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
        self.n_step = n_step
        self.roll_dims = roll_dims
        self.hidden_dims = hidden_dims
        self.eps = 100
        self.rhythm_dims = rhythm_dims
        self.sample = None
        self.iteration = 0
        self.z_dims = self.latent_dim = z_dims
        self.k = torch.FloatTensor([k])
        self.n_component = 1
        self._engine = None
        self._engine_key = None
        self._weights_version = None
        self._version = 0

    def _engine_params(self):
        p = {k: t.data for k, t in self.named_parameters()}
        dev = self.mu_r.weight.device
        for e in ("r", "n"):                     # the single dummy component of the reused mixture kernels
            p["mu_%s_lookup.weight" % e] = torch.zeros(1, self.latent_dim, device=dev)
            p["logvar_%s_lookup.weight" % e] = torch.zeros(1, self.latent_dim, device=dev)
        return p

    def encoder(self, x):
        """model_v2.py:81-97 (the sibling calls it ``encoder``)."""
        return self.encode(x)

    def approx_qy_x(self, *a, **k):
        raise AttributeError("MusicAttrRegVAE has no mixture posterior (model_v2.py:9-171)")

    def sub_decoders(self, rhythm, z_r, note, z_n):
        """model_v2.py:99-116 -> (rhythm_out, note_out)"""
        r_out, n_out, _, _ = MusicAttrRegGMVAE.sub_decoders(self, rhythm, z_r, note, z_n)
        return r_out, n_out

    def forward(self, x, rhythm, note, chroma, eps=None):
        """model_v2.py:145-171."""
        if self.training:
            self.sample = x
            self.iteration += 1
        self.engine()
        d = self._indices(x, self.roll_dims)
        r = self._indices(rhythm, 3)
        n = self._indices(note, 16)
        c = chroma.float().contiguous()
        B, T = d.shape
        if eps is None:
            eps = self._draw_eps(B, T, d.device)     # Normal(0,1).sample(size) consumes the generator like randn (model_v2.py:152-154)
        eps_r, eps_n = (e.float().contiguous() for e in eps)
        names = [k for k, _ in self.used_parameters()]
        plist = [p for _, p in self.used_parameters()]
        res = _GMVAEFunction.apply(self, names, d, r, n, c, eps_r, eps_n, *plist)
        out, r_out, n_out, mu_r, sg_r, mu_n, sg_n, z_r, z_n = res[:9]
        return ((out, r_out, n_out), (Normal(mu_r, sg_r), Normal(mu_n, sg_n)), (z_r, z_n))
