#!/usr/bin/env python3
"""Golden fixture for the BATCHED fader sweep: the reference's own eval-mode ``global_decoder`` on a (64, 280) batch at hidden 512.

Runs ONLY in the build container (imports /root/reference with the ``.cuda()`` no-op shim); writes ``sweep.npz`` next to itself.

What the reference does per sample (test_class.py:233-254, called 8 times per sample from the loop at :84-113 with the 8 values
``min_val + k * (max_val - min_val) / 8`` of :84-85): encode -> re-sample z from the posterior -> overwrite ``z_r[:, 0]`` with the target value ->
``model.eval()`` -> ``global_decoder(cat[z_r, z_n, c], steps=100)`` on ONE row.  The build decodes all rows at once; this fixture is those 64
single-row problems (8 samples x 8 values) stacked into one batch and pushed through the reference's decoder in ONE call: the reference's
decoder is row-independent (GRUCell / Linear / log_softmax(dim=1) / max(1) per row, gmm_model.py:119-149 and 73-80), which the script checks
against row-by-row calls for the first 3 rows.

Stored: the (64, 280) z batch, greedy tokens (64, 100), top-2 log-probability gaps (64, 100) and the log-probabilities of the first step.
Weights: seeded (torch.manual_seed(1234), the construction order of gmm_model.py:33-71), hidden 512, z 128, K = 2 - nothing but checksums stored.

Usage:  python tests/golden/make_golden_sweep.py
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.path.insert(0, REF)
import gmm_model as ref_gmm  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden.make_golden import synth_batch  # noqa: E402  (the seeded synthetic batch generator of the other fixtures)

H, Z, K, NS, NV, T, TR, STEPS = 512, 128, 2, 8, 8, 64, 16, 100


def main():
    torch.manual_seed(1234)
    model = ref_gmm.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=H, z_dims=Z, n_step=32, n_component=K)
    rng = np.random.RandomState(11)
    d_np, _, _, c_np, _, _, _ = synth_batch(rng, NS, T, TR)
    d = torch.from_numpy(d_np).long()
    c = torch.from_numpy(c_np).float()
    d_oh = torch.zeros(NS, T, 342).scatter_(-1, d.unsqueeze(-1), 1.0)
    min_val, max_val = -2.0, 2.0
    gap = (max_val - min_val) / 8
    values = np.array([min_val + k * gap for k in range(NV)])          # test_class.py:84-85
    torch.manual_seed(77)
    with torch.no_grad():
        model.train()
        dis_r, dis_n = model.encode(d_oh)
        rows = []
        for s in range(NS):
            for v in values:                                            # shift() re-samples z for every call (test_class.py:243-244)
                z_r = dis_r.mean[s:s + 1] + dis_r.stddev[s:s + 1] * torch.randn(1, Z)
                z_n = dis_n.mean[s:s + 1] + dis_n.stddev[s:s + 1] * torch.randn(1, Z)
                z_r[:, 0] = float(v)                                    # test_class.py:249
                rows.append(torch.cat([z_r, z_n, c[s:s + 1]], dim=1))
        z = torch.cat(rows, 0)                                          # (64, 280)
        model.eval()
        dec = model.global_decoder(z, steps=STEPS)                      # (64, 100, 342) log-probabilities
        for i in range(3):                                              # row independence of the reference's decoder
            one = model.global_decoder(z[i:i + 1], steps=STEPS)
            assert torch.equal(one.argmax(-1), dec[i:i + 1].argmax(-1)), i
            assert float((one - dec[i:i + 1]).abs().max()) < 1e-4
    tok = dec.argmax(-1)
    top2 = dec.topk(2, dim=-1).values
    out = dict(dims=np.array([H, Z, K, NS, NV, STEPS]), values=values, z=z.numpy(), tokens=tok.numpy().astype(np.int32),
               gap=(top2[..., 0] - top2[..., 1]).numpy(), logp_first=dec[:, 0, :].numpy())
    for k_, v_ in model.state_dict().items():
        vd = v_.double()
        out["w0sum/" + k_] = np.array([vd.sum().item(), vd.abs().sum().item(), (vd * vd).sum().item()])
    path = os.path.join(HERE, "sweep.npz")
    np.savez_compressed(path, **out)
    print("sweep ->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))
    print("  distinct tokens per row (first 8):", [len(set(t.tolist())) for t in tok[:8]])
    print("  min top-2 gap over all rows / steps: %.3e; rows whose gap drops below 1e-4 somewhere: %d" % (float(out["gap"].min()), int((out["gap"].min(1) < 1e-4).sum())))


if __name__ == "__main__":
    main()
