"""Dataset side: our YamahaDataset / VGMIDIDataset / chroma sanitisation vs the reference's classes run on the same
synthetic arrays (tests/golden/data.npz, produced by tests/golden/make_golden_data.py from ptb_v2.py:344-489)."""
import numpy as np
import torch
from torch.utils.data import DataLoader

from fake_ops import FakeOps
from helpers import load_golden, make_model
from mfn_import import load_package


def _ragged(flat, lens):
    out, o = np.empty(len(lens), dtype=object), 0
    for i, L in enumerate(lens):
        out[i] = flat[o:o + L].tolist()
        o += L
    return out


def test_yamaha_dataset_and_sanitisation():
    load_package()
    from music_fader_nets_amd import datasets as D
    g = load_golden("data")
    san = D.sanitize_chroma(g["y_in_data"], g["y_in_rhythm"], g["y_in_note"], g["y_in_chroma"])
    for got, k in zip(san, ("data_lst", "rhythm_lst", "note_density_lst", "chroma_lst")):
        assert np.array_equal(got, g["y_san_" + k]), k
    assert len(san[0]) == len(g["y_in_data"]) - 2                       # the two songs with an empty chroma vector are dropped
    assert np.array_equal(g["y_in_chroma"][4], np.zeros(24))            # and the input arrays are left alone
    for mode in ("train", "val", "test"):
        ds = D.YamahaDataset(*san, mode=mode)
        assert len(ds) == int(g["y_%s_len" % mode])
        for i in range(len(ds)):
            for j, nm in enumerate(("x", "r", "n", "c", "rd", "nd")):
                assert np.array_equal(np.asarray(ds[i][j]), g["y_%s_%s" % (mode, nm)][i]), (mode, i, nm)
    b = next(iter(DataLoader(D.YamahaDataset(*san), batch_size=4)))
    assert [str(t.dtype) for t in b] == [str(s) for s in g["y_batch_dtypes"]]


def test_vgmidi_dataset():
    load_package()
    from music_fader_nets_amd import datasets as D
    g = load_golden("data")
    toks = _ragged(g["v_in_tokens"], g["v_in_lens"])
    rl, nl = _ragged(g["v_in_rhythm"], g["v_in_rlens"]), _ragged(g["v_in_note"], g["v_in_rlens"])
    for mode in ("train", "val", "test"):
        ds = D.VGMIDIDataset(toks, rl, nl, g["v_in_chroma"], g["v_in_arousal"].copy(), g["v_in_valence"], mode=mode)
        assert len(ds) == int(g["v_%s_len" % mode])
        for i in range(len(ds)):
            for j, nm in enumerate(("x", "r", "n", "c", "a", "v", "rd", "nd")):
                assert np.array_equal(np.asarray(ds[i][j]), g["v_%s_%s" % (mode, nm)][i]), (mode, i, nm)
        x0 = np.asarray(ds[0][0])
        L = int(g["v_in_lens"][0]) if mode == "train" else None
        if L is not None:                                                # EOS sits in front of the LAST token (np.insert(k, -1, 1))
            assert x0[L - 1] == 1 and x0[L] == toks[0][-1] and (x0[L + 1:] == 0).all()
    b = next(iter(DataLoader(D.VGMIDIDataset(toks, rl, nl, g["v_in_chroma"], g["v_in_arousal"].copy(), g["v_in_valence"]), batch_size=4)))
    assert [str(t.dtype) for t in b] == [str(s) for s in g["v_batch_dtypes"]]
    assert set(np.unique(b[4].numpy())) <= {0.0, 1.0}


def test_loader_batches_feed_the_trainer():
    """a DataLoader batch of either dataset goes straight into GMVAETrainer.train (float token tensors, float64 densities)."""
    pkg = load_package()
    from music_fader_nets_amd import datasets as D
    g = load_golden("data")
    san = D.sanitize_chroma(g["y_in_data"], g["y_in_rhythm"], g["y_in_note"], g["y_in_chroma"])
    m = make_model(64, 32, ops=FakeOps())
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
    d, r, n, c, rd, nd = next(iter(DataLoader(D.YamahaDataset(*san), batch_size=4)))
    step, tup = tr.train(0, None, None, None, d, r, n, c.float(), rd, nd)
    assert step == 1 and len(tup) == 8 and np.isfinite(tup).all()
    toks = _ragged(g["v_in_tokens"], g["v_in_lens"])
    rl, nl = _ragged(g["v_in_rhythm"], g["v_in_rlens"]), _ragged(g["v_in_note"], g["v_in_rlens"])
    ds = D.VGMIDIDataset(toks, rl, nl, g["v_in_chroma"], g["v_in_arousal"].copy(), g["v_in_valence"])
    d, r, n, c, a, v, rd, nd = next(iter(DataLoader(ds, batch_size=4)))
    assert d.is_floating_point()          # float32 ids, as the reference's dataset yields them: no manual cast (trainer_gmm.py:323 does .long())
    step, tup = tr.train(step, None, None, None, d, r, n, c.float(), rd, nd, is_supervised=True, y_label=a)
    assert step == 2 and np.isfinite(tup).all()
