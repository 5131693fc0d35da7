"""Dataset side of the GM-VAE trainer: the reference's pre-processed ``.npy`` arrays -> the batches ``train`` / ``evaluate`` consume.

Semantics of ``ptb_v2.py:344-489`` (MIDI tokenisation itself stays out of scope; the released arrays are already tokenised):

* Yamaha (unsupervised): dense arrays ``data [N][T]``, ``rhythm [N][Tr]``, ``note_density [N][Tr]``, ``chroma [N][24]``; chroma is
  *sanitised* on load (keep the three largest entries of every vector, drop the songs whose vector is empty, ptb_v2.py:350-362); split
  80 / 10 / 10 in file order (:411); items ``(x, r, n, c, r_density, n_density)`` with ``r_density`` = share of 1s in the rhythm
  sequence and ``n_density`` = mean of the note sequence (:421-422).
* VGMIDI (supervised): ragged object arrays of token / rhythm / note sequences + ``chroma``, ``arousal``, ``valence``; split 90 / 5 / 5
  (:447); an EOS token 1 is inserted BEFORE the last token of every song (``np.insert(k, -1, 1)``, :459 - kept as is), sequences are
  right-padded with 0 to the longest of the split (:460,469-470), densities are taken before padding (:463-464), arousal is binarised
  in place to {0, 1} at ``>= 0`` (:472-473); items ``(x, r, n, c, a, v, r_density, n_density)``.

Both are ``torch.utils.data.Dataset``s, so the reference's ``DataLoader(ds, batch_size, shuffle, num_workers=0)`` lines keep working; token
tensors come out as float32 (as in the reference) and are converted to int32 by ``GMVAETrainer.prepare_batch``.
"""
import os
from collections import Counter

import numpy as np
import torch
from torch.utils.data import Dataset


def sanitize_chroma(data, rhythm, note, chroma):
    """ptb_v2.py:350-362: zero everything below the third-largest entry of each chroma vector, drop songs left with none."""
    chroma = np.array(chroma, copy=True)
    empty = []
    for i in range(len(chroma)):
        c = chroma[i]
        third = -np.sort(-c)[2]
        c[c < third] = 0
        if np.count_nonzero(c) == 0:
            empty.append(i)
    return tuple(np.delete(a, empty, axis=0) for a in (data, rhythm, note, chroma))


def load_yamaha_arrays(root="data/values_v3"):
    """ptb_v2.py:344-364 (the branch that reads the saved arrays)."""
    arrs = [np.load(os.path.join(root, f)) for f in ("data.npy", "rhythm.npy", "note_density.npy", "chroma.npy")]
    return sanitize_chroma(*arrs)


def load_vgmidi_arrays(root="data/filtered_songs_disambiguate"):
    """ptb_v2.py:371-397 -> (data, rhythm, note, arousal, valence, chroma); chroma_lst.npy must exist (building it needs MIDI tools)."""
    ld = lambda f, pickle=False: np.load(os.path.join(root, f), allow_pickle=pickle)
    return (ld("song_tokens.npy", True), ld("rhythm_lst.npy", True), ld("note_lst.npy", True), ld("arousal_lst.npy"),
            ld("valence_lst.npy"), ld("chroma_lst.npy"))


def _split(arrays, mode, fractions):
    n = len(arrays[0])
    a, b = int(fractions[0] * n), int(fractions[1] * n)
    sl = {"train": slice(0, a), "val": slice(a, b), "test": slice(b, None)}[mode]
    return [x[sl] for x in arrays]


def _densities(rhythm, note):
    r_density = [Counter(list(k))[1] / len(k) for k in rhythm]
    n_density = np.array([sum(k) / len(k) for k in note])
    return r_density, n_density


def _pad(seqs):
    return torch.nn.utils.rnn.pad_sequence([torch.Tensor(np.asarray(k, dtype=np.float32)) for k in seqs], batch_first=True)


class YamahaDataset(Dataset):
    """ptb_v2.py:400-436."""

    def __init__(self, data, rhythm, note, chroma, mode="train"):
        super().__init__()
        self.data, self.rhythm, self.note, self.chroma = _split([data, rhythm, note, chroma], mode, (0.8, 0.9))
        self.r_density, self.n_density = _densities(self.rhythm, self.note)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx], self.rhythm[idx], self.note[idx], self.chroma[idx], self.r_density[idx], self.n_density[idx]


class VGMIDIDataset(Dataset):
    """ptb_v2.py:439-489."""

    def __init__(self, data, rhythm, note, chroma, arousal, valence, mode="train"):
        super().__init__()
        data, rhythm, note, self.chroma, self.arousal, self.valence = _split([data, rhythm, note, chroma, arousal, valence], mode, (0.9, 0.95))
        self.data = _pad([np.insert(np.asarray(k), -1, 1) for k in data])       # EOS goes in front of the last token (:459)
        self.r_density, self.n_density = _densities(rhythm, note)              # before padding (:463-464)
        self.rhythm, self.note = _pad(rhythm), _pad(note)
        self.arousal[self.arousal >= 0] = 1                                     # in place, like the reference (:472-473)
        self.arousal[self.arousal < 0] = 0

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return (self.data[idx], self.rhythm[idx], self.note[idx], self.chroma[idx], self.arousal[idx], self.valence[idx],
                self.r_density[idx], self.n_density[idx])
