#!/usr/bin/env python3
"""Golden fixture for the dataset side: the reference's ``YamahaDataset`` / ``VGMIDIDataset`` classes
(ptb_v2.py:400-489, AST-extracted - the module itself needs MIDI libraries) and its chroma sanitisation (:350-362, the statements of the
load branch, executed as they stand) on small synthetic arrays in the released ``.npy`` layout.  Output: data.npz.

Runs ONLY in the build container.   python tests/golden/make_golden_data.py
"""
import ast
import os
from collections import Counter

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(os.path.join(REF, "ptb_v2.py")).read()
    tree = ast.parse(src)
    ns = dict(np=np, torch=torch, Dataset=Dataset, Counter=Counter)
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("YamahaDataset", "VGMIDIDataset")]
    exec(compile(ast.Module(body=body, type_ignores=[]), "ptb_v2.py[extract]", "exec"), ns)
    # the sanitisation statements of get_classic_piano's load branch (ptb_v2.py:350-362)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_classic_piano"][0]
    loops = [n for n in ast.walk(fn) if isinstance(n, ast.For) and "third_largest" in ast.unparse(n)]
    assert len(loops) == 1
    san_src = "idx = []\n" + ast.unparse(loops[0]).replace("tqdm(range(len(chroma_lst)))", "range(len(chroma_lst))") + "\n" + \
              "\n".join("%s = np.delete(%s, idx, axis=0)" % (v, v) for v in ("data_lst", "rhythm_lst", "note_density_lst", "chroma_lst"))

    rng = np.random.RandomState(11)
    out = {}
    # ---- Yamaha-style dense arrays ---------------------------------------------------------------------------------
    N, T, TR = 23, 20, 8
    data = rng.randint(0, 342, size=(N, T)).astype(np.int64)
    rhythm = rng.randint(0, 3, size=(N, TR)).astype(np.int64)
    note = rng.randint(0, 14, size=(N, TR)).astype(np.int64)
    chroma = rng.uniform(0, 1, size=(N, 24)) * (rng.uniform(size=(N, 24)) < 0.35)
    chroma[4] = 0.0
    chroma[17] = 0.0                                        # songs the sanitisation has to drop
    for k, v in (("data", data), ("rhythm", rhythm), ("note", note), ("chroma", chroma)):
        out["y_in_" + k] = v.copy()
    env = dict(np=np, data_lst=data.copy(), rhythm_lst=rhythm.copy(), note_density_lst=note.copy(), chroma_lst=chroma.copy())
    exec(san_src, env)
    for k in ("data_lst", "rhythm_lst", "note_density_lst", "chroma_lst"):
        out["y_san_" + k] = env[k]
    for mode in ("train", "val", "test"):
        ds = ns["YamahaDataset"](env["data_lst"], env["rhythm_lst"], env["note_density_lst"], env["chroma_lst"], mode=mode)
        items = [ds[i] for i in range(len(ds))]
        out["y_%s_len" % mode] = np.array(len(ds))
        for j, nm in enumerate(("x", "r", "n", "c", "rd", "nd")):
            out["y_%s_%s" % (mode, nm)] = np.array([np.asarray(it[j]) for it in items])
    b = next(iter(DataLoader(ns["YamahaDataset"](env["data_lst"], env["rhythm_lst"], env["note_density_lst"], env["chroma_lst"]), batch_size=4)))
    out["y_batch_dtypes"] = np.array([str(t.dtype) for t in b])
    # ---- VGMIDI-style ragged arrays ---------------------------------------------------------------------------------
    M = 41
    lens = rng.randint(5, 19, size=M)
    toks = np.empty(M, dtype=object)
    rl = np.empty(M, dtype=object)
    nl = np.empty(M, dtype=object)
    for i, L in enumerate(lens):
        toks[i] = rng.randint(2, 342, size=L).tolist()
        k = int(rng.randint(3, 9))
        rl[i] = rng.randint(0, 3, size=k).tolist()
        nl[i] = rng.randint(0, 14, size=k).tolist()
    vchroma = np.eye(24)[rng.randint(0, 24, size=M)]
    arousal = rng.uniform(-1, 1, size=M)
    arousal[3] = 0.0                                         # the ">= 0" edge
    valence = rng.uniform(-1, 1, size=M)
    out["v_in_lens"] = lens
    out["v_in_tokens"] = np.concatenate([np.asarray(t) for t in toks])
    out["v_in_rlens"] = np.array([len(x) for x in rl])
    out["v_in_rhythm"] = np.concatenate([np.asarray(t) for t in rl])
    out["v_in_note"] = np.concatenate([np.asarray(t) for t in nl])
    out["v_in_chroma"], out["v_in_arousal"], out["v_in_valence"] = vchroma, arousal.copy(), valence.copy()
    for mode in ("train", "val", "test"):
        ds = ns["VGMIDIDataset"](toks, rl, nl, vchroma, arousal.copy(), valence, mode=mode)
        items = [ds[i] for i in range(len(ds))]
        out["v_%s_len" % mode] = np.array(len(ds))
        for j, nm in enumerate(("x", "r", "n", "c", "a", "v", "rd", "nd")):
            out["v_%s_%s" % (mode, nm)] = np.array([np.asarray(it[j]) for it in items])
    b = next(iter(DataLoader(ns["VGMIDIDataset"](toks, rl, nl, vchroma, arousal.copy(), valence), batch_size=4)))
    out["v_batch_dtypes"] = np.array([str(t.dtype) for t in b])
    path = os.path.join(HERE, "data.npz")
    np.savez_compressed(path, **out)
    print("data ->", path, "%.1f KB" % (os.path.getsize(path) / 1e3))
    print({k: v.shape for k, v in out.items() if k.startswith("v_train") or k.startswith("y_train")}, out["y_batch_dtypes"], out["v_batch_dtypes"])


if __name__ == "__main__":
    main()
