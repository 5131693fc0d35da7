"""HOST twins of the C ABI (include/fadernets_host.h, libfadernets_host.so = an AddressSanitizer build): the arithmetic of the hot-path entry
points driven through ctypes in a machine without a GPU - the `small` golden fixture (encoder scans + heads + latent block), BASELINE
configs[0] (`c0`: sub-decoder with its time-axis head, regulariser, teacher-forced decoder cells, greedy decode tokens, weight-gradient
products, token sort + segment sums) and torch autograd references for the backward entry points.  Runs in a subprocess because the ASAN runtime has to be preloaded."""
import os
import subprocess
import sys

import pytest

from mfn_import import ROOT


def test_host_twins_under_asan():
    lib = os.path.join(ROOT, "music-fader-nets_amd", "libfadernets_host.so")
    if not os.path.exists(lib):
        pytest.skip("libfadernets_host.so missing (build() builds it best-effort: no sanitizer toolchain on this box?)")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")      # the interpreter itself is not leak-clean
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_twins_driver.py")], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "HOST TWINS OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
    assert "AddressSanitizer" not in p.stderr, p.stderr[-3000:]


def test_every_twin_mirrors_a_declared_entry_point():
    import re
    hdr = open(os.path.join(ROOT, "include", "fadernets.h")).read()
    twin = open(os.path.join(ROOT, "include", "fadernets_host.h")).read()
    declared = set(re.findall(r"\b(fn_[a-z0-9_]+)\s*\(", hdr))
    twins = set(re.findall(r"\b(fn_[a-z0-9_]+)_host\s*\(", twin))
    assert twins and twins <= declared, twins - declared
    assert {"fn_gru_seq_fwd", "fn_gru_seq_bwd", "fn_latent_fwd", "fn_latent_bwd", "fn_out_head_f32", "fn_clip_adam", "fn_gemm_f32", "fn_gru_dwhh_f32",
            "fn_token_sort", "fn_embed_grad_sorted", "fn_time_logsoftmax", "fn_time_logsoftmax_bwd", "fn_pairwise_reg", "fn_gru_cell_f32",
            "fn_decode_greedy"} <= twins
    # ... and is exported by the twin library with that name
    import ctypes
    lib_path = os.path.join(ROOT, "music-fader-nets_amd", "libfadernets_host.so")
    if os.path.exists(lib_path):
        src = open(os.path.join(ROOT, "music-fader-nets_amd", "csrc", "host", "fadernets_host.cpp")).read()
        for t in twins:
            assert re.search(r"\b%s_host\s*\(" % t, src), t
