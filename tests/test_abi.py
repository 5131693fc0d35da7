"""The C-ABI library loads and exports every symbol include/fadernets.h declares (no compute without a GPU)."""
import os
import re

from mfn_import import ROOT, load_package


def test_header_symbols_exported():
    load_package()
    from music_fader_nets_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "fadernets.h")).read()
    declared = set(re.findall(r"\b(fn_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fn_version() == 6
    assert lib.fn_strerror(-2) == b"unsupported or inconsistent sizes"


def test_argument_errors_without_gpu():
    """Argument validation happens before any launch, so it is testable on the CPU box."""
    load_package()
    from music_fader_nets_amd import _lib
    lib = _lib.load()
    assert lib.fn_gemm_f32(1, 1, 4, 4, 4, 1.0, None, 4, None, 4, 0.0, None, 4, None, 1, None, 0, None) == -1
    assert lib.fn_gru_seq_fwd(None, 1, None) == -1
    assert lib.fn_gemm_ws_bytes(128, 64, 4) == 4 * 128 * 64 * 4
    arr = (_lib.FnGruFwd * 9)()
    assert lib.fn_gru_seq_fwd(arr, 9, None) == -5
    # counters + sticky error word of the weight-stationary launches: whole 128-byte lines, error word in the last one
    nbytes = lib.fn_gru_sync_ws_bytes()
    assert nbytes % 128 == 0 and nbytes >= 64 * 128 + 128
    # fn_comm_* / fn_colsum_multi validate before they touch RCCL or the device
    assert lib.fn_comm_unique_id(None) == -1
    assert lib.fn_comm_all_reduce_f32(None, None, 4, None) == -1
    assert lib.fn_comm_destroy(None) == 0
    import ctypes as C
    h = C.c_void_p()
    assert lib.fn_comm_init(C.byref(h), 2, 2, C.create_string_buffer(_lib.FN_COMM_ID_BYTES)) == -2      # rank outside the world
    assert lib.fn_colsum_multi(None, 1, None) == -1
    assert lib.fn_colsum_multi((_lib.FnColsumJob * 1)(), 65, None) == -5
    assert lib.fn_occupy_cus(0, 1024, 10, None) == -2
    assert lib.fn_strerror(-7).startswith(b"librccl")


def test_build_entry_checks_the_current_abi_version():
    """__graft_entry__.build() must compare the library with _lib.ABI_VERSION, not with a literal (round 5: it still asked for 4 after the ABI went to 5)"""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")).read()
    assert "lib.fn_version() == _lib.ABI_VERSION" in src and not re.search(r"fn_version\(\) == \d", src)
