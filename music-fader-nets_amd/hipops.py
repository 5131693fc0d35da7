"""Tensor-level wrappers over the C ABI (include/fadernets.h).

PyTorch is only the owner of device memory and of the current HIP stream here: each method checks
dtype / device / layout, takes ``data_ptr()`` and calls the HIP kernel on
``torch.cuda.current_stream()``.  No method has a CPU or eager-torch fallback; a missing library or
a non-zero return code raises.
"""
import ctypes as C

import torch

from . import _lib, arith


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s): the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))


def _mat(t, name):
    """2-D view with unit inner stride -> (ptr, rows, cols, ld)."""
    _chk(t, name=name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError("%s must be 2-D with contiguous rows, got shape %s strides %s" % (name, tuple(t.shape), t.stride()))
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    return _p(t), t.shape[0], t.shape[1], ld


def _dense(t, dtype=torch.float32, name="tensor"):
    _chk(t, dtype, name)
    if t is not None and not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return _p(t)


class HipOps:
    """The product backend: every method is one C-ABI call on the current stream."""

    name = "hip"

    def __init__(self, device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipOps needs a GPU device")
        self._ws = {}
        self._retired = []
        self.cell_variant = 0     # FnGruCell.variant (tuning / tests)
        self.dw_x6 = arith.default() == arith.BF16X6        # arithmetic of the deep products (the package default; a model sets its own choice: arith.py): True = exact bf16 triple splits on the bf16 MFMA (FN_GEMM_BF16X6, FnGruFwd.variant bit 14), False = fp32 MFMA
        self.x6_wide = True       # with dw_x6: 128 x 256 output tiles (FN_GEMM_X6_WIDE) where the product has >= 256 columns - with TWICE the K ranges the caller asked for (the same number of workgroups); False: 128 x 128 tiles
        self.nt_x6 = True         # with dw_x6: the big Linear-forward / dX products (whole 128 x 128 tiles, K % 32 == 0) on the bf16 x 6 kernel too; False: fp32 MFMA (A/B measurements)
        self.x6_per_tile = False  # A/B, tests: the producer / consumer bf16 x 6 kernels as one workgroup per output tile / (tile, K range) item (round 6: one per CU walking its items)
        self.cell_x6_rows = 2048  # ... from this many rows on (tests: 0 = wherever the kernel takes the shape)
        self.cell_x6 = True       # with dw_x6: the large-batch decode cells (fn_gru_cell_f32) on the bf16 x 6 producer / consumer kernel (gru_cell_x6_kernel); False: fp32 MFMA cells
        self.x6_perwave = False   # with dw_x6: the round-5 weight-gradient kernel in which every wavefront splits its own operands (FN_GEMM_X6_PERWAVE; A/B measurements, tests)
        self.bwd_x6 = True        # with dw_x6: the backward scans on the bf16 x 6 kernel too (False: fp32 MFMA backward scans; A/B measurements, tests)
        self.variant = 0          # FnGruFwd.variant of every scan launch (tuning / tests only; results do not depend on it)
        self.lane = ""            # scratch namespace: kernels enqueued on different streams must not share workspaces

    # -- plumbing -------------------------------------------------------------------------------
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, nbytes, tag="ws"):
        """Grow-only scratch buffer per tag.  A captured hipGraph holds the raw pointer it was captured with, so a buffer that is
        outgrown is RETIRED (kept alive for the lifetime of this object), never freed: graphs of smaller batch shapes keep replaying
        on their own scratch."""
        tag = self.lane + tag
        cur = self._ws.get(tag)
        if cur is None or cur.numel() * 4 < nbytes:
            if cur is not None:
                self._retired.append(cur)
            cur = torch.empty((max(nbytes, 4) + 3) // 4, dtype=torch.float32, device=self.device)
            self._ws[tag] = cur
        return cur

    # -- dense ----------------------------------------------------------------------------------
    def gemm(self, A, B, Cm, a_k=True, b_k=True, alpha=1.0, beta=0.0, bias=None, splitk=1, lean=False, nt_x6=True):
        pa, ar, ac, lda = _mat(A, "A")
        pb, br, bc, ldb = _mat(B, "B")
        pc, M, N, ldc = _mat(Cm, "C")
        K = ac if a_k else ar
        if (ar if a_k else ac) != M or (br if b_k else bc) != N or (bc if b_k else br) != K:
            raise RuntimeError("gemm shape mismatch A%s B%s C%s a_k=%s b_k=%s" % (tuple(A.shape), tuple(B.shape), tuple(Cm.shape), a_k, b_k))
        _chk(bias, name="bias")
        ws, wsb = None, 0
        x6 = _lib.GEMM_BF16X6 if (self.dw_x6 and not a_k and not b_k and K >= 1024) else 0
        if x6:
            splitk, x6 = self._x6_mode(splitk, N, K, x6)
        elif self.dw_x6 and self.nt_x6 and nt_x6 and a_k and b_k and splitk <= 1 and M % 128 == 0 and N % 128 == 0 and K % 32 == 0 and K >= 128 and (M // 128) * (N // 128) >= 128:
            x6 = _lib.GEMM_BF16X6                  # Linear forward / dX of the decoder pipeline on the bf16 MFMA (gemm_nt_x6w_kernel; the lean instance is then moot)
            if self.x6_per_tile:                   # (A/B, tests: one workgroup per tile instead of one per CU walking its tiles)
                x6 |= _lib.GEMM_X6_PERTILE
        if splitk > 1:
            wsb = self.lib.fn_gemm_ws_bytes(M, N, splitk & 0xffff)
            ws = self.workspace(wsb, "gemm")
        _lib.check(self.lib.fn_gemm_f32(int(a_k), int(b_k), M, N, K, alpha, pa, lda, pb, ldb, beta, pc, ldc, _p(bias),
                                        splitk | (_lib.GEMM_LEAN if lean else 0) | x6,
                                        _p(ws), wsb, self.stream()), "fn_gemm_f32")

    def _x6_mode(self, splitk, N, K, x6):
        """kernel choice of a bf16 x 6 weight-gradient product: (K ranges, flags).  Wide (128 x 256) tiles halve the number of output tiles: twice the K
        ranges keep the number of workgroups (whole 32-k blocks per range permitting).  The per-wave kernel (tests, A/B) runs on the same K ranges."""
        if self.x6_wide == "force":                       # tests: the wide kernel on whatever shape and split, K ranges as given
            if not self.x6_perwave:
                x6 |= _lib.GEMM_X6_WIDE
        elif self.x6_wide and N >= 256 and splitk > 1 and K >= 32768:      # (K = 16384, the attribute decoders: 162 us on 128 x 128 tiles / 16 ranges, 172 on wide / 32)
            splitk *= 2
            if not self.x6_perwave:
                x6 |= _lib.GEMM_X6_WIDE
        if self.x6_perwave:
            x6 |= _lib.GEMM_X6_PERWAVE
        if self.x6_per_tile:
            x6 |= _lib.GEMM_X6_PERTILE
        return splitk, x6

    def gemm_multi(self, jobs, a_k=True, b_k=True):
        """jobs: dicts(C=2-D view, segs=[(A, B), ...] (<= 4 products summed), beta=0.0, bias=None) - independent GEMMs of one
        (a_k, b_k) form in ONE launch (fn_gemm_multi); more than 12 jobs are split into several launches"""
        for i0 in range(0, len(jobs), 12):
            part = jobs[i0:i0 + 12]
            arr = (_lib.FnGemmJob * len(part))()
            for d, j in zip(arr, part):
                pc, M, N, ldc = _mat(j["C"], "C")
                _chk(j.get("bias"), name="bias")
                d.M, d.N, d.C, d.ldc, d.bias, d.beta, d.n_seg = M, N, pc, ldc, _p(j.get("bias")), float(j.get("beta", 0.0)), len(j["segs"])
                if not 1 <= len(j["segs"]) <= 4:
                    raise RuntimeError("gemm_multi: 1..4 products per job")
                for sg, (A, B) in zip(d.seg, j["segs"]):
                    pa, ar, ac, lda = _mat(A, "A")
                    pb, br, bc, ldb = _mat(B, "B")
                    K = ac if a_k else ar
                    if (ar if a_k else ac) != M or (br if b_k else bc) != N or (bc if b_k else br) != K:
                        raise RuntimeError("gemm_multi shape mismatch A%s B%s C%s" % (tuple(A.shape), tuple(B.shape), (M, N)))
                    sg.A, sg.lda, sg.B, sg.ldb, sg.K = pa, lda, pb, ldb, K
            _lib.check(self.lib.fn_gemm_multi(int(a_k), int(b_k), arr, len(part), self.stream()), "fn_gemm_multi")

    def transpose(self, src, dst):
        ps, R, Cc, sld = _mat(src, "src")
        pd, dr, dc, dld = _mat(dst, "dst")
        if (dr, dc) != (Cc, R):
            raise RuntimeError("transpose shape mismatch %s -> %s" % (tuple(src.shape), tuple(dst.shape)))
        _lib.check(self.lib.fn_transpose_f32(ps, R, Cc, sld, pd, dld, self.stream()), "fn_transpose_f32")

    def colsum(self, X, out, beta=0.0):
        px, M, N, ld = _mat(X, "X")
        _dense(out, name="out")
        if out.numel() != N:
            raise RuntimeError("colsum: out has %d elements, expected %d" % (out.numel(), N))
        wsb = self.lib.fn_colsum_ws_bytes(M, N)
        ws = self.workspace(wsb, "colsum")
        _lib.check(self.lib.fn_colsum_f32(px, M, N, ld, beta, _p(out), _p(ws), wsb, self.stream()), "fn_colsum_f32")

    def colsum_multi(self, jobs):
        """jobs: (X 2-D view, out 1-D dense[, beta]) - all column sums in ONE launch per 64 jobs (fn_colsum_multi); meant for the
        bias gradients of a step (<= a few thousand rows each)"""
        for i0 in range(0, len(jobs), _lib.FN_COLSUM_MAX_JOBS):
            part = jobs[i0:i0 + _lib.FN_COLSUM_MAX_JOBS]
            arr = (_lib.FnColsumJob * len(part))()
            for d, j in zip(arr, part):
                X, out = j[0], j[1]
                px, M, N, ld = _mat(X, "X")
                _dense(out, name="out")
                if out.numel() != N:
                    raise RuntimeError("colsum_multi: out has %d elements, expected %d" % (out.numel(), N))
                d.X, d.M, d.N, d.ld, d.beta, d.out = px, M, N, ld, float(j[2]) if len(j) > 2 else 0.0, _p(out)
            _lib.check(self.lib.fn_colsum_multi(arr, len(part), self.stream()), "fn_colsum_multi")

    def occupy_cus(self, blocks, lds_bytes, cycles):
        """diagnostic (fn_occupy_cus): `blocks` workgroups holding `lds_bytes` of LDS each spin for ~`cycles` ticks on the current stream"""
        _lib.check(self.lib.fn_occupy_cus(int(blocks), int(lds_bytes), int(cycles), self.stream()), "fn_occupy_cus")

    def axpy(self, alpha, x, y):
        _dense(x, name="x"), _dense(y, name="y")
        if x.numel() != y.numel():
            raise RuntimeError("axpy size mismatch")
        _lib.check(self.lib.fn_axpy_f32(x.numel(), alpha, _p(x), _p(y), self.stream()), "fn_axpy_f32")

    def sum(self, x, out, scale=1.0):
        _dense(x, name="x"), _chk(out, name="out")
        _lib.check(self.lib.fn_sum_f32(_p(x), x.numel(), scale, _p(out), self.stream()), "fn_sum_f32")

    # -- GRU scans ------------------------------------------------------------------------------
    def gates_floats(self, B, H):
        """floats per time step of the (opaque, blocked) saved-gates buffer."""
        return int(self.lib.fn_gru_gates_floats(B, H))

    def frag_floats(self, rows, K):
        return int(self.lib.fn_frag_floats(rows, K))

    def frag_pack(self, src, dst):
        """row-major 2-D view -> fragment-major operand image (dst: 1-D buffer of frag_floats(rows, K))."""
        ps, rows, K, ld = _mat(src, "src")
        _dense(dst, name="dst")
        if dst.numel() < self.frag_floats(rows, K):
            raise RuntimeError("frag_pack: dst too small")
        _lib.check(self.lib.fn_frag_pack(ps, rows, K, ld, _p(dst), self.stream()), "fn_frag_pack")

    def weight_images(self, jobs):
        """jobs: (kind, src 2-D view, dst) with kind "transpose" (dst [C][R] contiguous), "frag" (fragment-major image of src), "frag_t"
        (fragment-major image of src^T), "frag3" / "frag3_t" (the same two as bf16 triple images), "copy" (dst [R][C] dense: an aligned image of a
        column slice) - all in ONE launch (fn_weight_images)"""
        kinds = {"transpose": 0, "frag": 1, "frag_t": 2, "frag3": 3, "frag3_t": 4, "copy": 5}
        for i0 in range(0, len(jobs), 56):
            part = jobs[i0:i0 + 56]
            arr = (_lib.FnWeightImage * len(part))()
            for d, (kind, src, dst) in zip(arr, part):
                ps, R, Cc, ld = _mat(src, "src")
                _dense(dst, name="dst")
                need = {"transpose": lambda: R * Cc, "copy": lambda: R * Cc, "frag": lambda: self.frag_floats(R, Cc), "frag_t": lambda: self.frag_floats(Cc, R),
                        "frag3": lambda: self.frag_floats(R, Cc) * 3 // 2, "frag3_t": lambda: self.frag_floats(Cc, R) * 3 // 2}[kind]()
                if dst.numel() < need:
                    raise RuntimeError("weight_images: dst too small for %s of %s" % (kind, tuple(src.shape)))
                d.src, d.dst, d.rows, d.cols, d.ld, d.kind = ps, _p(dst), R, Cc, ld, kinds[kind]
            _lib.check(self.lib.fn_weight_images(arr, len(part), self.stream()), "fn_weight_images")

    def _frag_ws(self, tag, i, n):
        return self.workspace(4 * n, "%s%d" % (tag, i))[:n]

    SYNC_REGIONS = 64

    def begin_capture(self):
        """call right before a hipGraph capture starts: the captured launches take counter regions 0, 1, ... of a pool whose
        zero-fill is the first node of the graph (see _sync_region)"""
        for ent in self.__dict__.get("_sync_pools", {}).values():
            ent["next"] = 0

    def _sync_region(self):
        """(counters of this launch, sticky error word, flag bits).  Arrival counters must be zero at launch.
        Eager launches: one private region per lane, zero-filled by the library in front of every launch (a memset node each).
        Captured launches (the training step replays ~50 weight-stationary launches): every lane owns a POOL of regions, each launch
        takes the next one, and ONE fill node zero-fills the whole pool at the start of the graph and whenever the rotation wraps -
        every replay therefore finds its regions zero, whatever ran in between."""
        words = int(self.lib.fn_gru_sync_ws_bytes()) // 4
        syncs = self.__dict__.setdefault("_syncs", {})
        if self.lane + "err" not in syncs:
            syncs[self.lane + "err"] = torch.zeros(32, dtype=torch.int32, device=self.device)
        err = syncs[self.lane + "err"]
        if not torch.cuda.is_current_stream_capturing():
            eager = self.__dict__.setdefault("_sync_eager", {})
            if self.lane not in eager:
                eager[self.lane] = torch.zeros(words, dtype=torch.int32, device=self.device)
            pools = self.__dict__.setdefault("_sync_pools", {})
            if self.lane not in pools:           # allocated here, outside of any capture
                pools[self.lane] = dict(pool=torch.zeros(self.SYNC_REGIONS, words, dtype=torch.int32, device=self.device), next=0)
            for ent in pools.values():
                ent["next"] = 0                  # whatever is captured next starts with the pool fill
            return eager[self.lane], err, 0
        pools = self.__dict__.setdefault("_sync_pools", {})
        ent = pools.get(self.lane)
        if ent is None:
            raise RuntimeError("weight-stationary launch captured before its lane ran once eagerly (the counter pool is allocated eagerly)")
        if ent["next"] == 0:
            ent["pool"].zero_()
        region = ent["pool"][ent["next"]]
        ent["next"] = (ent["next"] + 1) % self.SYNC_REGIONS
        return region, err, 0x200

    def gru_sync_error(self, clear=False):
        """True when a weight-stationary launch (scan or single-launch decode) gave up waiting: ONE D2H copy of all the sticky
        error words (host sync).  clear=True also resets them so that the caller can retry on the per-step kernels."""
        syncs = list(self.__dict__.get("_syncs", {}).values())
        if not syncs:
            return False
        words = torch.stack([t[-32] for t in syncs]).tolist()
        bad = any(int(w) != 0 for w in words)
        if bad and clear:
            for t in syncs:
                t[-32:].zero_()
        return bad

    def _fwd_descriptors(self, scans, cu_budget=0):
        arr = (_lib.FnGruFwd * len(scans))()
        for i, (d, s) in enumerate(zip(arr, scans)):
            for k in ("w_hh_frag", "b_hh", "b_ih", "h0", "gx_dense", "gx_table", "gx_rowbias", "h_all", "gates", "h0_frag", "h_last_frag"):
                _dense(s.get(k), name=k)
            for k in ("h0_frag", "h_last_frag"):       # (the size the launch's arithmetic needs is checked in gru_seq_fwd, once x6 is decided)
                if s.get(k) is not None and s[k].numel() * s[k].element_size() < self.frag_floats(s["B"], s["H"]) * 4:
                    raise RuntimeError("%s needs frag_floats(B, H) floats" % k)
            d.cu_budget = int(cu_budget)
            _dense(s.get("idx"), torch.int32, "idx")
            d.B, d.T, d.H, d.reverse = s["B"], s["T"], s["H"], int(s.get("reverse", 0))
            d.w_hh_frag, d.b_hh, d.b_ih, d.h0 = _p(s["w_hh_frag"]), _p(s["b_hh"]), _p(s.get("b_ih")), _p(s.get("h0"))
            d.gx_dense, d.gx_table, d.idx = _p(s.get("gx_dense")), _p(s.get("gx_table")), _p(s.get("idx"))
            d.idx_ld = s["idx"].shape[1] if s.get("idx") is not None else 0
            d.idx_shift, d.start_token = int(s.get("idx_shift", 0)), int(s.get("start_token", 0))
            d.gx_rowbias, d.h_all, d.gates = _p(s.get("gx_rowbias")), _p(s["h_all"]), _p(s.get("gates"))
            d.h0_frag, d.h_last_frag = _p(s.get("h0_frag")), _p(s.get("h_last_frag"))
        return arr

    def gru_fwd_x6_ok(self, scans, cu_budget=0):
        """would gru_seq_fwd run this launch on the bf16 x 6 kernels (fn_gru_fwd_x6_ok)?  The caller of a CHAIN of launches (time chunks that hand
        their state over as operand images) asks for every launch first and runs the whole chain on one arithmetic."""
        if not (self.dw_x6 and all(s.get("w_hh_frag3") is not None for s in scans)):
            return False
        return bool(self.lib.fn_gru_fwd_x6_ok(self._fwd_descriptors(scans, cu_budget), len(scans)))

    def gru_seq_fwd(self, scans, persistent=True, cu_budget=0, variant=None, x6=None):
        """x6: None = this launch decides for itself (bf16 x 6 where eligible, refused when it is part of a hand-over chain: h0_frag /
        h_last_frag images depend on the arithmetic), True / False = the decision the caller made for the whole chain"""
        arr = self._fwd_descriptors(scans, cu_budget)
        variant = self.variant if variant is None else variant
        sync = self._sync_region() if persistent else None
        if x6 is None:
            x6 = persistent and self.gru_fwd_x6_ok(scans, cu_budget)
            if x6 is False and self.dw_x6 and any(s.get("h0_frag") is not None or s.get("h_last_frag") is not None for s in scans) \
                    and all(s.get("w_hh_frag3") is not None for s in scans):
                raise RuntimeError("gru_seq_fwd: a launch of a hand-over chain must be told the chain's arithmetic (x6=True / False)")
        if x6 and not persistent:
            raise RuntimeError("gru_seq_fwd: the bf16 x 6 scans are weight-stationary launches")
        for i, (d, s) in enumerate(zip(arr, scans)):
            if x6:                               # hand-over images of a bf16 x 6 launch are triples: 6 bytes per value (an fp32-sized buffer would be overrun)
                for k in ("h0_frag", "h_last_frag"):
                    if s.get(k) is not None and s[k].numel() * s[k].element_size() < self.frag_floats(s["B"], s["H"]) * 6:
                        raise RuntimeError("%s of a bf16 x 6 launch needs 3/2 * frag_floats(B, H) floats (bf16 triples)" % k)
            d.frag_ws = _p(self._frag_ws("fragf", i, 3 * self.frag_floats(s["B"], s["H"])))      # 2 slabs of fp32 fragments, or 2 of bf16 triples (x6)
            d.sync_ws, d.err_ws, d.variant = (_p(sync[0]), _p(sync[1]), int(variant) | sync[2]) if persistent else (None, None, int(variant))
            if x6:
                d.w_hh_frag, d.variant = _p(s["w_hh_frag3"]), d.variant | 0x4000
        _lib.check(self.lib.fn_gru_seq_fwd(arr, len(scans), self.stream()), "fn_gru_seq_fwd (bf16 x 6)" if x6 else "fn_gru_seq_fwd")

    def frag3_pack(self, src, dst):
        """bf16 triple image of src [rows][K] (fn_frag3_pack); dst: 3/2 * frag_floats(rows, K) floats"""
        ps, R, K, ld = _mat(src, "src")
        _dense(dst, name="dst")
        if dst.numel() * dst.element_size() < self.frag_floats(R, K) * 6:
            raise RuntimeError("frag3_pack: dst too small")
        _lib.check(self.lib.fn_frag3_pack(ps, R, K, ld, _p(dst), self.stream()), "fn_frag3_pack")

    def gru_cell(self, h_prev, w_hh, b_hh, h_out, x=None, w_ih=None, b_ih=None, gx_table=None, idx=None, start_token=0, gx_rowbias=None, variant=None,
                 idx_best=None, best_v=0):
        """one GRUCell step of a large batch (fn_gru_cell_f32): h_out [B][H] from h_prev [B][H], optional dense input x [B][K1] with the
        torch matrix w_ih [3H][K1], optional token rows gx_table [V][3H] picked by idx (a [B] int32 column view, e.g. tokens[:, i - 1];
        None = start_token; or idx_best: the [B] int64 packed argmax words out_argmax left for the previous token, packed with best_v
        columns) and per-row constants gx_rowbias [B][3H]; w_hh [3H][H] is the torch matrix itself"""
        c = _lib.FnGruCell()
        ph, B, H, ldh = _mat(h_prev, "h_prev")
        pw, r3, Hw, ldw = _mat(w_hh, "w_hh")
        po, Bo, Ho, ldo = _mat(h_out, "h_out")
        if r3 != 3 * H or Hw != H or (Bo, Ho) != (B, H):
            raise RuntimeError("gru_cell shape mismatch h_prev%s w_hh%s h_out%s" % (tuple(h_prev.shape), tuple(w_hh.shape), tuple(h_out.shape)))
        _dense(b_hh, name="b_hh"), _dense(b_ih, name="b_ih"), _dense(gx_table, name="gx_table"), _dense(gx_rowbias, name="gx_rowbias")
        c.B, c.H, c.h_prev, c.ldh, c.w_hh, c.ldw_hh, c.b_hh, c.b_ih, c.h_out, c.ldo = B, H, ph, ldh, pw, ldw, _p(b_hh), _p(b_ih), po, ldo
        if x is not None:
            px, Bx, K1, ldx = _mat(x, "x")
            pwi, r3i, K1w, ldwi = _mat(w_ih, "w_ih")
            if Bx != B or r3i != 3 * H or K1w != K1:
                raise RuntimeError("gru_cell: x%s / w_ih%s do not match" % (tuple(x.shape), tuple(w_ih.shape)))
            c.x, c.ldx, c.K1, c.w_ih, c.ldw_ih = px, ldx, K1, pwi, ldwi
        c.gx_table, c.gx_rowbias, c.start_token = _p(gx_table), _p(gx_rowbias), int(start_token)
        c.variant = int(self.cell_variant if variant is None else variant)
        if self.dw_x6 and self.cell_x6 and variant is None and B >= self.cell_x6_rows:
            # FnGruCell.variant bit 14: the cell on the bf16 MFMA with exact triple splits where the shape allows it (B % 128 == 0, H % 32 == 0, K1 % 32 == 0), else
            # the fp32 cells.  Measured (scratch/r6_bench_decode_cells.py, us per token of the tokens-only decode): 2048 rows 90.0 against 95.7, 1536 rows 87.2
            # against 76.6, 1024 rows 87.2 against 59.5 - one 128-row workgroup per CU takes ~28 / 48 us for the 16 / 32 blocks of the two cells whatever the row
            # count, so it only pays where the fp32 cells need every CU: from 2048 rows on
            c.variant |= 0x4000
        if idx is not None:
            if idx.dtype != torch.int32 or idx.dim() != 1 or idx.shape[0] != B:
                raise RuntimeError("gru_cell: idx must be a [B] int32 column")
            c.idx, c.idx_ld = idx.data_ptr(), idx.stride(0)
        if idx_best is not None:
            if idx_best.dtype != torch.int64 or idx_best.dim() != 1 or idx_best.shape[0] != B or not idx_best.is_contiguous():
                raise RuntimeError("gru_cell: idx_best must be a contiguous [B] int64 row")
            c.idx_best, c.best_v = idx_best.data_ptr(), int(best_v)
        _lib.check(self.lib.fn_gru_cell_f32(C.byref(c), self.stream()), "fn_gru_cell_f32")

    def out_argmax(self, h, W, bias, best):
        """fn_out_argmax_f32: best[b] = max(best[b], packed (logit, column) words of h[b] W^T + bias) - best: a ZEROED contiguous [B] int64 row"""
        ph, B, K, ldh = _mat(h, "h")
        pw, V, Kw, ldw = _mat(W, "W")
        _dense(bias, name="bias")
        if Kw != K or bias.numel() != V or best.dtype != torch.int64 or best.dim() != 1 or best.shape[0] != B or not best.is_contiguous():
            raise RuntimeError("out_argmax shape mismatch h%s W%s best%s" % (tuple(h.shape), tuple(W.shape), tuple(best.shape)))
        _lib.check(self.lib.fn_out_argmax_f32(ph, ldh, pw, ldw, _p(bias), B, V, K, best.data_ptr(), self.stream()), "fn_out_argmax_f32")

    def best_tokens(self, best, V, tokens):
        """fn_best_tokens: best [steps][B] int64 packed words -> tokens [B][steps] int32 (row stride tokens.stride(0))"""
        if best.dtype != torch.int64 or best.dim() != 2 or not best.is_contiguous() or tokens.dtype != torch.int32 or tokens.dim() != 2:
            raise RuntimeError("best_tokens: best [steps][B] int64, tokens [B][>= steps] int32")
        steps, B = best.shape
        if tokens.shape[0] != B or tokens.shape[1] < steps or tokens.stride(1) != 1:
            raise RuntimeError("best_tokens: tokens%s does not take %d x %d words" % (tuple(tokens.shape), steps, B))
        _lib.check(self.lib.fn_best_tokens(best.data_ptr(), steps, B, int(V), tokens.data_ptr(), tokens.stride(0), self.stream()), "fn_best_tokens")

    def _bwd_descriptors(self, scans, cu_budget=0):
        arr = (_lib.FnGruBwd * len(scans))()
        for d, s in zip(arr, scans):
            for k in ("w_hh_t_frag", "h0", "h_all", "gates", "dh_last", "dh_ext", "dgx_all", "dghn_all", "dh0", "dgx_rowsum", "dghn_rowsum", "scratch"):
                _dense(s.get(k), name=k)
            d.cu_budget = int(cu_budget)
            d.B, d.T, d.H = s["B"], s["T"], s["H"]
            d.w_hh_t_frag, d.h0, d.h_all, d.gates = _p(s["w_hh_t_frag"]), _p(s.get("h0")), _p(s["h_all"]), _p(s["gates"])
            d.dh_last, d.dh_ext = _p(s.get("dh_last")), _p(s.get("dh_ext"))
            d.dgx_all, d.dghn_all, d.dh0 = _p(s["dgx_all"]), _p(s["dghn_all"]), _p(s.get("dh0"))
            d.dgx_rowsum, d.dghn_rowsum, d.scratch = _p(s.get("dgx_rowsum")), _p(s.get("dghn_rowsum")), _p(s["scratch"])
        return arr

    def gru_bwd_x6_ok(self, scans, cu_budget=0):
        """would gru_seq_bwd run this launch on the bf16 x 6 kernel (fn_gru_bwd_x6_ok)?"""
        if not (self.dw_x6 and self.bwd_x6 and all(s.get("w_hh_t_frag3") is not None for s in scans)):
            return False
        arr = self._bwd_descriptors(scans, cu_budget)
        arr[0].variant = int(self.variant)
        return bool(self.lib.fn_gru_bwd_x6_ok(arr, len(scans)))

    def gru_seq_bwd(self, scans, persistent=True, cu_budget=0, variant=None, x6=None):
        """x6: None = bf16 x 6 where the launch is eligible (a backward launch hands nothing but fp32 tensors to the next one: every launch decides for
        itself), True / False = forced"""
        arr = self._bwd_descriptors(scans, cu_budget)
        variant = self.variant if variant is None else variant
        sync = self._sync_region() if persistent else None
        auto = x6 is None
        if auto:
            x6 = persistent and self.gru_bwd_x6_ok(scans, cu_budget)
        for i, (d, s) in enumerate(zip(arr, scans)):
            d.frag_ws = _p(self._frag_ws("fragb", i, 3 * self.frag_floats(s["B"], 3 * s["H"])))      # 2 slabs of fp32 fragments, or 2 of bf16 triples (x6)
            d.sync_ws, d.err_ws, d.variant = (_p(sync[0]), _p(sync[1]), int(variant) | sync[2]) if persistent else (None, None, int(variant))
            if x6:
                d.w_hh_t_frag, d.variant = _p(s["w_hh_t_frag3"]), d.variant | 0x4000
        rc = self.lib.fn_gru_seq_bwd(arr, len(scans), self.stream())
        if rc == _lib.FN_E_UNSUPPORTED and x6 and auto:
            # the shape predicate said yes, a launch-time check (alignment, occupancy) said no: nothing was launched, and a backward launch hands only
            # fp32 tensors on - the fp32 kernels take it (a FORCED x6=True still raises)
            for d, s in zip(arr, scans):
                d.w_hh_t_frag, d.variant = _p(s["w_hh_t_frag"]), d.variant & ~0x4000
            x6 = False
            rc = self.lib.fn_gru_seq_bwd(arr, len(scans), self.stream())
        _lib.check(rc, "fn_gru_seq_bwd (bf16 x 6)" if x6 else "fn_gru_seq_bwd")

    def gru_dwhh(self, dgx, dghn, hprev, dW, beta=0.0, splitk=1, lean=False):
        """dW [3H][H] = beta*dW + [dgx[:, :2H] | dghn]^T hprev  (dgx [rows][3H], dghn / hprev [rows][H])."""
        for t, nm in ((dgx, "dgx"), (dghn, "dghn"), (hprev, "hprev"), (dW, "dW")):
            _dense(t, name=nm)
        rows, H = hprev.shape
        if tuple(dgx.shape) != (rows, 3 * H) or tuple(dghn.shape) != (rows, H) or tuple(dW.shape) != (3 * H, H):
            raise RuntimeError("gru_dwhh shape mismatch")
        x6 = _lib.GEMM_BF16X6 if (self.dw_x6 and rows >= 1024) else 0
        if x6:
            splitk, x6 = self._x6_mode(splitk, H, rows, x6)
        wsb = int(self.lib.fn_gru_dwhh_ws_bytes(H, splitk))
        ws = self.workspace(wsb, "gemm") if wsb else None
        _lib.check(self.lib.fn_gru_dwhh_f32(_p(dgx), _p(dghn), _p(hprev), rows, H, beta, _p(dW), splitk | (_lib.GEMM_LEAN if lean else 0) | x6,
                                            _p(ws), wsb, self.stream()), "fn_gru_dwhh_f32")

    def decode_greedy(self, B, steps, H, V, start_token, w_hh1_frag, b_hh1, b_ih1, table1, rowbias1, h0, w_ih2_frag, b_ih2, w_hh2_frag, b_hh2,
                      w_out_frag, b_out, tokens, logp=None):
        """single-launch greedy decode of <= 2048 sequences (fn_decode_greedy); tokens int32 [B][>=steps], logp [B][steps][V] or None"""
        d = _lib.FnDecode()
        for t, nm in ((w_hh1_frag, "w_hh1_frag"), (b_hh1, "b_hh1"), (b_ih1, "b_ih1"), (table1, "table1"), (rowbias1, "rowbias1"), (h0, "h0"),
                      (w_ih2_frag, "w_ih2_frag"), (b_ih2, "b_ih2"), (w_hh2_frag, "w_hh2_frag"), (b_hh2, "b_hh2"), (w_out_frag, "w_out_frag"),
                      (b_out, "b_out"), (logp, "logp")):
            _dense(t, name=nm)
            setattr(d, nm, _p(t))
        _dense(tokens, torch.int32, "tokens")
        d.B, d.steps, d.H, d.V, d.start_token = B, steps, H, V, start_token
        d.tokens, d.tok_ld = _p(tokens), tokens.shape[1]
        wsb = int(self.lib.fn_decode_ws_bytes(B, H, V))
        d.ws = _p(self.workspace(wsb, "decode"))
        syncs = self.__dict__.setdefault("_syncs", {})
        key = self.lane + "decode"
        if key not in syncs:
            syncs[key] = torch.zeros(int(self.lib.fn_decode_sync_ws_bytes()) // 4, dtype=torch.int32, device=self.device)
        d.sync_ws = _p(syncs[key])
        rc = self.lib.fn_decode_greedy(C.byref(d), self.stream())
        if rc == _lib.FN_E_UNSUPPORTED:            # not eligible on this device (needs one CU per role workgroup)
            return False
        _lib.check(rc, "fn_decode_greedy")
        return True

    def embed_grad(self, dgx_all, idx, idx_shift, start_token, reverse, V, out):
        _dense(dgx_all, name="dgx_all"), _dense(idx, torch.int32, "idx"), _dense(out, name="out")
        T, B, N3 = dgx_all.shape
        wsb = self.lib.fn_embed_grad_ws_bytes(T * B, V, N3)
        ws = self.workspace(wsb, "embed")
        _lib.check(self.lib.fn_embed_grad_f32(_p(dgx_all), B, T, N3, _p(idx), idx.shape[1], idx_shift, start_token, int(reverse), V,
                                              _p(out), _p(ws), wsb, self.stream()), "fn_embed_grad_f32")

    def token_sort(self, idx, V, img=None):
        """counting sort of the (time, batch) positions of idx [B][T] by token -> handle for embed_grad_sorted.  ONE sort serves
        every scan that reads this token matrix (encoder directions, decoder layer 1)."""
        _dense(idx, torch.int32, "idx")
        B, T = idx.shape
        n = int(self.lib.fn_token_sort_ints(B * T, V))
        if img is None:
            img = torch.empty(n, dtype=torch.int32, device=self.device)
        _dense(img, torch.int32, "img")
        if img.numel() < n:
            raise RuntimeError("token_sort: img too small")
        wsb = int(self.lib.fn_token_sort_ws_bytes(B * T, V))
        ws = self.workspace(wsb, "toksort")
        _lib.check(self.lib.fn_token_sort(_p(idx), B, T, idx.shape[1], V, _p(img), _p(ws), wsb, self.stream()), "fn_token_sort")
        return dict(img=img, B=B, T=T, V=V)

    def embed_grad_sorted(self, handle, jobs):
        """jobs: dicts(dgx [T][B][N3], out = [V][N3] table view, or with transposed=True the [N3][V] view (e.g. dW_ih[:, :V]),
        reverse, idx_shift, start_token) - all reading the token matrix behind `handle`; one launch pair for all jobs."""
        B, T, V = handle["B"], handle["T"], handle["V"]
        arr = (_lib.FnEmbedGrad * len(jobs))()
        N3 = jobs[0]["dgx"].shape[2]
        for d, j in zip(arr, jobs):
            _dense(j["dgx"], name="dgx")
            if tuple(j["dgx"].shape) != (T, B, N3):
                raise RuntimeError("embed_grad_sorted: dgx must be [T][B][N3] of the sorted token matrix")
            po, r, c, ld = _mat(j["out"], "out")
            tr = bool(j.get("transposed", False))
            if (r, c) != ((N3, V) if tr else (V, N3)):
                raise RuntimeError("embed_grad_sorted: out has shape %s" % ((r, c),))
            d.dgx_all, d.out, d.out_ld, d.transposed = _p(j["dgx"]), po, ld, int(tr)
            d.reverse, d.idx_shift, d.start_token = int(j.get("reverse", 0)), int(j.get("idx_shift", 0)), int(j.get("start_token", 0))
        wsb = int(self.lib.fn_embed_grad_sorted_ws_bytes(B * T, B, V, N3, len(jobs)))
        ws = self.workspace(wsb, "embed")
        _lib.check(self.lib.fn_embed_grad_sorted(arr, len(jobs), B, T, N3, V, _p(handle["img"]), _p(ws), wsb, self.stream()), "fn_embed_grad_sorted")

    def time_sum(self, X, out):
        """out[...] = sum over the leading (time) axis of X."""
        _dense(X, name="X"), _dense(out, name="out")
        T = X.shape[0]
        M = X.numel() // T
        if out.numel() != M:
            raise RuntimeError("time_sum: out has %d elements, expected %d" % (out.numel(), M))
        _lib.check(self.lib.fn_time_sum_f32(_p(X), T, M, _p(out), self.stream()), "fn_time_sum_f32")

    # -- heads ----------------------------------------------------------------------------------
    def vocab_logsoftmax(self, logits, B, T, E, logp_bt=None, target=None, nll_rows=None, grad_scale=0.0, dlogits=None):
        pl, rows, _, ld = _mat(logits, "logits")
        _dense(logp_bt, name="logp_bt"), _dense(target, torch.int32, "target"), _dense(nll_rows, name="nll_rows")
        if dlogits is not None and _mat(dlogits, "dlogits")[3] != ld:
            raise RuntimeError("dlogits must share the leading dimension of logits")
        _lib.check(self.lib.fn_vocab_logsoftmax(pl, B, T, E, ld, _p(logp_bt), _p(target), _p(nll_rows), grad_scale, _p(dlogits),
                                                self.stream()), "fn_vocab_logsoftmax")

    def out_head(self, h, W, bias, B, T, target, nll_rows=None, grad_scale=0.0, dlogits=None):
        """fused output head (fn_out_head_f32): h [T*B][H] time-major rows, W [V][H], bias [V], target [B][T] int32 ->
        nll_rows [T*B] and / or dlogits [T*B][ld] = grad_scale * (softmax(h W^T + b) - onehot); the logits are not materialised"""
        ph, R, H, ldh = _mat(h, "h")
        pw, V, Hw, ldw = _mat(W, "W")
        if R != B * T or Hw != H:
            raise RuntimeError("out_head shape mismatch h%s W%s B=%d T=%d" % (tuple(h.shape), tuple(W.shape), B, T))
        _dense(bias, name="bias"), _dense(target, torch.int32, "target"), _dense(nll_rows, name="nll_rows")
        pd, ld = None, 0
        if dlogits is not None:
            pd, rows, _, ld = _mat(dlogits, "dlogits")
            if rows != R:
                raise RuntimeError("out_head: dlogits has %d rows, expected %d" % (rows, R))
        _lib.check(self.lib.fn_out_head_f32(ph, ldh, pw, ldw, _p(bias), B, T, V, H, _p(target), grad_scale, _p(nll_rows), pd, ld,
                                            self.stream()), "fn_out_head_f32")

    def vocab_logsoftmax_bwd(self, logp_bt, gout_bt, dlogits):
        _dense(logp_bt, name="logp_bt"), _dense(gout_bt, name="gout_bt")
        B, T, E = logp_bt.shape
        pd, _, _, ld = _mat(dlogits, "dlogits")
        _lib.check(self.lib.fn_vocab_logsoftmax_bwd(_p(logp_bt), _p(gout_bt), B, T, E, ld, pd, self.stream()), "fn_vocab_logsoftmax_bwd")

    def vocab_argmax(self, logits, E, logp_out, tok_out):
        """logits [B][ld]; logp_out: 2-D view [B][E] (any row stride) or None; tok_out: int32 1-D view (any stride)."""
        pl, B, _, ld = _mat(logits, "logits")
        _chk(tok_out, torch.int32, "tok_out")
        lp_ld = 0
        if logp_out is not None:
            _chk(logp_out, name="logp_out")
            lp_ld = logp_out.stride(0)
        _lib.check(self.lib.fn_vocab_argmax(pl, B, E, ld, _p(logp_out), lp_ld, _p(tok_out), tok_out.stride(0) if tok_out.dim() else 1,
                                            self.stream()), "fn_vocab_argmax")

    def time_logsoftmax(self, logits, logp_bt=None, target=None, nll_bc=None, grad_scale=0.0, dlogits=None):
        _dense(logits, name="logits"), _dense(logp_bt, name="logp_bt"), _dense(target, torch.int32, "target")
        _dense(nll_bc, name="nll_bc"), _dense(dlogits, name="dlogits")
        Tr, B, Cc = logits.shape
        _lib.check(self.lib.fn_time_logsoftmax(_p(logits), B, Tr, Cc, _p(logp_bt), _p(target), _p(nll_bc), grad_scale, _p(dlogits),
                                               self.stream()), "fn_time_logsoftmax")

    def time_logsoftmax_bwd(self, logp_bt, gout_bt, dlogits):
        _dense(logp_bt, name="logp_bt"), _dense(gout_bt, name="gout_bt"), _dense(dlogits, name="dlogits")
        B, Tr, Cc = logp_bt.shape
        _lib.check(self.lib.fn_time_logsoftmax_bwd(_p(logp_bt), _p(gout_bt), B, Tr, Cc, _p(dlogits), self.stream()), "fn_time_logsoftmax_bwd")

    # -- latent ---------------------------------------------------------------------------------
    def latent_fwd(self, pre, eps, mu_lk, lv_lk, labels, sigma, z, ll, qy, y, terms):
        for t, n in ((pre, "pre"), (eps, "eps"), (mu_lk, "mu_lk"), (lv_lk, "lv_lk"), (sigma, "sigma"), (z, "z"), (ll, "ll"), (qy, "qy"),
                     (terms, "terms")):
            _dense(t, name=n)
        _dense(labels, torch.int32, "labels"), _dense(y, torch.int32, "y")
        B, Z = eps.shape
        K = mu_lk.shape[0]
        _lib.check(self.lib.fn_latent_fwd(_p(pre), _p(eps), _p(mu_lk), _p(lv_lk), B, Z, K, _p(labels), _p(sigma), _p(z), _p(ll), _p(qy),
                                          _p(y), _p(terms), self.stream()), "fn_latent_fwd")

    def latent_bwd(self, pre, eps, mu_lk, lv_lk, labels, z, qy, g_z, g_mu, g_sigma, g_ll, g_qy, w3, dpre, dmu_lk_rows):
        """w3: device float tensor {w_lat, w_cls, w_clf} (see fn_step_params) or None (= zeros)"""
        for t, n in ((pre, "pre"), (eps, "eps"), (mu_lk, "mu_lk"), (lv_lk, "lv_lk"), (z, "z"), (qy, "qy"), (g_z, "g_z"), (g_mu, "g_mu"),
                     (g_sigma, "g_sigma"), (g_ll, "g_ll"), (g_qy, "g_qy"), (dpre, "dpre"), (dmu_lk_rows, "dmu_lk_rows"), (w3, "w3")):
            _dense(t, name=n)
        _dense(labels, torch.int32, "labels")
        B, Z = eps.shape
        K = mu_lk.shape[0]
        _lib.check(self.lib.fn_latent_bwd(_p(pre), _p(eps), _p(mu_lk), _p(lv_lk), B, Z, K, _p(labels), _p(z), _p(qy), _p(g_z), _p(g_mu),
                                          _p(g_sigma), _p(g_ll), _p(g_qy), _p(w3), _p(dpre), _p(dmu_lk_rows),
                                          self.stream()), "fn_latent_bwd")

    def masked_prob(self, logits, E, ranges, sums=None, w=None, dlogits=None):
        """softmax mass of two token ranges per logits row (fn_masked_prob) and / or the gradient of w0 P_0 + w1 P_1 wrt the logits"""
        pl, rows, _, ld = _mat(logits, "logits")
        _dense(sums, name="sums"), _dense(w, name="w")
        pd = None
        if dlogits is not None:
            pd, r2, _, ld2 = _mat(dlogits, "dlogits")
            if (r2, ld2) != (rows, ld):
                raise RuntimeError("masked_prob: dlogits must have the layout of logits")
        (lo0, hi0), (lo1, hi1) = ranges
        _lib.check(self.lib.fn_masked_prob(pl, rows, E, ld, lo0, hi0, lo1, hi1, _p(sums), _p(w), pd, self.stream()), "fn_masked_prob")

    def adv_head(self, z, w_r, w_n, b_r, b_n, mask, dens, lam_dev, inv_global_batch, o, loss_rows, da=None, g_z=None):
        """adversarial heads of the Fader sibling (fn_adv_head): z [B][>=Z] row view, g_z likewise (gradient is SUBTRACTED)"""
        pz, B, Zc, ldz = _mat(z, "z")
        for t, nm in ((w_r, "w_r"), (w_n, "w_n"), (b_r, "b_r"), (b_n, "b_n"), (mask, "mask"), (dens, "dens"), (o, "o"), (loss_rows, "loss_rows"),
                      (da, "da"), (lam_dev, "lam_dev")):
            _dense(t, name=nm)
        Z = w_r.numel()
        if Zc < Z or tuple(mask.shape) != (B, 2) or tuple(dens.shape) != (B, 2):
            raise RuntimeError("adv_head shape mismatch")
        pg, ldg = (None, 0)
        if g_z is not None:
            pg, _, _, ldg = _mat(g_z, "g_z")
        _lib.check(self.lib.fn_adv_head(pz, ldz, Z, B, _p(w_r), _p(w_n), _p(b_r), _p(b_n), _p(mask), _p(dens), _p(lam_dev), inv_global_batch,
                                        _p(o), _p(loss_rows), _p(da), pg, ldg, self.stream()), "fn_adv_head")

    def pairwise_reg(self, z0_all, attr_all, row0, nrows, loss_rows, grad_scale=0.0, dz0=None):
        _dense(z0_all, name="z0_all"), _dense(attr_all, torch.float64, "attr_all"), _dense(loss_rows, name="loss_rows"), _dense(dz0, name="dz0")
        _lib.check(self.lib.fn_pairwise_reg(_p(z0_all), _p(attr_all), z0_all.numel(), row0, nrows, _p(loss_rows), grad_scale, _p(dz0),
                                            self.stream()), "fn_pairwise_reg")

    # -- optimiser ------------------------------------------------------------------------------
    def sumsq(self, g, out):
        _dense(g, name="g"), _chk(out, name="out")
        wsb = self.lib.fn_sumsq_ws_bytes(g.numel())
        ws = self.workspace(wsb, "sumsq")
        _lib.check(self.lib.fn_sumsq_f32(_p(g), g.numel(), _p(out), _p(ws), wsb, self.stream()), "fn_sumsq_f32")

    def step_params(self, counters, beta, lr, beta1, beta2, supervised, inv_global_batch, advance, out):
        """device-resident step counters -> {w_lat, w_cls, w_clf, lr/(1-b1^t), 1/sqrt(1-b2^t), beta0} (see fn_step_params)"""
        _dense(counters, torch.int64, "counters"), _dense(out, name="out")
        if counters.numel() < 2 or out.numel() < 8:
            raise RuntimeError("step_params: counters[2] / out[8] expected")
        _lib.check(self.lib.fn_step_params(_p(counters), beta, lr, beta1, beta2, int(supervised), inv_global_batch, int(advance), _p(out),
                                           self.stream()), "fn_step_params")

    def clip_adam(self, p, g, m, v, sumsq, max_norm, hyper, beta1, beta2, eps):
        for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
            _dense(t, name=n)
        _chk(sumsq, name="sumsq"), _chk(hyper, name="hyper")
        _lib.check(self.lib.fn_clip_adam(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(sumsq), max_norm, _p(hyper), beta1, beta2, eps,
                                         self.stream()), "fn_clip_adam")

    def onehot_to_index(self, oh, idx):
        _dense(oh, name="oh"), _dense(idx, torch.int32, "idx")
        V = oh.shape[-1]
        _lib.check(self.lib.fn_onehot_to_index(_p(oh), oh.numel() // V, V, _p(idx), self.stream()), "fn_onehot_to_index")
