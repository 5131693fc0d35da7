"""Forward / backward orchestration of the MusicAttrRegGMVAE path on top of the HIP ops.

This is the host-side schedule: which C-ABI kernel runs on which buffer, in which order.  All
arithmetic happens in the kernels (``ops`` = hipops.HipOps); torch is used for buffer ownership and
a few strided copies.  Reference semantics reproduced (file:line = /root/reference):

  encode            gmm_model.py:82-98      4 concurrent GRU scans + mu/var heads
  repar / q(y|x)    gmm_model.py:229-242    fn_latent_fwd (one wavefront per sample)
  sub_decoders      gmm_model.py:100-117    2 scans + TIME-axis log_softmax
  global_decoder    gmm_model.py:119-149    teacher forced (eps=100): L1 scan, batched W_ih2 GEMM, L2 scan,
                                            batched 512->342 output GEMM + log_softmax
  backward          autograd of the above   reverse scans, batched dW GEMMs, token-segment sums

Layouts: per-step tensors are time-major [T][B][..]; token tensors are [B][T] int32.
"""
import math

import torch

E_VOCAB, R_DIMS, N_DIMS, C_DIMS = 342, 3, 16, 24
LOGIT_LD = 352            # 342 padded to a multiple of the GEMM K tile (32): the input gradient dlogits W_out then takes the branch-free operand loads (K = 344 did not: 308 -> 250 us); the pad columns are zero


def ops_sort(eng, key, tokens, V):
    """fn_token_sort of one token matrix into a shape-keyed image buffer"""
    n = tokens.numel() + 2 * (V + 1) + 2
    img = eng.buf("sort_img_" + key, (n,), torch.int32) if tokens.is_cuda else None
    return eng.ops.token_sort(tokens, V, img)


class Engine:
    def __init__(self, ops, params, hidden, zdim, n_component, device):
        self.ops = ops
        self.p = params                 # name -> device tensor (the nn.Parameter storage)
        self.H, self.Z, self.K = hidden, zdim, n_component
        self.ZG = 2 * zdim + C_DIMS
        self.dev = torch.device(device)
        self._bufs = {}
        self.tab = {}                   # transposed one-hot columns of W_ih:  key -> [V][3H]
        self.whh_f = {}                 # W_hh in the kernels' fragment-major operand layout (forward scans)
        self.whh_f3 = {}                # ... as bf16 triple images (bf16 x 6 forward scans: model.set_arith("bf16x6") -> HipOps.dw_x6)
        self.whh_t = {}                 # W_hh^T, fragment-major (backward scans)
        self.whh_t3 = {}                # ... as bf16 triple images (bf16 x 6 backward scans)
        self.packs = {}                 # fragment-major W_ih2 / W_out (single-launch greedy decode)
        self.saved = None
        self.splitk_big = 16            # K ranges of the T*B-deep weight-gradient products (see _splitk)
        self.chunk = 32                 # time steps per pipeline chunk of the two decoder layers (layer 2 lags layer 1 by two chunks)
        # compute-unit budget of the decoder pipeline's launches (FnGruFwd / FnGruBwd.cu_budget; 0 = the whole chip).  With 128 a launch of two 256-row scans
        # takes the bf16 x 6 kernels in their 128-row (forward) / 64-row (backward) group forms on HALF of the compute units of every XCD, and the lane's
        # projection / state-gradient GEMM - which needs whole CUs (144 KB of LDS) and otherwise runs between the whole-chip launches - runs BESIDE the scan.
        # Measured (round 6): one launch + its GEMM in isolation 442 -> 396 us (backward), 365 -> 353 us (forward), but the captured step gets SLOWER
        # (17.28 - 17.57 -> 17.80 - 17.83 ms backward, 17.93 both): off.  EXPERIMENTS.md R6.3.
        self.dec_bwd_budget = 0
        self.dec_fwd_budget = 0
        self.persist_dec = True         # decoder scans as weight-stationary launches (False: per-step kernels; debug / tests)
        self.fill_edges = True          # the attribute decoders' chunks ride in the half-empty head / tail launches of the global decoder's
                                        # two-layer pipeline (same results; False: one launch of their own)
        self.single_launch_decode = True   # decode.py: small / medium batches decode as ONE launch (False: per-token kernels; tests)
        self.single_launch_rows = 704      # ... up to this many sequences (fn_decode_greedy takes <= 2048; 32-row blocks below 353 rows, 64-row blocks from there), above: fn_gru_cell_f32 per token.  Measured us per token, one launch vs the per-token cells with the tokens-only output layer (scratch/decode_crossover.py, round 4, final): 512 rows 41 / 78, 640 rows 51 / 59, 768 rows 60 / 59, 800 rows 71 / 59, 1024 rows 80 / 60, 1280 rows 100 / 78, 1536 rows 120 / 80, 2048 rows 157 / 99
        self.single_launch_skip = (0, -1)  # (a window of row counts inside single_launch_rows that goes to the cells anyway; unused since the cells with ceil(rows / 512) x 64 rows per workgroup)
        self._lane_alias = {}           # lane -> lane it is folded into (debug)
        self.fused_argmax = True        # decode.py, tokens-only decode on the per-token cells: output layer + argmax as ONE launch (fn_out_argmax_f32); False: GEMM + fn_vocab_argmax (tests)
        self.cell_decode_rows = self.single_launch_rows + 1     # (= 705: no window of row counts left to the T = 1 scan-step path) decode.py: from this many sequences on, the per-token cells are one-launch GEMM cells (fn_gru_cell_f32: LDS-free loop above 512 rows, staged below); measured crossover against the scan-step kernels (scratch/bench_decode_rows.py): 512 rows 67 vs 87 us per token, 768 rows 97 vs 88
        self.fused_head = True          # trainers: output projection + log-softmax + NLL + gradient seed as ONE kernel (fn_out_head_f32); False: GEMM -> logits in HBM -> fn_vocab_logsoftmax
        self.lean_dw = False            # decoder-side weight-gradient GEMMs as the <= 128-register instance.  Paid while an encoder-scan wavefront left 138 of a SIMD's 512 registers free (round 2: 9 % packing gain); the hand-placed K loops hold all of them, the side lane's GEMMs run once the scan has ended and the 194-register instance is the faster one (A/B in one session, scratch/ab_dw.py: 23.65 vs 24.29 ms per step)
        self.lean_proj = True           # layer-2 input projection beside the decoder pipeline's forward launches as the <= 128-register instance of the LDS-free NT kernel: 4 workgroups per CU (768 tiles = one round) and one of its wavefronts fits a SIMD beside a forward-scan wavefront (377 registers): 107 us beside a 351 us launch (the 248-register instance: 125 us in front of the launch)
        self.prepack_h0 = True          # global_decoder_tf: operand images of the initial states of later launches are packed on the aux lane after launch 0 (False: by fn_gru_seq_fwd in front of the launch)
        self.proj_x6 = True             # layer-2 input projection beside the decoder pipeline's FORWARD launches: True = the bf16x6 producer / consumer kernel (78 us, but its 144 KB / 512-thread workgroups cannot share a CU with a scan wavefront: the launch serialises with the scans), False = the lean fp32 instance above, which does (a bf16x6 forward-scan wavefront holds 328 of a SIMD's 512 registers)
        self.dw_order = "side+aux"      # decoder-side weight-gradient GEMMs: "side+aux" = the global decoder's on the side stream, issued in front of the encoder backward, the attribute decoders' on the aux stream behind the encoder scans; "side" = all of them on the side stream; "before" / "after" = caller's stream, in front of / behind the encoder block
        self.losses_early = False       # trainer: True = the side lane's loss-term launches start in FRONT of the fused output head (they need nothing from it) instead of behind it; measured: the graph runs them behind the head's and the dhx1 product's workgroups either way (17.70 vs 17.68 ms), and with 16 hardware queues they land beside the first backward launch (20.4 ms): off
        self.losses_on_side = True      # trainer: the small loss-term launches run on the side lane beside the decoder backward's first launches
        self.buf_ns = ""                # namespace of buf(): a second decoder pass (GLSR) must not overwrite the saved activations of the first
        self.serialize_lanes = False    # True: every lane runs on the caller's stream (per-kernel measurements: each kernel alone)
        if hidden % 32 != 0:
            raise ValueError("hidden_dims must be a multiple of 32 (K chunks of the MFMA step kernels)")
        if n_component > 8:
            raise ValueError("n_component > 8 not supported by fn_latent_*")

    # ------------------------------------------------------------------------------------------
    # Two HIP streams: the caller's current stream ("main") and one side stream.  Independent pieces of the schedule
    # (decoder layer 2 one time-chunk behind layer 1; decoder weight-gradient GEMMs under the encoder backward scans) are
    # enqueued on the side stream so that two kernels are resident at once - the scan steps are latency-bound and leave
    # most of the chip idle.  Each lane has its own scratch (ops.lane) so concurrent kernels never share a workspace.
    def _lane_stream(self, lane):
        """lane: "side" or "aux" -> its HIP stream (None on CPU test backends)."""
        if self.dev.type != "cuda" or self.serialize_lanes:
            return None
        lane = self._lane_alias.get(lane, lane)
        streams = self.__dict__.setdefault("_streams", {})
        if lane not in streams:
            streams[lane] = torch.cuda.Stream(device=self.dev)
        return streams[lane]

    def _side_stream(self):
        return self._lane_stream("side")

    class _Lane:
        def __init__(self, eng, side, lane="side"):
            self.eng, self.side, self.ctx, self.lane = eng, side, None, lane

        def __enter__(self):
            self.prev = getattr(self.eng.ops, "lane", "")
            self.eng.ops.lane = self.eng._lane_alias.get(self.lane, self.lane) + "/" if self.side else ""
            st = self.eng._lane_stream(self.lane) if self.side else None
            if st is not None:
                self.ctx = torch.cuda.stream(st)
                self.ctx.__enter__()
            return self

        def __exit__(self, *a):
            if self.ctx is not None:
                self.ctx.__exit__(*a)
            self.eng.ops.lane = self.prev

    def _cu_count(self):
        return torch.cuda.get_device_properties(self.dev).multi_processor_count if self.dev.type == "cuda" else 256

    def on_side(self):
        return Engine._Lane(self, True)

    def on_aux(self):
        return Engine._Lane(self, True, "aux")

    def lane_wait(self, waiter, waited):
        """stream of lane `waiter` waits for everything enqueued so far on lane `waited` ("main" = the current stream)."""
        if self.dev.type != "cuda":
            return
        get = lambda l: torch.cuda.current_stream(self.dev) if l == "main" else self._lane_stream(l)
        a, b = get(waiter), get(waited)
        if a is not None and b is not None and a != b:
            a.wait_stream(b)

    def side_wait_main(self):
        st = self._side_stream()
        if st is not None:
            st.wait_stream(torch.cuda.current_stream(self.dev))

    def main_wait_side(self):
        st = self._side_stream()
        if st is not None:
            torch.cuda.current_stream(self.dev).wait_stream(st)

    def buf(self, name, shape, dtype=torch.float32, zero_init=False):
        """Named scratch tensor.  The key includes the shape: a captured hipGraph holds raw pointers into these buffers, so a
        buffer is NEVER dropped or reallocated once handed out - another batch shape gets its own set (the epoch driver alternates
        train / validation / ragged-tail shapes, trainer_gmm.py:320-440, and replays the graph of each)."""
        shape = tuple(int(s) for s in shape)
        key = (self.buf_ns + name, shape, dtype)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero_init else torch.empty)(shape, dtype=dtype, device=self.dev)
            self._bufs[key] = t
        return t

    # Accumulators a step starts from zero (per-sequence row sums of the backward scans, upstream latent gradients) live in ONE arena:
    # begin_step() clears it with a single fill instead of one fill kernel per buffer (19 per step).  A buffer that is asked for a second
    # time within a step (or was never part of a cleared arena) is still cleared on its own - zbuf() always returns zeros.
    ARENA_BYTES = 48 << 20

    def zbuf(self, name, shape):
        shape = tuple(int(x) for x in shape)
        key = (self.buf_ns + name, shape, torch.float32)
        t = self._bufs.get(key)
        if t is None:
            n = 1
            for x in shape:
                n *= x
            n_al = (n + 63) // 64 * 64                       # 256-byte aligned carve-outs
            ar = self.__dict__.setdefault("_arena", dict(chunks=[], used=0))
            if not ar["chunks"] or ar["used"] + n_al > ar["chunks"][-1].numel():
                ar["chunks"].append(torch.zeros(max(self.ARENA_BYTES // 4, n_al), device=self.dev))
                ar["used"] = 0
            chunk = ar["chunks"][-1]
            t = chunk[ar["used"]:ar["used"] + n].view(shape)
            ar["used"] += n_al
            ar.setdefault("extent", {})[id(chunk)] = ar["used"]
            self._bufs[key] = t
            return t                                         # fresh chunks are zero
        clean = self.__dict__.get("_arena_clean")
        if clean is not None and key in clean:
            clean.discard(key)
        else:
            t.zero_()
        return t

    def begin_step(self):
        """one fill per arena chunk; every zbuf() of this step that has not been handed out since is then served without a fill kernel"""
        ar = self.__dict__.get("_arena")
        if ar is None:
            return
        for chunk in ar["chunks"]:
            chunk[:ar["extent"][id(chunk)]].zero_()
        self._arena_clean = {k for k in self._bufs if k[2] == torch.float32 and self._in_arena(self._bufs[k])}

    def _in_arena(self, t):
        ar = self.__dict__.get("_arena")
        if ar is None:
            return False
        p = t.data_ptr()
        return any(c.data_ptr() <= p < c.data_ptr() + c.numel() * 4 for c in ar["chunks"])

    # Bias gradients = column sums of small matrices: collected per lane and issued as ONE launch (fn_colsum_multi) where the lane's
    # inputs are complete, instead of a launch pair each (76 launches per step).  Tall inputs keep the two-phase kernel.
    def colsum(self, X, out, beta=0.0):
        if X.shape[0] > 4096 or not hasattr(self.ops, "colsum_multi"):
            self.ops.colsum(X, out, beta)
            return
        self.__dict__.setdefault("_cs_jobs", {}).setdefault(getattr(self.ops, "lane", ""), []).append((X, out, beta))

    def flush_colsums(self):
        jobs = self.__dict__.get("_cs_jobs", {}).pop(getattr(self.ops, "lane", ""), None)
        if jobs:
            self.ops.colsum_multi(jobs)

    # GRU parameter sets on the path: key -> (prefix, suffix, one-hot width V)
    def _gru_sets(self):
        return {
            "r": ("gru_r.", "_l0", E_VOCAB), "r_reverse": ("gru_r.", "_l0_reverse", E_VOCAB),
            "n": ("gru_n.", "_l0", E_VOCAB), "n_reverse": ("gru_n.", "_l0_reverse", E_VOCAB),
            "d_r": ("gru_d_r.", "_l0", R_DIMS), "d_n": ("gru_d_n.", "_l0", N_DIMS),
            "g": ("grucell_g.", "", E_VOCAB), "g2": ("grucell_g_2.", "", 0),
        }

    def refresh_weights(self, need_backward=True):
        """Re-derive the transposed / fragment-major weight images after the parameters changed (once per optimiser step): ONE launch
        for all of them (fn_weight_images)."""
        H = self.H
        jobs = []
        for key, (pfx, sfx, V) in self._gru_sets().items():
            w_ih = self.p[pfx + "weight_ih" + sfx]
            if V > 0:
                self.tab[key] = self.buf("tab_" + key, (V, 3 * H))
                jobs.append(("transpose", w_ih[:, :V], self.tab[key]))
            w_hh = self.p[pfx + "weight_hh" + sfx]
            self.whh_f[key] = self.buf("whhf_" + key, (self.ops.frag_floats(3 * H, H),))
            jobs.append(("frag", w_hh, self.whh_f[key]))
            if need_backward:
                self.whh_t[key] = self.buf("whht_" + key, (self.ops.frag_floats(H, 3 * H),))
                jobs.append(("frag_t", w_hh, self.whh_t[key]))
        # operand images of the two dense matrices of the single-launch greedy decode (decode.py): refreshed here so that the
        # captured training step keeps them current - a decode after training must not see the weights of an earlier step
        for key, name in (("ih2", "grucell_g_2.weight_ih"), ("out", "linear_out_g.weight")):
            w = self.p[name]
            if key not in self.packs:
                self.packs[key] = torch.zeros(self.ops.frag_floats(w.shape[0], w.shape[1]), device=self.dev)
            jobs.append(("frag", w, self.packs[key]))
        if need_backward:                                    # linear_out_g.weight^T [H][LOGIT_LD], pad columns zero: B operand of the input gradient of the output layer
            if getattr(self, "wout_t", None) is None:
                self.wout_t = torch.zeros(H, LOGIT_LD, device=self.dev)
            # grucell_g_2.weight_ih^T [H][3H]: with it the layer-2 input gradient dhx0 = dgx2 W_ih2 is a product of two K-contiguous
            # operands (the LDS-free fn_gemm_f32 path: 104 us per 32-step chunk instead of 132)
            w2 = self.p.get("grucell_g_2.weight_ih")
            if w2 is not None:
                self.wih2_t = self.buf("wih2_t", (w2.shape[1], w2.shape[0]))
                jobs.append(("transpose", w2, self.wih2_t))
        if need_backward:
            # the z columns of the three cells' input matrices W_ih[:, V:] as dense, 16-byte aligned matrices: dz = drb W_ih[:, V:] sits on the critical
            # path between the decoder and the encoder backward, and a slice that starts at column 342 / 3 takes fn_gemm_multi's element-wise loads
            self.wz = {}
            for key, name, V in (("g", "grucell_g.weight_ih", E_VOCAB), ("d_r", "gru_d_r.weight_ih_l0", R_DIMS), ("d_n", "gru_d_n.weight_ih_l0", N_DIMS)):
                w = self.p.get(name)
                if w is not None and hasattr(self.ops, "weight_images"):
                    self.wz[key] = self.buf("wz_" + key, (w.shape[0], w.shape[1] - V))
                    jobs.append(("copy", w[:, V:], self.wz[key]))
        if getattr(self.ops, "dw_x6", False) and H == 512:
            # bf16 x 6 arithmetic: bf16 triple images of the recurrent matrices (forward scans) and of their transposes (backward scans)
            for key, (pfx, sfx, V) in self._gru_sets().items():
                w_hh = self.p[pfx + "weight_hh" + sfx]
                self.whh_f3[key] = self.buf("whhf3_" + key, (self.ops.frag_floats(3 * H, H) * 3 // 2,))
                jobs.append(("frag3", w_hh, self.whh_f3[key]))
                if need_backward:
                    self.whh_t3[key] = self.buf("whht3_" + key, (self.ops.frag_floats(H, 3 * H) * 3 // 2,))
                    jobs.append(("frag3_t", w_hh, self.whh_t3[key]))
        else:
            self.whh_f3.clear()
            self.whh_t3.clear()
        self.ops.weight_images(jobs)
        if need_backward:
            self.ops.transpose(self.p["linear_out_g.weight"], self.wout_t[:, :E_VOCAB])

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def encode(self, d, save=True):
        """4 concurrent scans (gru_r / gru_n x fwd / reverse) + mu|var heads -> pre_e [B][2Z] (mu | log-sigma)."""
        ops, P, H, Z = self.ops, self.p, self.H, self.Z
        B, T = d.shape
        scans, hall = [], {}
        for e in ("r", "n"):
            for key, sfx, rev in ((e, "_l0", 0), (e + "_reverse", "_l0_reverse", 1)):
                pfx = "gru_%s." % e
                hall[key] = self.buf("enc_h_" + key, (T, B, H))
                scans.append(dict(B=B, T=T, H=H, reverse=rev, w_hh_frag=self.whh_f[key], w_hh_frag3=self.whh_f3.get(key), b_hh=P[pfx + "bias_hh" + sfx],
                                  b_ih=P[pfx + "bias_ih" + sfx], gx_table=self.tab[key], idx=d, idx_shift=0,
                                  h_all=hall[key], gates=self.buf("enc_g_" + key, (T, ops.gates_floats(B, H))) if save else None))
        ops.gru_seq_fwd(scans)
        pre = {"h_all": hall, "gates": {sc_key: sc["gates"] for sc_key, sc in zip(hall, scans)}}
        jobs = []
        for e in ("r", "n"):
            hf, hb = hall[e][T - 1], hall[e + "_reverse"][T - 1]
            pre[e] = self.buf("pre_" + e, (B, 2 * Z))
            for head, c0 in (("mu_", 0), ("var_", Z)):          # [h_fwd | h_rev] W^T + b: two products into one output (gmm_model.py:85-86)
                W = P[head + e + ".weight"]
                jobs.append(dict(C=pre[e][:, c0:c0 + Z], segs=[(hf, W[:, :H]), (hb, W[:, H:])], bias=P[head + e + ".bias"]))
        ops.gemm_multi(jobs)                                     # the four heads of both encoders: one launch
        return pre

    def latent(self, pre, eps, labels=None):
        ops, P, Z, K = self.ops, self.p, self.Z, self.K
        out = {}
        for e in ("r", "n"):
            B = pre[e].shape[0]
            o = dict(sigma=self.buf("sigma_" + e, (B, Z)), z=self.buf("z_" + e, (B, Z)), ll=self.buf("ll_" + e, (B, K)),
                     qy=self.buf("qy_" + e, (B, K)), y=self.buf("y_" + e, (B,), torch.int32), terms=self.buf("terms_" + e, (B, 4)))
            ops.latent_fwd(pre[e], eps[e], P["mu_%s_lookup.weight" % e], P["logvar_%s_lookup.weight" % e], labels,
                           o["sigma"], o["z"], o["ll"], o["qy"], o["y"], o["terms"])
            out[e] = o
        return out

    def sub_decoders_fwd(self, r, n, z_r, z_n, save=True, defer=False):
        """gmm_model.py:100-117 up to the pre-softmax logits: both attribute decoders as ONE weight-stationary launch (whole chip).
        defer=True: only the initial states / per-sequence input parts; the scan descriptors are left in sd[e]['scan'] for
        global_decoder_tf(fill=...) and the logits for _sub_decoder_logits."""
        ops, P, H = self.ops, self.p, self.H
        B, Tr = r.shape
        scans, sd = [], {}
        jobs = []
        for e, attr, Ce, z in (("r", r, R_DIMS, z_r), ("n", n, N_DIMS, z_n)):
            h0 = self.buf("sd_h0_" + e, (B, H))
            jobs.append(dict(C=h0, segs=[(z, P["linear_init_%s.weight" % e])], bias=P["linear_init_%s.bias" % e]))
            w_ih = P["gru_d_%s.weight_ih_l0" % e]
            rb = self.buf("sd_rb_" + e, (B, 3 * H))
            jobs.append(dict(C=rb, segs=[(z, w_ih[:, Ce:])]))
            sd[e] = dict(h0=h0, rb=rb, h_all=self.buf("sd_h_" + e, (Tr, B, H)),
                         gates=self.buf("sd_g_" + e, (Tr, ops.gates_floats(B, H))) if save else None)
            scans.append(dict(B=B, T=Tr, H=H, w_hh_frag=self.whh_f["d_" + e], w_hh_frag3=self.whh_f3.get("d_" + e), b_hh=P["gru_d_%s.bias_hh_l0" % e],
                              b_ih=P["gru_d_%s.bias_ih_l0" % e], h0=h0, gx_table=self.tab["d_" + e], idx=attr, gx_rowbias=rb,
                              h_all=sd[e]["h_all"], gates=sd[e]["gates"]))
        if defer:                                                # (the four products ride in global_decoder_tf's launch of its own two)
            for e, sc in zip(("r", "n"), scans):
                sc["tag"] = "sd_" + e
                sd[e]["scan"] = sc
            sd["init_jobs"] = jobs
            return sd
        ops.gemm_multi(jobs)                                     # initial states + per-sequence input parts of both decoders: one launch
        ops.gru_seq_fwd(scans)
        return self._sub_decoder_logits(sd, Tr, B)

    def _sub_decoder_logits(self, sd, Tr, B):
        ops, P, H = self.ops, self.p, self.H
        for e, Ce in (("r", R_DIMS), ("n", N_DIMS)):
            sd[e]["logits"] = self.buf("sd_logits_" + e, (Tr, B, Ce))
            ops.gemm(sd[e]["h_all"].view(Tr * B, H), P["linear_out_%s.weight" % e], sd[e]["logits"].view(Tr * B, Ce),
                     bias=P["linear_out_%s.bias" % e])
        return sd

    def pack_zc(self, z_r, z_n, c):
        """z = cat([z_r, z_n, chroma], 1) (gmm_model.py:249)"""
        Z = self.Z
        zc = self.buf("zc", (z_r.shape[0], self.ZG))
        zc[:, :Z].copy_(z_r)
        zc[:, Z:2 * Z].copy_(z_n)
        zc[:, 2 * Z:].copy_(c)
        return zc

    def _fill_ok(self, T, Tr):
        """the attribute decoders (Tr steps) fit the two head and two tail launches of the global decoder's pipeline"""
        CH = self.chunk
        return bool(self.fill_edges and self.persist_dec and Tr <= 2 * CH and T >= 2 * CH and Tr > 1)

    def global_decoder_tf(self, d, zc, save=True, head=True, fill=None, more_jobs=()):
        """gmm_model.py:119-149 in train mode (teacher forced with d, start token 341, input shifted by one step) up to the
        pre-softmax logits [T*B][LOGIT_LD].  head=False: the output projection is left to the caller's fused head
        (ops.out_head on dec['hx1'], which writes the gradient seed into dec['logits']); the buffer is returned unwritten."""
        ops, P, H = self.ops, self.p, self.H
        B, T = d.shape
        h0g = self.buf("g_h0", (B, H))
        rbg = self.buf("g_rb", (B, 3 * H))
        ops.gemm_multi([dict(C=h0g, segs=[(zc, P["linear_init_global.weight"])], bias=P["linear_init_global.bias"]),
                        dict(C=rbg, segs=[(zc, P["grucell_g.weight_ih"][:, E_VOCAB:])])] + list(more_jobs))
        hx0 = self.buf("g_hx0", (T, B, H))
        g1 = self.buf("g_gates1", (T, ops.gates_floats(B, H))) if save else None
        l1 = dict(B=B, T=T, H=H, w_hh_frag=self.whh_f["g"], w_hh_frag3=self.whh_f3.get("g"), b_hh=P["grucell_g.bias_hh"], b_ih=P["grucell_g.bias_ih"],
                  h0=h0g, gx_table=self.tab["g"], idx=d, idx_shift=-1, start_token=E_VOCAB - 1, gx_rowbias=rbg, h_all=hx0, gates=g1)
        gx2 = self.buf("g_gx2", (T, B, 3 * H))
        hx1 = self.buf("g_hx1", (T, B, H))
        g2 = self.buf("g_gates2", (T, ops.gates_floats(B, H))) if save else None
        l2 = dict(B=B, T=T, H=H, w_hh_frag=self.whh_f["g2"], w_hh_frag3=self.whh_f3.get("g2"), b_hh=P["grucell_g_2.bias_hh"], h0=None, gx_dense=gx2, h_all=hx1, gates=g2)
        # Time is cut into chunks and the two layers run as ONE weight-stationary launch per chunk step k:
        #     launch k = [layer 1, chunk k] + [layer 2, chunk k-2]      (two independent scans: 8 row groups of 64 = one per XCD, so a
        #                                                                row group's state exchange stays inside one XCD's L2)
        #     aux lane  : gx2[chunk k] = hx0[chunk k] W_ih2^T + b_ih2   (one GEMM, beside launch k+1)
        # Layer 2 lags layer 1 by two chunks; its state starts from hx0[0] (gmm_model.py:134-135).
        CH = self.chunk
        pd = self.persist_dec
        l1["tag"], l2["tag"] = "dec_l1", "dec_l2"
        starts = list(range(0, T, CH))
        nch = len(starts)
        # a chunk leaves its final state ALSO in the fragment-major exchange layout; the next chunk of that layer starts from it
        # without a packing launch (FnGruFwd.h_last_frag -> h0_frag)
        nf = ops.frag_floats(B, H) * 3 // 2          # fp32 fragments, or bf16 triples (the opt-in bf16 x 6 scans)
        hand = {name: [self.buf("g_hand_%s_%d" % (name, i), (nf,)) for i in range(2)] for name in ("l1", "l2", "sd_r", "sd_n")}

        def chunk(name, sc, ci, n=None):
            n = nch if n is None else n
            c = self._fwd_chunk(sc, ci * CH, ci * CH + CH)
            if ci > 0:
                c["h0_frag"] = hand[name][(ci - 1) & 1]
            if ci + 1 < n:
                c["h_last_frag"] = hand[name][ci & 1]
            return c

        # launches 0, 1 (layer 1 alone) and nch, nch+1 (layer 2 alone) would leave half of the chip idle: the attribute decoders' chunks
        # (independent scans of the same batch) ride there - 'r' in the head, 'n' in the tail
        fill = fill or {}
        nsd = {e: (sc["T"] + CH - 1) // CH for e, sc in fill.items()}
        launches = []
        for k in range(nch + 2):
            part = []
            if k < nch:
                part.append(chunk("l1", l1, k))
            if k < 2 and "r" in fill and k < nsd["r"]:
                part.append(chunk("sd_r", fill["r"], k, nsd["r"]))
            if k >= nch and "n" in fill and k - nch < nsd["n"]:
                part.append(chunk("sd_n", fill["n"], k - nch, nsd["n"]))
            if k >= 2:
                c2 = chunk("l2", l2, k - 2)
                if k == 2:
                    c2["h0"] = hx0[0]
                part.append(c2)
            launches.append(part)
        # ONE arithmetic for the whole pipeline: the chunks of a scan hand their state over as operand images (bf16 triples or fp32
        # fragments), so a single launch the bf16 x 6 kernels do not take (a one-step tail chunk, more row groups than compute units) puts
        # every launch on the fp32 MFMA
        x6 = bool(pd and getattr(ops, "gru_fwd_x6_ok", None) and all(ops.gru_fwd_x6_ok(part) for part in launches if part))
        xkw = {"x6": x6} if hasattr(ops, "gru_fwd_x6_ok") else {}
        if x6 and self.dec_fwd_budget and all(ops.gru_fwd_x6_ok(part, self.dec_fwd_budget) for part in launches if part):
            xkw["cu_budget"] = self.dec_fwd_budget
        # Initial states that a LATER launch starts from (layer 2: hx0[0], known after launch 0; the 'n' attribute decoder in the tail) are
        # turned into operand images here, on the aux lane in front of the first projection - left to fn_gru_seq_fwd the packing launch
        # sits between two scan launches beside a projection that saturates the memory pipes: 109 us instead of 6, twice per step
        prepack = []
        if self.prepack_h0 and pd and nch >= 2 and hasattr(ops, "frag3_pack"):
            for k, part in enumerate(launches):
                for c in part:
                    if k >= 2 and c.get("h0") is not None and c.get("h0_frag") is None:
                        img = self.buf("g_h0img_%s" % c["tag"], (nf,))
                        prepack.append((c["h0"], img))
                        c["h0_frag"] = img
        for k, part in enumerate(launches):
            if k >= 2:
                self.lane_wait("main", "aux%d" % (k & 1))            # the projection of chunk k-2 (issued two launches ago)
            if part:
                ops.gru_seq_fwd(part, persistent=pd, **xkw)
            if k < nch:
                t0, t1 = starts[k], min(T, starts[k] + CH)
                lane = "aux%d" % (k & 1)
                self.lane_wait(lane, "main")
                with Engine._Lane(self, True, lane):
                    if k == 0:
                        for src, img in prepack:
                            (ops.frag3_pack if x6 else ops.frag_pack)(src, img)
                    ops.gemm(hx0[t0:t1].view(-1, H), P["grucell_g_2.weight_ih"], gx2[t0:t1].view(-1, 3 * H), bias=P["grucell_g_2.bias_ih"],
                             lean=self.lean_proj, **({} if self.proj_x6 else {"nt_x6": False}))
        logits = self.buf("g_logits", (T * B, LOGIT_LD), zero_init=True)     # columns [342, 352) stay zero: every writer leaves them alone or writes zeros
        if head:
            ops.gemm(hx1.view(T * B, H), P["linear_out_g.weight"], logits[:, :E_VOCAB], bias=P["linear_out_g.bias"])
        return dict(zc=zc, h0g=h0g, rbg=rbg, hx0=hx0, g1=g1, gx2=gx2, hx1=hx1, g2=g2, logits=logits)

    def decoders(self, d, r, n, c, z_r, z_n, save=True, head=True, sd_logits=True):
        """sub-decoders + teacher-forced global decoder up to the (pre-softmax) logits.  sd_logits=False: the attribute decoders' output
        layers are left to the caller (``sub_decoder_logits(S)``: the trainer issues them on the side lane, beside its fused head)."""
        if self._fill_ok(d.shape[1], r.shape[1]):
            sd = self.sub_decoders_fwd(r, n, z_r, z_n, save, defer=True)
            dec = self.global_decoder_tf(d, self.pack_zc(z_r, z_n, c), save, head, fill={e: sd[e].pop("scan") for e in ("r", "n")}, more_jobs=sd.pop("init_jobs"))
            if sd_logits:
                self._sub_decoder_logits(sd, r.shape[1], r.shape[0])
        else:
            sd = self.sub_decoders_fwd(r, n, z_r, z_n, save)
            dec = self.global_decoder_tf(d, self.pack_zc(z_r, z_n, c), save, head)
        dec["sd"] = sd
        return dec

    @staticmethod
    def _fwd_chunk(sc, t0, t1):
        """steps [t0, t1) of a forward (non-reverse) scan descriptor: the state carries over through h_all[t0-1]"""
        t1 = min(t1, sc["T"])
        c = dict(sc)
        c["T"] = t1 - t0
        if t0 > 0:
            c["h0"] = sc["h_all"][t0 - 1]
        c["h_all"] = sc["h_all"][t0:t1]
        if sc.get("gates") is not None:
            c["gates"] = sc["gates"][t0:t1]
        if sc.get("gx_dense") is not None:
            c["gx_dense"] = sc["gx_dense"][t0:t1]
        c["idx_shift"] = sc.get("idx_shift", 0) + t0
        return c

    @staticmethod
    def _bwd_chunk(sc, t0, t1, carry_in, carry_out):
        """steps [t0, t1) of a backward scan descriptor; the gradient wrt the state before t0 leaves through carry_out and
        enters the previous chunk as dh_last"""
        c = dict(sc)
        c["T"] = t1 - t0
        if t0 > 0:
            c["h0"] = sc["h_all"][t0 - 1]
        for k in ("h_all", "gates", "dh_ext", "dgx_all", "dghn_all"):
            if sc.get(k) is not None:
                c[k] = sc[k][t0:t1]
        c["dh_last"] = carry_in
        c["dh0"] = carry_out
        return c

    def sub_decoder_logits(self, S):
        """the output layers forward(sd_logits=False) left out (no-op when they were not deferred)"""
        sd = S["dec"]["sd"]
        if "logits" not in sd["r"]:
            self._sub_decoder_logits(sd, S["r"].shape[1], S["r"].shape[0])
        return sd

    def forward(self, d, r, n, c, eps_r, eps_n, labels=None, save=True, head=True, sd_logits=True):
        """Full training-mode forward up to logits; everything backward needs stays in named buffers (save=False: forward only,
        the gate tensors are not written).  head=False: see global_decoder_tf; sd_logits=False: see decoders()."""
        sort = None
        if save:
            # token sorts for the backward's segment sums (embed.hip): tiny kernels, side stream, beside the encoder scans
            self.side_wait_main()
            with self.on_side():
                sort = {k: ops_sort(self, k, t, V) for k, t, V in (("d", d, E_VOCAB), ("r", r, R_DIMS), ("n", n, N_DIMS))}
        pre = self.encode(d, save)
        lat = self.latent(pre, {"r": eps_r, "n": eps_n}, labels)
        dec = self.decoders(d, r, n, c, lat["r"]["z"], lat["n"]["z"], save, head, sd_logits)
        self.main_wait_side()
        S = dict(d=d, r=r, n=n, c=c, eps={"r": eps_r, "n": eps_n}, labels=labels, pre=pre, lat=lat, dec=dec, sort=sort)
        self.saved = S if save else None
        return S

    # ------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------
    def _gru_weight_grads(self, key, pfx, sfx, T, B, dgx, dghn, h_all, h0, G, splitk, rs, rsn, lean=False):
        """dW_hh / db_hh of one scan from the saved per-step gate gradients (batched over all steps); the bias gradients
        are column sums of the per-sequence row sums rs [B][3H] / rsn [B][H] accumulated by the backward scan."""
        ops, H = self.ops, self.H
        dW = G[pfx + "weight_hh" + sfx]
        dgx2 = dgx.view(T * B, 3 * H)
        dgn2 = dghn.view(T * B, H)
        if T > 1:
            ops.gru_dwhh(dgx2[B:], dgn2[B:], h_all.view(T * B, H)[: (T - 1) * B], dW, splitk=splitk, lean=lean)
        else:
            dW.zero_()
        if h0 is not None:
            ops.gru_dwhh(dgx2[:B], dgn2[:B], h0, dW, beta=1.0)
        db = G[pfx + "bias_hh" + sfx]
        self.colsum(rs[:, : 2 * H], db[: 2 * H])
        self.colsum(rsn, db[2 * H:])

    def _splitk(self, rows):
        # 48 tiles x split depth workgroups at two per CU: 8 ranges = 384 workgroups leave half of the second round empty (rows = 16384: 264 us = 0.62 of the MFMA
        # peak against 210 us = 0.78 with 16; scratch/sweep_dwhh_splitk.py)
        return self.splitk_big if rows >= 8192 else (8 if rows >= 4096 else (4 if rows >= 1024 else 1))

    def _bwd_global_decoder_scans(self, S, fill=None, before_first_launch=None):
        """Backward of global_decoder_tf up to the gate gradients (dlogits must already be in S['dec']['logits'], in place):
        output layer dX, then the two cells, chunk-pipelined.  Returns the gate-gradient buffers, the per-sequence row sums
        (drb_g = d(W_ih[:, V:] z) rows, i.e. the gradient wrt the conditioning projection) and dh0_g = dL/d(linear_init_global(z))."""
        ops, P, H = self.ops, self.p, self.H
        dec = S["dec"]
        B, T = S["d"].shape
        dlog = dec["logits"]                                   # [T*B][LOGIT_LD], holds dlogits
        dhx1 = self.buf("g_dhx1", (T, B, H))
        ops.gemm(dlog, self.wout_t, dhx1.view(T * B, H), a_k=True, b_k=True)         # K = LOGIT_LD incl. the zero pad columns: whole K tiles
        if before_first_launch is not None:
            before_first_launch()          # backward(): joins the side lane (loss terms, the attribute decoders' gradient seeds) behind this GEMM
        dgx2 = self.buf("g_dgx2", (T, B, 3 * H))
        dghn2 = self.buf("g_dghn2", (T, B, H))
        dhx0 = self.buf("g_dhx0", (T, B, H))
        dgx1 = self.buf("g_dgx1", (T, B, 3 * H))
        dghn1 = self.buf("g_dghn1", (T, B, H))
        # per-sequence sums over time of the gate gradients (bias / z-projection gradients) are accumulated by the scans
        rs2, rsn2 = self.zbuf("g_rs2", (B, 3 * H)), self.zbuf("g_rsn2", (B, H))
        drb_g, rsn_g = self.zbuf("g_drb", (B, 3 * H)), self.zbuf("g_rsn1", (B, H))
        l2 = dict(B=B, T=T, H=H, w_hh_t_frag=self.whh_t["g2"], w_hh_t_frag3=self.whh_t3.get("g2"), h0=dec["hx0"][0], h_all=dec["hx1"], gates=dec["g2"], dh_ext=dhx1,
                  dgx_all=dgx2, dghn_all=dghn2, scratch=self.buf("g_scr2", (B, H)), dgx_rowsum=rs2, dghn_rowsum=rsn2, tag="dec_l2")
        l1 = dict(B=B, T=T, H=H, w_hh_t_frag=self.whh_t["g"], w_hh_t_frag3=self.whh_t3.get("g"), h0=dec["h0g"], h_all=dec["hx0"], gates=dec["g1"], dh_ext=dhx0,
                  dgx_all=dgx1, dghn_all=dghn1, scratch=self.buf("g_scr1", (B, H)), dgx_rowsum=drb_g, dghn_rowsum=rsn_g, tag="dec_l1")
        CH = self.chunk
        carry = {k: [self.buf("carry_%s_%d" % (k, i), (B, H)) for i in range(2)] for k in ("l2", "l1")}

        def chunk(name, sc, t0):
            """chunk descriptor with ping-pong carries of the state gradient (slot = chunk parity)"""
            t1 = min(t0 + CH, T)
            slot = (t0 // CH) & 1
            return self._bwd_chunk(sc, t0, t1, None if t1 >= T else carry[name][slot ^ 1], carry[name][slot])

        # the attribute decoders' reverse scans ride in the half-empty launches (see global_decoder_tf): 'n' beside layer 2's first two
        # chunks, 'r' beside layer 1's last two; fill[e] = (descriptor, buffer for dL/dh0)
        fill = fill or {}
        fch = {}
        for e, (sc, dh0) in fill.items():
            ts = list(reversed(range(0, sc["T"], CH)))
            cb = self.buf("carry_sd_" + e, (B, H))
            fch[e] = [self._bwd_chunk(sc, t0, min(t0 + CH, sc["T"]), None if t0 + CH >= sc["T"] else cb, dh0 if t0 == 0 else cb) for t0 in ts]

        pd = self.persist_dec
        # launch k = [layer 2, chunk js[k]] + [layer 1, chunk js[k-2]] (time runs backwards: js = last chunk .. first), ONE weight-
        # stationary launch of two independent scans (8 row groups = one per XCD); beside launch k+1 the aux lane turns the layer-2
        # gate gradients of chunk js[k] into layer 1's incoming state gradient: dhx0[chunk] = dgx2[chunk] W_ih2 (one GEMM)
        js = list(reversed(range(0, T, CH)))
        nch = len(js)
        # (gru_bwd_rs_kernel claims the whole register file: the aux lane's projection GEMM runs in the gaps of these launches, never on a SIMD
        # beside a scan wave - gru_persist.hip, profiles/r05_eager_nondeterminism.txt; eager and captured launches take the same kernel)
        for k in range(nch + 2):
            part = []
            if k < nch:
                part.append(chunk("l2", l2, js[k]))
            if k < 2 and "n" in fch and k < len(fch["n"]):
                part.append(fch["n"][k])
            if k >= 2:
                self.lane_wait("main", "auxb%d" % (k & 1))
                part.append(chunk("l1", l1, js[k - 2]))
            if k >= nch and "r" in fch and k - nch < len(fch["r"]):
                part.append(fch["r"][k - nch])
            if part:
                bkw = {}
                if pd and self.dec_bwd_budget and getattr(ops, "dw_x6", False) and getattr(ops, "bwd_x6", False) and len(part) == 2 \
                        and ops.gru_bwd_x6_ok(part, self.dec_bwd_budget):
                    bkw = {"cu_budget": self.dec_bwd_budget}
                ops.gru_seq_bwd(part, persistent=pd, **bkw)
            if k < nch:
                t0, t1 = js[k], min(T, js[k] + CH)
                lane = "auxb%d" % (k & 1)
                self.lane_wait(lane, "main")
                with Engine._Lane(self, True, lane):
                    ops.gemm(dgx2[t0:t1].view(-1, 3 * H), self.wih2_t, dhx0[t0:t1].view(-1, H), a_k=True, b_k=True)
                    if t0 == 0:
                        ops.axpy(1.0, carry["l2"][0], dhx0[0])          # hx1 was initialised with hx0[0]: dL/dh_init of layer 2
        return dict(dgx1=dgx1, dghn1=dghn1, dgx2=dgx2, dghn2=dghn2, rs2=rs2, rsn2=rsn2, drb_g=drb_g, rsn_g=rsn_g, dh0_g=carry["l1"][0])

    def _bwd_global_decoder_params(self, G, S, gd, flush=True):
        """parameter gradients of the global decoder (linear_out_g, grucell_g_2, grucell_g, linear_init_global) from the gate gradients;
        flush=False: the caller issues the collected bias-gradient column sums itself (together with more of them)"""
        ops, H = self.ops, self.H
        dec = S["dec"]
        B, T = S["d"].shape
        sk_T = self._splitk(T * B)
        dlog = dec["logits"]
        hx1f, hx0f = dec["hx1"].view(T * B, H), dec["hx0"].view(T * B, H)
        dgx1, dghn1, dgx2, dghn2, rs2, rsn2, drb_g, rsn_g, dh0_g = (gd[k] for k in ("dgx1", "dghn1", "dgx2", "dghn2", "rs2", "rsn2", "drb_g", "rsn_g", "dh0_g"))
        # 342 x 512 output: only 12 tiles of 128 x 128 - a deeper K split fills the chip (42 x 12 = 504 workgroups: 258 vs 325 us)
        ln = self.lean_dw
        ops.gemm(dlog[:, :E_VOCAB], hx1f, G["linear_out_g.weight"], a_k=False, b_k=False, splitk=42 if T * B >= 32768 else sk_T, lean=ln)
        self.colsum(dlog[:, :E_VOCAB], G["linear_out_g.bias"])
        self._gru_weight_grads("g2", "grucell_g_2.", "", T, B, dgx2, dghn2, dec["hx1"], dec["hx0"][0], G, sk_T, rs2, rsn2, lean=ln)
        ops.gemm(dgx2.view(T * B, 3 * H), hx0f, G["grucell_g_2.weight_ih"], a_k=False, b_k=False, splitk=sk_T, lean=ln)
        self.colsum(rs2, G["grucell_g_2.bias_ih"])
        self._gru_weight_grads("g", "grucell_g.", "", T, B, dgx1, dghn1, dec["hx0"], dec["h0g"], G, sk_T, drb_g, rsn_g, lean=ln)
        dWg = G["grucell_g.weight_ih"]                          # [3H][E+ZG]: token columns = segment sums, written in place
        ops.embed_grad_sorted(S["sort"]["d"], [dict(dgx=dgx1, out=dWg[:, :E_VOCAB], transposed=True, idx_shift=-1, start_token=E_VOCAB - 1)])
        ops.gemm(drb_g, dec["zc"], dWg[:, E_VOCAB:], a_k=False, b_k=False)
        self.colsum(drb_g, G["grucell_g.bias_ih"])
        ops.gemm(dh0_g, dec["zc"], G["linear_init_global.weight"], a_k=False, b_k=False)
        self.colsum(dh0_g, G["linear_init_global.bias"])
        if flush:
            self.flush_colsums()

    def backward(self, G, dlogits_sd, lat_up, w3=None, after_decoders=None, after_encoder_r=None):
        """Backward of forward().

        G            name -> gradient tensor to FILL (views of the flat gradient buffer)
        (dlogits of the global decoder must already be in saved['dec']['logits'], in place)
        dlogits_sd   {'r','n'} -> [Tr][B][Ce] gradient wrt the sub-decoder pre-softmax logits
        lat_up       {'r','n'} -> dict(g_z, g_mu, g_sigma, g_ll, g_qy) upstream gradients (entries may be None;
                     g_z is a REQUIRED zero-or-filled [B][Z] buffer that decoder gradients are accumulated into)
        w3           device tensor {w_lat, w_cls, w_clf}: fused loss weights of fn_latent_bwd (None = only the upstream gradients)
        after_decoders  optional callback fired once every decoder-side parameter gradient is enqueued
                     (data parallel: start reducing that bucket while the encoder scans run)
        after_encoder_r optional callback fired once everything but the note encoder's gradients is enqueued (heads, component means,
                     rhythm encoder: the second bucket, reduced beside the note encoder's weight-gradient GEMMs)
        """
        ops, P, H, Z, ZG, K = self.ops, self.p, self.H, self.Z, self.ZG, self.K
        S = self.saved
        d, r, n = S["d"], S["r"], S["n"]
        dec, lat, pre = S["dec"], S["lat"], S["pre"]
        B, T = d.shape
        Tr = r.shape[1]
        sk_T, sk_Tr = self._splitk(T * B), self._splitk(Tr * B)

        # ---- global decoder: output layer + both cells, chunk-pipelined (see _bwd_global_decoder_scans) ----------------
        sd = dec["sd"]
        if self._fill_ok(T, Tr):
            # ---- both attribute decoders ride in the half-empty launches of the global decoder's pipeline ------------------------
            # the caller's loss terms / gradient seeds of the sub-decoders and of z sit on the side lane (beside the fused head): the two
            # output-layer products of the attribute decoders follow them THERE, and the lane is joined behind the dhx1 product
            self.side_wait_main()        # (a caller that wrote the seeds on its own stream)
            with self.on_side():
                sdb, sds = self._bwd_sub_decoder_scans(sd, dlogits_sd, B, Tr, defer=True)
            gd = self._bwd_global_decoder_scans(S, fill={e: (sds[e], sdb[e]["dh0"]) for e in ("r", "n")}, before_first_launch=self.main_wait_side)
        else:
            gd = self._bwd_global_decoder_scans(S)
            self.main_wait_side()
            # ---- sub-decoders: both attribute decoders, all Tr steps, ONE whole-chip launch ----------------------------------
            sdb = self._bwd_sub_decoder_scans(sd, dlogits_sd, B, Tr)
        dgx1, dghn1, dgx2, dghn2, rs2, rsn2, drb_g, rsn_g, dh0_g = (gd[k] for k in ("dgx1", "dghn1", "dgx2", "dghn2", "rs2", "rsn2", "drb_g", "rsn_g", "dh0_g"))
        dlog = dec["logits"]
        pd = self.persist_dec
        # ---- what the encoder side needs from the decoders (main stream, critical path): dz ---------------------
        Wz_g, Wig = P["grucell_g.weight_ih"], P["linear_init_global.weight"]
        # four products into each g_z (it already holds the regulariser's part).  As ONE job per latent that is 8 workgroups walking
        # K = 4096 on the critical path (245 us); instead 6 partial products per latent with K <= 3H/2 (12 jobs, one launch) and one
        # column-sum launch that adds the partials to g_z in a fixed order
        NP = 6
        part = self.buf("gz_part", (2, NP, B, Z))
        jobs, sums = [], []
        for ei, (e, c0, Ce) in enumerate((("r", 0, R_DIMS), ("n", Z, N_DIMS))):
            wz = getattr(self, "wz", {})          # aligned images of the z columns (refresh_weights)
            prods = [(drb_g, wz["g"][:, c0:c0 + Z] if "g" in wz else Wz_g[:, E_VOCAB + c0:E_VOCAB + c0 + Z]), (dh0_g, Wig[:, c0:c0 + Z]),
                     (sdb[e]["drb"], wz.get("d_" + e, P["gru_d_%s.weight_ih_l0" % e][:, Ce:])), (sdb[e]["dh0"], P["linear_init_%s.weight" % e])]
            pieces = []
            for A, W in prods:
                Kp = A.shape[1]
                if Kp >= 2 * H and Kp % 2 == 0:
                    pieces += [(A[:, :Kp // 2], W[:Kp // 2]), (A[:, Kp // 2:], W[Kp // 2:])]
                else:
                    pieces.append((A, W))
            assert len(pieces) == NP
            jobs += [dict(C=part[ei, i], beta=0.0, segs=[pc]) for i, pc in enumerate(pieces)]
            sums.append((part[ei].view(NP, B * Z), lat_up[e]["g_z"].view(-1), 1.0))
        ops.gemm_multi(jobs, a_k=True, b_k=False)
        ops.colsum_multi(sums)
        # ---- decoder-side PARAMETER gradients (dw_order: where they run relative to the latent block and the encoder scans) ---
        def sub_decoder_params():
            self._bwd_sub_decoder_params(G, sd, sdb, dlogits_sd, S["sort"], {"r": lat["r"]["z"], "n": lat["n"]["z"]}, B, Tr)
            self.flush_colsums()

        def decoder_params():
            self._bwd_global_decoder_params(G, S, gd, flush=False)
            sub_decoder_params()
            if after_decoders is not None:
                after_decoders()           # data parallel: this bucket's all-reduce is ordered behind these launches

        if self.dw_order == "side+aux":
            # three lanes: the global decoder's deep products on the side lane (issued in front of the encoder block, they run once its
            # scans have ended); the attribute decoders' ~25 small launches (K = Tr*B products, segment sums, column sums: a 0.6 ms
            # dependent chain that fills a fraction of the chip) on the aux lane, issued BEHIND the encoder scans so that they run in
            # the gaps of the deep products instead of alone at the end of the step
            self.side_wait_main()
            with self.on_side():
                self._bwd_global_decoder_params(G, S, gd, flush=True)

            def after_scans():
                self.lane_wait("aux", "main")
                with self.on_aux():
                    sub_decoder_params()
                self.lane_wait("side", "aux")
                if after_decoders is not None:
                    with self.on_side():
                        after_decoders()
            self.backward_encoder(G, S, lat_up, w3, after_encoder_r, after_scans=after_scans)
            self.main_wait_side()
        elif self.dw_order in ("side", "side_late"):
            def on_side_lane():
                self.side_wait_main()
                with self.on_side():
                    decoder_params()
            if self.dw_order == "side":
                on_side_lane()
            # "side_late": the side lane starts behind the latent block / heads, so its GEMMs do not sit beside those small
            # launches of the critical path
            self.backward_encoder(G, S, lat_up, w3, after_encoder_r, before_scans=on_side_lane if self.dw_order == "side_late" else None)
            self.main_wait_side()
        elif self.dw_order == "before":
            decoder_params()
            self.backward_encoder(G, S, lat_up, w3, after_encoder_r)
        else:
            self.backward_encoder(G, S, lat_up, w3, after_encoder_r)
            decoder_params()

    def _bwd_sub_decoder_scans(self, sd, dlogits_sd, B, Tr, defer=False):
        """output layers' input gradients + the reverse scans of both attribute decoders (ONE launch); -> per decoder dict(dgx, dghn,
        drb = per-sequence sums of the gate gradients (= gradient wrt the z projection), rsn, dh0 = dL/d linear_init(z))"""
        ops, P, H = self.ops, self.p, self.H
        dh_sd, sdb, sds = {}, {}, {}
        jobs = []
        for e, Ce in (("r", R_DIMS), ("n", N_DIMS)):
            dl = dlogits_sd[e].view(Tr * B, Ce)
            dh_sd[e] = self.buf("sd_dh_" + e, (Tr, B, H))
            jobs.append(dict(C=dh_sd[e].view(Tr * B, H), segs=[(dl, P["linear_out_%s.weight" % e])]))
            sdb[e] = dict(dgx=self.buf("sd_dgx_" + e, (Tr, B, 3 * H)), dghn=self.buf("sd_dghn_" + e, (Tr, B, H)),
                          drb=self.zbuf("sd_drb_" + e, (B, 3 * H)), rsn=self.zbuf("sd_rsn_" + e, (B, H)), dh0=self.buf("sd_dh0_" + e, (B, H)))
            sds[e] = dict(B=B, T=Tr, H=H, w_hh_t_frag=self.whh_t["d_" + e], w_hh_t_frag3=self.whh_t3.get("d_" + e), h0=sd[e]["h0"], h_all=sd[e]["h_all"], gates=sd[e]["gates"],
                          dh_ext=dh_sd[e], dgx_all=sdb[e]["dgx"], dghn_all=sdb[e]["dghn"], scratch=self.buf("sd_scr_" + e, (B, H)),
                          dgx_rowsum=sdb[e]["drb"], dghn_rowsum=sdb[e]["rsn"], tag="sd_" + e)
        ops.gemm_multi(jobs, a_k=True, b_k=False)                # both output layers' input gradients (K = 3 / 16): one launch
        if defer:                                                # the scans are left to _bwd_global_decoder_scans(fill=...)
            return sdb, sds
        ops.gru_seq_bwd([self._bwd_chunk(sds[e], 0, Tr, None, sdb[e]["dh0"]) for e in ("r", "n")], persistent=self.persist_dec)
        return sdb

    def _bwd_sub_decoder_params(self, G, sd, sdb, dlogits_sd, sorts, z, B, Tr):
        """parameter gradients of both attribute decoders (gru_d_*, linear_out_*, linear_init_*) from the gate gradients"""
        ops, H = self.ops, self.H
        sk_Tr = self._splitk(Tr * B)
        for e, Ce in (("r", R_DIMS), ("n", N_DIMS)):
            pfx = "gru_d_%s." % e
            dl = dlogits_sd[e].view(Tr * B, Ce)
            ops.gemm(dl, sd[e]["h_all"].view(Tr * B, H), G["linear_out_%s.weight" % e], a_k=False, b_k=False, splitk=sk_Tr)
            self.colsum(dl, G["linear_out_%s.bias" % e])
            self._gru_weight_grads("d_" + e, pfx, "_l0", Tr, B, sdb[e]["dgx"], sdb[e]["dghn"], sd[e]["h_all"], sd[e]["h0"], G, sk_Tr,
                                   sdb[e]["drb"], sdb[e]["rsn"], lean=self.lean_dw)
            dW = G[pfx + "weight_ih_l0"]                        # [3H][Ce+Z]
            ops.embed_grad_sorted(sorts[e], [dict(dgx=sdb[e]["dgx"], out=dW[:, :Ce], transposed=True)])
            ops.gemm(sdb[e]["drb"], z[e], dW[:, Ce:], a_k=False, b_k=False)
            self.colsum(sdb[e]["drb"], G[pfx + "bias_ih_l0"])
            ops.gemm(sdb[e]["dh0"], z[e], G["linear_init_%s.weight" % e], a_k=False, b_k=False)
            self.colsum(sdb[e]["dh0"], G["linear_init_%s.bias" % e])

    def backward_encoder(self, G, S, lat_up, w3=None, after_encoder_r=None, before_scans=None, after_scans=None):
        """latent block + heads + the four encoder scans and their parameter gradients (the last third of backward(); also the whole
        backward of a direct ``model.encode(x)`` call: S then holds d, pre, lat, eps, labels, sort['d'] only)"""
        ops, P, H, Z, K = self.ops, self.p, self.H, self.Z, self.K
        pre, lat = S["pre"], S["lat"]
        B, T = S["d"].shape
        sk_T = self._splitk(T * B)
        # ---- latent block + heads -----------------------------------------------------------------
        scans = []
        encb = {}
        dh_jobs, head_jobs, head_bias = [], [], []
        for e in ("r", "n"):
            up = lat_up[e]
            dpre = self.buf("dpre_" + e, (B, 2 * Z))
            dmu_rows = self.buf("dmulk_rows_" + e, (B, K * Z))
            ops.latent_bwd(pre[e], S["eps"][e], P["mu_%s_lookup.weight" % e], P["logvar_%s_lookup.weight" % e], S["labels"],
                           lat[e]["z"], lat[e]["qy"], up["g_z"], up.get("g_mu"), up.get("g_sigma"), up.get("g_ll"), up.get("g_qy"),
                           w3, dpre, dmu_rows)
            if "mu_%s_lookup.weight" % e in G:           # the plain-VAE sibling has no component means to train
                self.colsum(dmu_rows, G["mu_%s_lookup.weight" % e].view(-1))
            hf = pre["h_all"][e][T - 1]
            hb = pre["h_all"][e + "_reverse"][T - 1]
            dhf, dhb = self.buf("enc_dhf_" + e, (B, H)), self.buf("enc_dhb_" + e, (B, H))
            Wm, Wv = P["mu_" + e + ".weight"], P["var_" + e + ".weight"]      # [Z][2H]
            dpm, dpv = dpre[:, :Z], dpre[:, Z:]
            # the scans wait for dhf / dhb only: both encoders' four products are ONE launch behind the two latent launches, the heads'
            # own parameter gradients follow the scan launch (they used to sit in front of it, 2 x 20 us of the critical path)
            dh_jobs += [dict(C=dhf, segs=[(dpm, Wm[:, :H]), (dpv, Wv[:, :H])]), dict(C=dhb, segs=[(dpm, Wm[:, H:]), (dpv, Wv[:, H:])])]
            head_jobs += [dict(C=G[head + e + ".weight"][:, c0:c0 + H], segs=[(dp, hh)])
                          for head, dp in (("mu_", dpm), ("var_", dpv)) for c0, hh in ((0, hf), (H, hb))]
            head_bias += [(dp, G[head + e + ".bias"]) for head, dp in (("mu_", dpm), ("var_", dpv))]
            for key, dh in ((e, dhf), (e + "_reverse", dhb)):
                encb[key] = dict(dgx=self.buf("enc_dgx_" + key, (T, B, 3 * H)), dghn=self.buf("enc_dghn_" + key, (T, B, H)),
                                 rs=self.zbuf("enc_rs_" + key, (B, 3 * H)), rsn=self.zbuf("enc_rsn_" + key, (B, H)))
                scans.append(dict(B=B, T=T, H=H, w_hh_t_frag=self.whh_t[key], w_hh_t_frag3=self.whh_t3.get(key), h0=None, h_all=pre["h_all"][key],
                                  gates=pre["gates"][key], dh_last=dh, dgx_all=encb[key]["dgx"], dghn_all=encb[key]["dghn"],
                                  scratch=self.buf("enc_scr_" + key, (B, H)), dgx_rowsum=encb[key]["rs"], dghn_rowsum=encb[key]["rsn"]))
        ops.gemm_multi(dh_jobs, a_k=True, b_k=False)
        if before_scans is not None:
            before_scans()
        ops.gru_seq_bwd(scans)        # 4 concurrent reverse scans, one weight-stationary launch (the side lane's GEMMs run once it has ended)
        if after_scans is not None:
            after_scans()
        ops.gemm_multi(head_jobs, a_k=False, b_k=False)
        for dp, gb in head_bias:
            self.colsum(dp, gb)
        enc_keys = [(e, "gru_%s." % e, key, sfx, rev) for e in ("r", "n") for key, sfx, rev in ((e, "_l0", 0), (e + "_reverse", "_l0_reverse", 1))]
        # one-hot columns of the four W_ih: token-segment sums of the gate gradients (ONE launch pair, the batch's token sort is shared)
        ops.embed_grad_sorted(S["sort"]["d"], [dict(dgx=encb[key]["dgx"], out=G[pfx + "weight_ih" + sfx], transposed=True, reverse=rev)
                                               for e, pfx, key, sfx, rev in enc_keys])
        for i, (e, pfx, key, sfx, rev) in enumerate(enc_keys):
            self._gru_weight_grads(key, pfx, sfx, T, B, encb[key]["dgx"], encb[key]["dghn"], pre["h_all"][key], None, G, sk_T,
                                   encb[key]["rs"], encb[key]["rsn"])
            self.colsum(encb[key]["rs"], G[pfx + "bias_ih" + sfx])
            if i == 1 and after_encoder_r is not None:           # both directions of gru_r done (enc_keys: r, r_reverse, n, n_reverse)
                self.flush_colsums()
                after_encoder_r()
        self.flush_colsums()
