"""Fused training / evaluation step of the GM-VAE, mirroring trainer_gmm.py:220-293.

``GMVAETrainer.train(step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, is_supervised, y_label)`` keeps
the reference's argument list and returns ``(step + 1, (loss, CE_X, CE_R, CE_N, l_r, l_n, kld_latent, kld_class))``
(trainer_gmm.py:252-258) but runs without autograd: forward kernels -> fused loss / gradient-seed kernels ->
backward kernels -> clip + Adam kernel over one flat parameter buffer.

Loss terms (trainer_gmm.py:109-217); every reduction is normalised by the GLOBAL batch so that data-parallel
ranks simply SUM their gradients and statistics:
  CE = 5*CE_X + CE_R + CE_N (means over B*T / B*Tr incl. pad, :131-138)
  beta0 schedule :125-128 (negative between step 1000 and 10000 - reproduced)
  unsupervised: + beta0 * (sum_k mean_b q_k KL_k  [r,n]  +  mean_b(mean_k q log q) - log(1/K)  [r,n])  (:150-178)
  supervised:   + beta0 * mean_b KL_label [r,n] + CrossEntropy(softmax(qy), label) [r,n]            (:181-194)
  + pairwise tanh/sign regulariser on z[:,0] for r and n                                             (:199-217)
"""
import contextlib
import math
import os

import numpy as np

import torch

from .engine import E_VOCAB

# parameters whose gradient is complete once the decoder backward has run (first all-reduce bucket)
DECODER_PREFIXES = ("linear_out_g.", "grucell_g_2.", "grucell_g.", "linear_init_global.", "gru_d_r.", "gru_d_n.",
                    "linear_out_r.", "linear_out_n.", "linear_init_r.", "linear_init_n.")

# layout of the device-side statistics vector
S_CE_X, S_CE_R, S_CE_N, S_L_R, S_L_N, S_TERMS_R, S_TERMS_N, S_LEN = 0, 1, 2, 3, 4, 5, 9, 16


def beta_schedule(step, beta):
    """trainer_gmm.py:125-128."""
    return 0.0 if step < 1000 else min((step - 10000) / 10000 * beta, beta)


class FlatParams:
    """Flat fp32 images of the trainable, used parameters: parameter / gradient / Adam m, v.

    The module's parameters are re-pointed at views of the flat parameter buffer (``state_dict`` keeps
    working, one kernel updates everything) and the gradient buffer is what data-parallel all-reduces.
    Order = the order in which the backward completes the gradients, so that data parallel can reduce them in three
    contiguous buckets while later work still runs: [decoder side | heads, component means, rhythm encoder | note encoder]
    (bucket 1 overlaps the encoder backward scans, bucket 2 the note encoder's weight-gradient GEMMs; only bucket 3 is exposed)."""

    def __init__(self, model):
        named = model.used_parameters()
        dec = [kp for kp in named if kp[0].startswith(DECODER_PREFIXES)]
        last = [kp for kp in named if kp[0].startswith("gru_n.")]
        mid = [kp for kp in named if not kp[0].startswith(DECODER_PREFIXES) and not kp[0].startswith("gru_n.")]
        mid = [kp for kp in mid if not kp[0].startswith("gru_r.")] + [kp for kp in mid if kp[0].startswith("gru_r.")]
        named = dec + mid + last
        dev = named[0][1].device
        self.offsets, off = {}, 0
        self.bucket_split = self.bucket_split2 = None
        for i, (k, p) in enumerate(named):
            if i == len(dec):
                self.bucket_split = off
            if i == len(dec) + len(mid):
                self.bucket_split2 = off
            self.offsets[k] = off
            off += (p.numel() + 3) // 4 * 4          # 16-byte aligned segment starts
        self.n = off
        self.bucket_split = off if self.bucket_split is None else self.bucket_split
        self.bucket_split2 = off if self.bucket_split2 is None else self.bucket_split2
        self.param = torch.zeros(off, device=dev)
        self.grad = torch.zeros(off, device=dev)
        self.m = torch.zeros(off, device=dev)
        self.v = torch.zeros(off, device=dev)
        self.names = [k for k, _ in named]
        self.G = {}
        for k, p in named:
            o, s = self.offsets[k], p.numel()
            self.param[o:o + s].copy_(p.data.reshape(-1))
            p.data = self.param[o:o + s].view_as(p.data)
            self.G[k] = self.grad[o:o + s].view_as(p.data)
        self.t = 0
        model._engine = None           # parameter storage moved: rebuild the engine's pointer table
        model.weights_changed()


class GMVAETrainer:
    def __init__(self, model, lr=1e-3, beta=0.2, max_norm=1.0, dist_ctx=None):
        self.model = model
        self.lr, self.beta, self.max_norm = lr, beta, max_norm
        self.flat = FlatParams(model)
        self.dist = dist_ctx                         # parallel.DataParallelContext or None
        dev = self.flat.param.device
        self.stats = torch.zeros(S_LEN, device=dev)  # partial sums of the loss terms (device side)
        self.sumsq = torch.zeros(1, device=dev)
        # step counters live on the device (fn_step_params): [training step, Adam t]; sp = derived scalars
        self.counters = torch.zeros(2, dtype=torch.int64, device=dev)
        self.sp = torch.zeros(8, device=dev)
        self._dev_step = 0                          # host mirror of counters[0]
        self.use_graph = dev.type == "cuda"          # replay the whole step as ONE hipGraph (no per-launch host cost)
        if dist_ctx is not None:
            # Data parallel: the collectives are RCCL calls on our own streams (parallel.DirectRccl, fn_comm_*) and CAN be part of the captured
            # step.  (Through torch.distributed's process group they were not capturable reliably: about 1 capture in 25 ended with
            # hipErrorStreamCaptureUnjoined on torch 2.10 / RCCL 2.26 and its watchdog thread then died on an event "last recorded in a
            # capturing stream" - round 2, DESIGN.md.)  The captured form has been run and soaked with ONE rank only (the pool hands out
            # 1-GPU boxes), so with real peers (world > 1) the default is eager launches; FN_DP_GRAPH=1 opts in to the single-graph step,
            # FN_DP_GRAPH=0 forces eager launches everywhere.  DataParallelContext(direct=False) (torch.distributed collectives): eager.
            want = os.environ.get("FN_DP_GRAPH")
            if not getattr(dist_ctx, "want_direct", False) or want == "0" or (want is None and dist_ctx.world > 1):
                self.use_graph = False
        self._dens_all = None                        # data parallel: the batch's densities of ALL ranks (gathered once per batch)
        self._graphs = {}
        self._static = {}
        model.train()

    # ------------------------------------------------------------------------------------------
    def prepare_batch(self, d, r, n, c, r_density, n_density, y_label=None):
        """Upload one batch: token tensors -> int32 [B][T]; densities stay float64 (only the SIGN of their
        pairwise float64 differences is used, trainer_gmm.py:208-210)."""
        m = self.model
        dev = self.flat.param.device
        # [B][T] tensors are token ids whatever their dtype (the loaders yield float32 ids, cast by `.long()` at
        # trainer_gmm.py:323,397); [B][T][V] tensors are the convert_to_one_hot images
        d = m._indices(torch.as_tensor(d).to(dev), 342, ids_ndim=2)
        r = m._indices(torch.as_tensor(r).to(dev), 3, ids_ndim=2)
        n = m._indices(torch.as_tensor(n).to(dev), 16, ids_ndim=2)
        c = torch.as_tensor(c).to(dev).float().contiguous()
        f64 = lambda x: (x.to(dev).double() if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, dtype=np.float64)).to(dev)).contiguous()
        rd, nd = f64(r_density), f64(n_density)
        lab = None if y_label is None else torch.as_tensor(y_label).to(dev).to(torch.int32).contiguous()
        return d, r, n, c, rd, nd, lab

    def draw_eps(self, B, T):
        """eps in the reference's CPU-generator order (see MusicAttrRegGMVAE._draw_eps).  Data parallel: the draw is made for the GLOBAL
        batch and this rank keeps its rows - with lockstep generators (train.py seeds every rank alike) the concatenation over ranks is the
        single-process draw, and no two shards share their noise."""
        dev = self.flat.param.device
        if self.dist is None or self.dist.world == 1:
            return self.model._draw_eps(B, T, dev)
        lo = self.dist.rank * B
        return tuple(e[lo:lo + B].contiguous() for e in self.model._draw_eps(B * self.dist.world, T, dev))

    def _forward_losses(self, step, batch, eps, want_grads):
        m = self.model
        eng = m.engine()
        ops = eng.ops
        d, r, n, c, rd, nd, labels = batch
        B, T = d.shape
        Tr = r.shape[1]
        Bg = B if self.dist is None else self.dist.global_batch(B)
        Z = eng.Z
        beta0 = beta_schedule(step, self.beta)
        self._step_now = step
        fused = eng.fused_head
        if want_grads:
            eng.begin_step()                       # one fill for every zero-initialised accumulator of the step
        side = eng.losses_on_side
        S = eng.forward(d, r, n, c, eps[0], eps[1], labels, head=not fused, sd_logits=not side)
        dec, lat = S["dec"], S["lat"]
        st = self.stats
        if side and eng.losses_early:
            eng.side_wait_main()                   # the side lane starts HERE: its ~25 small launches run beside the output head below
        # reconstruction terms; the gradient seeds land where the logits would be (fused head: the logits are never written)
        nll = eng.buf("nll_rows", (T * B,))
        gs = 5.0 / (Bg * T) if want_grads else 0.0
        if fused:
            ops.out_head(dec["hx1"].view(T * B, eng.H), eng.p["linear_out_g.weight"], eng.p["linear_out_g.bias"], B, T, d, nll_rows=nll,
                         grad_scale=gs, dlogits=dec["logits"] if want_grads else None)
        else:
            ops.vocab_logsoftmax(dec["logits"], B, T, E_VOCAB, target=d, nll_rows=nll, grad_scale=gs, dlogits=dec["logits"] if want_grads else None)
        ops.sum(nll, st[S_CE_X:S_CE_X + 1], 1.0 / (Bg * T))
        # Everything below is ~25 dependent launches of a few microseconds each that need nothing from the output head: the attribute
        # decoders' output layers and loss terms, the latent terms, the regulariser.  They run on the side lane beside the head and the
        # first product of the decoder backward (Engine.backward joins the lane in front of its first scan launch).
        if side and not eng.losses_early:
            eng.side_wait_main()                   # (A/B switch: the lane starts behind the head, as up to round 5)
        with (eng.on_side() if side else contextlib.nullcontext()):
            # first the launches that hold no LDS (column sums of the latent terms, regulariser): they fit a CU beside a workgroup of the dhx1 product that
            # Engine.backward issues next (144 of 160 KB); the two output-layer products of the attribute decoders (37 KB of LDS) only start once it has ended
            for slot, e in ((S_TERMS_R, "r"), (S_TERMS_N, "n")):
                ops.colsum(lat[e]["terms"], st[slot:slot + 4])      # latent terms: column sums of the per-row terms written by fn_latent_fwd
            lat_up = self._regulariser(eng, S, batch, Bg, want_grads)
            eng.sub_decoder_logits(S)
            dl_sd = {}
            for slot, e, attr, Ce in ((S_CE_R, "r", r, 3), (S_CE_N, "n", n, 16)):
                nbc = eng.buf("nll_bc_" + e, (B, Ce))
                dl_sd[e] = eng.buf("sd_dlogits_" + e, (Tr, B, Ce)) if want_grads else None
                ops.time_logsoftmax(dec["sd"][e]["logits"], target=attr, nll_bc=nbc, grad_scale=1.0 / (Bg * Tr), dlogits=dl_sd[e])
                ops.sum(nbc, st[slot:slot + 1], 1.0 / (Bg * Tr))
        if not want_grads:
            eng.main_wait_side()
        return dl_sd, lat_up, self.sp[0:3], beta0, Bg

    def _regulariser(self, eng, S, batch, Bg, want_grads):
        """pairwise regulariser on z[:, 0] against the GLOBAL batch (trainer_gmm.py:199-217): fills stats[S_L_R / S_L_N] and returns
        the upstream gradient buffers {'r','n'} -> dict(g_z=[B][Z]) the backward accumulates the decoders' dz into"""
        ops, lat, Z = eng.ops, S["lat"], eng.Z
        d, r, n, c, rd, nd, labels = batch
        B = d.shape[0]
        st = self.stats
        lat_up = {}
        # column 0 of both latents: one [2][B] buffer, and data parallel ONE all-gather for the two of them (it sits on the critical path
        # between the encoder and the decoders; the densities - constants of the batch - were gathered once per batch, _gather_densities)
        z0 = eng.buf("reg_z0", (2, B))
        z0[0].copy_(lat["r"]["z"][:, 0])
        z0[1].copy_(lat["n"]["z"][:, 0])
        if self.dist is not None:
            W = self.dist.world
            z0_all = eng.buf("reg_z0_all", (2, W * B))
            z0_all.view(2, W, B).copy_(self.dist.gather(z0.view(-1)).view(W, 2, B).permute(1, 0, 2))      # [rank][latent][row] -> [latent][global row]
            row0 = self.dist.rank * B
        else:
            z0_all, row0 = z0, 0
        for i, (slot, e, attr) in enumerate(((S_L_R, "r", rd), (S_L_N, "n", nd))):
            if self.dist is not None:
                held = self._dens_all[i] if self._dens_all is not None else None
                a_all = held if held is not None else self.dist.gather(attr)
            else:
                a_all = attr
            lrow = eng.buf("reg_rows_" + e, (B,))
            dz0 = eng.buf("reg_dz0_" + e, (B,)) if want_grads else None
            ops.pairwise_reg(z0_all[i], a_all, row0, B, lrow, 1.0 / (Bg * Bg), dz0)
            ops.sum(lrow, st[slot:slot + 1], 1.0 / (Bg * Bg))
            if want_grads:
                gz = eng.zbuf("g_z_" + e, (B, Z))
                gz[:, 0].copy_(dz0)
                lat_up[e] = dict(g_z=gz)
        return lat_up

    def _tuple8(self, beta0, Bg, supervised):
        """device partial sums -> the reference's 8 numbers (ONE D2H copy; the reference does 8 .item() calls, :257)."""
        if self.dist is not None:
            self.dist.all_reduce_sum(self.stats)
        s = self.stats.tolist()
        # the host is synchronised here anyway: a weight-stationary scan whose workgroups gave up waiting (bounded spins) has left
        # garbage behind - fail loudly instead of returning it
        check = getattr(self.model.engine().ops, "gru_sync_error", None)
        if check is not None and check(clear=True):
            # e.g. another process / stream held CUs for seconds.  The flag is cleared so that the NEXT step can run; this one is lost.
            raise RuntimeError("a weight-stationary GRU launch timed out waiting for its row group (bounded spin, sync error word set): "
                               "the results of this step are invalid; the error word was cleared, the step can be repeated "
                               "(Engine.persist_dec = False / HipOps.gru_seq_*(persistent=False) select the per-step kernels)")
        K = self.model.n_component
        ce_x, ce_r, ce_n, l_r, l_n = s[S_CE_X], s[S_CE_R], s[S_CE_N], s[S_L_R], s[S_L_N]
        tr_, tn_ = s[S_TERMS_R:S_TERMS_R + 4], s[S_TERMS_N:S_TERMS_N + 4]
        if not supervised:
            kld_lat = tr_[0] / Bg + tn_[0] / Bg
            kld_cls = (tr_[1] / Bg - math.log(1.0 / K)) + (tn_[1] / Bg - math.log(1.0 / K))
            loss = 5 * ce_x + ce_r + ce_n + beta0 * (kld_lat + kld_cls)
        else:
            kld_lat = tr_[2] / Bg + tn_[2] / Bg
            kld_cls = 0.0
            loss = 5 * ce_x + ce_r + ce_n + beta0 * kld_lat + (tr_[3] / Bg + tn_[3] / Bg)
        loss += l_r + l_n
        return (loss, ce_x, ce_r, ce_n, l_r, l_n, kld_lat, kld_cls)

    # ------------------------------------------------------------------------------------------
    def loss_and_grads(self, step, batch, eps):
        """forward + losses + backward WITHOUT the optimiser update: fills flat.grad, returns the reference's 8 numbers
        (used by the parity tests; the step counters are not advanced)"""
        m = self.model
        eng = m.engine()
        self._sync_step_counter(step)
        supervised = batch[6] is not None
        Bg = batch[0].shape[0] if self.dist is None else self.dist.global_batch(batch[0].shape[0])
        eng.ops.step_params(self.counters, self.beta, self.lr, 0.9, 0.999, supervised, 1.0 / Bg, False, self.sp)
        self._gather_densities(batch, None)
        fw = self._forward_losses(step, batch, eps, want_grads=True)
        self._run_backward(fw, None)
        eng.ops.sumsq(self.flat.grad, self.sumsq)
        return self._tuple8(fw[3], fw[4], supervised)

    def _gather_densities(self, batch, static):
        """data parallel: the pairwise regulariser needs the densities of the GLOBAL batch - constants of the batch, so they are
        all-gathered once per batch (not once per step inside the step's graph); `static` holds the buffers a captured graph reads"""
        if self.dist is None:
            self._dens_all = None
            return
        outs = []
        for i, t in enumerate((batch[4], batch[5])):
            out = None
            if static is not None:
                key = "dens_all%d" % i
                if key not in static:
                    static[key] = torch.empty(t.numel() * self.dist.world, dtype=t.dtype, device=t.device)
                out = static[key]
            outs.append(self.dist.gather(t, out))
        self._dens_all = tuple(outs)

    def _run_backward(self, fw, hook, hook2=None):
        """fw = what _forward_losses returned; fills flat.G"""
        self.model.engine().backward(self.flat.G, fw[0], fw[1], fw[2], after_decoders=hook, after_encoder_r=hook2)

    def _sync_step_counter(self, step):
        """the caller owns `step` (trainer_gmm.py:60,252); the device counter follows it (one tiny copy only when they differ)"""
        if step != self._dev_step:
            self.counters[0:1].copy_(torch.tensor([step], dtype=torch.int64))
            self._dev_step = step

    def _step_body(self, step, batch, eps, advance=True):
        """everything one optimisation step enqueues (no host sync, no step-dependent host scalar in a kernel argument)"""
        m = self.model
        eng = m.engine()
        ops = eng.ops
        supervised = batch[6] is not None
        Bg = batch[0].shape[0] if self.dist is None else self.dist.global_batch(batch[0].shape[0])
        ops.step_params(self.counters, self.beta, self.lr, 0.9, 0.999, supervised, 1.0 / Bg, advance, self.sp)
        fw = self._forward_losses(step, batch, eps, want_grads=True)
        beta0, Bg = fw[3], fw[4]
        hook = hook2 = None
        if self.dist is not None:
            f = self.flat
            hook = lambda: self.dist.start_bucket(f.grad[:f.bucket_split], "bucket1")
            hook2 = lambda: self.dist.start_bucket(f.grad[f.bucket_split:f.bucket_split2], "bucket2")
        if hook2 is not None and type(self)._run_backward is GMVAETrainer._run_backward:
            self._run_backward(fw, hook, hook2)
        else:                                      # trainers of other model families: two buckets
            self._run_backward(fw, hook)
            hook2 and hook2()
        if self.dist is not None:
            self.dist.start_bucket(self.flat.grad[self.flat.bucket_split2:], "bucket3")
            self.dist.finish_buckets()
        ops.sumsq(self.flat.grad, self.sumsq)      # norm of the (all-reduced) gradient: identical on every rank
        ops.clip_adam(self.flat.param, self.flat.grad, self.flat.m, self.flat.v, self.sumsq, self.max_norm, self.sp[3:5], 0.9, 0.999, 1e-8)
        eng.refresh_weights()                      # packed / transposed weight images for the next step
        return beta0, Bg

    def step_device(self, step, batch, eps):
        """One optimisation step, fully asynchronous (no host sync); statistics stay on the device.

        On the GPU the step is captured ONCE per (shapes, supervised) into a hipGraph (two streams included) and replayed:
        ~2400 kernel launches otherwise cost more host time than the GPU needs to run them.  Call 1 runs eagerly (buffers get
        allocated), call 2 is captured, later calls copy the batch into the captured buffers and replay."""
        m = self.model
        self._sync_step_counter(step)
        beta0 = beta_schedule(step, self.beta)
        B = batch[0].shape[0]
        Bg = B if self.dist is None else self.dist.global_batch(B)
        self._dev_step = step + 1
        self.flat.t += 1
        if not self.use_graph:
            m.engine()
            self._gather_densities(batch, None)
            self._step_body(step, batch, eps)
            m._weights_version = (m._version, m._param_versions())      # refresh_weights() already ran at the end of the step
            return beta0, Bg
        key = (tuple(batch[0].shape), tuple(batch[1].shape), batch[6] is not None, bool(getattr(m.engine().ops, "dw_x6", False)))
        st = self._static.get(key)
        if st is None:                              # first call with these shapes: static input buffers + one eager run
            st = dict(batch=[None if t is None else t.clone() for t in batch], eps=[None if e is None else e.clone() for e in eps], runs=0)
            self._static[key] = st
        # the batch into the captured step's input buffers: one multi-tensor copy per dtype (3 launches instead of 9 in front of every replay)
        groups = {}
        for dst, src in zip(st["batch"] + st["eps"], list(batch) + list(eps)):
            if dst is not None and dst.data_ptr() != src.data_ptr():
                if src.device == dst.device and src.dtype == dst.dtype and src.shape == dst.shape:
                    g = groups.setdefault(dst.dtype, ([], []))
                    g[0].append(dst), g[1].append(src)
                else:
                    dst.copy_(src, non_blocking=True)
        for dsts, srcs in groups.values():
            if len(dsts) > 1:
                torch._foreach_copy_(dsts, srcs, non_blocking=True)
            else:
                dsts[0].copy_(srcs[0], non_blocking=True)
        sbatch, seps = tuple(st["batch"]), tuple(st["eps"])
        m.engine()                                  # (re)builds the engine / weight images outside of any capture
        self._gather_densities(sbatch, st)          # per-batch constants: gathered eagerly into static buffers, outside of the graph
        if key in self._graphs:
            self._graphs[key].replay()
        elif st["runs"] == 0:
            self._step_body(step, sbatch, seps)
        else:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            getattr(m.engine().ops, "begin_capture", lambda: None)()
            try:
                # thread_local: other threads (the RCCL watchdog polls the events of earlier collectives) may keep calling HIP
                # while this thread captures; in the default "global" mode such a call invalidates the capture (flaky)
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._step_body(step, sbatch, seps)
            except Exception as e:                  # e.g. a collective library that cannot be captured: keep training, eagerly
                import warnings
                warnings.warn("hipGraph capture of the training step failed (%s: %s); continuing with eager launches" % (type(e).__name__, e))
                self.use_graph = False
                if self.dist is not None:
                    self.dist.abort_capture()
                torch.cuda.synchronize()
                self._step_body(step, sbatch, seps)
            else:
                self._graphs[key] = g               # the capture itself does not execute: run the step now
                g.replay()
        st["runs"] += 1
        m._weights_version = (m._version, m._param_versions())
        return beta0, Bg

    def train(self, step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, is_supervised=False, y_label=None, eps=None):
        """trainer_gmm.py:220-258 (same argument order; the one-hot arguments are accepted and ignored when the
        integer tensors are given, since convert_to_one_hot(d) carries no more information than d)."""
        batch = self.prepare_batch(d if d is not None else d_oh, r if r is not None else r_oh, n if n is not None else n_oh,
                                   c, r_density, n_density, y_label if is_supervised else None)
        if eps is None:
            eps = self.draw_eps(*batch[0].shape)
        beta0, Bg = self.step_device(step, batch, eps)
        return step + 1, self._tuple8(beta0, Bg, is_supervised)

    @torch.no_grad()
    def evaluate(self, step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, is_supervised=False, y_label=None, eps=None):
        """trainer_gmm.py:261-293: the same forward + losses (still train mode, as the reference), no update."""
        batch = self.prepare_batch(d if d is not None else d_oh, r if r is not None else r_oh, n if n is not None else n_oh,
                                   c, r_density, n_density, y_label if is_supervised else None)
        if eps is None:
            eps = self.draw_eps(*batch[0].shape)
        self.model.engine()
        self._gather_densities(batch, None)
        _, _, _, beta0, Bg = self._forward_losses(step, batch, eps, want_grads=False)
        return self._tuple8(beta0, Bg, is_supervised)

    def grad_norm(self):
        """L2 norm of the last (all-reduced, unclipped) gradient - one host sync."""
        return math.sqrt(float(self.sumsq.item()))


class VAETrainer(GMVAETrainer):
    """Step of the vanilla-VAE sibling (reference ``trainer.py:87-187``): CE + KL-to-N(0,1) + the pairwise regulariser.

    Quirk reproduced on purpose: the reference's ``loss_function`` reads the MODULE-LEVEL ``step`` (trainer.py:57,93), which its
    ``train(step, ...)`` never updates (it increments its own argument, :135,159), so ``beta0`` is 0 for the whole run and the KL term
    never contributes to the loss or the gradient.  ``train`` returns ``(step + 1, (loss, CE_X, CE_R, CE_N, l_r, l_n))`` (:161-162) and
    ``evaluate`` takes no step (:165)."""

    def __init__(self, model, lr=1e-3, beta=0.1, max_norm=1.0, dist_ctx=None):
        super().__init__(model, lr=lr, beta=0.0, max_norm=max_norm, dist_ctx=dist_ctx)
        self.beta_arg = beta                         # accepted like args['beta'] (trainer.py:150); without effect, see above

    def train(self, step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=None):
        step, t8 = super().train(step, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=eps)
        return step, t8[:6]

    @torch.no_grad()
    def evaluate(self, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=None):
        return super().evaluate(0, d_oh, r_oh, n_oh, d, r, n, c, r_density, n_density, eps=eps)[:6]


def convert_to_one_hot(input, dims):
    """trainer_gmm.py:296-303 (kept for callers that still build one-hot tensors; the kernels read indices)."""
    input = input.long()
    oh = torch.zeros(tuple(input.shape) + (dims,), device=input.device)
    return oh.scatter_(-1, input.unsqueeze(-1), 1.0)
