"""Data parallelism for the fused step: one process per GPU, weights replicated, minibatch rows sharded.  Data plane: RCCL over
xGMI called directly through the C ABI (fn_comm_*, include/fadernets.h) on our own streams - part of the step's hipGraph; control
plane (rendezvous, the 128-byte RCCL id): torch.distributed, gloo.  CPU tensors (the gloo tests of the host logic) use
torch.distributed collectives.

The reference has no distributed code at all - this is new work.  What has to be exchanged:
  * gradients: SUM all-reduce of the flat gradient buffer in two buckets - the decoder-side bucket is started as
    soon as the decoder backward is enqueued and overlaps the 4 encoder backward scans; every loss reduction is
    normalised by the GLOBAL batch, so the sum over ranks IS the global-batch gradient (no 1/n rescale);
  * the pairwise regulariser (trainer_gmm.py:199-217) couples all samples: z[:,0] and the densities are
    all-gathered (2 small vectors per encoder) and each rank evaluates its own rows against the global columns;
  * clip_grad_norm_ uses the norm of the REDUCED gradient, which is identical on every rank;
  * the 8 reported loss numbers: one SUM all-reduce of a 16-float vector.
"""
import ctypes as C

import torch
import torch.distributed as dist


class DirectRccl:
    """RCCL through the C ABI (fn_comm_*, csrc/comm.hip): collectives are plain work on OUR streams - capturable into the hipGraph of
    the step, no process-group watchdog thread.  torch.distributed is only the rendezvous (its store carries the 128-byte unique id)."""

    def __init__(self, world, rank, device, group=None):
        from . import _lib
        self.lib = _lib.load()
        self._check = _lib.check
        self.world, self.rank, self.device = world, rank, torch.device(device)
        buf = C.create_string_buffer(_lib.FN_COMM_ID_BYTES)
        if rank == 0:
            self._check(self.lib.fn_comm_unique_id(buf), "fn_comm_unique_id")
        box = [buf.raw if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(box, src=0, group=group)           # out-of-band: pickled bytes through the process group
        self.handle = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.fn_comm_init(C.byref(self.handle), world, rank, C.create_string_buffer(box[0], _lib.FN_COMM_ID_BYTES)), "fn_comm_init")
        self.stream = torch.cuda.Stream(device=self.device)               # gradient buckets run here, beside the backward kernels

    def _st(self, stream=None):
        return C.c_void_p((stream or torch.cuda.current_stream(self.device)).cuda_stream)

    def all_reduce_(self, t, stream=None):
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            raise RuntimeError("DirectRccl.all_reduce_: contiguous fp32 device tensor expected")
        self._check(self.lib.fn_comm_all_reduce_f32(self.handle, C.c_void_p(t.data_ptr()), t.numel(), self._st(stream)), "fn_comm_all_reduce_f32")

    def all_gather(self, out, t, stream=None):
        if not (t.is_contiguous() and out.is_contiguous() and out.numel() * out.element_size() == self.world * t.numel() * t.element_size()):
            raise RuntimeError("DirectRccl.all_gather: out must hold world x the (contiguous) input")
        self._check(self.lib.fn_comm_all_gather(self.handle, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), t.numel() * t.element_size(),
                                                self._st(stream)), "fn_comm_all_gather")

    def ranks(self):
        """(number of ranks, this rank) as RCCL itself reports them for the communicator (fn_comm_count / fn_comm_rank)"""
        n, r = C.c_int(-1), C.c_int(-1)
        self._check(self.lib.fn_comm_count(self.handle, C.byref(n)), "fn_comm_count")
        self._check(self.lib.fn_comm_rank(self.handle, C.byref(r)), "fn_comm_rank")
        return int(n.value), int(r.value)

    def close(self):
        if self.handle:
            self.lib.fn_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class DataParallelContext:
    """Collectives of the data-parallel step.  Device tensors go through DirectRccl (created on first use); CPU tensors (the gloo /
    FakeOps tests of the host logic) through torch.distributed."""

    def __init__(self, group=None, direct=True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.want_direct = direct
        self.rccl = None
        self._pending = []
        self._forked = False
        self.timing = None                       # bench.py: dict of (start, end) event pairs per bucket

    def _direct(self, t):
        if not (self.want_direct and t.is_cuda):
            return None
        if self.rccl is None:
            self.rccl = DirectRccl(self.world, self.rank, t.device, self.group)
        return self.rccl

    def global_batch(self, local_batch):
        return local_batch * self.world          # equal shards (the loader drops the ragged tail per rank)

    def gather(self, t, out=None):
        """all-gather of one contiguous vector -> [world * n] (into `out` when given: a captured graph reads static buffers)"""
        if out is None:
            out = torch.empty(t.numel() * self.world, dtype=t.dtype, device=t.device)
        r = self._direct(t)
        if r is not None:
            r.all_gather(out, t.contiguous())
        else:
            dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def gather_rows(self, z0, attr, attr_all=None):
        """all-gather a float32 [B] vector (and, unless the caller already holds it, a float64 [B] vector)
        -> ([B*world], [B*world], first global row of this rank)."""
        B = z0.numel()
        return self.gather(z0), (attr_all if attr_all is not None else self.gather(attr)), self.rank * B

    def all_reduce_sum(self, t):
        r = self._direct(t)
        if r is not None:
            r.all_reduce_(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def start_bucket(self, flat_view, tag=None):
        """SUM all-reduce of one gradient bucket, ordered after the kernels already enqueued on the current stream; it runs on the
        communicator's stream, beside whatever the current stream does next (finish_buckets joins)."""
        if not flat_view.numel():
            return
        r = self._direct(flat_view)
        if r is None:
            self._pending.append(dist.all_reduce(flat_view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        cur = torch.cuda.current_stream(flat_view.device)
        r.stream.wait_stream(cur)
        ev = None
        if self.timing is not None and not torch.cuda.is_current_stream_capturing():
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(r.stream)
        r.all_reduce_(flat_view, stream=r.stream)
        if ev is not None:
            ev[1].record(r.stream)
            self.timing.setdefault(tag or "bucket%d" % len(self.timing), []).append(ev)
        self._forked = True

    def finish_buckets(self):
        """Join of the gradient buckets, called in front of clip + Adam.  For the direct RCCL path this is also the guarantee that a
        collective's kernels have LEFT the compute units before the next weight-stationary launch asks for all 256 of them: an event
        recorded on the communicator's stream behind the last all-reduce is waited on by the step's stream (``wait_stream``), and
        everything the step enqueues afterwards - clip + Adam, the weight images, the NEXT step's encoder forward scan - is ordered
        behind it; inside a captured step the same edge is a graph dependency."""
        for w in self._pending:
            w.wait()                              # current stream waits for the collective; the host does not block on NCCL
        self._pending = []
        if self._forked:
            cur = torch.cuda.current_stream(self.rccl.device)
            timed = self.timing is not None and not torch.cuda.is_current_stream_capturing()
            if timed:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(cur)
            cur.wait_stream(self.rccl.stream)
            if timed:
                ev[1].record(cur)
                self.timing.setdefault("exposed", []).append(ev)       # how long the step's stream stood still for the last bucket
            self._forked = False

    def abort_capture(self):
        """a hipGraph capture that contained collectives was abandoned: forget its half-built fork / work handles"""
        self._pending = []
        self._forked = False


def init_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torch.distributed.run)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get("FN_FORCE_DIST", "0") != "1":     # FN_FORCE_DIST=1: exercise the collectives with 1 rank
        return None, 0
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        # the process group is only the rendezvous / control plane (the 128-byte RCCL id, barriers, host-side scalars): gloo.  The
        # data plane is RCCL called directly (DirectRccl).  "nccl" still works (then torch's own RCCL communicator idles beside ours).
        backend = "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return DataParallelContext(), local
