"""Data parallelism for the fused step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in the CPU tests), weights replicated, minibatch rows sharded.

The reference has no distributed code at all - this is new work.  What has to be exchanged:
  * gradients: SUM all-reduce of the flat gradient buffer in two buckets - the decoder-side bucket is started as
    soon as the decoder backward is enqueued and overlaps the 4 encoder backward scans; every loss reduction is
    normalised by the GLOBAL batch, so the sum over ranks IS the global-batch gradient (no 1/n rescale);
  * the pairwise regulariser (trainer_gmm.py:199-217) couples all samples: z[:,0] and the densities are
    all-gathered (2 small vectors per encoder) and each rank evaluates its own rows against the global columns;
  * clip_grad_norm_ uses the norm of the REDUCED gradient, which is identical on every rank;
  * the 8 reported loss numbers: one SUM all-reduce of a 16-float vector.
"""
import torch
import torch.distributed as dist


class DataParallelContext:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._pending = []

    def global_batch(self, local_batch):
        return local_batch * self.world          # equal shards (the loader drops the ragged tail per rank)

    def gather_rows(self, z0, attr):
        """all-gather a float32 [B] and a float64 [B] vector -> ([B*world], [B*world], first global row of this rank)."""
        B = z0.numel()
        z_all = torch.empty(B * self.world, dtype=z0.dtype, device=z0.device)
        a_all = torch.empty(B * self.world, dtype=attr.dtype, device=attr.device)
        dist.all_gather_into_tensor(z_all, z0.contiguous(), group=self.group)
        dist.all_gather_into_tensor(a_all, attr.contiguous(), group=self.group)
        return z_all, a_all, self.rank * B

    def all_reduce_sum(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def start_bucket(self, flat_view):
        """async SUM all-reduce of one gradient bucket; the collective is ordered after the kernels already enqueued
        on the current stream and runs on the communicator's own stream."""
        if flat_view.numel():
            self._pending.append(dist.all_reduce(flat_view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish_buckets(self):
        for w in self._pending:
            w.wait()                              # current stream waits for the collective; the host does not block on NCCL
        self._pending = []


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
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return DataParallelContext(), local
