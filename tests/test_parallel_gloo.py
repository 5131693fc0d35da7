"""Data-parallel path with world_size 2 over gloo on the CPU (kernels replaced by their documented semantics,
tests/fake_ops.py): two ranks with half the batch each must reproduce the single-process step on the full batch -
gradients (SUM all-reduce of global-batch-normalised terms), the pairwise regulariser across the shard boundary,
the clip norm, the 8 reported numbers and the updated weights."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from fake_ops import FakeOps
    from helpers import batch_of, load_golden, make_model, sd_from
    from mfn_import import load_package
    pkg = load_package()
    from music_fader_nets_amd import parallel
    ctx, _ = parallel.init_from_env("gloo")
    gold = load_golden("small")
    b = batch_of(gold)
    B = b["d"].shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world
    m = make_model(64, 32, sd_from(gold, "w0/"), ops=FakeOps())
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2, dist_ctx=ctx)
    tuples = []
    step = 19999
    for it in range(2):
        torch.manual_seed(99 + it)
        eps_r, eps_n = torch.randn(B, 32), torch.randn(B, 32)          # global draw, sliced per rank
        step, tup = tr.train(step, None, None, None, b["d"][lo:hi], b["r"][lo:hi], b["n"][lo:hi], b["c"][lo:hi],
                             b["r_density"][lo:hi], b["n_density"][lo:hi], eps=(eps_r[lo:hi].contiguous(), eps_n[lo:hi].contiguous()))
        tuples.append(tup)
    torch.save(dict(tuples=tuples, flat=tr.flat.param.clone(), gn=tr.grad_norm(), names=tr.flat.names), os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process(tmp_path):
    sys.path.insert(0, HERE)
    from fake_ops import FakeOps
    from helpers import batch_of, load_golden, make_model, sd_from
    from mfn_import import load_package
    pkg = load_package()
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"), weights_only=False)
    # both ranks hold identical replicas and report identical numbers
    assert torch.equal(r0["flat"], r1["flat"])
    np.testing.assert_allclose(r0["tuples"], r1["tuples"], rtol=0, atol=0)
    # single process, full batch
    gold = load_golden("small")
    b = batch_of(gold)
    m = make_model(64, 32, sd_from(gold, "w0/"), ops=FakeOps())
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
    step = 19999
    for it in range(2):
        torch.manual_seed(99 + it)
        step, tup = tr.train(step, None, None, None, b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
        np.testing.assert_allclose(r0["tuples"][it], tup, rtol=2e-5)
        np.testing.assert_allclose(tup, gold["train_tuples"][it], rtol=3e-4)       # = the reference's own train()
    np.testing.assert_allclose(r0["gn"], tr.grad_norm(), rtol=1e-4)
    diff = (r0["flat"] - tr.flat.param).abs()
    assert float(diff.max()) <= 2.1e-3 and float((diff > 1e-5).float().mean()) < 2e-3    # Adam noise on ~zero gradients only
