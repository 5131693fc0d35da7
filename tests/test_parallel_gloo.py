"""Data-parallel path with world_size 2 and 4 over gloo on the CPU (kernels replaced by their documented semantics,
tests/fake_ops.py): the ranks with their rows of the batch must reproduce the single-process step on the full batch -
gradients (SUM all-reduce of global-batch-normalised terms), the pairwise regulariser across the shard boundaries,
the clip norm, the 8 reported numbers and the updated weights; world 4 also covers the supervised branch, a ragged last batch
(train.RankShard drops the remainder), the trainer's own noise draw (global draw, sliced per rank) and the replica sync."""
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


# ---- world 4: unsupervised + supervised steps over a loader whose last batch is ragged, eps drawn by the trainer itself -------------------------
def _w4_batches():
    sys.path.insert(0, ROOT)
    from mfn_import import load_package
    load_package()
    from music_fader_nets_amd.synth import synth_batch
    rng = np.random.RandomState(5)
    out = []
    for B in (8, 6):                                                   # 6 rows over 4 ranks: one row each, two rows dropped
        b = synth_batch(rng, B, 20, 8)
        a = rng.randint(0, 2, size=B)
        out.append(tuple(torch.from_numpy(np.asarray(b[k])) for k in ("d", "r", "n", "c")) + (torch.from_numpy(a), torch.from_numpy(b["r_density"]),
                                                                                              torch.from_numpy(b["n_density"])))
    return out


def _run_w4(tr, loader, supervised):
    tuples, step = [], 19999
    torch.manual_seed(77)                                              # lockstep generators (train.py seeds every rank alike)
    for d, r, n, c, a, rd, nd in loader:
        kw = dict(is_supervised=True, y_label=a) if supervised else {}
        step, tup = tr.train(step, None, None, None, d, r, n, c, rd, nd, **kw)      # eps=None: GMVAETrainer.draw_eps
        tuples.append(tup)
    return tuples


def _worker4(rank, world, port, out_dir, supervised):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from fake_ops import FakeOps
    from helpers import make_model
    from mfn_import import load_package
    pkg = load_package()
    from music_fader_nets_amd import parallel
    from music_fader_nets_amd.train import RankShard, sync_replicas
    ctx, _ = parallel.init_from_env("gloo")
    m = make_model(64, 32, ops=FakeOps(), seed=1234 + rank)            # replicas that do NOT agree at first (a rank that missed the checkpoint)
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2, dist_ctx=ctx)
    sync_replicas(tr, ctx)                                             # ... start from rank 0's weights
    w0 = tr.flat.param.clone()
    shard = RankShard(_w4_batches(), rank, world)
    tuples = _run_w4(tr, shard, supervised)
    torch.save(dict(tuples=tuples, flat=tr.flat.param.clone(), w0=w0, gn=tr.grad_norm()), os.path.join(out_dir, "w4_r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("supervised", [False, True])
def test_four_ranks_ragged_last_batch_equals_single_process(tmp_path, supervised):
    sys.path.insert(0, HERE)
    from fake_ops import FakeOps
    from helpers import make_model
    from mfn_import import load_package
    pkg = load_package()
    mp.start_processes(_worker4, args=(4, _free_port(), str(tmp_path), supervised), nprocs=4, join=True, start_method="spawn")
    rs = [torch.load(os.path.join(tmp_path, "w4_r%d.pt" % r), weights_only=False) for r in range(4)]
    m = make_model(64, 32, ops=FakeOps(), seed=1234)
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
    for r in rs:                                                       # replica sync: everyone started from rank 0's (= seed 1234) weights
        assert torch.equal(r["w0"], tr.flat.param)
        assert torch.equal(r["flat"], rs[0]["flat"])
        np.testing.assert_allclose(r["tuples"], rs[0]["tuples"], rtol=0, atol=0)
    # single process on the rows the four shards cover: all 8 of the first batch, the first 4 of the ragged one
    full = [tuple(t[:(len(t) // 4) * 4] for t in x) for x in _w4_batches()]
    tuples = _run_w4(tr, full, supervised)
    assert len(rs[0]["tuples"]) == len(tuples) == 2
    np.testing.assert_allclose(rs[0]["tuples"], tuples, rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(rs[0]["gn"], tr.grad_norm(), rtol=1e-4)
    diff = (rs[0]["flat"] - tr.flat.param).abs()
    assert float(diff.max()) <= 2.1e-3 and float((diff > 1e-5).float().mean()) < 2e-3
