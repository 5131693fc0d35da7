"""Data-parallel path through real RCCL on the GPU box (called directly through the C ABI, fn_comm_*), the HIP kernels underneath.

* one rank (any GPU box): the step with the collectives in place - all-reduce buckets beside the weight-stationary scans, all-gather
  of the regulariser inputs, all-reduce of the statistics - must be BIT-identical to the plain single-GPU step, captured into ONE
  hipGraph (the default) and with eager launches (FN_DP_GRAPH=0), with the control plane on gloo (default) or on torch's nccl backend;
* two ranks (skipped below 2 visible GPUs): half the batch each == the single-process step on the full batch, as the gloo test
  checks on the CPU backend;
* bench.py launches its own ranks when asked for --gpus N outside a launcher.
Every case runs in spawned processes so that the process group never leaks into the pytest process.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from test_parallel_gloo import HERE, ROOT, _free_port


def _rccl_worker(rank, world, port, out_dir, dp_graph="1", backend=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      FN_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", FN_DP_GRAPH=dp_graph)
    from helpers import batch_of, load_golden, make_model, sd_from
    from mfn_import import load_package
    pkg = load_package()
    from music_fader_nets_amd import parallel
    ctx, local = parallel.init_from_env(backend)
    assert ctx is not None and ctx.world == world
    dev = "cuda:%d" % local
    gold = load_golden("small")
    b = batch_of(gold)
    B = b["d"].shape[0]
    lo, hi = rank * B // world, (rank + 1) * B // world
    m = make_model(64, 32, sd_from(gold, "w0/"), device=dev)
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2, dist_ctx=ctx)
    tuples, step = [], 19999
    for it in range(4):                                       # eager, capture, replay, replay
        torch.manual_seed(99 + it)
        eps_r, eps_n = torch.randn(B, 32), torch.randn(B, 32)
        step, tup = tr.train(step, None, None, None, b["d"][lo:hi], b["r"][lo:hi], b["n"][lo:hi], b["c"][lo:hi],
                             b["r_density"][lo:hi], b["n_density"][lo:hi], eps=(eps_r[lo:hi].contiguous().to(dev), eps_n[lo:hi].contiguous().to(dev)))
        tuples.append(tup)
    torch.save(dict(tuples=tuples, flat=tr.flat.param.cpu(), gn=tr.grad_norm(), graphs=len(tr._graphs), use_graph=tr.use_graph,
                    direct=ctx.rccl is not None), os.path.join(out_dir, "r%d.pt" % rank))
    if ctx.rccl is not None:
        ctx.rccl.close()
    assert not m.engine().ops.gru_sync_error()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _single_process_run(steps, dev="cuda:0"):
    sys.path.insert(0, HERE)
    from helpers import batch_of, load_golden, make_model, sd_from
    from mfn_import import load_package
    pkg = load_package()
    gold = load_golden("small")
    b = batch_of(gold)
    m = make_model(64, 32, sd_from(gold, "w0/"), device=dev)
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
    tuples, step = [], 19999
    for it in range(steps):
        torch.manual_seed(99 + it)
        step, tup = tr.train(step, None, None, None, b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
        tuples.append(tup)
    return tuples, tr.flat.param.cpu(), tr.grad_norm(), gold


@pytest.mark.gpu
@pytest.mark.parametrize("dp_graph,backend", [("1", None), ("0", None), ("1", "nccl")])
def test_single_rank_rccl_step_is_bit_identical(tmp_path, dp_graph, backend):
    mp.start_processes(_rccl_worker, args=(1, _free_port(), str(tmp_path), dp_graph, backend), nprocs=1, join=True, start_method="spawn")
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"), weights_only=False)
    tuples, flat, gn, gold = _single_process_run(4)
    np.testing.assert_array_equal(np.asarray(r0["tuples"]), np.asarray(tuples))
    assert torch.equal(r0["flat"], flat)
    np.testing.assert_allclose(tuples[:3], gold["train_tuples"], rtol=5e-4)       # and both are the reference's train()
    assert r0["direct"]                                                           # the collectives went through fn_comm_* (RCCL)
    if dp_graph == "0":
        assert not r0["use_graph"] and r0["graphs"] == 0
    else:
        assert r0["use_graph"] and r0["graphs"] == 1                              # the data-parallel step IS one hipGraph


@pytest.mark.gpu
def test_two_rank_rccl_step_equals_single_process(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 visible GPUs (the driver's multi-GPU node); the 1-GPU box runs the single-rank RCCL test")
    mp.start_processes(_rccl_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0 = torch.load(os.path.join(tmp_path, "r0.pt"), weights_only=False)
    r1 = torch.load(os.path.join(tmp_path, "r1.pt"), weights_only=False)
    assert torch.equal(r0["flat"], r1["flat"])
    np.testing.assert_array_equal(np.asarray(r0["tuples"]), np.asarray(r1["tuples"]))
    tuples, flat, gn, gold = _single_process_run(4)
    np.testing.assert_allclose(r0["tuples"], tuples, rtol=2e-5)
    np.testing.assert_allclose(r0["gn"], gn, rtol=1e-4)
    diff = (r0["flat"] - flat).abs()
    assert float(diff.max()) <= 4.1e-3 and float((diff > 1e-5).float().mean()) < 2e-3    # Adam noise on ~zero gradients only


def _run_bench(extra, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2])
def test_bench_launches_its_own_ranks(n):
    """`python bench.py --gpus N` with no launcher environment prints ONE JSON line with n_gpus = N (N = 1 also runs with the
    collectives forced on, FN_FORCE_DIST=1, i.e. through RCCL)."""
    if torch.cuda.device_count() < n:
        pytest.skip("needs %d visible GPUs" % n)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if n == 1:
        env["FN_FORCE_DIST"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["config"]["global_batch"] == 256 * n and out["value"] > 0
    assert 0 < out["roofline"]["frac"] < 1 and out["roofline_all"]


def test_bench_respawns_under_the_launcher_without_gpus():
    """CPU container: `--gpus 2` with no WORLD_SIZE must reach the per-rank "needs an MI355X" exit through torch.distributed.run,
    not die on a WORLD_SIZE mismatch before any rank starts (round-1 behaviour)."""
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode != 0
    assert "needs an MI355X" in p.stderr and "but WORLD_SIZE" not in p.stderr, p.stderr[-2000:]
