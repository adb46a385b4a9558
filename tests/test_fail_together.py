"""A scorer call on a context with a communicator is collective (one all-reduce of {sum lnL, rejects} inside cafe_score,
SURVEY 8e); the reference has no counterpart (OpenMP over families, base_model.cpp:81-107).  These tests pin what the
library promises about it (include/cafe_mi355x.h, "Ranks fail TOGETHER"):
  * a rank whose own enqueue fails in the middle of a call still enters the all-reduce, with rejects = NaN: it returns its
    own error promptly, every other rank returns "another rank ... failed", nobody waits;
  * the contexts stay usable afterwards;
  * two ranks sharing GPU 0 (gloo carries the pair, as `bench.py --backend gloo` does) with REAL device partials from
    cafe_score_partial over the library's own shard plan reproduce the single-context value -- and fail together when one
    of them fails.
RCCL refuses two ranks on one device, so the N = 2 collective itself cannot run on a one-GPU box; what runs here is every
line a rank executes, with a world of one for RCCL and a world of two over gloo."""
import math
import os
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cafexp_amd import problem as P, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    from cafexp_amd import capi as C
    C.load()
    return C


def _case(oracle):
    pb, _ = synth.make_problem(n_taxa=14, n_families=900, max_count=50, lam_sim=0.004, seed=9, root_cap=35)
    probs, mult = oracle.discrete_gamma(3, 1.5)
    return pb, P.Params(lambdas=np.array([0.004]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)


def test_a_failing_rank_enters_the_collective_and_everyone_returns(capi, oracle):
    pb, pr = _case(oracle)
    ctx = capi.Context(pb, max_categories=3)
    want = ctx.score(pr, alpha=1.5)
    ctx.comm_attach(capi.comm_unique_id(), 1, 0)
    assert ctx.score(pr, alpha=1.5) == want
    ctx.debug_fail_next(2)                                   # the second call from now fails behind its K1 launch
    assert ctx.score(pr, alpha=1.5) == want
    t0 = time.perf_counter()
    with pytest.raises(capi.CafeError, match=r"injected failure.*told through the all-reduce"):
        ctx.score(pr, alpha=1.5)
    assert time.perf_counter() - t0 < 5.0
    with pytest.raises(capi.CafeError):                      # no results of a failed call
        ctx.family_results(3)
    assert ctx.score(pr, alpha=1.5) == want                  # communicator and context are intact
    ctx.comm_detach()
    # without a communicator the same failure is a plain error, and the partial API marks its pair
    ctx.debug_fail_next(1)
    with pytest.raises(capi.CafeError, match="injected failure"):
        ctx.score(pr, alpha=1.5)
    buf = torch.zeros(2, dtype=torch.float64, device="cuda")
    ctx.debug_fail_next(1)
    with pytest.raises(capi.CafeError, match="injected failure"):
        ctx.score_partial(pr, buf.data_ptr(), torch.cuda.current_stream().cuda_stream, alpha=1.5)
    torch.cuda.synchronize()
    pair = buf.cpu().numpy()
    assert pair[0] == 0.0 and math.isnan(pair[1]) and math.isnan(ctx.finish(pair))
    ctx.score_partial(pr, buf.data_ptr(), torch.cuda.current_stream().cuda_stream, alpha=1.5)
    torch.cuda.synchronize()
    assert ctx.finish(buf.cpu().numpy()) == want
    ctx.close()


def test_a_failing_shard_of_the_in_process_scorer(capi, oracle):
    pb, pr = _case(oracle)
    sh = capi.Sharded(pb, [0], max_categories=3)
    want = sh.score(pr, alpha=1.5)
    sh.shard(0).debug_fail_next(1)
    t0 = time.perf_counter()
    with pytest.raises(capi.CafeError, match=r"shard 0: injected failure"):
        sh.score(pr, alpha=1.5)
    assert time.perf_counter() - t0 < 5.0
    assert sh.score(pr, alpha=1.5) == want
    sh.close()


def test_the_wait_has_a_deadline(capi, oracle, monkeypatch):
    """CAFE_COMM_TIMEOUT_S (read at attach): with a deadline the wait polls the stream and the communicator's asynchronous
    error state instead of blocking in hipStreamSynchronize; same value, and <= 0 restores the blocking wait."""
    pb, pr = _case(oracle)
    ctx = capi.Context(pb, max_categories=3)
    want = ctx.score(pr, alpha=1.5)
    for t in ("30", "0"):
        monkeypatch.setenv("CAFE_COMM_TIMEOUT_S", t)
        ctx.comm_attach(capi.comm_unique_id(), 1, 0)
        assert ctx.score(pr, alpha=1.5) == want
        ctx.comm_detach()
    ctx.close()


# ------------------------------------------------------------------ two ranks on GPU 0, gloo carries the pair
def _rank(rank, world, port, fail_rank, out):
    sys.path.insert(0, ROOT)
    import dataclasses
    from cafexp_amd import capi
    from cafexp_amd.gamma_rates import discrete_gamma
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    pb, _ = synth.make_problem(n_taxa=16, n_families=1500, max_count=60, lam_sim=0.004, seed=31, root_cap=40)
    probs, mult = discrete_gamma(4, 2.0)
    pr = P.Params(lambdas=np.array([0.004]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
    mine = capi.shard_plan(pb, world, 4)[rank]               # the library's plan, as bench.py's ranks take it
    shard = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[mine]), family_ids=[pb.family_ids[i] for i in mine])
    ctx = capi.Context(shard, max_categories=4, device=0)
    buf = torch.zeros(2, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        try:
            ctx.score_partial(pr, buf.data_ptr(), stream.cuda_stream, alpha=2.0)
            err = None
        except capi.CafeError as e:                          # the pair now holds {0, NaN}: the reduction tells the others
            err = e
        pair = buf.cpu()
        dist.all_reduce(pair)
        return ctx.finish(pair.numpy()), err

    v, err = step()
    res = {"rank": rank, "n": len(mine), "value": v, "err": None if err is None else str(err)}
    if fail_rank >= 0:
        if rank == fail_rank:
            ctx.debug_fail_next(1)
        t0 = time.perf_counter()
        v2, err2 = step()
        res.update(value2=v2, err2=None if err2 is None else str(err2), seconds2=time.perf_counter() - t0)
        v3, err3 = step()                                    # and the call after it is whole again
        res.update(value3=v3, err3=None if err3 is None else str(err3))
    if rank == 0:
        whole = capi.Context(pb, max_categories=4, device=0)
        res["whole"] = whole.score(pr, alpha=2.0)
        whole.close()
    out.put(res)
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def _run_two(fail_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_rank, args=(r, 2, port, fail_rank, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300), q.get(timeout=300)], key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_ranks_on_one_gpu_with_real_device_partials():
    r0, r1 = _run_two(-1)
    assert r0["err"] is None and r1["err"] is None
    assert r0["n"] + r1["n"] == 1500 and min(r0["n"], r1["n"]) > 100
    assert r0["value"] == r1["value"] and math.isfinite(r0["value"])
    assert abs(r0["value"] - r0["whole"]) <= 1e-13 * abs(r0["whole"])


def test_two_ranks_on_one_gpu_fail_together():
    r0, r1 = _run_two(1)
    assert r0["value"] == r1["value"] and abs(r0["value"] - r0["whole"]) <= 1e-13 * abs(r0["whole"])
    assert "injected failure" in r1["err2"] and r0["err2"] is None
    assert math.isnan(r0["value2"]) and math.isnan(r1["value2"])          # BOTH ranks learn of it from the reduced pair
    assert r0["seconds2"] < 5.0 and r1["seconds2"] < 5.0
    assert r0["err3"] is None and r1["err3"] is None and r0["value3"] == r0["value"] == r1["value3"]
