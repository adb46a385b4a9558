"""The N > 1 path on CPU: two gloo ranks shard the families exactly like bench.py, each produces the pair
{sum lnL, rejects} for its shard, one all-reduce (gloo here, RCCL on the GPUs) combines them, and the
result equals the single-process score.  There is no GPU in this container, so the per-shard pair comes
from the CPU oracle -- it stands in for cafe_score_partial's output; the sharding, the reduction and
cafe_finish_partial's rule are the code under test."""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _finish(pair):            # cafe_finish_partial (cafexp_amd/csrc/cafe_ctx.hip)
    return math.inf if pair[1] > 0 else -pair[0]


def _worker(rank, world, port, case, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dataclasses
    from cafexp_amd import problem as P, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pb, _ = synth.make_problem(n_taxa=10, n_families=101, max_count=40, lam_sim=0.004, seed=21, root_cap=30)
    if case == "gamma":
        probs, mult = O.discrete_gamma(3, 1.5)
        pr = P.Params(lambdas=np.array([0.004]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
    elif case == "reject":
        pr = P.Params(lambdas=np.array([-0.1]), prior=P.prior_uniform(pb.max_root_family_size))
    else:
        pr = P.Params(lambdas=np.array([0.004]), prior=P.prior_uniform(pb.max_root_family_size))
    lo, hi = P.shard_families(pb.n_families, world, rank)
    shard = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[lo:hi]), family_ids=pb.family_ids[lo:hi])
    v = O.score(shard, pr)
    pair = torch.tensor([0.0, 1.0] if math.isinf(v) else [-v, 0.0], dtype=torch.float64)
    dist.all_reduce(pair)
    if rank == 0:
        whole = O.score(pb, pr)
        out.put((_finish(pair.tolist()), whole, hi - lo))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["base", "gamma", "reject"])
def test_two_rank_shards_allreduce_to_the_whole(case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + {"base": 0, "gamma": 1, "reject": 2}[case]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, whole, n0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if case == "reject":
        assert got == whole == math.inf
    else:
        assert abs(got - whole) / whole < 1e-13
