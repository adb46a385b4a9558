"""The N > 1 path on CPU.

(1) cafe_shard_plan -- the library's own family partition (host code of libcafe_mi355x.so, no GPU needed): a
    partition, deterministic, balanced by distinct subtree patterns.
(2) Two gloo ranks shard the families with that plan exactly like bench.py, each produces the pair {sum lnL, rejects}
    for its shard, one all-reduce (gloo here; on the GPUs the library's own ncclAllReduce inside cafe_score) combines
    them, the library's cafe_finish_partial turns the pair into the scorer value, and the result equals the
    single-process score.  There is no GPU in this container, so the per-shard pair comes from the CPU oracle -- it
    stands in for cafe_score_partial's output; the plan, the reduction and cafe_finish_partial are the code under test.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _distinct_columns(pb, fams, tile=1):
    """Distinct leaf-count patterns under every interior non-root node (padded to the device's column tile), summed: the
    device's unit of work."""
    C = np.ascontiguousarray(pb.counts[fams])
    n = pb.n_nodes
    children = [[] for _ in range(n)]
    for v in range(n):
        if pb.parent[v] >= 0:
            children[int(pb.parent[v])].append(v)
    leafset = [None] * n
    total = 0
    for v in range(n):
        leafset[v] = [int(pb.leaf_taxon[v])] if pb.leaf_taxon[v] >= 0 else [t for c in children[v] for t in leafset[c]]
        if pb.leaf_taxon[v] < 0 and pb.parent[v] >= 0:
            total += -(-len(np.unique(C[:, leafset[v]], axis=0)) // tile) * tile
    return total


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shard_plan_is_a_balanced_partition(world):
    sys.path.insert(0, ROOT)
    from cafexp_amd import capi, synth
    pb, _ = synth.make_problem(n_taxa=24, n_families=4000, max_count=80, lam_sim=0.004, seed=5, root_cap=50)
    plan = capi.shard_plan(pb, world)
    again = capi.shard_plan(pb, world)
    assert len(plan) == world
    assert all(len(p) > 0 for p in plan)
    assert np.array_equal(np.sort(np.concatenate(plan)), np.arange(pb.n_families))      # every family exactly once
    assert all(np.array_equal(a, b) for a, b in zip(plan, again))                        # every rank derives the same plan
    if world > 1:
        cost = np.array([_distinct_columns(pb, p, tile=128) for p in plan], dtype=float)   # K2's tiles are 128 columns wide
        assert cost.max() / cost.min() < 1.25, cost                                       # by (predicted time over) distinct patterns, not family count
        assert max(len(p) for p in plan) > min(len(p) for p in plan)                      # ... which is not an even family count
        # look-alikes share a shard: the shards together hold fewer distinct columns than contiguous blocks of the table
        blocks = np.array_split(np.arange(pb.n_families), world)
        assert cost.sum() < sum(_distinct_columns(pb, b) for b in blocks)


def test_shard_plan_of_identical_families_is_even():
    """64 copies of one family: all columns are shared, what is left is the per-family term -- the shards must come out
    about equal (a start vector that counted first-seen patterns only gave [1, 1, 17, 45])."""
    sys.path.insert(0, ROOT)
    import dataclasses
    from cafexp_amd import capi, synth
    pb, _ = synth.make_problem(n_taxa=12, n_families=64, max_count=60, lam_sim=0.004, seed=3, root_cap=40)
    same = dataclasses.replace(pb, counts=np.repeat(pb.counts[:1], 64, axis=0).copy())
    sizes = [len(x) for x in capi.shard_plan(same, 4)]
    assert sum(sizes) == 64 and min(sizes) >= 10 and max(sizes) <= 22, sizes


def test_shard_plan_rejects_more_shards_than_families():
    sys.path.insert(0, ROOT)
    from cafexp_amd import capi, synth
    pb, _ = synth.make_problem(n_taxa=6, n_families=5, max_count=20, lam_sim=0.004, seed=2, root_cap=10)
    with pytest.raises(capi.CafeError):
        capi.shard_plan(pb, 6)


def _worker(rank, world, port, case, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dataclasses
    from cafexp_amd import capi, problem as P, synth
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
    mine = capi.shard_plan(pb, world)[rank]                 # bench.py's sharding: the library's plan
    shard = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[mine]), family_ids=[pb.family_ids[i] for i in mine])
    v = O.score(shard, pr)
    pair = torch.tensor([0.0, 1.0] if math.isinf(v) else [-v, 0.0], dtype=torch.float64)
    dist.all_reduce(pair)
    if rank == 0:
        whole = O.score(pb, pr)
        lib = capi.load()
        a = np.ascontiguousarray(pair.numpy(), dtype=np.float64)
        import ctypes
        got = lib.cafe_finish_partial(a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))     # the library's own rule
        out.put((got, whole, len(mine)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["base", "gamma", "reject"])
def test_two_rank_shards_allreduce_to_the_whole(case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:                              # a port nobody holds right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
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


def test_bench_parent_starts_its_ranks_without_touching_the_gpu():
    """`python bench.py --gpus N` run plainly must launch N ranks itself (the driver invokes it that way) and must do so
    before importing torch / touching the GPU: in a FRESH interpreter, import bench, call its launcher with subprocess.run
    replaced, and check that torch was never imported and what the command line is (the launcher owns the rendezvous port)."""
    import json
    import subprocess
    code = """
import json, subprocess, sys
sys.path.insert(0, %r)
import bench
seen = {}
class R:
    returncode = 0
    stdout = '{"metric": "x", "n_gpus": 4}\\n'
def fake_run(cmd, **kw):
    seen["cmd"] = cmd
    seen["env"] = kw.get("env", {}).get("HSA_ENABLE_IPC_MODE_LEGACY")
    return R()
subprocess.run = fake_run
rc = bench.launch_ranks(type("A", (), {"gpus": 4})())
assert rc == 0
loaded = [m for m in ("torch", "cafexp_amd.capi", "numpy") if m in sys.modules]
print(json.dumps({"cmd": seen["cmd"], "env": seen["env"], "loaded": loaded}))
""" % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr[-1500:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{") and "cmd" in ln][-1])
    assert d["loaded"] == []                                  # nothing that could touch the GPU was even imported
    cmd = d["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    assert any(os.path.basename(c) == "bench.py" for c in cmd) and d["env"] == "0"


def test_scaled_plan_moves_the_cuts_towards_the_shards_that_ran_long():
    """cafe_shard_plan_scaled: families of a shard that took longer than the mean weigh more in the next plan, so that shard
    gets fewer of them (and the plan stays a partition in the same family order); a scale of one everywhere reproduces
    cafe_shard_plan; scales outside (0.1, 10) are refused."""
    from cafexp_amd import capi, synth
    pb, _ = synth.make_problem(n_taxa=12, n_families=3000, max_count=90, lam_sim=0.003, seed=4, root_cap=50)
    plan = capi.shard_plan(pb, 4, 2)
    same = capi.shard_plan(pb, 4, 2, family_scale=np.ones(pb.n_families))
    assert all(np.array_equal(a, b) for a, b in zip(plan, same))
    again = capi.rebalanced_plan(pb, plan, [1.0, 1.0, 1.0, 1.3], 2)
    assert np.array_equal(np.concatenate(again), np.concatenate(plan))            # same order, other cuts
    assert len(again[3]) < len(plan[3]) and sum(len(x) for x in again) == pb.n_families
    assert len(again[0]) >= len(plan[0])
    with pytest.raises(capi.CafeError):
        capi.shard_plan(pb, 4, 2, family_scale=np.full(pb.n_families, 20.0))
