#!/usr/bin/env python3
"""bench.py -- gene-family likelihoods / second per optimizer_scorer call on MI355X.

One "step" = one scorer call (model::infer_family_likelihoods through the C ABI): build every
transition matrix, prune every family x gamma category, reduce to -lnL, all-reduce the shard sums
(N > 1) and bring the value to the host -- with the families already resident in HBM.

Workload (BASELINE.json configs[3], the configuration the metric is quoted on; it fits one GPU):
synthetic 50 000 families / 100 taxa / max family size 600 (M=720, R=750, matrix order 751), gamma
model K=8, from cafexp_amd/synth.py (seed 20251004), scored at lambda=0.002, alpha=2.0.  Strong
scaling: the 50 000 families are sharded contiguously over the N ranks (one process per GPU), each
rank builds all matrices, and one RCCL all-reduce of {sum lnL, rejects} closes the call.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (fp64 MFMA, from
HIP events the library records around every K2 launch in the timed region) and `cpu_baseline`
(the CPU restatement of the reference algorithm timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # AMD MI355X datasheet: FP64 matrix = FP64 vector = 78.6 TFLOP/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--families", type=int, default=50000)
    ap.add_argument("--taxa", type=int, default=100)
    ap.add_argument("--max-count", type=int, default=600)
    ap.add_argument("--categories", type=int, default=8)
    ap.add_argument("--lam", type=float, default=0.002)
    ap.add_argument("--alpha", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-shard", default="", help="R/W: rehearsal on one GPU of what rank R of W would run (no collective)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the real thing); gloo only to rehearse the N>1 path on a box with fewer GPUs than ranks")
    ap.add_argument("--cpu-matrices", type=int, default=3, help="matrices built by the O(N^3) reference algorithm in the CPU sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU sample (0 = the host CPU share)")
    ap.add_argument("--cpu-families", type=int, default=0, help="families pruned in the CPU sample (0 = 4 per host thread)")
    return ap.parse_args()


def cpu_baseline(pb, pr, args, n_matrices_call):
    """CPU restatement of the reference path (oracle/, "port") on a bounded sample, all host cores.
    matrices: the reference's O(N^3) log-space entries (matrix_cache.cpp:121); prune: per-family
    post-order mat-vecs (core.cpp:133).  Extrapolated linearly to the whole call."""
    import dataclasses
    import numpy as np
    from oracle import oracle as O
    threads = min(O.host_cpu_share(), args.cpu_threads or 10 ** 6)
    O.set_threads(threads)                            # OpenMP would otherwise take every logical CPU of the host
    n = pb.matrix_size
    ts = sorted({float(t) for t in pb.branch_length if t > 0})
    pick = [ts[(i * len(ts)) // args.cpu_matrices] for i in range(args.cpu_matrices)]
    t_mat = O.time_matrices(n, args.lam, pick, fast=False)
    per_matrix = t_mat / len(pick)
    nf = args.cpu_families or 4 * threads
    nf = min(nf, pb.n_families)
    sub = dataclasses.replace(pb, counts=pb.counts[:nf].copy(), family_ids=pb.family_ids[:nf])
    O.score(sub, pr, fast=True)                       # matrices by the O(N^2) build: only the prune is timed
    t_m, t_all = O.last_timings()
    per_family = (t_all - t_m) / nf                   # all K categories of one family, on `threads` threads
    call_s = per_matrix * n_matrices_call + per_family * pb.n_families
    K = 1 if pr.multipliers is None else len(pr.multipliers)
    return {
        "value": pb.n_families / call_s, "unit": "families/s", "cores": threads, "kind": "port",
        "sample": "%d of %d transition matrices (order %d, reference O(N^3) algorithm, %.2f s each) + %d of %d families x %d "
                  "categories pruned (%.3f s per family); whole call extrapolated linearly to %.0f s"
                  % (len(pick), n_matrices_call, n, per_matrix, nf, pb.n_families, K, per_family, call_s),
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import dataclasses
    import numpy as np
    import torch
    import torch.distributed as dist
    from cafexp_amd import capi, problem as P, synth
    from cafexp_amd.gamma_rates import discrete_gamma

    device = local_rank % torch.cuda.device_count()          # == local_rank on a real N-GPU launch
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    pb, _ = synth.make_problem(n_taxa=args.taxa, n_families=args.families, max_count=args.max_count)
    F = pb.n_families
    K = args.categories
    if K > 1:
        probs, mult = discrete_gamma(K, args.alpha)
        pr = P.Params(lambdas=np.array([args.lam]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
    else:
        pr = P.Params(lambdas=np.array([args.lam]), prior=P.prior_uniform(pb.max_root_family_size))
    # every rank computes the same partition; shards are balanced by distinct subtree patterns (the device's unit of
    # work), not by family count
    if args.emulate_shard:
        er, ew = (int(x) for x in args.emulate_shard.split("/"))
        mine = P.shard_families_by_pattern_cost(pb, ew)[er]
    else:
        mine = P.shard_families_by_pattern_cost(pb, world)[rank]
    shard = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[mine]), family_ids=[pb.family_ids[i] for i in mine])
    ctx = capi.Context(shard, max_categories=max(1, K), device=device)

    buf = torch.zeros(2, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        ctx.score_partial(pr, buf.data_ptr(), stream.cuda_stream, alpha=args.alpha)
        if world > 1 and args.backend == "nccl":
            dist.all_reduce(buf)                      # RCCL over xGMI: {sum lnL, rejects}
        pair = buf.cpu()                              # host read-back = the scorer's return value
        if world > 1 and args.backend != "nccl":
            dist.all_reduce(pair)
        return ctx.finish(pair.numpy())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    value = None
    for _ in range(args.warmup):
        value = step()
    fence()
    flops = flops_fam = ms_gemm = 0.0
    launches = 0
    ms_mat = ms_prune = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        value = step()
        st = ctx.stats()                              # HIP events of this call, recorded on the launch stream
        flops += st["gemm_flops"]; flops_fam += st["gemm_flops_per_family"]; ms_gemm += st["ms_gemm"]; launches += st["gemm_launches"]
        ms_mat += st["ms_matrices"]; ms_prune += st["ms_prune"]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        st = ctx.stats()
        achieved = flops / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):                     # PMC passes of this same workload (profiles/)
            try:
                t = json.load(open(tpath))
                if t.get("workload") == {"families": args.families, "taxa": args.taxa, "max_count": args.max_count,
                                         "categories": args.categories, "gpus": world}:
                    traffic = t.get("prune_gemm_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "gene-family likelihoods/sec per optimizer_scorer call",
            "value": F * args.steps / elapsed, "unit": "families/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic %d families / %d taxa / max family size %d (matrix order %d), gamma K=%d, lambda=%g alpha=%g, "
                                   "family-sharded over %d GPU(s)" % (F, args.taxa, args.max_count, pb.matrix_size, K, args.lam, args.alpha, world),
                       "families": F, "taxa": args.taxa, "max_family_size": args.max_count, "gamma_categories": K,
                       "parallelism": "family-shard x%d + 1 all-reduce" % world},
            "neg_lnl": value,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": "prune_gemm_kernel", "launches_per_step": launches // max(1, args.steps),
                         "avg_launch_ms": ms_gemm / max(1, launches), "flops_per_launch": flops / max(1, launches),
                         # columns = distinct subtree patterns: the flops the launches execute; with
                         # one column per family at every node (SURVEY 8d's per-family figure) they would be:
                         "flops_per_launch_one_column_per_family": flops_fam / max(1, launches)},
            "phases_ms_per_step": {"bd_matrix_build": ms_mat / args.steps, "prune_total": ms_prune / args.steps,
                                   "prune_gemm": ms_gemm / args.steps},
            "n_matrices": st["n_matrices"],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pb, pr, args, st["n_matrices"])
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
