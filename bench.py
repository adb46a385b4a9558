#!/usr/bin/env python3
"""bench.py -- gene-family likelihoods / second per optimizer_scorer call on MI355X.

One "step" = one scorer call (model::infer_family_likelihoods through the C ABI): build every
transition matrix, prune every family x gamma category, reduce to -lnL, all-reduce the shard sums
(N > 1) and bring the value to the host -- with the families already resident in HBM.

Workload (BASELINE.json configs[3], the configuration the metric is quoted on; it fits one GPU):
synthetic 50 000 families / 100 taxa / max family size 600 (M=720, R=750, matrix order 751), gamma
model K=8, from cafexp_amd/synth.py (seed 20251004), scored at lambda=0.002, alpha=2.0.  Strong
scaling: the 50 000 families are sharded over the N ranks by the library's plan (cafe_shard_plan), one
process per GPU, each rank builds all matrices, and ONE RCCL all-reduce of {sum lnL, rejects} -- issued
inside cafe_score by the library itself (cafe_comm_attach) -- closes the call.

`python bench.py --gpus N` run plainly starts its N ranks itself (torch.distributed.run as a child
process, before this process touches torch or the GPU) and relays rank 0's JSON line; launched under
torch.distributed.run it is one of the ranks.  `--single-process` drives the N GPUs from ONE process
instead (cafe_create_sharded: a host thread and stream per device, ncclCommInitAll).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (fp64 MFMA, from HIP
events the library records around every K2 launch in the timed region) and, at N = 1, `cpu_baseline`
(the CPU restatement of the reference algorithm timed on this host's cores on a bounded sample),
`parity_sample` (the CPU-pruned families against the GPU's values for the same families, and the
reference-algorithm matrices against the GPU's), `one_column_per_family` (the same call without the
subtree-level column sharing: the data-independent figure) and `other_configs` (BASELINE configs 2, 3
on the mammals fixture and config 5's shape).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # AMD MI355X datasheet: FP64 matrix = FP64 vector = 78.6 TFLOP/s
PARITY_TOL = 1e-10                # per-family values against the CPU restatement (BASELINE asks 1e-6 on -lnL)


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
    ap.add_argument("--no-extras", action="store_true", help="skip one_column_per_family and other_configs (N = 1 only)")
    ap.add_argument("--single-process", action="store_true", help="N > 1: one process drives all GPUs (cafe_create_sharded)")
    ap.add_argument("--emulate-shard", default="", help="R/W: rehearsal on one GPU of what rank R of W would run (no collective)")
    ap.add_argument("--force-comm", action="store_true", help="diagnostic, under a launcher with ONE rank: take the N > 1 path (process group, "
                    "communicator inside the library, all-reduce in cafe_score) on a one-GPU box")
    ap.add_argument("--shard-times", default="", help="with --emulate-shard: ms per shard measured under the default plan (comma list; "
                    "several steps separated by ';': each measured under the plan the steps before it give): use the plan rebalanced by "
                    "them, as the ranks of an N > 1 run do after their first calls")
    ap.add_argument("--rebalance-steps", type=int, default=1, help="N > 1: steps of measured rebalancing during set-up (a second step measured no better: the residual 2 % is run-to-run noise)")
    ap.add_argument("--no-rebalance", dest="rebalance", action="store_false", help="N > 1: keep the predicted shard plan.  Default: one step "
                    "of measured rebalancing during set-up (every rank times a few calls of its predicted shard, the plan is corrected by the "
                    "gathered times)")
    ap.add_argument("--rebalance", dest="rebalance", action="store_true", help="(the default)")
    ap.set_defaults(rebalance=True)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI, issued by the library (the real thing); gloo: rehearsal of the N>1 path on a box "
                         "with fewer GPUs than ranks (all ranks share GPU 0, the pair is summed on the host)")
    ap.add_argument("--cpu-matrices", type=int, default=3, help="matrices built by the O(N^3) reference algorithm in the CPU sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU sample (0 = the host CPU share)")
    ap.add_argument("--cpu-families", type=int, default=256, help="families pruned in the CPU sample (SURVEY 8d: 256)")
    return ap.parse_args()


def launch_ranks(args):
    """N > 1 and not under a launcher: start the ranks as children BEFORE anything here touches the GPU."""
    # the launcher owns the rendezvous port (--standalone: a c10d store on a port it binds itself), no bind-close-reuse race
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    for ln in p.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if line:
        print(line[-1], flush=True)
    return p.returncode if p.returncode else (0 if line else 1)


def cpu_baseline(pb, pr, args, n_matrices_call, ctx_values):
    """CPU restatement of the reference path (oracle/, "port") on a bounded sample, all host cores.
    matrices: the reference's O(N^3) log-space entries (matrix_cache.cpp:121); prune: per-family
    post-order mat-vecs (core.cpp:133).  Extrapolated linearly to the whole call.  The sample is also the
    full-size parity check: the pruned families' values and the O(N^3) matrices against the GPU's."""
    import dataclasses
    import numpy as np
    from oracle import oracle as O
    threads = min(O.host_cpu_share(), args.cpu_threads or 10 ** 6)
    O.set_threads(threads)                            # OpenMP would otherwise take every logical CPU of the host
    n = pb.matrix_size
    K = 1 if pr.multipliers is None else len(pr.multipliers)
    # matrices: branches spread over the tree; category 0 carries the smallest multiplier, K-1 the largest
    branches = [v for v in range(pb.n_nodes) if pb.parent[v] >= 0 and pb.branch_length[v] > 0]
    pick = [branches[(i * len(branches)) // args.cpu_matrices] for i in range(args.cpu_matrices)]
    mat_rel = 0.0
    t_mat = 0.0
    for i, v in enumerate(pick):
        k = (i * max(1, K - 1)) // max(1, args.cpu_matrices - 1) if K > 1 else 0
        lam = float(pr.lambdas[pb.lambda_index[v]]) * (float(pr.multipliers[k]) if pr.multipliers is not None else 1.0)
        t0 = time.perf_counter()
        want = O.build_matrix(n, lam, float(pb.branch_length[v]), fast=False)      # O(N^3), OpenMP over rows
        t_mat += time.perf_counter() - t0
        got = ctx_values["matrix"](v, k)
        cols = n if pb.leaf_taxon[v] >= 0 else pb.max_family_size + 1               # interior branches: columns <= M only
        sel = want[:, :cols] > 1e-290                                               # (below that the O(N^3) sum itself loses digits)
        mat_rel = max(mat_rel, float(np.max(np.abs(got[:, :cols][sel] / want[:, :cols][sel] - 1.0))))
    per_matrix = t_mat / len(pick)
    nf = min(args.cpu_families, pb.n_families)
    sel = np.unique(np.linspace(0, pb.n_families - 1, nf).astype(np.int64))       # spread over the table
    sub = dataclasses.replace(pb, counts=pb.counts[sel].copy(), family_ids=[pb.family_ids[i] for i in sel])
    if pr.multipliers is not None:
        _, cat, fam = O.score_gamma(sub, pr, fast=True, per_family=True)            # matrices by the O(N^2) build: only the prune is timed
        got_fam = ctx_values["family_likelihood"][sel]
        got_cat = ctx_values["category_likelihood"][sel]
        fam_rel = float(np.max(np.abs(got_fam / fam - 1.0)))
        ok = cat > 0
        fam_rel = max(fam_rel, float(np.max(np.abs(got_cat[ok] / cat[ok] - 1.0))))
    else:
        _, fam = O.score_base(sub, pr, fast=True, per_family=True)
        fam_rel = float(np.max(np.abs(ctx_values["family_lnl"][sel] / fam - 1.0)))
    t_m, t_all = O.last_timings()
    per_family = (t_all - t_m) / len(sel)             # all K categories of one family, on `threads` threads
    call_s = per_matrix * n_matrices_call + per_family * pb.n_families
    # the same on ONE thread, smaller sample: one matrix, 8 families
    O.set_threads(1)
    t0 = time.perf_counter()
    O.build_matrix(n, float(pr.lambdas[0]), float(pb.branch_length[pick[0]]), fast=False)
    per_matrix_1 = time.perf_counter() - t0
    sel1 = sel[:: max(1, len(sel) // 8)][:8]
    sub1 = dataclasses.replace(pb, counts=pb.counts[sel1].copy(), family_ids=[pb.family_ids[i] for i in sel1])
    O.score(sub1, pr, fast=True)
    t_m1, t_all1 = O.last_timings()
    per_family_1 = (t_all1 - t_m1) / len(sel1)
    call_s1 = per_matrix_1 * n_matrices_call + per_family_1 * pb.n_families
    O.set_threads(threads)
    base = {
        "value": pb.n_families / call_s, "unit": "families/s", "cores": threads, "kind": "port",
        "sample": "%d of %d transition matrices (order %d, reference O(N^3) algorithm, %.2f s each) + %d of %d families x %d "
                  "categories pruned (%.3f s per family); whole call extrapolated linearly to %.0f s"
                  % (len(pick), n_matrices_call, n, per_matrix, len(sel), pb.n_families, K, per_family, call_s),
        "one_thread": {"value": pb.n_families / call_s1, "unit": "families/s", "cores": 1,
                       "sample": "1 matrix (%.2f s) + %d families (%.3f s per family); extrapolated to %.0f s per call"
                                 % (per_matrix_1, len(sel1), per_family_1, call_s1)},
        "vs_compiled_reference": "the restatement runs 0.94x (prune) to 0.69x (matrix build) the compiled reference's time on equal "
                                 "threads (BASELINE.md section 3): the true reference is that much slower than this figure",
    }
    parity = {"families_checked": int(len(sel)), "family_values_max_rel": fam_rel, "matrices_checked": len(pick),
              "matrix_entries_max_rel": mat_rel, "tolerance": PARITY_TOL,
              "against": "oracle/ CPU restatement: per-family (and per-category) likelihoods of the sampled families; "
                         "matrix entries against the reference's O(N^3) log-space sum"}
    return base, parity


def timed_calls(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        v = fn()
    return (time.perf_counter() - t0) / reps, v


def other_configs(args):
    """BASELINE configs 2, 3 (mammals fixture) and the shape of config 5, each a few timed calls on GPU 0."""
    import numpy as np
    from cafexp_amd import capi, problem as P, synth
    from cafexp_amd.gamma_rates import discrete_gamma
    out = {}
    data = os.path.join(ROOT, "tests", "golden", "data")
    rd = lambda n: open(os.path.join(data, n)).read()
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_golden.json")))["scores"]
    tree = P.parse_newick(rd("mammals_tree.txt"))
    species, ids, counts = P.read_family_table(rd("mammal_gene_families.txt"))
    pb = P.build_problem(tree, species, ids, counts)
    prior = P.prior_uniform(pb.max_root_family_size)
    ctx = capi.Context(pb, max_categories=4)
    ctx.set_profiling(False)
    for name, gold, pr, alpha in [
        ("config2_mammals_base", "mammals_base_l0.01", P.Params(lambdas=np.array([0.01]), prior=prior), 1.0),
        ("config3_mammals_gamma_k4", "mammals_gamma_k4_a2",
         (lambda pm: P.Params(lambdas=np.array([0.005]), prior=prior, multipliers=pm[1], cat_probs=pm[0]))(discrete_gamma(4, 2.0)), 2.0),
    ]:
        sec, v = timed_calls(lambda: ctx.score(pr, alpha=alpha), 50)
        ref = golden[gold]["neg_lnl"]
        out[name] = {"families": pb.n_families, "ms_per_call": 1e3 * sec, "families_per_s": pb.n_families / sec, "neg_lnl": v,
                     "rel_err_vs_reference": abs(v - ref) / abs(ref), "reference_neg_lnl": ref}
    ctx.close()
    def sample_parity(pb_, pr_, res_, n=64):
        """The same full-size check the headline gets (`parity_sample`): n families spread over the table, pruned by the CPU
        restatement, against the GPU's per-family values of this very call."""
        import dataclasses
        from oracle import oracle as O
        O.set_threads(min(O.host_cpu_share(), args.cpu_threads or 10 ** 6))
        sel = np.unique(np.linspace(0, pb_.n_families - 1, n).astype(np.int64))
        sub = dataclasses.replace(pb_, counts=pb_.counts[sel].copy(), family_ids=[pb_.family_ids[i] for i in sel])
        if pr_.multipliers is not None:
            _, cat, fam = O.score_gamma(sub, pr_, fast=True, per_family=True)
            rel = float(np.max(np.abs(res_["family_likelihood"][sel] / fam - 1.0)))
            ok = cat > 0
            rel = max(rel, float(np.max(np.abs(res_["category_likelihood"][sel][ok] / cat[ok] - 1.0))))
        else:
            _, fam = O.score_base(sub, pr_, fast=True, per_family=True)
            rel = float(np.max(np.abs(res_["family_lnl"][sel] / fam - 1.0)))
        return {"families_checked": int(len(sel)), "family_values_max_rel": rel, "tolerance": PARITY_TOL,
                "against": "oracle/ CPU restatement, per-family values of the sampled families"}

    check = not args.no_cpu_baseline
    # config 5: lambda tree with two rates + 3-tap error model, base model, 100 000 families (the bench's generator)
    pb5, _ = synth.make_problem(n_taxa=args.taxa, n_families=2 * args.families, max_count=args.max_count, lambda_clade_min=10, n_deviations=3)
    em = P.error_model_table(P.default_error_model(pb5.max_family_size)[:1] + [[0.05, 0.9, 0.05]], pb5.max_family_size)
    pr5 = P.Params(lambdas=np.array([args.lam, 2 * args.lam]), prior=P.prior_uniform(pb5.max_root_family_size), error_model=em)
    ctx = capi.Context(pb5)
    ctx.set_profiling(False)
    sec, v = timed_calls(lambda: ctx.score(pr5), 3)
    out["config5_shape_two_lambdas_error_model"] = {"families": pb5.n_families, "ms_per_call": 1e3 * sec, "families_per_s": pb5.n_families / sec, "neg_lnl": v}
    if check:
        out["config5_shape_two_lambdas_error_model"]["parity_sample"] = sample_parity(pb5, pr5, ctx.family_results(0))
    ctx.close()
    # SURVEY 8d's generator parameters for config 4 (lambda_sim 0.003, root sizes capped at 480; the headline uses 0.002 / 300)
    # scored at SURVEY 8d's point lambda 0.003 / alpha 1.5: larger families, fewer shared subtree patterns, wider non-zero
    # extents.  (Finite: profiles/r03_zero_categories.json lists which (lambda, alpha) points have a zero category.)
    pb4, _ = synth.make_problem(n_taxa=args.taxa, n_families=args.families, max_count=args.max_count, lam_sim=0.003, root_cap=480)
    K = args.categories
    for key, lam, alpha in [("config4_survey_generator_lambda_sim_0.003_root_cap_480", args.lam, args.alpha),
                            ("config4_survey_generator_and_scoring_point_lambda_0.003_alpha_1.5", 0.003, 1.5)]:
        probs, mult = discrete_gamma(K, alpha)
        pr4 = P.Params(lambdas=np.array([lam]), prior=P.prior_uniform(pb4.max_root_family_size), multipliers=mult, cat_probs=probs)
        ctx = capi.Context(pb4, max_categories=K)
        ctx.set_profiling(False)
        sec, v = timed_calls(lambda: ctx.score(pr4, alpha=alpha), 3)
        out[key] = {"families": pb4.n_families, "lambda": lam, "alpha": alpha, "ms_per_call": 1e3 * sec, "families_per_s": pb4.n_families / sec, "neg_lnl": v}
        if check:
            out[key]["parity_sample"] = sample_parity(pb4, pr4, ctx.family_results(K))
        ctx.close()
    return out


def main():
    args = parse()
    under_launcher = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher and not args.single_process and not args.emulate_shard:
        sys.exit(launch_ranks(args))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # (dmabuf IPC: RCCL across processes needs it on this driver)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if under_launcher:
        args.gpus = world
    n_gpus = args.gpus

    import dataclasses
    import numpy as np
    import torch
    import torch.distributed as dist
    from cafexp_amd import capi, problem as P, synth
    from cafexp_amd.gamma_rates import discrete_gamma

    dist_on = world > 1 or args.force_comm            # (--force-comm: the N > 1 code path with a world of one rank, for a one-GPU box)
    native_comm = dist_on and args.backend == "nccl"
    lib_reduce = native_comm                          # the all-reduce is issued inside cafe_score (cafe_comm_attach)
    device = local_rank % max(1, torch.cuda.device_count()) if native_comm or world == 1 else 0
    torch.cuda.set_device(device)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if native_comm:
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

    sharded = None
    plan_note = None
    if args.single_process and n_gpus > 1:
        sharded = capi.Sharded(pb, list(range(n_gpus)), max_categories=max(1, K))
        ctx = sharded.shard(0)
        for r in range(n_gpus):
            sharded.shard(r).set_profiling(r == 0)
    else:
        # every rank derives the same plan; shards are balanced by predicted device time (distinct subtree patterns)
        def shard_of(fam):
            return dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[fam]), family_ids=[pb.family_ids[i] for i in fam])

        if args.emulate_shard:
            er, ew = (int(x) for x in args.emulate_shard.split("/"))
            plan = capi.shard_plan(pb, ew, max(1, K))
            if args.shard_times:
                scale = None
                for step_times in args.shard_times.split(";"):
                    plan, scale = capi.rebalanced_plan(pb, plan, [float(x) for x in step_times.split(",")], max(1, K), scale=scale, return_scale=True)
                plan_note = "rebalanced %d time(s) from measured shard times" % len(args.shard_times.split(";"))
            mine = plan[er]
        elif dist_on:
            plan = capi.shard_plan(pb, world, max(1, K))
            if args.rebalance:
                # Measured rebalancing, part of the set-up: what the prediction cannot see (how many K tiles the zero extents leave
                # at these parameters) is in the time a shard's call takes.  Every rank times a few calls of its shard, the times
                # are gathered, every rank derives the same corrected plan; a second step takes out most of what the first left.
                scale, notes = None, []
                for _step in range(max(1, args.rebalance_steps)):
                    probe = capi.Context(shard_of(plan[rank]), max_categories=max(1, K), device=device)
                    for _ in range(2):
                        probe.score(pr, alpha=args.alpha)
                    t0p = time.perf_counter()
                    for _ in range(3):
                        probe.score(pr, alpha=args.alpha)
                    mine_ms = (time.perf_counter() - t0p) / 3 * 1e3
                    probe.close()
                    tt = torch.zeros(world, dtype=torch.float64, device="cuda" if native_comm else "cpu")
                    tt[rank] = mine_ms
                    dist.all_reduce(tt)
                    times = [float(x) for x in tt.cpu().numpy()]
                    plan, scale = capi.rebalanced_plan(pb, plan, times, max(1, K), scale=scale, return_scale=True)
                    notes.append([round(x, 2) for x in times])
                plan_note = "rebalanced %d time(s) from measured shard times %s ms" % (len(notes), notes)
            mine = plan[rank]
        else:
            mine = np.arange(F)
        shard = pb if len(mine) == F and world == 1 else shard_of(mine)
        ctx = capi.Context(shard, max_categories=max(1, K), device=device)
        ctx.set_profiling(True)
        if native_comm:
            # the communicator of the data path lives in the library: rank 0's id travels over the launcher's channel
            idt = torch.zeros(capi.CAFE_COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            ok = torch.ones(1, dtype=torch.int32, device="cuda")
            try:
                ctx.comm_attach(bytes(idt.cpu().numpy().tobytes()), world, rank)
            except capi.CafeError as e:                # every rank must take the same path: agree on it
                print("rank %d: cafe_comm_attach failed (%s)" % (rank, e), file=sys.stderr, flush=True)
                ok.zero_()
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                # the library's communicator could not be set up: the pair is reduced by torch.distributed instead (the same
                # RCCL all-reduce of two doubles, issued from here rather than from inside cafe_score)
                try:
                    ctx.comm_detach()
                except capi.CafeError:
                    pass
                lib_reduce = False

    buf = torch.zeros(2, dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        if sharded is not None:
            return sharded.score(pr, alpha=args.alpha)
        if world == 1 or lib_reduce:
            return ctx.score(pr, alpha=args.alpha)    # N > 1: ends with the library's ncclAllReduce of {sum lnL, rejects}; ranks fail together
        # the pair is reduced from here.  A rank whose call fails leaves {0, NaN} in its pair and STILL takes part in the
        # reduction, so that every rank reads NaN and raises instead of waiting for it.
        err = None
        try:
            ctx.score_partial(pr, buf.data_ptr(), stream.cuda_stream, alpha=args.alpha)
        except capi.CafeError as e:
            err = e
        if native_comm:                               # (only if cafe_comm_attach failed) RCCL all-reduce issued by torch
            dist.all_reduce(buf)
            pair = buf.cpu().numpy()
        else:
            pair = buf.cpu()                          # gloo rehearsal
            dist.all_reduce(pair)
            pair = pair.numpy()
        if err is not None or pair[1] != pair[1]:
            raise RuntimeError("rank %d: %s" % (rank, err if err is not None else "another rank's call failed (poisoned pair)"))
        return ctx.finish(pair)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    # The roofline of the N = 1 line is measured live: HIP events on every K2 dispatch of every timed step.  A shard's call is
    # six times shorter and the 200 events cost it 2 % (0.5 ms; 0.3 ms = 0.2 % of the whole table's call), so the ranks of an
    # N > 1 run (and the one-GPU rehearsal of a rank) time their steps without events and take the per-launch figures from one
    # more, profiled, step behind the timed region (`roofline.measured`).
    live_events = not dist_on and sharded is None and not args.emulate_shard
    if not live_events:
        for r in range(n_gpus if sharded is not None else 1):
            (sharded.shard(r) if sharded is not None else ctx).set_profiling(False)
    value = None
    for _ in range(args.warmup):
        value = step()
    fence()
    flops = flops_fam = flops_dense = ms_gemm = 0.0
    launches = 0
    ms_mat = ms_prune = 0.0

    def account():
        nonlocal flops, flops_fam, flops_dense, ms_gemm, launches, ms_mat, ms_prune
        st = ctx.stats()                              # HIP events of this call, recorded on the launch stream
        flops += st["gemm_flops"]; flops_fam += st["gemm_flops_per_family"]; ms_gemm += st["ms_gemm"]; launches += st["gemm_launches"]
        flops_dense += st["gemm_flops_dense"]
        ms_mat += st["ms_matrices"]; ms_prune += st["ms_prune"]

    t0 = time.perf_counter()
    for _ in range(args.steps):
        value = step()
        if live_events:
            account()
    fence()
    elapsed = time.perf_counter() - t0
    n_prof = args.steps
    if not live_events:
        ctx.set_profiling(True)
        step()
        account()
        n_prof = 1
        fence()
    # what the K2 launches of a step really executed (the same every step: same parameters); counted once, outside the timing
    flops_executed = ctx.executed_flops() * n_prof
    flops_tile_ranges = ctx.tile_range_flops() * n_prof
    if dist_on:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if native_comm else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    rc = 0
    if rank == 0:
        st = ctx.stats()
        achieved = flops_executed / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):                     # PMC passes of this same workload (profiles/)
            try:
                t = json.load(open(tpath))
                if t.get("workload") == {"families": args.families, "taxa": args.taxa, "max_count": args.max_count,
                                         "categories": args.categories, "gpus": n_gpus}:
                    traffic = t.get("prune_gemm_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "gene-family likelihoods/sec per optimizer_scorer call",
            "value": F * args.steps / elapsed, "unit": "families/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic %d families / %d taxa / max family size %d (matrix order %d), gamma K=%d, lambda=%g alpha=%g, "
                                   "family-sharded over %d GPU(s)" % (F, args.taxa, args.max_count, pb.matrix_size, K, args.lam, args.alpha, n_gpus),
                       "families": F, "taxa": args.taxa, "max_family_size": args.max_count, "gamma_categories": K,
                       "parallelism": "family-shard x%d + 1 RCCL all-reduce %s (%s)"
                                      % (n_gpus, "inside cafe_score" if (lib_reduce or sharded is not None or world == 1) else
                                         "issued by torch.distributed (cafe_comm_attach failed)" if native_comm else "on the host (gloo rehearsal)",
                                         "one process, a host thread per GPU" if sharded is not None else "one process per GPU")},
            "neg_lnl": value,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": "prune_gemm_kernel", "launches_per_step": launches // max(1, n_prof),
                         "measured": "HIP events on every K2 dispatch of the timed steps" if live_events
                                     else "HIP events on every K2 dispatch of one step behind the timed region (rank 0's shard)",
                         "avg_launch_ms": ms_gemm / max(1, launches), "flops_per_launch": flops_executed / max(1, launches),
                         # flops_per_launch = the flops of the MFMAs the launches ISSUE: columns = distinct subtree patterns, K
                         # tiles = those inside a row tile's non-zero extent, each 16-row block of the tile over its own extent
                         # only (DESIGN.md sections 2, 3).  Then: every block over its tile's whole K range (the count rounds 2
                         # and 3a quoted `frac` on: the kernel issued those MFMAs then), every K tile of the distinct columns,
                         # one column per family at every node (SURVEY 8d's per-family figure):
                         "flops_per_launch_whole_tile_ranges": flops_tile_ranges / max(1, launches),
                         "frac_over_whole_tile_ranges": flops_tile_ranges / (ms_gemm * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS if ms_gemm > 0 else None,
                         "flops_per_launch_all_k_tiles": flops / max(1, launches),
                         "flops_per_launch_one_column_per_family": flops_fam / max(1, launches)},
            "phases_ms_per_step": {"bd_matrix_build": ms_mat / n_prof, "prune_total": ms_prune / n_prof,
                                   "prune_gemm": ms_gemm / n_prof},
            "n_matrices": st["n_matrices"],
        }
        if n_gpus > 1:
            out["roofline"]["note"] = "rank 0's shard"
        if dist_on or n_gpus > 1 or args.emulate_shard:
            out["config"]["shard_plan"] = plan_note or "predicted device time (cafe_shard_plan)"
        if not dist_on and sharded is None and not args.emulate_shard:
            if not args.no_cpu_baseline:
                res = ctx.family_results(K if K > 1 else 0)
                res["matrix"] = ctx.matrix
                out["cpu_baseline"], out["parity_sample"] = cpu_baseline(pb, pr, args, st["n_matrices"], res)
                worst = max(out["parity_sample"]["family_values_max_rel"], out["parity_sample"]["matrix_entries_max_rel"])
                if not worst <= PARITY_TOL:
                    print("bench.py: PARITY FAILURE at full size: %r" % out["parity_sample"], file=sys.stderr)
                    rc = 3
            ctx.close()
            if not args.no_extras:
                # the same call with one column per family at every node (no subtree-level sharing): data-independent
                plain = capi.Context(pb, max_categories=max(1, K), device=device, subtree_dedup=False)
                plain.set_profiling(True)
                sec, v = timed_calls(lambda: plain.score(pr, alpha=args.alpha), 2)
                ps = plain.stats()
                out["one_column_per_family"] = {"ms_per_step": 1e3 * sec, "value": F / sec, "neg_lnl": v,
                                                "prune_gemm_tflops": plain.executed_flops() / (ps["ms_gemm"] * 1e-3) / 1e12 if ps["ms_gemm"] > 0 else None,
                                                "identical_to_headline": v == value}
                plain.close()
                # and with every K tile of every column run (CAFE_NO_KSKIP, read at cafe_create: no zero extents, no planned
                # tile lists): what the call costs when the matrices have no exact zeros to skip (large lambda * t)
                os.environ["CAFE_NO_KSKIP"] = "1"
                try:
                    dense = capi.Context(pb, max_categories=max(1, K), device=device)
                finally:
                    del os.environ["CAFE_NO_KSKIP"]
                dense.set_profiling(True)
                sec, v = timed_calls(lambda: dense.score(pr, alpha=args.alpha), 2)
                ds = dense.stats()
                out["every_k_tile"] = {"ms_per_step": 1e3 * sec, "value": F / sec, "neg_lnl": v,
                                       "prune_gemm_tflops": ds["gemm_flops"] / (ds["ms_gemm"] * 1e-3) / 1e12 if ds["ms_gemm"] > 0 else None,
                                       "identical_to_headline": v == value}
                dense.close()
                # and with one op per launch (CAFE_NO_GROUPS: the post-order schedule of rounds 1-2, 98 K2 launches): what the
                # level-batched launches buy; no per-launch events on either side
                per_op = {}
                for flag in ("0", "1"):
                    if flag == "1":
                        os.environ["CAFE_NO_GROUPS"] = "1"
                    try:
                        cs = capi.Context(pb, max_categories=max(1, K), device=device)
                    finally:
                        os.environ.pop("CAFE_NO_GROUPS", None)
                    sec, v = timed_calls(lambda: cs.score(pr, alpha=args.alpha), 3)
                    per_op[flag] = (sec, v, cs.stats()["gemm_launches"])
                    cs.close()
                out["one_op_per_launch"] = {"ms_per_step": 1e3 * per_op["1"][0], "k2_launches": per_op["1"][2],
                                            "grouped_no_events_ms_per_step": 1e3 * per_op["0"][0], "grouped_k2_launches": per_op["0"][2],
                                            "identical_to_headline": per_op["1"][1] == value and per_op["0"][1] == value}
                for name in ("one_column_per_family", "every_k_tile", "one_op_per_launch"):
                    if name in out and not out[name]["identical_to_headline"]:
                        print("bench.py: %s is not bit-identical to the headline call: %r" % (name, out[name]), file=sys.stderr)
                        rc = 3
                out["other_configs"] = other_configs(args)
                for name, leg in out["other_configs"].items():       # every -lnL this line prints is checked
                    worst = leg.get("rel_err_vs_reference", leg.get("parity_sample", {}).get("family_values_max_rel", None))
                    if worst is None and not args.no_cpu_baseline:
                        worst = float("inf")
                    if worst is not None and not worst <= PARITY_TOL:
                        print("bench.py: PARITY FAILURE in other_configs[%s]: %r" % (name, leg), file=sys.stderr)
                        rc = 3
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        if lib_reduce:
            ctx.comm_detach()
        dist.destroy_process_group()
    if sharded is not None:
        sharded.close()
    sys.exit(rc)


if __name__ == "__main__":
    main()
