"""One 1/8 shard of the bench table: ms per scorer call with per-launch HIP events (what bench.py measures with), without
them, and with the call's launch sequence replayed from a hipGraph."""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import capi, problem as P, synth
from cafexp_amd.gamma_rates import discrete_gamma
r, w = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3/8").split("/"))
pb, _ = synth.make_problem(n_families=50000)
mine = capi.shard_plan(pb, w, 8)[r] if w > 1 else np.arange(pb.n_families)
pb = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[mine]), family_ids=[pb.family_ids[i] for i in mine])
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
for prof, graphs in ((True, False), (False, False), (False, True)):
    ctx.set_profiling(prof); ctx.set_graphs(graphs)
    for _ in range(3): v = ctx.score(pr, alpha=2.0)
    t = time.perf_counter()
    for _ in range(10): v = ctx.score(pr, alpha=2.0)
    print("shard %d/%d  events=%-5s graph=%-5s  %.3f ms per call   -lnL %.6f" % (r, w, prof, graphs, (time.perf_counter() - t) * 100, v), flush=True)
