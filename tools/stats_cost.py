import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from cafexp_amd.gamma_rates import discrete_gamma
import dataclasses
pb, _ = synth.make_problem(n_families=50000)
mine = P.shard_families_by_pattern_cost(pb, 8)[3]
pb = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[mine]), family_ids=[pb.family_ids[i] for i in mine])
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
for prof in (True, False):
    ctx.set_profiling(prof)
    ctx.score(pr, alpha=2.0)
    t = time.perf_counter()
    for _ in range(10): ctx.score(pr, alpha=2.0)
    a = (time.perf_counter() - t) / 10
    t = time.perf_counter()
    for _ in range(10): ctx.score(pr, alpha=2.0); ctx.stats()
    b = (time.perf_counter() - t) / 10
    print("profiling", prof, "score %.3f ms, score+stats %.3f ms" % (a * 1e3, b * 1e3))
