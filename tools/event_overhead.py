"""Scratch: what the per-launch HIP events of cafe_stats cost per scorer call (profiling on vs off)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from cafexp_amd.gamma_rates import discrete_gamma
F = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
pb, _ = synth.make_problem(n_families=F)
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
for on in (True, False, True, False):
    ctx.set_profiling(on)
    ctx.score(pr, alpha=2.0)
    t = time.perf_counter()
    for _ in range(5):
        ctx.score(pr, alpha=2.0)
    print("profiling", on, "ms per call %.3f" % ((time.perf_counter() - t) / 5 * 1e3))
