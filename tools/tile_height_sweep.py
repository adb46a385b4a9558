"""Forced K2 tile heights (CAFE_FORCE_TILE) and planner constants at the bench shape: ms per call, same -lnL."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import capi, problem as P, synth
from cafexp_amd.gamma_rates import discrete_gamma
pb, _ = synth.make_problem(n_families=50000)
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
sweeps = ({}, {"CAFE_FORCE_TILE": "3"}, {"CAFE_FORCE_TILE": "4"}, {"CAFE_FORCE_TILE": "5"}, {"CAFE_FORCE_TILE": "7"}, {"CAFE_FORCE_TILE": "9"}, {"CAFE_KB": "16"},
          {"CAFE_PLAN_FIXED": "2"}, {"CAFE_PLAN_FIXED": "6"}, {})
if len(sys.argv) > 1 and sys.argv[1] == "bias3":
    sweeps = tuple({"CAFE_PLAN_BIAS3": v} for v in (sys.argv[2:] or ["100,100,100", "89,103,110", "83,105,116", "78,106,124"]))
if len(sys.argv) > 1 and sys.argv[1] == "bias4":
    sweeps = tuple({"CAFE_PLAN_BIAS4": v} for v in sys.argv[2:])
for env in sweeps:
    for k, v in env.items():
        os.environ[k] = v
    ctx = capi.Context(pb, max_categories=8)
    for k in env:
        del os.environ[k]
    for _ in range(2):
        v = ctx.score(pr, alpha=2.0)
    t0 = time.perf_counter()
    for _ in range(5):
        v = ctx.score(pr, alpha=2.0)
    print(env, "%.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3), v, flush=True)
    ctx.close()
