"""Randomised cross-check of everything that decides WHICH work K2 / the assemble pass do (zero extents, planned tile lists,
rows left out by the assemble pass, tile heights, grouped launches): random trees, tables, rates, categories, error models, several calls per
context; every per-family value must have the bits of a context created with all of it switched off (CAFE_NO_KSKIP).
Usage: fuzz_extents.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import capi, problem as P, synth
from cafexp_amd.gamma_rates import discrete_gamma

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    n_taxa = int(rng.choice([4, 7, 16, 33, 60]))
    n_fam = int(rng.choice([130, 700, 2500, 6000]))
    max_count = int(rng.choice([230, 300, 420, 600]))
    n_dev = int(rng.choice([0, 0, 3]))
    two = bool(rng.integers(0, 2)) and n_taxa >= 16
    K = int(rng.choice([1, 2, 4, 8]))
    pb, _ = synth.make_problem(n_taxa=n_taxa, n_families=n_fam, max_count=max_count, lam_sim=float(rng.choice([0.001, 0.003])),
                               seed=int(rng.integers(1, 1 << 30)), root_cap=int(rng.choice([80, 200])),
                               lambda_clade_min=4 if two else 0, n_deviations=n_dev)
    em = None
    if n_dev:
        em = P.error_model_table(P.default_error_model(pb.max_family_size)[:1] + [[0.05, 0.9, 0.05]], pb.max_family_size)
    calls = []
    for _ in range(3):
        lam = np.array([10 ** rng.uniform(-3.6, -2.0) for _ in range(pb.n_lambdas)])
        alpha = float(rng.uniform(0.4, 3.0))
        if K > 1:
            probs, mult = discrete_gamma(K, alpha)
            calls.append((P.Params(lambdas=lam, prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs, error_model=em), alpha))
        else:
            calls.append((P.Params(lambdas=lam, prior=P.prior_uniform(pb.max_root_family_size), error_model=em), 1.0))
    for k in ("CAFE_NO_KSKIP", "CAFE_FORCE_TILE"):
        os.environ.pop(k, None)
    if rng.integers(0, 3) == 0:
        os.environ["CAFE_FORCE_TILE"] = str(int(rng.choice([2, 3, 4, 6, 7, 8])))
    os.environ["CAFE_KB"] = str(int(rng.choice([8, 16])))   # depth of K2's K tiles (normally by matrix order)
    os.environ["CAFE_LEAF_T_MIN"] = str(int(rng.choice([0, 0, 6])))   # transposed leaf matrices for the assemble passes: every eligible branch / the default rule
    fast = capi.Context(pb, max_categories=8)
    os.environ.pop("CAFE_FORCE_TILE", None)
    os.environ.pop("CAFE_LEAF_T_MIN", None)
    os.environ["CAFE_KB"] = "16" if os.environ["CAFE_KB"] == "8" else "8"
    os.environ["CAFE_NO_KSKIP"] = "1"
    if rng.integers(0, 2) == 0:
        os.environ["CAFE_NO_GROUPS"] = "1"                   # ... and one op per launch from the slot pool (the grouped schedule is the default)
    if rng.integers(0, 2) == 0:
        os.environ["CAFE_NO_LEAF_T"] = "1"                   # ... and assemble passes that gather their leaf from the row-major matrix
    plain = capi.Context(pb, max_categories=8)
    os.environ.pop("CAFE_NO_KSKIP")
    os.environ.pop("CAFE_NO_GROUPS", None)
    os.environ.pop("CAFE_NO_LEAF_T", None)
    os.environ.pop("CAFE_KB", None)
    ok = True
    for pr, alpha in calls + calls[:1]:
        def run(ctx):
            try:
                return ctx.score(pr, alpha=alpha, per_family=True)
            except capi.CafeError:                           # a rejected call (lambda too large for the tree) has no family results
                return ctx.score(pr, alpha=alpha), {}
        (v1, r1), (v2, r2) = run(fast), run(plain)
        same = (v1 == v2 or (v1 != v1 and v2 != v2)) and r1.keys() == r2.keys() and all(np.array_equal(r1[k], r2[k], equal_nan=True) for k in r1)
        ok = ok and same
    n_planned = fast.plan_check()[0]
    st = fast.stats()
    print("case %2d: taxa %2d families %5d N %d K %d lambdas %d error model %d: planned launches %d / %d, assemble passes %d -> %s"
          % (case, n_taxa, n_fam, pb.matrix_size, K, pb.n_lambdas, n_dev, n_planned, st["gemm_launches"], st["n_assemble_passes"], "identical" if ok else "DIFFERENT"), flush=True)
    bad += not ok
print("FAILED" if bad else "all identical")
sys.exit(1 if bad else 0)
