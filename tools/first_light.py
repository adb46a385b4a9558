"""Scratch GPU check: K1 matrices, mammals scores and the MFMA probe against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi
from oracle import oracle as O

D = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
print("fp64 mfma probe TF:", capi.probe_fp64_mfma())
for n, lam, t in [(5, 0.05, 5.0), (141, 0.006335, 68.7105), (141, 0.01, 96.435575), (60, 0.05, 25.0), (32, 0.001, 0.0005), (300, 0.003, 40.0)]:
    g = capi.build_matrices(n, [lam], [t])[0]
    o = O.build_matrix(n, lam, t)
    mask = o > 1e-290
    rel = np.abs(g - o)[mask] / o[mask] if mask.any() else np.zeros(1)
    print(n, lam, t, "max rel", rel.max(), "zeros equal", np.array_equal(g == 0, o == 0), "maxabs", np.abs(g-o).max())

tree = P.parse_newick(open(os.path.join(D, "mammals_tree.txt")).read())
sp, ids, counts = P.read_family_table(open(os.path.join(D, "mammal_gene_families.txt")).read())
pb = P.build_problem(tree, sp, ids, counts)
ctx = capi.Context(pb, max_categories=4)
pr = P.Params(lambdas=np.array([0.01]), prior=P.prior_uniform(pb.max_root_family_size))
for i in range(3):
    t0 = time.time(); v = ctx.score(pr); dt = time.time() - t0
    print("base", repr(v), "ref 207724.99537424563", "rel", abs(v - 207724.99537424563) / 207724.99537424563, "sec", dt)
print(ctx.stats())
probs, mult = O.discrete_gamma(4, 2.0)
pr = P.Params(lambdas=np.array([0.005]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
for i in range(2):
    t0 = time.time(); v = ctx.score(pr, alpha=2.0); dt = time.time() - t0
    print("gamma", repr(v), "ref 161407.8507684404", "rel", abs(v - 161407.8507684404) / 161407.8507684404, "sec", dt)
print(ctx.stats())
probs, mult = O.discrete_gamma(4, 0.5)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
print("gamma inf:", ctx.score(pr, alpha=0.5))
# root vector parity for one family
pr = P.Params(lambdas=np.array([0.01]), prior=P.prior_uniform(pb.max_root_family_size))
ctx.score(pr)
g = ctx.root_likelihoods(7)
o = O.prune(pb, pr, 7)
print("root vec max rel", (np.abs(g - o) / np.maximum(o, 1e-300)).max())
