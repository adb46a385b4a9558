"""Scratch: K1 (bd_matrix_build) alone -- the HIP-event time of the matrix build inside scorer calls (stats `ms_matrices`) at
the bench's matrix shape (order 751, 1320 matrices: independent of the number of families) and on the mammals fixture
(order 141).  With a library built with -D'CAFE_EXPERIMENT_K1_STORE_IF=&& n < 0' (bd_matrix.hip) the same figure is the build
without its global stores: how much of K1 is the latency chain of the row steps and how much the write of the pools."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import capi, problem as P, synth
from cafexp_amd.gamma_rates import discrete_gamma

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def k1(ctx, pr, alpha):
    ctx.set_profiling(True)
    ms = []
    for i in range(reps + 2):
        try:
            ctx.score(pr, alpha=alpha)
        except capi.CafeError:
            pass                       # the no-store build scores garbage
        if i >= 2:
            ms.append(ctx.stats()["ms_matrices"])
    return min(ms), float(np.median(ms)), ctx.stats()["n_matrices"]


pb, _ = synth.make_problem(n_families=2048)
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
print("order %d, K=8: K1 min %.4f ms, median %.4f ms (%d matrices)" % ((pb.matrix_size,) + k1(ctx, pr, 2.0)), flush=True)
ctx.close()

data = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
rd = lambda n: open(os.path.join(data, n)).read()
tree = P.parse_newick(rd("mammals_tree.txt"))
species, ids, counts = P.read_family_table(rd("mammal_gene_families.txt"))
pb = P.build_problem(tree, species, ids, counts)
prior = P.prior_uniform(pb.max_root_family_size)
probs, mult = discrete_gamma(4, 2.0)
ctx = capi.Context(pb, max_categories=4)
print("order %d, base: K1 min %.4f ms, median %.4f ms (%d matrices)" % ((pb.matrix_size,) + k1(ctx, P.Params(lambdas=np.array([0.01]), prior=prior), 1.0)), flush=True)
print("order %d, K=4:  K1 min %.4f ms, median %.4f ms (%d matrices)" % ((pb.matrix_size,) + k1(ctx, P.Params(lambdas=np.array([0.005]), prior=prior, multipliers=mult, cat_probs=probs), 2.0)), flush=True)
ctx.close()
