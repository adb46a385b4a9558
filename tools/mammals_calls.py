"""BASELINE configs 2 and 3 (mammals fixture, base / gamma K=4) through the C ABI: ms per scorer call with the call's launch
sequence enqueued launch by launch (default) and replayed from its hipGraph.  Run under rocprofv3 --kernel-trace --stats for
the per-kernel picture (profiles/r02_mammals_kernel_stats.csv)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import capi, problem as P
from cafexp_amd.gamma_rates import discrete_gamma

data = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "data")
rd = lambda n: open(os.path.join(data, n)).read()
tree = P.parse_newick(rd("mammals_tree.txt"))
species, ids, counts = P.read_family_table(rd("mammal_gene_families.txt"))
pb = P.build_problem(tree, species, ids, counts)
prior = P.prior_uniform(pb.max_root_family_size)
probs, mult = discrete_gamma(4, 2.0)
cases = [("base lambda=0.01", P.Params(lambdas=np.array([0.01]), prior=prior), 1.0),
         ("gamma K=4 lambda=0.005 alpha=2", P.Params(lambdas=np.array([0.005]), prior=prior, multipliers=mult, cat_probs=probs), 2.0)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for graphs in (True, False):
    ctx = capi.Context(pb, max_categories=4)
    ctx.set_graphs(graphs)
    for name, pr, alpha in cases:
        for _ in range(5):
            v = ctx.score(pr, alpha=alpha)
        t0 = time.perf_counter()
        for _ in range(reps):
            v = ctx.score(pr, alpha=alpha)
        dt = (time.perf_counter() - t0) / reps
        st = ctx.stats()
        print("%-34s graphs=%-5s %.4f ms per call  (%d K2 launches, %d assemble, %d gathered-factor epilogues, %d leaf passes)  -lnL %.10f"
              % (name, graphs, dt * 1e3, st["gemm_launches"], st["n_assemble_passes"], st["n_gather_epilogues"], st["n_leaf_passes"], v), flush=True)
    ctx.close()
