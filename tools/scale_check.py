"""Scratch: config-4 scale run on the GPU + parity of a family subset against the oracle."""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from oracle import oracle as O

F = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
pb, tree = synth.make_problem(n_families=F)
print("gen sec", time.time() - t0, pb.n_families, pb.max_family_size, pb.max_root_family_size)
probs, mult = O.discrete_gamma(K, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
t0 = time.time()
ctx = capi.Context(pb, max_categories=K)
print("create sec", time.time() - t0)
for i in range(3):
    t0 = time.time(); v = ctx.score(pr, alpha=2.0); dt = time.time() - t0
    st = ctx.stats()
    print("score", repr(v), "sec", dt, "fam/s", pb.n_families / dt)
    print({k: (round(x, 3) if isinstance(x, float) else x) for k, x in st.items()})
    print("GEMM TF/s", st["gemm_flops"] / (st["ms_gemm"] * 1e-3) / 1e12)
res = ctx.family_results(K)
sel = np.array([0, 1, 2, 3, 5, 8, 13, 21])
sel = sel[sel < pb.n_families]
sub = dataclasses.replace(pb, counts=pb.counts[sel].copy(), family_ids=[pb.family_ids[i] for i in sel])
t0 = time.time()
v, cat, fam = O.score_gamma(sub, pr, fast=True, per_family=True)
print("oracle subset sec", time.time() - t0)
rel = np.abs(res["category_likelihood"][sel] - cat) / cat
print("cat lik max rel", rel.max(), "fam lik max rel", (np.abs(res["family_likelihood"][sel] - fam) / fam).max())
print("failed any", res["failed"].any(), "finite", np.isfinite(res["family_lnl"]).all())
