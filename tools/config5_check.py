"""Scratch: BASELINE config 5 shape (multi-lambda tree + error model, base model, 100k families) on the GPU."""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from oracle import oracle as O
F = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pb, tree = synth.make_problem(n_families=F, lambda_clade_min=10, n_deviations=3)
print(pb.n_families, pb.n_lambdas, pb.single_lambda, pb.max_family_size, np.bincount(pb.lambda_index))
em = P.error_model_table(P.default_error_model(pb.max_family_size)[:1] + [[0.05, 0.9, 0.05]], pb.max_family_size)
pr = P.Params(lambdas=np.array([0.002, 0.004]), prior=P.prior_uniform(pb.max_root_family_size), error_model=em)
ctx = capi.Context(pb)
for i in range(3):
    t0 = time.time(); v = ctx.score(pr); dt = time.time() - t0
    st = ctx.stats()
    print("score", repr(v), "sec", dt, "fam/s", pb.n_families / dt, {k: round(st[k], 2) for k in ("ms_matrices", "ms_prune", "ms_gemm", "ms_reduce")}, "gemm TF", st["gemm_flops"] / st["ms_gemm"] / 1e9)
res = ctx.family_results(0)
sel = np.array([0, 1, 2, 3, 5, 8, 13, 21, 34, 55])
sub = dataclasses.replace(pb, counts=pb.counts[sel].copy(), family_ids=[pb.family_ids[i] for i in sel])
v, fam = O.score_base(sub, pr, fast=True, per_family=True)
print("family lnL max rel vs oracle", np.abs(res["family_lnl"][sel] / fam - 1).max())
