"""Scratch: Pupko reconstruction and p-value root maxima at the bench's config-4 shape (100 taxa, M=720), timing the
device path and spot-checking it against the oracle on a few families."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from oracle import oracle as O

F = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
pb, _ = synth.make_problem(n_families=F)
lam = np.array([0.002])
jmax = min(pb.max_family_size, pb.max_root_family_size)
rp = np.zeros(jmax + 1, dtype=np.float32)
rp[:pb.max_root_family_size] = P.prior_uniform(pb.max_root_family_size)[:jmax + 1]
ctx = capi.Context(pb)
for rep in range(2):
    t = time.time(); st = ctx.reconstruct(lam, rp); dt = time.time() - t
    nI = int((pb.leaf_taxon < 0).sum())
    elems = (nI - 1) * (pb.max_family_size + 1) ** 2 * ctx.stats()["n_unique_families"]
    print("reconstruct: %.3f s for %d families (%d unique), %.2f T (i,j,f) elements/s in the max-product kernel bound" % (dt, F, ctx.stats()["n_unique_families"], elems / dt / 1e12))
t = time.time(); rm = ctx.root_max(lam); print("root_max: %.3f s" % (time.time() - t))
t = time.time(); bp = ctx.branch_probabilities(lam, st[0]); print("branch_probabilities: %.3f s, valid %.1f%%" % (time.time() - t, 100 * np.mean(~np.isnan(bp))))
# spot check
idx = np.array([0, 1, 2, F // 2, F - 1])
sub = P.Problem(parent=pb.parent, branch_length=pb.branch_length, lambda_index=pb.lambda_index, leaf_taxon=pb.leaf_taxon,
                counts=np.ascontiguousarray(pb.counts[idx]), max_family_size=pb.max_family_size, max_root_family_size=pb.max_root_family_size,
                taxa=pb.taxa, family_ids=[pb.family_ids[i] for i in idx], node_names=pb.node_names)
want = O.reconstruct(sub, lam, rp, fast=True)[0]
print("states equal to the oracle on the sample:", np.array_equal(st[0][idx], want), int((st[0][idx] != want).sum()), "entries differ")
wm = O.root_max(sub, lam, fast=True)
print("root_max rel err on the sample:", float(np.max(np.abs(rm[idx] / wm - 1))))
wb = O.branch_probabilities(sub, lam, want, fast=True)
m = ~np.isnan(wb)
print("branch prob max rel err on the sample:", float(np.max(np.abs(bp[idx][m] - wb[m]) / np.maximum(wb[m], 1e-300))) if np.array_equal(st[0][idx], want) else "n/a")
