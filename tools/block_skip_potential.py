"""How many (16-row block, 8-deep K tile) MFMA groups K2 would run if each row block of a tile skipped the K tiles outside ITS OWN
matrix extent, against what it runs now (the hull over the tile's 5 blocks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import capi, problem as P, synth
from cafexp_amd.gamma_rates import discrete_gamma
pb, _ = synth.make_problem(n_families=50000)
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
ctx.score(pr, alpha=2.0)
KB, MI = 8, 5
root = int(np.nonzero(pb.parent < 0)[0][0])
tot_hull = tot_blk = 0
for cat in (0, 3, 7):
    for v in range(pb.n_nodes):
        if pb.leaf_taxon[v] >= 0 or v == root:
            continue
        m, pt = ctx.extents(v, cat)
        if pt is None:
            continue
        nb = len(m)
        blo, bhi = pt[:, 0].astype(np.int64), pt[:, 1].astype(np.int64)
        ok = bhi >= blo
        h = b = 0
        for rt in range(0, nb, MI):
            blk = m[rt:rt + MI]
            live = blk[:, 1] >= blk[:, 0]
            if live.any():
                lo, hi = blk[live, 0].min(), blk[live, 1].max()
                l2, h2 = np.maximum(lo, blo), np.minimum(hi, bhi)
                kt = np.where(ok & (h2 >= l2), h2 // KB - l2 // KB + 1, 1)
            else:
                kt = np.ones(len(blo), dtype=np.int64)
            h += int(kt.sum()) * len(blk)
            for i in range(len(blk)):
                if not live[i]:
                    continue
                l2, h2 = np.maximum(blk[i, 0], blo), np.minimum(blk[i, 1], bhi)
                b += int(np.where(ok & (h2 >= l2), h2 // KB - l2 // KB + 1, 0).sum())
        tot_hull += h; tot_blk += b
    print("cat", cat, "hull", tot_hull, "per block", tot_blk, "ratio %.3f" % (tot_blk / tot_hull), flush=True)
