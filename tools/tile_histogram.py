"""Per interior branch of the bench problem: how many (category, row tile, column tile) output tiles of its K2 launch run how
many K tiles (80-row tiles): the shape of the work after the zero extents."""
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
M = pb.max_family_size
MI = 5
inner = [v for v in range(pb.n_nodes) if pb.parent[v] >= 0 and pb.leaf_taxon[v] < 0]
tot = np.zeros(48, dtype=np.int64)
for v in inner:
    hist = np.zeros(48, dtype=np.int64)
    nt = 0
    for k in range(8):
        ext, pt = ctx.extents(v, k)
        if pt is None:
            continue
        nt = len(pt)
        nb = len(ext)
        for rt in range((720 + 16 * MI - 1) // (16 * MI)):
            blk = ext[rt * MI:min(rt * MI + MI, nb)]
            ok = blk[:, 1] >= blk[:, 0]
            alo, ahi = (blk[ok, 0].min(), blk[ok, 1].max()) if ok.any() else (1 << 30, -1)
            lo = np.maximum(alo, pt[:, 0]); hi = np.minimum(np.minimum(ahi, pt[:, 1]), M)
            nkt = np.where(hi >= lo, hi // 16 - lo // 16 + 1, 1)
            hist += np.bincount(nkt, minlength=48)[:48]
    tot += hist
    if nt >= 380:
        n = hist.sum()
        print("node %3d tiles %6d: 1 K tile %4.1f%%, 2-10 %4.1f%%, 11-30 %4.1f%%, 31-46 %4.1f%%, mean %.1f" %
              (v, n, 100 * hist[1] / n, 100 * hist[2:11].sum() / n, 100 * hist[11:31].sum() / n, 100 * hist[31:].sum() / n, (hist * np.arange(48)).sum() / n), flush=True)
n = tot.sum()
print("all launches: tiles %d: 1 K tile %.1f%%, 2-10 %.1f%%, 11-30 %.1f%%, 31-46 %.1f%%, mean %.1f K tiles" %
      (n, 100 * tot[1] / n, 100 * tot[2:11].sum() / n, 100 * tot[11:31].sum() / n, 100 * tot[31:].sum() / n, (tot * np.arange(48)).sum() / n))
