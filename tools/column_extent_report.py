"""How much of K2's work is the price of tiling?  Per interior node of the bench problem: the mean width of the per-column
zero extents (the least any tiling could stage), of the hulls of the 128-column tiles as the columns are ordered now, and of
the hulls if the columns were ordered by their extents instead (lo, then hi)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import capi, problem as P, synth
from cafexp_amd.gamma_rates import discrete_gamma
pb, _ = synth.make_problem(n_families=int(sys.argv[1]) if len(sys.argv) > 1 else 50000)
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
ctx.score(pr, alpha=2.0)
M = pb.max_family_size
inner = [v for v in range(pb.n_nodes) if pb.parent[v] >= 0 and pb.leaf_taxon[v] < 0]
tot = np.zeros(4)
def hull16(lo, hi):          # K tiles (16 rows) a tile with this hull stages
    return np.where(hi >= lo, hi // 16 - lo // 16 + 1, 1)
for v in inner:
    for k in (0, 3, 7):
        ce = ctx.column_extents(v, k).astype(np.int64)
        n = len(ce) // 128 * 128
        lo, hi = ce[:n, 0].clip(0, M), ce[:n, 1].clip(-1, M)
        live = hi >= lo
        per_col = hull16(lo, hi).mean()
        t_lo = np.where(live, lo, 1 << 30).reshape(-1, 128).min(axis=1)
        t_hi = np.where(live, hi, -1).reshape(-1, 128).max(axis=1)
        now = hull16(t_lo, t_hi).mean()
        order = np.lexsort((hi, lo))
        s_lo = np.where(live, lo, 1 << 30)[order].reshape(-1, 128).min(axis=1)
        s_hi = np.where(live, hi, -1)[order].reshape(-1, 128).max(axis=1)
        by_ext = hull16(s_lo, s_hi).mean()
        order2 = np.lexsort((lo, hi))
        s_lo2 = np.where(live, lo, 1 << 30)[order2].reshape(-1, 128).min(axis=1)
        s_hi2 = np.where(live, hi, -1)[order2].reshape(-1, 128).max(axis=1)
        by_hi = hull16(s_lo2, s_hi2).mean()
        tot += np.array([per_col, now, by_ext, by_hi]) * n
        if k == 3 and v % 7 == 0:
            print("node %3d cat %d cols %6d: K tiles per column %.1f | per tile now %.1f | sorted by (lo,hi) %.1f | by (hi,lo) %.1f" % (v, k, n, per_col, now, by_ext, by_hi), flush=True)
w = sum(len(ctx.column_extents(v, 0)) // 128 * 128 for v in inner) * 3
print("column-weighted over all nodes and categories 0/3/7 (K tiles, of %d): per column %.2f, tiles now %.2f, ordered by (lo,hi) %.2f, by (hi,lo) %.2f"
      % ((M + 16) // 16, tot[0] / w, tot[1] / w, tot[2] / w, tot[3] / w))
