"""How tight are the zero extents?  For a few interior branches of the bench problem: the matrix's exact non-zero band
(from cafe_get_matrix) against the block extents K1 published, and the panel tile extents of the node."""
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
for v in inner[::12]:
    for k in (0, 7):
        Pm = ctx.matrix(v, k)[1:M + 1, :M + 1]
        nz = Pm != 0
        ext, pt = ctx.extents(v, k)
        exact = []
        for b in range(len(ext)):
            blk = nz[16 * b:16 * b + 16]
            cols = np.nonzero(blk.any(axis=0))[0]
            exact.append((cols[0], cols[-1]) if len(cols) else (1 << 30, -1))
        exact = np.array(exact)
        w_ext = np.maximum(0, ext[:, 1] - ext[:, 0] + 1).mean()
        w_exact = np.maximum(0, exact[:, 1] - exact[:, 0] + 1).mean()
        ok = np.all((ext[:, 0] <= exact[:, 0]) & (ext[:, 1] >= exact[:, 1]) | (exact[:, 1] < 0))
        line = "node %3d t %.3f cat %d: matrix band per 16-row block: published %.0f wide, exact %.0f wide, conservative %s" % (v, pb.branch_length[v], k, w_ext, w_exact, ok)
        if pt is not None:
            w = np.maximum(0, pt[:, 1] - pt[:, 0] + 1)
            line += " | panel: %d tiles, extent width mean %.0f (min %d max %d), lo mean %.0f" % (len(pt), w.mean(), w.min(), w.max(), pt[:, 0].clip(0, M).mean())
        print(line, flush=True)
