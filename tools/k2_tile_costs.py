"""Where a K2 workgroup's time goes (CAFE_GEMM_STAMPS=1: the last K2 launch of a call -- the root's second child, a
full-width launch): per workgroup the time inside the K loops, inside the epilogues, and the rest (tile hand-over, list and
descriptor fetch), per output tile and per K tile.  Usage: k2_tile_costs.py [R/W]"""
import os, sys
os.environ["CAFE_GEMM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from cafexp_amd.gamma_rates import discrete_gamma
pb, _ = synth.make_problem(n_families=50000)
if len(sys.argv) > 1:
    import dataclasses
    r_, w_ = (int(x) for x in sys.argv[1].split("/"))
    mine = capi.shard_plan(pb, w_, 8)[r_]
    pb = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[mine]), family_ids=[pb.family_ids[i] for i in mine])
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
for _ in range(3):
    ctx.score(pr, alpha=2.0)
w = ctx.debug_stamps(6 * 1024).reshape(-1, 6)
w = w[w[:, 5] > 0]
life = (w[:, 5].astype(np.int64) - w[:, 2].astype(np.int64)) / 100.0            # us
ep = (w[:, 3].astype(np.int64) & 0xFFFFFFFF) / 100.0
loop = (w[:, 3].astype(np.int64) >> 32) / 100.0
tiles = (w[:, 4].astype(np.int64) & 0xFFFFF).astype(float)
kt = (w[:, 4].astype(np.int64) >> 20).astype(float)
rest = life - ep - loop
print("workgroups %d, tiles per workgroup %.1f, K tiles per tile %.1f, lifetime us p50 %.1f max %.1f" % (len(w), tiles.mean(), kt.sum() / tiles.sum(), np.median(life), life.max()))
print("per K tile inside the loops: %.3f us;  per output tile: epilogue %.2f us, rest (hand-over, fetches) %.2f us, K loops %.2f us"
      % (loop.sum() / kt.sum(), ep.sum() / tiles.sum(), rest.sum() / tiles.sum(), loop.sum() / tiles.sum()))
print("share of the workgroups' lifetime: K loops %.1f %%, epilogues %.1f %%, rest %.1f %%" % (100 * loop.sum() / life.sum(), 100 * ep.sum() / life.sum(), 100 * rest.sum() / life.sum()))
print("in K tiles of loop time: epilogue %.2f, rest %.2f per output tile" % (ep.sum() / tiles.sum() / (loop.sum() / kt.sum()), rest.sum() / tiles.sum() / (loop.sum() / kt.sum())))
# finish times by dispatch order: workgroups j, j + 32, j + 64 of an XCD (local index = blockIdx.x >> 3) share a CU
raw = ctx.debug_stamps(6 * 1024).reshape(-1, 6)
idx = np.nonzero(raw[:, 5] > 0)[0]
raw = raw[idx]
t0 = raw[:, 2].astype(np.int64)
end = (raw[:, 5].astype(np.int64) - t0.min()) / 100.0
local = idx >> 3
n_per = max(1, (local.max() + 1 + 2) // 3) if local.max() >= 64 else 32
for g in range((local.max() // 32) + 1):
    m = (local // 32) == g
    if m.any():
        print("local index %2d..%2d: %3d workgroups, finish us min %.0f median %.0f max %.0f; tiles %.1f, K tiles %.0f, K-tile time in loops %.3f us"
              % (32 * g, 32 * g + 31, m.sum(), end[m].min(), np.median(end[m]), end[m].max(), tiles[m].mean(), kt[m].mean(), loop[m].sum() / kt[m].sum()))
print("launch span %.0f us; mean finish %.0f; sum of K tiles per workgroup: min %.0f max %.0f" % (end.max(), end.mean(), kt.min(), kt.max()))
