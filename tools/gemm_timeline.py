"""Scratch: per-workgroup lifetime and placement of the last K2 launch (CAFE_GEMM_STAMPS=1).

The persistent K2 kernel writes one record per workgroup: HW_ID, XCC_ID, start, epilogue | K-loop ticks, tiles | K tiles, end
(100 MHz ticks).
Reports how evenly the workgroups finish (tail of the launch) and whether blockIdx.x & 7 picks the XCD.
"""
import os, sys, collections
os.environ["CAFE_GEMM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from cafexp_amd.gamma_rates import discrete_gamma

pb, _ = synth.make_problem(n_families=50000)
if len(sys.argv) > 1:                                   # R/W: one shard of the library's plan
    import dataclasses
    r_, w_ = (int(x) for x in sys.argv[1].split("/"))
    mine = capi.shard_plan(pb, w_, 8)[r_]
    pb = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[mine]), family_ids=[pb.family_ids[i] for i in mine])
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
for _ in range(3):
    ctx.score(pr, alpha=2.0)
    print("ms_gemm", ctx.stats()["ms_gemm"])
nblk = 1024
w = ctx.debug_stamps(6 * nblk).reshape(-1, 6)
idx = np.nonzero(w[:, 5] > 0)[0]
w = w[idx]
hw, xcc, t0, t3 = (w[:, i].astype(np.int64) for i in (0, 1, 2, 5))
# word 3: epilogue ticks (low 32 bits) | K-loop ticks << 32; word 4: tiles (low 20 bits) | K tiles << 20 (tools/k2_tile_costs.py)
ep, nt = w[:, 3].astype(np.int64) & 0xFFFFFFFF, w[:, 4].astype(np.int64) & 0xFFFFF
print("tiles per workgroup: min %d max %d; epilogue us per tile (issue to last store issued): mean %.2f p90 %.2f; share of lifetime %.2f%%"
      % (nt.min(), nt.max(), (ep / np.maximum(nt, 1)).mean() / 100.0, np.percentile(ep / np.maximum(nt, 1), 90) / 100.0, 100.0 * ep.sum() / (t3 - t0).sum()))
print("workgroups recorded", len(w))
base = t0.min()
print("start spread us %.2f" % ((t0.max() - base) / 100.0))
life = (t3 - t0) / 100.0
print("lifetime us: min %.1f p50 %.1f max %.1f" % (life.min(), np.percentile(life, 50), life.max()))
end = (t3 - base) / 100.0
print("finish us: min %.1f p50 %.1f max %.1f  (launch span %.1f, idle tail of the median block %.1f%%)"
      % (end.min(), np.percentile(end, 50), end.max(), end.max(), 100 * (1 - np.percentile(end, 50) / end.max())))
cu = (xcc & 0xF) * 4096 + ((hw >> 8) & 0xF) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7)
print("distinct CUs", len(np.unique(cu)), "workgroups per CU", collections.Counter(collections.Counter(cu.tolist()).values()))
tab = collections.Counter(zip((idx & 7).tolist(), (xcc & 0xF).tolist()))
print("(blockIdx.x & 7, XCC_ID) -> workgroups:", sorted(tab.items()), "distinct pairs", len(tab))
x_of = (xcc & 0xF)
print("finish us by XCD (min / median / max):", [(int(x), round(float(end[x_of == x].min()), 0), round(float(np.median(end[x_of == x])), 0), round(float(end[x_of == x].max()), 0)) for x in np.unique(x_of)])
se = (hw >> 12) & 1
print("by shader-engine bit of HW_ID: median finish", [round(float(np.median(end[se == b])), 0) for b in (0, 1)])
pairs = collections.defaultdict(list)
for c_, e_ in zip(cu.tolist(), end.tolist()):
    pairs[c_].append(e_)
d = np.array([abs(v[0] - v[1]) for v in pairs.values() if len(v) == 2])
print("two workgroups of a CU finish %.1f us apart on average (all pairs of workgroups: %.1f)" % (d.mean(), np.abs(end[:, None] - end[None, :]).mean()))
locs = collections.defaultdict(list)
for c_, i_ in zip(cu.tolist(), idx.tolist()):
    locs[c_].append(i_ >> 3)
diffs = collections.Counter(abs(v[0] - v[1]) for v in locs.values() if len(v) == 2)
print("local index distance of the two workgroups of a CU:", diffs.most_common(5))
cu_end = np.array([max(v) for v in pairs.values()])
print("per-CU finish (later of its two workgroups) us: min %.1f p50 %.1f max %.1f; earlier one: p50 %.1f" % (cu_end.min(), np.median(cu_end), cu_end.max(), np.median([min(v) for v in pairs.values()])))
first_lower = 0
for c_ in pairs:
    (e0, e1), (l0, l1) = pairs[c_], locs[c_]
    if len(pairs[c_]) == 2:
        first_lower += (e0 < e1) == (l0 < l1)
print("CUs whose lower-index workgroup finishes first: %d of %d" % (first_lower, len(pairs)))
