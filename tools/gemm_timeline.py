"""Scratch: per-block timeline of the last K2 launch (CAFE_GEMM_STAMPS=1)."""
import os, sys
os.environ["CAFE_GEMM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from cafexp_amd.gamma_rates import discrete_gamma
pb, _ = synth.make_problem(n_families=50000)
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
import time
for _ in range(3):
    t=time.time(); ctx.score(pr, alpha=2.0); print('call s', time.time()-t, ctx.stats()['ms_gemm'])
nblk = 8 * 392 * 6 * 8
w = ctx.debug_stamps(6 * nblk).reshape(-1, 6)
w = w[w[:, 5] > 0]
w = w[w[:, 2] >= w[:, 2].max() - 1500000]   # keep the stamped launch only (15 ms window)
hw, xcc, t0, t1, t2, t3 = (w[:, i].astype(np.int64) for i in range(6))
print("blocks recorded", len(w))
base = t0.min()
dur = (t3 - t0) / 100.0
print("block duration us: mean %.1f p10 %.1f p50 %.1f p90 %.1f" % (dur.mean(), *np.percentile(dur, [10, 50, 90])))
print("prologue us  mean %.2f p50 %.2f p90 %.2f" % (((t1 - t0) / 100).mean(), *np.percentile((t1 - t0) / 100, [50, 90])))
print("mainloop us  mean %.2f p50 %.2f p90 %.2f" % (((t2 - t1) / 100).mean(), *np.percentile((t2 - t1) / 100, [50, 90])))
print("epilogue us  mean %.2f p50 %.2f p90 %.2f" % (((t3 - t2) / 100).mean(), *np.percentile((t3 - t2) / 100, [50, 90])))
print("launch span us", (t3.max() - base) / 100.0)
cu = (xcc & 0xF) * 4096 + ((hw >> 8) & 0xF) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7)
slot = hw & 0xF
print("distinct CUs", len(np.unique(cu)), "wave slots seen", np.unique(slot))
gaps = []
conc = []
for c in np.unique(cu)[:64]:
    m = cu == c
    for s in np.unique(slot[m]):
        mm = m & (slot == s)
        order = np.argsort(t0[mm])
        a0, a3 = t0[mm][order], t3[mm][order]
        gaps.extend(((a0[1:] - a3[:-1]) / 100.0).tolist())
gaps = np.array(gaps)
print("gap between consecutive blocks in the same (CU, slot) us: mean %.2f p50 %.2f p90 %.2f n %d" % (gaps.mean(), *np.percentile(gaps, [50, 90]), len(gaps)))
# phase relation of co-resident blocks: for each block, is another block of the same CU in its main loop while this one is in epilogue?
c0 = np.unique(cu)[3]
m = cu == c0
o = np.argsort(t0[m])
for i in o[:12]:
    print("cu", c0, "slot", slot[m][i], "t0 %.1f t1 %.1f t2 %.1f t3 %.1f" % tuple((x[m][i] - base) / 100.0 for x in (t0, t1, t2, t3)))

# --- dispatch placement: does blockIdx.x & 7 pick the XCD?
allw = ctx.debug_stamps(6 * nblk).reshape(-1, 6)
gx = 8 * 49 * 5          # gridDim.x of an MI=9 launch at 50k families
idxs = np.nonzero(allw[:, 5] > 0)[0]
bx = idxs % gx
xcc_id = (allw[idxs, 1].astype(np.int64)) & 0xF
import collections
tab = collections.Counter(zip((bx & 7).tolist(), xcc_id.tolist()))
print("(blockIdx.x & 7, XCC_ID) -> blocks:", sorted(tab.items())[:24], "... distinct pairs", len(tab))
