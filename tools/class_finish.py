import os, sys, collections
os.environ["CAFE_GEMM_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import problem as P, capi, synth
from cafexp_amd.gamma_rates import discrete_gamma
pb, _ = synth.make_problem(n_families=50000)
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
ctx = capi.Context(pb, max_categories=8)
for _ in range(3):
    ctx.score(pr, alpha=2.0)
w = ctx.debug_stamps(6 * 1024).reshape(-1, 6)
idx = np.nonzero(w[:, 5] > 0)[0]
w = w[idx]
t0, t3 = w[:, 2].astype(np.int64), w[:, 5].astype(np.int64)
end = (t3 - t0.min()) / 100.0
cls = (idx >> 3) >> 5
kt = w[:, 4].astype(np.int64) >> 20
nt = w[:, 4].astype(np.int64) & 0xFFFFF
print("launch", os.environ.get("CAFE_GEMM_STAMPS_LAUNCH"), "span %.0f us" % end.max(), "finish by class (median, max) / K tiles (mean) / tiles:",
      [(int(c), round(float(np.median(end[cls == c])) / end.max(), 3), round(float(end[cls == c].max()) / end.max(), 3), int(kt[cls == c].mean()), round(float(nt[cls == c].mean()), 1)) for c in range(4)])
