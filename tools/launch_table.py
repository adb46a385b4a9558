"""Per K2 launch of one scorer call at the bench shape (or a shard of it): columns, tile height, executed share of the K
tiles, duration (HIP events on the dispatch) and rate over executed flops.  Usage: launch_table.py [R/W]"""
import os, sys, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from cafexp_amd import capi, problem as P, synth
from cafexp_amd.gamma_rates import discrete_gamma
r, w = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0/1").split("/"))
pb, _ = synth.make_problem(n_families=50000)
if w > 1:
    mine = capi.shard_plan(pb, w, 8)[r]
    pb = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[mine]), family_ids=[pb.family_ids[i] for i in mine])
probs, mult = discrete_gamma(8, 2.0)
pr = P.Params(lambdas=np.array([0.002]), prior=P.prior_uniform(750), multipliers=mult, cat_probs=probs)
os.environ["CAFE_DUMP_LAUNCH_MS"] = "1"
ctx = capi.Context(pb, max_categories=8)
ctx.set_profiling(True)
for _ in range(3):
    ctx.score(pr, alpha=2.0)
ex, al, mi = ctx.launch_flops()
st = ctx.stats()
ms = np.zeros(len(ex))
lib = capi.load()
lib.cafe_debug_launch_ms.restype = C.c_int
lib.cafe_debug_launch_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
lib.cafe_debug_launch_ms(ctx._h, ms.ctypes.data_as(C.POINTER(C.c_double)), len(ms))
cols = al / (2.0 * 720 * 721 * 8)
order = np.argsort(cols)
print("launch  ~cols  mi  executed/all  ms      TF/s(executed)")
for i in order:
    print("%4d %7.0f  %d   %.3f       %.4f  %.1f" % (i, cols[i], mi[i], ex[i] / al[i], ms[i], ex[i] / ms[i] / 1e9 if ms[i] > 0 else 0))
print("total: %.2f ms, %.1f TF/s over executed flops, executed share %.3f" % (ms.sum(), ex.sum() / ms.sum() / 1e9, ex.sum() / al.sum()))
for lo, hi in ((0, 1024), (1024, 8192), (8192, 32768), (32768, 1 << 30)):
    m = (cols >= lo) & (cols < hi)
    if m.any():
        print("cols [%d, %d): %d launches, %.2f ms, %.1f TF/s, executed share %.3f" % (lo, hi, m.sum(), ms[m].sum(), ex[m].sum() / ms[m].sum() / 1e9, ex[m].sum() / al[m].sum()))
