"""Scratch: cafe_create wall time (family and subtree de-duplication, tables, pools) at the bench shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cafexp_amd import capi, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
pb, _ = synth.make_problem(n_families=F)
for sd in (True, False, True):
    t = time.perf_counter(); ctx = capi.Context(pb, max_categories=8, subtree_dedup=sd); dt = time.perf_counter() - t
    print("subtree_dedup", sd, "cafe_create %.3f s" % dt)
    del ctx
