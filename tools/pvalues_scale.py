"""Scratch: device-side p-values at the bench's config-4 shape (100 taxa, N = 751): R x 1000 simulated families."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cafexp_amd import capi, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
nsim = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
pb, _ = synth.make_problem(n_families=F)
ctx = capi.Context(pb)
for rep in range(4):
    t = time.time(); pv = ctx.pvalues(np.array([0.002]), n_simulations=nsim, seed=3 + rep); dt = time.time() - t
    print("pvalues: %.3f s for %d observed + %d x %d simulated families; significant at 0.05: %d; mean p %.3f"
          % (dt, F, pb.max_root_family_size, nsim, int((pv < 0.05).sum()), pv.mean()))
