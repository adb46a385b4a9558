"""Scratch: device-side p-values against the reference's golden p-values (statistical agreement) and timing."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import case_from_args
from oracle import oracle as O
from cafexp_amd import capi
gp = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_pvalues.json")))["cases"]
for name in ("mammals", "mammals_lambda_tree", "synth20"):
    e = gp[name]
    pb, pr, _ = case_from_args(e["args"], O)
    ctx = capi.Context(pb)
    t = time.time(); got = ctx.pvalues(pr.lambdas, n_simulations=e["nsim"], seed=12345); dt = time.time() - t
    t = time.time(); got2 = ctx.pvalues(pr.lambdas, n_simulations=e["nsim"], seed=777); dt2 = time.time() - t
    want = np.array(e["pvalues"])
    d = got - want
    print(name, "n", len(want), "nsim", e["nsim"], "sec %.3f %.3f" % (dt, dt2), "mean diff %.4f mean|d| %.4f max|d| %.3f" % (d.mean(), np.abs(d).mean(), np.abs(d).max()),
          "sig<0.05: got %d want %d" % ((got < 0.05).sum(), (want < 0.05).sum()), "seed-to-seed mean|d| %.4f max %.3f" % (np.abs(got - got2).mean(), np.abs(got - got2).max()))
