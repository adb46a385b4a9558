"""P-value path (src/probability.cpp:255-454).

CPU: the oracle's restatement (root maximum, pvalue, tree p-value) against the reference's known-answer tests
(test.cpp:1175, :1185, :2229) and the compiled reference's outputs at a fixed seed (tests/golden/ref_pvalues.json,
written by tests/golden/make_golden.py pvalues).
GPU: cafe_root_max against the oracle, and the cafexp_hip driver's compute_pvalues -- host draws that consume the
engine like the reference, both prune batches on the GPU -- against the same golden p-values and conditional
distributions.
"""
import json
import os
import subprocess

import numpy as np
import pytest

from helpers import case_from_args
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cafexp_amd", "host")
DATA = os.path.join(ROOT, "tests", "golden", "data")


@pytest.fixture(scope="module")
def gp():
    with open(os.path.join(ROOT, "tests", "golden", "ref_pvalues.json")) as f:
        return json.load(f)["cases"]


def _problem(e):
    pb, pr, _ = case_from_args(e["args"], O)
    if "m" in e["args"]:
        pb.max_family_size, pb.max_root_family_size = int(e["args"]["m"]), int(e["args"]["r"])
    assert (pb.n_families, pb.max_family_size, pb.max_root_family_size) == (e["n_families"], e["max_family_size"], e["max_root_family_size"])
    return pb, pr


def _cond(e):
    return np.array(e["cond"]).reshape(e["ncond"], e["nsim"])


# ------------------------------------------------------------------------------------------------ CPU
def test_oracle_pvalue_known_answers():
    cd = np.cumsum(np.full(10, 0.01))                       # test.cpp:1175-1183
    assert abs(O.pvalue(0.05, cd) - 0.5) < 1e-3
    assert abs(O.pvalue(0.0001, cd) - 0.0) < 1e-3
    assert abs(O.pvalue(0.099, cd) - 0.9) < 1e-3
    assert O.pvalue(5.0, cd) == 0.9                         # past the end: index size-1 (probability.cpp:381)
    cond = np.tile(cd, (10, 1))                             # test.cpp:1185-1202
    assert abs(O.tree_pvalues([0.05], cond)[0] - 0.5) < 1e-3


def test_oracle_reproduces_reference_pvalue_of_its_own_test(gp):
    """test.cpp:2229: tree (A:1,B:1), lambda 0.05, M=10, R=8, family A=1 B=2, 3 simulations at seed 10 -> 0.666667.
    The golden case holds all R conditional distributions, so the deterministic half is checked end to end."""
    e = gp["test2229"]
    assert abs(e["pvalues"][0] - 0.666667) < 1e-5
    pb, pr = _problem(e)
    obs = O.root_max(pb, pr.lambdas)
    assert O.tree_pvalues(obs, _cond(e)).tolist() == e["pvalues"]


@pytest.mark.parametrize("name", ["test2229", "mammals", "mammals_lambda_tree", "synth20"])
def test_oracle_root_max_matches_reference_on_extinct_families(gp, name):
    """Root size 0 simulates only all-zero families: every entry of the reference's conditional distribution 0 is
    max_j L_root[j] of that one family."""
    e = gp[name]
    pb, pr = _problem(e)
    pb.counts = np.zeros((1, pb.n_taxa), dtype=np.int32)
    pb.family_ids = ["zero"]
    c0 = _cond(e)[0]
    assert c0.min() == c0.max()
    assert abs(O.root_max(pb, pr.lambdas)[0] / c0[0] - 1) < 1e-12


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mammals", "mammals_lambda_tree", "synth20"])
def test_root_max_matches_oracle(gp, name):
    from cafexp_amd import capi
    pb, pr = _problem(gp[name])
    ctx = capi.Context(pb)
    got = ctx.root_max(pr.lambdas)
    want = O.root_max(pb, pr.lambdas)
    assert np.max(np.abs(got / want - 1)) < 1e-11
    # a scorer call and a root-max call may alternate on one context
    a = ctx.score(pr)
    assert np.array_equal(ctx.root_max(pr.lambdas), got)
    assert ctx.score(pr) == a


@pytest.mark.gpu
def test_root_max_ignores_the_error_model(gp, golden):
    """compute_pvalues prunes with a NULL error model even when the model has one (probability.cpp:429, :443)."""
    from cafexp_amd import capi
    e = golden["scores"]["mammals_multilambda_err"]
    pb, pr, _ = case_from_args(e["args"], O)
    pb.counts = np.ascontiguousarray(pb.counts[:400])
    pb.family_ids = pb.family_ids[:400]
    ctx = capi.Context(pb)
    got = ctx.root_max(pr.lambdas)
    want = O.root_max(pb, pr.lambdas)
    assert np.max(np.abs(got / want - 1)) < 1e-11
    with pytest.raises(capi.CafeError):
        ctx.family_results()                                # no scorer results after a root-max call
    assert np.isfinite(ctx.score(pr))                       # and the error-model scorer path still works afterwards


@pytest.mark.gpu
def test_root_max_rejects_invalid_lambda(gp):
    from cafexp_amd import capi
    pb, pr = _problem(gp["synth20"])
    ctx = capi.Context(pb)
    with pytest.raises(capi.CafeError):
        ctx.root_max(np.array([-0.1]))


def _driver(tmp_path, e, *extra):
    exe = os.path.join(HOST, "cafexp_hip")
    assert os.path.exists(exe), "cafexp_hip missing: run __graft_entry__.build()"
    a = e["args"]
    pv, cd = str(tmp_path / "pv.txt"), str(tmp_path / "cond.txt")
    cmd = [exe, "-t", os.path.join(DATA, a["tree"]), "-i", os.path.join(DATA, a["families"]), "-s", a["seed"],
           "--pvalues", a["nsim"], "--pvalues-out", pv, "--pvalues-cond", "%s:%d" % (cd, e["ncond"])]
    if "lambdas" in a:
        cmd += ["-m", a["lambdas"], "-y", os.path.join(DATA, a["lambda_tree"])]
    else:
        cmd += ["-l", repr(float(a["lambda"]))]
    if "limit" in a:
        cmd += ["--limit", a["limit"]]
    if "m" in a:
        cmd += ["--sizes", "%d,%d" % (a["m"], a["r"])]
    if int(a.get("rootfilter", 1)) == 0:
        cmd += ["-z"]
    out = subprocess.run([str(x) for x in cmd] + list(extra), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    rows = [l.split("\t") for l in open(pv).read().splitlines()[1:]]
    cond = np.array([[float(x) for x in l.split("\t")] for l in open(cd).read().splitlines()])
    return d, np.array([float(r[1]) for r in rows]), cond


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["test2229", "mammals", "mammals_lambda_tree", "synth20"])
def test_driver_pvalues_match_compiled_reference(gp, tmp_path, name):
    """Same seed, same draw sequence as the reference: the simulated families are the same, so the sorted conditional
    distributions agree to rounding and the p-values are the reference's (a draw can only differ where a uniform
    variate falls within ~1e-13 of a cumulative-probability boundary)."""
    e = gp[name]
    d, pv, cond = _driver(tmp_path, e)
    assert d["n_families"] == e["n_families"] and d["pvalues"]["simulations"] == e["nsim"]
    want_c = _cond(e)
    assert cond.shape == want_c.shape
    assert np.max(np.abs(cond / want_c - 1)) < 1e-10
    want = np.array(e["pvalues"])
    assert len(pv) == len(want)
    # observed == simulated value ties are decided by the last bits of two different summation orders
    differ = np.nonzero(pv != want)[0]
    assert len(differ) <= max(1, len(want) // 100), (len(differ), pv[differ][:5], want[differ][:5])
    assert np.max(np.abs(pv - want)) <= 0.05


@pytest.mark.gpu
def test_device_side_simulation_agrees_statistically(gp):
    """cafe_pvalues draws on the device (Philox keyed by the seed): a different sample of the same distribution as the
    reference's, so agreement is statistical.  Seeds are fixed, so the numbers below are reproducible; the bounds leave
    several standard errors of a 1000-sample tail estimate (sqrt(p(1-p)/n) <= 0.016)."""
    from cafexp_amd import capi
    e = gp["mammals"]
    pb, pr = _problem(e)
    ctx = capi.Context(pb)
    got = ctx.pvalues(pr.lambdas, n_simulations=e["nsim"], seed=12345)
    want = np.array(e["pvalues"])
    d = got - want
    assert abs(d.mean()) < 0.003 and np.abs(d).mean() < 0.005 and np.abs(d).max() < 0.08
    assert abs(int((got < 0.05).sum()) - int((want < 0.05).sum())) <= max(3, len(want) // 50)
    assert np.array_equal(got, ctx.pvalues(pr.lambdas, n_simulations=e["nsim"], seed=12345))      # a function of the seed only
    other = ctx.pvalues(pr.lambdas, n_simulations=e["nsim"], seed=777)
    assert not np.array_equal(got, other) and np.abs(got - other).mean() < 0.005
    # the scorer still works on the same context, and so does the lambda-tree case
    assert np.isfinite(ctx.score(pr))
    e2 = gp["mammals_lambda_tree"]
    pb2, pr2 = _problem(e2)
    got2 = capi.Context(pb2).pvalues(pr2.lambdas, n_simulations=1000, seed=5)
    assert np.abs(got2 - np.array(e2["pvalues"])).mean() < 0.02                                  # the reference used 100 simulations here
    with pytest.raises(capi.CafeError):
        ctx.pvalues(pr.lambdas, n_simulations=5000)
