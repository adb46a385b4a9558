"""Pins the CPU oracle (oracle/cafe_oracle.c) to the reference:
  (1) the reference's own known-answer tests (values and tolerances from test.cpp, cited per test);
  (2) golden vectors printed by the real reference compiled in the build container
      (tests/golden/ref_golden.json, made by tests/golden/make_golden.py).
CPU only; no product code is exercised here except the input flattening in cafexp_amd/problem.py.
"""
import math

import os

import numpy as np
import pytest

from cafexp_amd import problem as P
from helpers import case_from_args, rel_err

TIGHT = 1e-12       # oracle vs real reference, same libm: expected identical to the last bits


# ---------------------------------------------------------------- reference known answers (test.cpp)
def test_probability_of_some_values(oracle):                    # test.cpp:601-612
    assert oracle.bd_prob(0.05, 5, 5, 9) == pytest.approx(0.0152237, abs=1e-5)
    assert oracle.bd_prob(0.05, 5, 10, 9) == pytest.approx(0.17573, abs=1e-5)
    assert oracle.bd_prob(0.05, 5, 10, 10) == pytest.approx(0.182728, abs=1e-5)
    assert oracle.bd_prob(0.05, 1, 10, 10) == pytest.approx(0.465565, abs=1e-5)


def test_the_probability_of_going_from_parent_fam_size_to_c(oracle):   # test.cpp:641-644
    assert oracle.bd_prob(.006335, 68.7105, 5, 5) == pytest.approx(0.194661, abs=1e-5)


def test_birthdeath_rate_with_log_alpha(oracle):                # test.cpp:1287-1300
    for s, c, la, co, exp in [(46, 45, -3.672556, 0.949177, -1.55455), (44, 46, -2.617970, 0.854098, -2.20436),
                              (43, 43, -1.686354, 0.629613, -2.39974), (43, 44, -1.686354, 0.629613, -2.44301),
                              (13, 14, -2.617970, 0.854098, -1.58253)]:
        assert math.log(oracle.bd_log_alpha(s, c, la, co)) == pytest.approx(exp, abs=1e-5)
    assert oracle.bd_log_alpha(40, 42, -1.37, 0.5) == pytest.approx(0.107, abs=1e-3)
    assert oracle.bd_log_alpha(41, 34, -1.262, 0.4) == pytest.approx(0.006, abs=1e-3)
    assert oracle.bd_log_alpha(5, 5, -1.1931291703283662, 0.39345841643135504) == pytest.approx(0.194661, abs=1e-4)


def test_probability_of_matrix(oracle):                         # test.cpp:646-663
    expected = np.array([[1, 0, 0, 0, 0], [0.2, 0.64, 0.128, 0.0256, 0.00512], [0.04, 0.256, 0.4608, 0.17408, 0.0512],
                         [0.008, 0.0768, 0.26112, 0.36352, 0.187392], [0.0016, 0.02048, 0.1024, 0.249856, 0.305562]])
    assert np.abs(oracle.build_matrix(5, 0.05, 5) - expected).max() < 1e-5
    assert np.abs(oracle.build_matrix(5, 0.05, 5, fast=True) - expected).max() < 1e-5


def test_matrices_take_fractional_branch_lengths_into_account(oracle):   # test.cpp:631-639
    assert oracle.build_matrix(141, 0.006335, 68.7105)[5, 5] == pytest.approx(0.194661, abs=1e-5)
    assert oracle.build_matrix(141, 0.006335, 68.0)[5, 5] == pytest.approx(0.195791, abs=1e-5)


def test_matrix_multiply(oracle):                               # test.cpp:1390-1400
    m = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=float)
    assert np.allclose(oracle.matvec(m, np.array([7., 9., 11.]), 0, 2, 0, 2), [58, 139, 220], atol=1e-3)


def test_matrix_is_saturated(oracle):                           # test.cpp:1619-1624
    assert oracle.is_saturated(25, 0.05)
    assert not oracle.is_saturated(25, 0.01)


def test_matrix_cache_key_handles_floating_point_imprecision(oracle):    # test.cpp:1271-1285 (lambda slot = t, t slot = 0.3)
    keys, t = set(), 0.0
    for _ in range(31):
        t += 0.1
        keys.add(oracle.quantize(t, 0.3))
    assert len(keys) == 31
    assert oracle.quantize(3.0, 0.3) in keys


def _ab_problem(newick, counts, M, R):
    tree = P.parse_newick(newick)
    species = sorted(counts[0].keys())
    table = np.array([[fam[s] for s in species] for fam in counts], dtype=np.int32)
    return P.build_problem(tree, species, ["f%d" % i for i in range(len(counts))], table, root_filter=False,
                           max_family_size=M, max_root_family_size=R)


def test_infer_processes(oracle):                               # test.cpp:519-547 -> 41.7504 +- 1e-3
    pb = _ab_problem("(A:1,B:1);", [{"A": 1, "B": 2}, {"A": 2, "B": 1}, {"A": 3, "B": 6}, {"A": 6, "B": 3}], 56, 30)
    pr = P.Params(lambdas=np.array([0.01]), prior=P.prior_uniform(30))
    v = oracle.score_base(pb, pr)
    assert v == pytest.approx(41.7504, abs=1e-3)
    assert v == pytest.approx(41.75042830803, rel=1e-12)        # compiled reference


def test_gamma_lambda_optimizer(oracle):                        # test.cpp:2240-2248 -> 6.4168 +- 1e-4
    pb = _ab_problem("(A:1,B:1);", [{"A": 1, "B": 2}], 10, 10)
    probs, mult = oracle.discrete_gamma(4, 0.25)
    pr = P.Params(lambdas=np.array([0.01]), prior=P.prior_uniform(10), multipliers=mult, cat_probs=probs)
    v = oracle.score_gamma(pb, pr)
    assert v == pytest.approx(6.4168, abs=1e-4)
    assert v == pytest.approx(6.4168193056853, rel=1e-12)


def test_prune(oracle):                                         # test.cpp:1642-1663
    pb = _ab_problem("(A:1,B:3):7", [{"A": 3, "B": 6}], 20, 20)
    pr = P.Params(lambdas=np.array([0.03]), prior=P.prior_uniform(20))
    got = oracle.prune(pb, pr, 0, mult=1.5)
    log_expected = [-17.2771, -10.0323, -5.0695, -4.91426, -5.86062, -7.75163, -10.7347, -14.2334, -18.0458, -22.073, -26.2579,
                    -30.5639, -34.9663, -39.4472, -43.9935, -48.595, -53.2439, -57.9338, -62.6597, -67.4173]
    assert np.abs(np.log(got) - log_expected).max() < 1e-4


def test_likelihood_computer_sets_root_nodes_correctly(oracle):    # test.cpp:1709-1743 (lambda 0.03, no multiplier)
    pb = _ab_problem("(A:1,B:3):7", [{"A": 3, "B": 6}], 20, 20)
    pr = P.Params(lambdas=np.array([0.03]), prior=P.prior_uniform(20))
    got = oracle.prune(pb, pr, 0, mult=1.0)
    log_expected = [-19.7743, -11.6688, -5.85672, -5.66748, -6.61256, -8.59725, -12.2301, -16.4424, -20.9882, -25.7574, -30.6888,
                    -35.7439, -40.8971, -46.1299, -51.4289, -56.7837, -62.1863, -67.6304, -73.1106, -78.6228]
    assert np.abs(np.log(got) - log_expected).max() < 1e-4


def test_gamma_model_prune(oracle):                             # test.cpp:1224-1248
    pb = _ab_problem("(A:1,B:3):7", [{"A": 3, "B": 6}], 10, 8)
    rd = [1, 2, 3, 4, 5, 4, 3, 2, 1]                             # root_distribution::vector
    prior = np.zeros(8, dtype=np.float32)
    for j in range(8):
        prior[j] = np.float32(rd[j]) / np.float32(sum(rd))
    pr = P.Params(lambdas=np.array([0.005]), prior=prior, multipliers=np.array([0.1, 0.5]), cat_probs=np.array([0.01, 0.05]))
    v, cat, fam = oracle.score_gamma(pb, pr, per_family=True)
    assert math.log(cat[0, 0]) == pytest.approx(-23.3728, abs=1e-4)
    assert math.log(cat[0, 1]) == pytest.approx(-17.0086, abs=1e-4)


def test_gamma_model_prune_returns_false_if_saturated(oracle):  # test.cpp:1250-1269: lambda 0.9 x {0.1,0.5}
    pb = _ab_problem("(A:1,B:3):7", [{"A": 3, "B": 6}], 10, 8)
    pr = P.Params(lambdas=np.array([0.9]), prior=P.prior_uniform(8), multipliers=np.array([0.1, 0.5]), cat_probs=np.array([1.0, 1.0]))
    assert math.isinf(oracle.score_gamma(pb, pr))


def test_uniform_distribution(oracle):                          # test.cpp:549-556, :2345
    assert oracle.prior_uniform(10)[5] == pytest.approx(0.1, abs=1e-4)
    assert float(oracle.prior_uniform(112)[0]) == 0.0089285718277096748     # float rounding of 1/112


def test_error_model_leaf_taps(oracle):                         # test.cpp:1745-1790: {0.2,0.6,0.2} at sizes 2,3,4
    from helpers import read
    _, dev, dists = P.read_error_model(read("errormodel_small.txt"))
    tab = P.error_model_table(dists, 20)
    assert tab[3].tolist() == [0.2, 0.6, 0.2] and tab[0].tolist() == [0.0, 0.8, 0.2] and tab[12].tolist() == [0.2, 0.6, 0.2]


# ---------------------------------------------------------------- golden vectors from the compiled reference
def test_golden_bd(oracle, golden):
    for e in golden["bd"]:
        assert rel_err(oracle.bd_prob(e["lambda"], e["t"], e["s"], e["c"]), e["value"]) <= TIGHT, e
    for e in golden["bdlog"]:
        assert rel_err(oracle.bd_log_alpha(e["s"], e["c"], e["log_alpha"], e["coeff"]), e["value"]) <= TIGHT, e


def test_golden_keys(oracle, golden):
    for e in golden["keys"]:
        assert oracle.quantize(e["lambda"], e["t"]) == (e["lambda_q"], e["t_q"]), e


def test_golden_matrices(oracle, golden):
    for e in golden["matrices"]:
        for fast, tol in ((False, TIGHT), (True, 2e-11)):       # fast = O(N^2) convolution build (not the reference algorithm)
            m = oracle.build_matrix(e["n"], e["lambda"], e["t"], fast=fast)
            checks = [(m.diagonal(), np.array(e["diag"]))]
            if "full" in e:
                checks.append((m, np.array(e["full"])))
            for r, row in e.get("rows", {}).items():
                checks.append((m[int(r)], np.array(row)))
            for got, exp in checks:
                assert np.array_equal(got == 0, exp == 0)
                big = exp > 1e-290
                assert (np.abs(got - exp)[big] / exp[big]).max(initial=0.0) <= tol, (e["n"], e["lambda"], e["t"], fast)


def test_golden_discrete_gamma(oracle, golden):
    for e in golden["gamma"]:
        probs, mult = oracle.discrete_gamma(e["k"], e["alpha"])
        assert np.array_equal(probs, np.array(e["cat_probs"]))
        assert np.abs(mult / np.array(e["multipliers"]) - 1).max() <= 1e-13, e


def test_golden_prune(oracle, golden):
    from helpers import read
    for e in golden["prune"]:
        counts = dict((kv.split(":")[0], int(kv.split(":")[1])) for kv in e["counts"].split(","))
        pb = _ab_problem(e["newick"], [counts], e["m"], e["r"])
        pr = P.Params(lambdas=np.array([e["lambda"]]), prior=P.prior_uniform(e["r"]))
        if "errfile" in e:
            _, dev, dists = P.read_error_model(read(e["errfile"]))
            pb.n_deviations = len(dev)
            pr.error_model = P.error_model_table(dists, e["m"])
        got = oracle.prune(pb, pr, 0, mult=e["mult"])
        exp = np.array(e["root"])
        assert np.array_equal(got == 0, exp == 0)
        assert (np.abs(got - exp) / np.maximum(exp, 1e-300)).max() <= TIGHT, e["newick"]


@pytest.mark.parametrize("name", [
    "mammals_base_l0.01", "mammals_base_nofilter", "mammals_gamma_k4_a2", "mammals_gamma_k4_inf", "mammals_multilambda_err",
    "mammals_err_poisson10", "mammals_rootdist", "mammals_first200_base", "mammals_first200_gamma", "mammals_first200_err",
    "synth20_base", "synth20_gamma_k8", "synth20_multilambda_err", "synth100_base"])
def test_golden_scores(oracle, golden, name):
    e = golden["scores"][name]
    pb, pr, alpha = case_from_args(e["args"], oracle)
    assert (pb.n_families, pb.max_family_size, pb.max_root_family_size) == (e["n_families"], e["max_family_size"], e["max_root_family_size"])
    if pr.multipliers is not None:
        assert np.abs(pr.multipliers / np.array(e["multipliers"]) - 1).max() <= 1e-13
        v, cat, fam = oracle.score_gamma(pb, pr, per_family=True)
        if "category_likelihood" in e:
            assert np.abs(cat.ravel() / np.array(e["category_likelihood"]) - 1).max() <= TIGHT
            assert np.abs(fam / np.array(e["family_likelihood"]).reshape(cat.shape)[:, 0] - 1).max() <= TIGHT     # one row per (family, category)
    else:
        v, fam = oracle.score_base(pb, pr, per_family=True)
        if "family_lnl" in e:
            assert np.abs(fam / np.array(e["family_lnl"]) - 1).max() <= TIGHT
    assert rel_err(v, e["neg_lnl"]) <= TIGHT, (v, e["neg_lnl"])


def test_gamma_zero_sum_rule_at_the_edge_of_fp64(oracle):
    """gamma_core.cpp:152 rejects a call when a category's root vector sums to exactly 0.  tests/golden/ref_borderline.json
    holds the REAL reference on both sides of that edge (family `border`: at lambda 0.00085 the slowest category's largest
    root entry is 5 denormal units, at 0.0008 it is 0): the restatement must classify every case alike and reproduce the
    finite values."""
    import json
    from helpers import read
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_borderline.json")))
    tree = P.parse_newick(read(g["args"]["tree"]))
    species, ids, counts = P.read_family_table(read(g["args"]["families"]))
    pb = P.build_problem(tree, species, ids, counts)
    probs, mult = oracle.discrete_gamma(g["args"]["k"], g["args"]["alpha"])
    seen_inf = seen_denormal = False
    for lam, e in g["cases"].items():
        pr = P.Params(lambdas=np.array([float(lam)]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
        want = float(e["neg_lnl"])
        got = oracle.score(pb, pr)
        if np.isinf(want):
            assert got == want
            seen_inf = True
            continue
        assert abs(got - want) <= 1e-12 * abs(want)
        _, cat, fam = oracle.score_gamma(pb, pr, per_family=True)
        wc = np.array(e["category_likelihood"], dtype=float).reshape(cat.shape)
        big = wc > 1e-290
        assert np.allclose(cat[big], wc[big], rtol=1e-10)
        assert np.all((cat[~big] == 0) == (wc[~big] == 0)) or np.all(np.abs(cat[~big] - wc[~big]) <= 1e-320)    # denormal range: few bits
        seen_denormal = seen_denormal or bool(np.any((wc > 0) & (wc < 2.3e-308)))
    assert seen_inf and seen_denormal
