"""Fixtures FROM THE REFERENCE at the matrix order of BASELINE.json's configs 4 and 5 (max count 600 -> M = 720, R = 750,
N = 751; user_data.cpp:45-46, base_model.cpp:77).  tests/golden/ref_n751.json was printed by the compiled reference
(oracle/_ref/ref_harness, generator tests/golden/make_n751_golden.py):
  * seven full rows, three columns and the diagonal of three N = 751 transition matrices at config-4-like (lambda * m_k, t),
    incl. the smallest and the largest multiplier of K = 8 at alpha = 2          (matrix_cache.cpp:121-171)
  * a gamma K = 8 score, per family and category, on a 12-taxon tree whose table holds a 600   (gamma_core.cpp:169-244)
  * a base-model score with a lambda tree (two rates) and a 3-tap error model whose last row differs from the others and
    has a family sitting on it (config 5's shape; probability.cpp:182-193, error_model.cpp:52-57), and the plain base score
  * the bench's OWN 100-taxon tree (config 4's shape exactly: 198 branches x 8 categories of order-751 matrices, half an hour
    of reference time per score) with four families of the bench tables: per family and category at the bench's scoring
    point, and a point where the reference rejects the call because one family's slowest category has an all-zero root
    vector (gamma_core.cpp:152, :227) -- found on the GPU over all 50 000 families (profiles/r03_zero_categories.json).
CPU tests pin the oracle to them; the -m gpu tests compare the HIP path with the same reference outputs."""
import json
import math
import os

import numpy as np
import pytest

from helpers import case_from_args, rel_err

TIGHT = 1e-12          # oracle (same algorithm as the reference) against the reference
SCORE_TOL = 1e-10      # HIP path: -lnL and per-family values
VEC_TOL = 5e-11        # HIP path: matrix entries (O(N^2) recurrence against the reference's O(N^3) log-space sums)
SCORES = ["big12_gamma_k8", "big12_multilambda_err", "big12_base", "bench100_gamma_k8_l0.002_a2"]
INF = "bench100_gamma_k8_l0.002_a1.5_inf"


@pytest.fixture(scope="module")
def n751():
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_n751.json")) as f:
        return json.load(f)


def _check_matrix(got, e, tol):
    checks = [(got.diagonal(), np.array(e["diag"]))]
    checks += [(got[int(r)], np.array(row)) for r, row in e["rows"].items()]
    checks += [(got[:, int(c)], np.array(col)) for c, col in e["cols"].items()]
    worst = 0.0
    for g, exp in checks:
        big = exp > 1e-290                       # (below that the log-space sum itself loses digits; the two flush to 0 at different k)
        worst = max(worst, float((np.abs(g - exp)[big] / exp[big]).max(initial=0.0)))
        assert g[~big].max(initial=0.0) <= 1e-280
    assert worst <= tol, worst
    return worst


def test_fixture_is_at_the_bench_matrix_order(n751):
    assert [m["n"] for m in n751["matrices"]] == [751, 751, 751]
    for name in SCORES:
        e = n751["scores"][name]
        assert (e["max_family_size"], e["max_root_family_size"]) == (720, 750) and math.isfinite(e["neg_lnl"])
    lams = sorted(m["lambda"] for m in n751["matrices"])
    assert lams[0] < 0.0005 and lams[-1] > 0.004     # the smallest and the largest multiplier of K = 8 at lambda 0.002, alpha 2


def test_oracle_rejects_where_the_reference_does_at_100_taxa(oracle, n751):
    e = n751["scores"][INF]
    assert e["neg_lnl"] == math.inf and (e["max_family_size"], e["max_root_family_size"]) == (720, 750)
    pb, pr, alpha = case_from_args(e["args"], oracle)
    assert pb.n_taxa == 100 and pb.n_families == 4
    v, cat, fam = oracle.score_gamma(pb, pr, fast=True, per_family=True)
    assert v == math.inf
    # the slowest category of the large family from the SURVEY-8d table (the restatement, like gamma_model::prune, stops there)
    assert cat[3, 0] == 0 and (cat[:3] > 0).all()


@pytest.mark.parametrize("i", [0, 1, 2])
def test_oracle_matrices_at_751(oracle, n751, i):
    e = n751["matrices"][i]
    _check_matrix(oracle.build_matrix(751, e["lambda"], e["t"], fast=False), e, TIGHT)        # the reference's own algorithm
    _check_matrix(oracle.build_matrix(751, e["lambda"], e["t"], fast=True), e, 2e-11)         # the O(N^2) recurrence the device uses


@pytest.mark.parametrize("name", SCORES)
def test_oracle_scores_at_751(oracle, n751, name):
    e = n751["scores"][name]
    pb, pr, alpha = case_from_args(e["args"], oracle)
    assert (pb.n_families, pb.max_family_size, pb.max_root_family_size, pb.matrix_size) == (e["n_families"], 720, 750, 751)
    if pr.multipliers is not None:
        assert np.abs(pr.multipliers / np.array(e["multipliers"]) - 1).max() <= 1e-13
        v, cat, fam = oracle.score_gamma(pb, pr, fast=True, per_family=True)
        assert np.abs(cat.ravel() / np.array(e["category_likelihood"]) - 1).max() <= 2e-11
        assert np.abs(fam / np.array(e["family_likelihood"]).reshape(cat.shape)[:, 0] - 1).max() <= 2e-11
    else:
        v, fam = oracle.score_base(pb, pr, fast=True, per_family=True)
        assert np.abs(fam / np.array(e["family_lnl"]) - 1).max() <= 2e-11
    assert rel_err(v, e["neg_lnl"]) <= 2e-11, (v, e["neg_lnl"])


# ------------------------------------------------------------------ the HIP path against the same reference outputs
@pytest.fixture(scope="module")
def capi():
    from cafexp_amd import capi as C
    C.load()
    return C


@pytest.mark.gpu
@pytest.mark.parametrize("layout", [0, 1])
def test_gpu_matrices_at_751_against_the_reference(capi, n751, layout):
    """layout 0: row-major (leaf branches); layout 1: the k-major operand of K2, built through reversibility."""
    for e in n751["matrices"]:
        got = capi.build_matrices(751, [e["lambda"]], [e["t"]], layout=layout)[0]
        _check_matrix(got, e, VEC_TOL)
        assert got.min() >= 0.0 and got.max() <= 1.0 and got[0, 0] == 1.0 and not got[0, 1:].any()


@pytest.mark.gpu
def test_gpu_matrix_at_751_against_the_cubic_oracle(capi, oracle, n751):
    """Every entry of one N = 751 matrix against the reference ALGORITHM (O(N^3) log-space sums, ~seconds on the box's cores)."""
    e = n751["matrices"][2]
    want = oracle.build_matrix(751, e["lambda"], e["t"], fast=False)
    _check_matrix(want, e, TIGHT)
    for layout in (0, 1):
        got = capi.build_matrices(751, [e["lambda"]], [e["t"]], layout=layout)[0]
        big = want > 1e-290
        assert (np.abs(got - want)[big] / want[big]).max() <= VEC_TOL
        assert got[~big].max(initial=0.0) <= 1e-280


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCORES)
def test_gpu_scores_at_751_against_the_reference(capi, oracle, n751, name):
    e = n751["scores"][name]
    pb, pr, alpha = case_from_args(e["args"], oracle)
    K = len(pr.multipliers) if pr.multipliers is not None else 1
    for subtree_dedup in (True, False):
        ctx = capi.Context(pb, max_categories=K, subtree_dedup=subtree_dedup)
        v = ctx.score(pr, alpha=alpha)
        assert rel_err(v, e["neg_lnl"]) <= SCORE_TOL, (v, e["neg_lnl"])
        res = ctx.family_results(K if pr.multipliers is not None else 0)
        if pr.multipliers is not None:
            assert np.abs(res["category_likelihood"].ravel() / np.array(e["category_likelihood"]) - 1).max() <= VEC_TOL
            assert np.abs(res["family_likelihood"] / np.array(e["family_likelihood"]).reshape(-1, K)[:, 0] - 1).max() <= VEC_TOL
        else:
            assert np.abs(res["family_lnl"] / np.array(e["family_lnl"]) - 1).max() <= SCORE_TOL
        ctx.close()


@pytest.mark.gpu
def test_gpu_rejects_where_the_reference_does_at_100_taxa(capi, oracle, n751):
    """The same four families on the bench's 100-taxon tree: +inf at lambda 0.002 / alpha 1.5 like the reference (one
    category of one family is exactly zero -- and only that one), finite at alpha 2 (the SCORES case above)."""
    e = n751["scores"][INF]
    pb, pr, alpha = case_from_args(e["args"], oracle)
    ctx = capi.Context(pb, max_categories=8)
    assert ctx.score(pr, alpha=alpha) == math.inf
    # a zero-sum category is a numeric rejection, not an error: the per-family values are there (gamma_core.cpp:227-236)
    res = ctx.family_results(8)
    assert list(res["failed"]) == [0, 0, 0, 1] and res["category_likelihood"][3, 0] == 0 and (res["category_likelihood"][:3] > 0).all()
    ctx.close()
