"""Ancestral reconstruction (src/gene_family_reconstructor.cpp:13-165, :361-400).

Golden values: tests/golden/ref_reconstruct.json, printed by the compiled reference running the tail of
estimator::execute (tests/golden/make_golden.py reconstruct).
CPU: the oracle's restatement against those.  GPU: cafe_reconstruct / cafe_branch_probabilities through the C ABI
against the oracle and the golden values.  States are integers: exact, except that an arg max between two candidates
whose products agree to rounding may fall either way (the matrices come from a different but equivalent recurrence);
such a difference must be a near-tie in likelihood, which the test checks.
"""
import json
import os

import numpy as np
import pytest

from helpers import case_from_args
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gr():
    with open(os.path.join(ROOT, "tests", "golden", "ref_reconstruct.json")) as f:
        return json.load(f)["cases"]


def _case(e):
    pb, pr, alpha = case_from_args(e["args"], O)
    assert (pb.n_families, pb.max_family_size, pb.max_root_family_size) == (e["n_families"], e["max_family_size"], e["max_root_family_size"])
    jmax = min(pb.max_family_size, pb.max_root_family_size)
    rp = np.zeros(jmax + 1, dtype=np.float32)                     # compute(j) is 0 past the R entries of the prior
    n = min(len(pr.prior), jmax + 1)
    rp[:n] = pr.prior[:n]
    perm = [pb.node_names.index(name) for name in e["nodes"]]    # golden node order (reverse level order) -> ours
    return pb, pr, rp, perm


def _golden_states(e, perm, n_nodes):
    g = np.array(e["states"], dtype=np.int64).reshape(e["n_families"], len(perm))
    out = np.empty((e["n_families"], n_nodes), dtype=np.int64)
    out[:, perm] = g
    return out


def _golden_cat_states(e, perm, n_nodes, K):
    g = np.array(e["category_states"], dtype=np.int64).reshape(e["n_families"], K, len(perm))
    out = np.empty((K, e["n_families"], n_nodes), dtype=np.int64)
    for k in range(K):
        out[k][:, perm] = g[:, k, :]
    return out


def _golden_bp(e, perm, n_nodes):
    g = np.array(e["branch_probabilities"], dtype=np.float64).reshape(e["n_families"], len(perm))
    out = np.empty((e["n_families"], n_nodes))
    out[:, perm] = g
    return out


def _check_bp(got, want, tol):
    assert np.array_equal(np.isnan(got), np.isnan(want))
    m = ~np.isnan(want)
    assert np.max(np.abs(got[m] - want[m]) / np.maximum(np.abs(want[m]), 1e-300)) < tol


CASES = ["mammals_base", "mammals_gamma_k3", "mammals_lambda_tree", "mammals_poisson", "synth20", "mammals_saturated"]


# ------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("name", CASES)
def test_oracle_reconstruction_matches_reference(gr, name):
    e = gr[name]
    pb, pr, rp, perm = _case(e)
    if pr.multipliers is not None:
        got = O.reconstruct(pb, pr.lambdas, rp, multipliers=pr.multipliers)
        assert np.array_equal(got, _golden_cat_states(e, perm, pb.n_nodes, len(pr.multipliers)))
        # weighted averages (gamma_core.cpp:283-299) and the reported integer sizes (:407-420: truncation)
        avg = np.zeros((pb.n_families, pb.n_nodes))
        for k in range(len(pr.multipliers)):
            avg += pr.cat_probs[k] * got[k].astype(np.float64)
        want_avg = np.empty_like(avg)
        want_avg[:, perm] = np.array(e["averages"]).reshape(pb.n_families, len(perm))
        assert np.max(np.abs(avg - want_avg)) < 1e-9
    else:
        got = O.reconstruct(pb, pr.lambdas, rp)
        assert np.array_equal(got[0], _golden_states(e, perm, pb.n_nodes))


@pytest.mark.parametrize("name", CASES)
def test_oracle_branch_probabilities_match_reference(gr, name):
    e = gr[name]
    pb, pr, rp, perm = _case(e)
    sizes = _golden_states(e, perm, pb.n_nodes)
    got = O.branch_probabilities(pb, pr.lambdas, sizes)
    _check_bp(got, _golden_bp(e, perm, pb.n_nodes), 1e-11)


def _tiny(newick, counts, M, R):
    from cafexp_amd import problem as P
    tree = P.parse_newick(newick)
    species = sorted(counts)
    pb = P.build_problem(tree, species, ["fam"], np.array([[counts[s] for s in species]], dtype=np.int32), root_filter=False)
    pb.max_family_size, pb.max_root_family_size = M, R
    return pb


def test_oracle_reconstruct_gene_family_known_answer():
    """test.cpp:1040: (A:1,B:3):7, A=3 B=6, lambda 0.005, M=10, R=8, root distribution {1,2,3,4,5,4,3,2,1} -> AB = 4."""
    pb = _tiny("(A:1,B:3):7;", {"A": 3, "B": 6}, 10, 8)
    rd = np.array([1, 2, 3, 4, 5, 4, 3, 2, 1], dtype=np.float32)
    got = O.reconstruct(pb, [0.005], rd / np.float32(rd.sum()))[0][0]
    assert got[pb.node_names.index("AB")] == 4


def test_oracle_viterbi_sum_known_answers():
    """test.cpp:1145-1173: ((A:1,B:3):7,(C:11,D:17):23), A=11, AB=10, lambda 0.05, M=24 -> 0.2182032 for branch A; invalid
    when parent and child sizes are equal, and at the root."""
    pb = _tiny("((A:1,B:3):7,(C:11,D:17):23);", {"A": 11, "B": 2, "C": 5, "D": 6}, 24, 24)
    sizes = np.zeros((1, pb.n_nodes), dtype=np.int32)
    for name, v in {"A": 11, "B": 2, "C": 5, "D": 6, "AB": 10, "CD": 6, "ABCD": 7}.items():
        sizes[0, pb.node_names.index(name)] = v
    bp = O.branch_probabilities(pb, [0.05], sizes)[0]
    assert abs(bp[pb.node_names.index("A")] - 0.2182032) < 1e-6
    assert np.isnan(bp[pb.node_names.index("ABCD")]) and np.isnan(bp[pb.node_names.index("D")])      # root; CD == D == 6
    sizes[0, pb.node_names.index("AB")] = 11
    assert np.isnan(O.branch_probabilities(pb, [0.05], sizes)[0][pb.node_names.index("A")])


# ------------------------------------------------------------------------------------------------ GPU
def _near_tie_only(pb, pr, rp, got, want, mult):
    """Every family whose reconstruction differs from the reference's must be a rounding-level tie: both joint
    assignments have the same likelihood to 1e-9 (the reference's own argmax scans decide by the last bits)."""
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) <= max(1, len(got) // 200), "%d of %d families differ" % (len(bad), len(got))
    for f in bad:
        la, lb = (_joint_likelihood(pb, pr, rp, s[f], mult) for s in (got, want))
        assert abs(la - lb) <= 1e-9 * max(la, lb), (f, la, lb)


def _joint_likelihood(pb, pr, rp, state, mult):
    lik = float(rp[state[np.nonzero(pb.parent < 0)[0][0]]])
    for v in range(pb.n_nodes):
        par = pb.parent[v]
        if par < 0:
            continue
        lam = pr.lambdas[pb.lambda_index[v]] * mult
        if state[par] == 0:
            lik *= 1.0 if state[v] == 0 else 0.0
        else:
            lik *= O.bd_prob(*O.quantize(lam, pb.branch_length[v]), int(state[par]), int(state[v]))
    return lik


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_reconstruction_matches_reference(gr, name):
    from cafexp_amd import capi
    e = gr[name]
    pb, pr, rp, perm = _case(e)
    ctx = capi.Context(pb, max_categories=1 if pr.multipliers is None else len(pr.multipliers))
    if pr.multipliers is not None:
        got = ctx.reconstruct(pr.lambdas, rp, multipliers=pr.multipliers)
        want = _golden_cat_states(e, perm, pb.n_nodes, len(pr.multipliers))
        for k in range(len(pr.multipliers)):
            _near_tie_only(pb, pr, rp, got[k], want[k], pr.multipliers[k])
    else:
        got = ctx.reconstruct(pr.lambdas, rp)
        assert got.shape == (1, pb.n_families, pb.n_nodes)
        _near_tie_only(pb, pr, rp, got[0], _golden_states(e, perm, pb.n_nodes), 1.0)
    # leaves carry the observed counts
    leaves = np.nonzero(pb.leaf_taxon >= 0)[0]
    assert np.array_equal(got[0][:, leaves], pb.counts[:, pb.leaf_taxon[leaves]])
    # the scorer path still works on the same context afterwards
    assert np.isfinite(ctx.score(pr)) or pr.multipliers is not None or name == "mammals_saturated"


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_branch_probabilities_match_reference(gr, name):
    from cafexp_amd import capi
    e = gr[name]
    pb, pr, rp, perm = _case(e)
    ctx = capi.Context(pb)
    sizes = _golden_states(e, perm, pb.n_nodes)
    got = ctx.branch_probabilities(pr.lambdas, sizes)
    _check_bp(got, _golden_bp(e, perm, pb.n_nodes), 1e-9)


@pytest.mark.gpu
def test_reconstruction_chunked_and_duplicated_families(gr):
    """A small workspace forces several column chunks; duplicated families share one device column."""
    from cafexp_amd import capi
    e = gr["synth20"]
    pb, pr, rp, perm = _case(e)
    pb.counts = np.ascontiguousarray(np.concatenate([pb.counts, pb.counts[::3]]))
    pb.family_ids = pb.family_ids + ["dup%d" % i for i in range(len(pb.counts) - len(pb.family_ids))]
    ctx = capi.Context(pb)
    got = ctx.reconstruct(pr.lambdas, rp)[0]
    want = O.reconstruct(pb, pr.lambdas, rp)[0]
    _near_tie_only(pb, pr, rp, got, want, 1.0)
    # room for one 128-column chunk only: more than 128 distinct families go through in several passes
    pb, pr, rp, perm = _case(gr["mammals_base"])
    whole = capi.Context(pb)
    assert whole.stats()["n_unique_families"] > 128
    n_interior = int((pb.leaf_taxon < 0).sum())
    rows = (pb.max_family_size + 1 + 3) // 4 * 4
    small = capi.Context(pb, workspace_limit=(n_interior * rows * 8 + 2 * pb.n_nodes * 4 + n_interior * 4 + 1) * 128 + 64)
    assert np.array_equal(small.reconstruct(pr.lambdas, rp)[0], whole.reconstruct(pr.lambdas, rp)[0])


@pytest.mark.gpu
def test_repeated_calls_on_one_context_change_no_state(gr):
    """cafe_reconstruct shares the context's matrix pools and stream with the scorer: a second call, a call with categories
    after a base call, and a call after a scorer call on the same context return what a fresh context returns."""
    from cafexp_amd import capi, problem as P
    from cafexp_amd.gamma_rates import discrete_gamma
    pb, pr, rp, perm = _case(gr["mammals_base"])
    _, mult = discrete_gamma(3, 0.7)
    ctx = capi.Context(pb, max_categories=3)
    first = ctx.reconstruct(pr.lambdas, rp)
    assert np.array_equal(ctx.reconstruct(pr.lambdas, rp), first)
    cats = ctx.reconstruct(pr.lambdas, rp, multipliers=mult)
    ctx.score(P.Params(lambdas=pr.lambdas, prior=P.prior_uniform(pb.max_root_family_size)))
    assert np.array_equal(ctx.reconstruct(pr.lambdas, rp), first)
    assert np.array_equal(ctx.reconstruct(pr.lambdas, rp, multipliers=mult), cats)
    ctx.close()
    fresh = capi.Context(pb, max_categories=3)
    assert np.array_equal(fresh.reconstruct(pr.lambdas, rp, multipliers=mult), cats)
    assert np.array_equal(fresh.reconstruct(pr.lambdas, rp), first)
    fresh.close()


@pytest.mark.gpu
def test_polytomies_and_extinct_families():
    """Multifurcating nodes (a leaf child of the root, a 3-way interior node) and all-zero / single-taxon families:
    reconstruction, Viterbi sums and root maxima against the oracle (the reference walks `_descendants` generically)."""
    from cafexp_amd import capi, problem as P
    tree = P.parse_newick("((A:3,B:5,C:2):4,(D:6,E:1):2,F:9);")
    species = ["A", "B", "C", "D", "E", "F"]
    rng = np.random.default_rng(5)
    counts = rng.integers(0, 12, size=(300, 6)).astype(np.int32)
    counts[0] = 0
    counts[1] = [0, 0, 0, 0, 0, 7]
    counts[2] = [9, 0, 0, 0, 0, 0]
    pb = P.build_problem(tree, species, ["f%d" % i for i in range(300)], counts, root_filter=False)
    lam = np.array([0.02])
    jmax = min(pb.max_family_size, pb.max_root_family_size)
    rp = np.zeros(jmax + 1, dtype=np.float32)
    rp[:pb.max_root_family_size] = P.prior_uniform(pb.max_root_family_size)[:jmax + 1]
    ctx = capi.Context(pb)
    got = ctx.reconstruct(lam, rp)[0]
    want = O.reconstruct(pb, lam, rp)[0]
    pr = P.Params(lambdas=lam, prior=P.prior_uniform(pb.max_root_family_size))
    _near_tie_only(pb, pr, rp, got, want, 1.0)
    _check_bp(ctx.branch_probabilities(lam, want), O.branch_probabilities(pb, lam, want), 1e-9)
    assert np.max(np.abs(ctx.root_max(lam) / O.root_max(pb, lam) - 1)) < 1e-11


@pytest.mark.gpu
def test_bench_shape_sample_against_oracle():
    """The bench's 100-taxon / M = 720 shape (matrix order 751, 99 interior nodes): reconstruction, root maxima and
    Viterbi sums of sampled families against the oracle (its O(N^2) matrix build, itself pinned to the O(N^3) one)."""
    from cafexp_amd import capi, problem as P, synth
    pb, _ = synth.make_problem(n_families=2048)
    lam = np.array([0.002])
    jmax = min(pb.max_family_size, pb.max_root_family_size)
    rp = np.zeros(jmax + 1, dtype=np.float32)
    rp[:pb.max_root_family_size] = P.prior_uniform(pb.max_root_family_size)[:jmax + 1]
    ctx = capi.Context(pb)
    st = ctx.reconstruct(lam, rp)[0]
    rm = ctx.root_max(lam)
    idx = np.array([0, 1, 777, 2047])
    sub = P.Problem(parent=pb.parent, branch_length=pb.branch_length, lambda_index=pb.lambda_index, leaf_taxon=pb.leaf_taxon,
                    counts=np.ascontiguousarray(pb.counts[idx]), max_family_size=pb.max_family_size,
                    max_root_family_size=pb.max_root_family_size, taxa=pb.taxa, family_ids=[pb.family_ids[i] for i in idx],
                    node_names=pb.node_names)
    want = O.reconstruct(sub, lam, rp, fast=True)[0]
    pr = P.Params(lambdas=lam, prior=P.prior_uniform(pb.max_root_family_size))
    _near_tie_only(sub, pr, rp, st[idx], want, 1.0)
    assert np.max(np.abs(rm[idx] / O.root_max(sub, lam, fast=True) - 1)) < 1e-10
    bp = ctx.branch_probabilities(lam, st)
    _check_bp(bp[idx], O.branch_probabilities(sub, lam, st[idx], fast=True), 1e-9)


@pytest.mark.gpu
def test_row_group_flags_change_no_state(monkeypatch):
    """K5 walks a child panel only between its first and last non-zero row group (flags written by whoever wrote the panel,
    reconstruct.hip), and the walk down scans a matrix row only inside K1's extent: every product left out has an exact
    zero in it, so every state of every family equals the one found with the flags switched off (CAFE_NO_RECON_FLAGS) --
    short and long branches, two rates, gamma categories, families that are extinct or sit on one taxon."""
    from cafexp_amd import capi, problem as P, synth
    pb, _ = synth.make_problem(n_taxa=33, n_families=1500, max_count=420, lam_sim=0.003, seed=5, root_cap=200, lambda_clade_min=4)
    assert pb.matrix_size >= 256 and pb.n_lambdas == 2
    pb.counts[3, :] = 0
    pb.counts[4, :] = 0
    pb.counts[4, 7] = 300
    jmax = min(pb.max_family_size, pb.max_root_family_size)
    rp = np.zeros(jmax + 1, dtype=np.float32)
    rp[:pb.max_root_family_size] = P.prior_uniform(pb.max_root_family_size)[:jmax + 1]
    _, mult = O.discrete_gamma(3, 0.7)
    calls = [(np.array([0.0004, 0.003]), None), (np.array([0.006, 0.0002]), mult)]
    ctx = capi.Context(pb, max_categories=3)
    got = [ctx.reconstruct(lam, rp, multipliers=m) for lam, m in calls]
    monkeypatch.setenv("CAFE_NO_RECON_FLAGS", "1")
    plain = capi.Context(pb, max_categories=3)
    for (lam, m), g in zip(calls, got):
        assert np.array_equal(g, plain.reconstruct(lam, rp, multipliers=m))
    idx = np.array([3, 4, 5, 900])
    import dataclasses
    sub = dataclasses.replace(pb, counts=np.ascontiguousarray(pb.counts[idx]), family_ids=[pb.family_ids[i] for i in idx])
    lam = calls[0][0]
    pr = P.Params(lambdas=lam, prior=P.prior_uniform(pb.max_root_family_size))
    _near_tie_only(sub, pr, rp, got[0][0][idx], O.reconstruct(sub, lam, rp, fast=True)[0], 1.0)


@pytest.mark.gpu
def test_reconstruct_argument_errors(gr):
    from cafexp_amd import capi
    pb, pr, rp, perm = _case(gr["synth20"])
    ctx = capi.Context(pb)
    with pytest.raises(capi.CafeError):
        ctx.reconstruct(np.array([-1.0]), rp)
    with pytest.raises(capi.CafeError):
        ctx.branch_probabilities(pr.lambdas, np.full((pb.n_families, pb.n_nodes), pb.max_family_size + 1))


# ------------------------------------------------------------------------------------------------ driver + reports
def _run_driver(tmp_path, e, *extra):
    import subprocess
    exe = os.path.join(ROOT, "cafexp_amd", "host", "cafexp_hip")
    assert os.path.exists(exe), "cafexp_hip missing: run __graft_entry__.build()"
    data = os.path.join(ROOT, "tests", "golden", "data")
    a = e["args"]
    cmd = [exe, "-t", os.path.join(data, a["tree"]), "-i", os.path.join(data, a["families"]), "-l", repr(float(a["lambda"])), "--limit", a["limit"],
           "-s", a["seed"], "--pvalues", a["nsim"], "--reconstruct", "-o", str(tmp_path)]
    if a.get("model") == "gamma":
        cmd += ["-k", a["k"], "-a", a["alpha"]]
    out = subprocess.run([str(x) for x in cmd] + list(extra), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("name,model", [("mammals_base", "Base"), ("mammals_gamma_k3", "Gamma")])
def test_driver_reports_match_compiled_reference(gr, tmp_path, name, model):
    """The whole tail of estimator::execute through cafexp_hip: p-values at the reference's seed, reconstruction,
    Viterbi branch probabilities of the significant families and every report of reconstruction::write_results,
    compared as text with what the compiled reference writes for the same inputs."""
    e = gr[name]
    d = _run_driver(tmp_path, e)
    assert abs(d["neg_lnl"] - e["neg_lnl"]) / e["neg_lnl"] < 1e-10
    read = lambda f: open(os.path.join(str(tmp_path), f)).read()
    assert read(model + "_family_results.txt") == e["family_results_txt"]            # p-values: same draws as the reference
    assert read(model + "_count.tab") == e["count_tab"]
    assert read(model + "_change.tab") == e["change_tab"]
    assert read(model + "_asr.tre") == e["asr_tre"]
    # the reference lists clades in the order of their heap addresses: same lines, another order
    assert sorted(read(model + "_clade_results.txt").splitlines()) == sorted(e["clade_results_txt"].splitlines())
    got, want = read(model + "_branch_probabilities.tab").splitlines(), e["branch_probabilities_tab"].splitlines()
    assert len(got) == len(want) and got[0] == want[0]
    for g, w in zip(got[1:], want[1:]):                                               # 6 significant digits in the file
        gt, wt = g.split("\t"), w.split("\t")
        assert gt[0] == wt[0] and len(gt) == len(wt)
        for x, y in zip(gt[1:], wt[1:]):
            assert (x == y) or (x != "N/A" and y != "N/A" and abs(float(x) - float(y)) <= 2e-6 * abs(float(y)))
    if model == "Gamma":
        got, want = read("Gamma_category_likelihoods.txt").splitlines(), e["category_likelihoods_txt"].splitlines()
        assert got[0] == want[0] and len(got) == len(want)
        for g, w in zip(got[1:], want[1:]):
            gt, wt = g.split("\t"), w.split("\t")
            assert gt[0] == wt[0]
            for x, y in zip(gt[1:], wt[1:]):
                assert x == y or abs(float(x) - float(y)) <= 2e-6 * abs(float(y))
