"""CPU-only tests of the host plumbing: input flattening against the reference's rules, family sharding,
and that the C-ABI library loads and exports every symbol include/cafe_mi355x.h declares (no compute)."""
import os
import re

import numpy as np
import pytest

from cafexp_amd import problem as P
from helpers import DATA, read

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mammals_flattening_matches_reference_sizes(golden):
    tree = P.parse_newick(read("mammals_tree.txt"))
    sp, ids, counts = P.read_family_table(read("mammal_gene_families.txt"))
    assert counts.shape == (12653, 12)
    pb = P.build_problem(tree, sp, ids, counts)
    e = golden["scores"]["mammals_base_l0.01"]
    assert (pb.n_families, pb.max_family_size, pb.max_root_family_size) == (e["n_families"], e["max_family_size"], e["max_root_family_size"]) == (10956, 140, 112)
    assert pb.n_nodes == 23 and pb.matrix_size == 141
    assert len({round(t, 9) for t in pb.branch_length if t > 0}) == 18          # 18 distinct branch lengths
    nofilter = P.build_problem(tree, sp, ids, counts, root_filter=False)
    assert nofilter.n_families == golden["scores"]["mammals_base_nofilter"]["n_families"] == 12653
    # children precede parents, single root last
    assert pb.parent[-1] == -1 and all(pb.parent[i] > i for i in range(pb.n_nodes - 1))
    assert sorted(pb.taxa) == sorted(s for s in sp)


def test_max_size_rules():
    assert P.max_sizes(np.array([[90]])) == (140, 112)          # user_data.cpp:45-46
    assert P.max_sizes(np.array([[600]])) == (720, 750)
    assert P.max_sizes(np.array([[10]])) == (60, 30)
    assert P.max_sizes(np.array([[26]])) == (76, 32)            # rint(32.5) = 32 (ties to even)
    assert P.max_sizes(np.array([[300]])) == (360, 375)


def test_newick_variants_and_names():
    t = P.parse_newick("(A:1,B:3):7")                            # test.cpp:1642: root branch length is kept
    assert [c.name for c in t.children] == ["A", "B"] and t.length == 7 and t.key() == "AB"
    t = P.parse_newick("((A:1,B:1):2,(C:3,D:0.5):1,E:4)")
    assert len(t.children) == 3 and t.key() == "ABCDE"
    t = P.parse_newick("((E:0.36,D:0.30)abc:1.00,(C:0.85,(A:0.59,B:0.35):0.42):0.39);")
    assert sorted(l.name for l in t.leaves()) == ["A", "B", "C", "D", "E"]
    with pytest.raises(ValueError):
        P.parse_newick("(A:1,B:0);")                             # clade.cpp:397: non-positive branch length
    lt = P.parse_newick(read("chimphuman_separate_lambda.txt"), lambda_tree=True)
    idx = {n.key(): n.lambda_index for n in lt.postorder()}
    assert idx["chimp"] == 2 and idx["human"] == 2 and idx["chimphuman"] == 2 and idx["orang"] == 1
    assert lt.lambda_index == 1                                  # root defaults to the first lambda (clade.cpp:383)


def test_lambda_index_map_and_validation():
    tree = P.parse_newick(read("mammals_tree.txt"))
    sp, ids, counts = P.read_family_table(read("mammal_gene_families.txt"))
    lt = P.parse_newick(read("chimphuman_separate_lambda.txt"), lambda_tree=True)
    pb = P.build_problem(tree, sp, ids, counts[:50], lambda_tree=lt)
    assert pb.n_lambdas == 2 and not pb.single_lambda
    by_name = dict(zip(pb.node_names, pb.lambda_index))
    assert by_name["chimp"] == 1 and by_name["chimphuman"] == 1 and by_name["cat"] == 0
    bad = P.parse_newick("((A:1,B:1):1,C:1);", lambda_tree=True)
    with pytest.raises(ValueError):
        P.build_problem(tree, sp, ids, counts[:50], lambda_tree=bad)          # clade::validate_lambda_tree


def test_error_model_file_and_fill_rule():
    maxcnt, dev, dists = P.read_error_model(read("errormodel_0.1.txt"))
    assert maxcnt == 90 and dev == [-1, 0, 1] and len(dists) == 91
    tab = P.error_model_table(dists, 140)
    assert tab.shape == (141, 3) and tab[0].tolist() == [0.0, 0.95, 0.05] and tab[140].tolist() == [0.05, 0.9, 0.05]
    _, _, small = P.read_error_model(read("errormodel_small.txt"))           # rows 0, 1, 20: 2..19 repeat row 1
    assert small[7] == [0.2, 0.6, 0.2] and len(small) == 21
    d = P.default_error_model(60)                                             # core.cpp:39-44 + the fill rule
    assert d[0] == [0, .95, .05] and d[30] == [0, .95, .05] and d[60] == [.05, .9, .05]
    with pytest.raises(ValueError):
        P.read_error_model("maxcnt: 3\ncntdiff -1 0 1\n0 0.1 0.8 0.1\n")   # size 0 cannot lose a gene


def test_priors_match_oracle(oracle):
    for R in (10, 30, 112, 750):
        assert np.array_equal(P.prior_uniform(R), oracle.prior_uniform(R))
        assert np.array_equal(P.prior_poisson(R, 10.0), oracle.prior_poisson(R, 10.0))
    rd = {1: 1, 2: 5, 3: 10, 4: 15, 5: 42}
    assert np.array_equal(P.prior_rootdist(30, rd), oracle.prior_rootdist(30, rd))
    assert P.prior_uniform(112).dtype == np.float32


def test_exists_at_root_filter():
    tree = P.parse_newick("((A:1,B:1):1,(C:1,D:1):1);")
    counts = np.array([[1, 0, 0, 1], [1, 1, 0, 0], [0, 0, 0, 0], [0, 3, 2, 0]], dtype=np.int32)
    pb = P.build_problem(tree, ["A", "B", "C", "D"], list("wxyz"), counts)
    assert pb.family_ids == ["w", "z"]


def test_species_lookup_is_case_insensitive():
    tree = P.parse_newick("(Cat:1,DOG:1);")
    pb = P.build_problem(tree, ["dog", "CAT"], ["f"], np.array([[3, 5]], dtype=np.int32), root_filter=False)
    assert pb.taxa == ["Cat", "DOG"] and pb.counts.tolist() == [[5, 3]]


def test_shard_families_partitions_exactly():
    for F, W in [(50000, 8), (10, 3), (7, 8), (1, 1), (100001, 4)]:
        spans = [P.shard_families(F, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == F
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_pattern_cost_sharding_is_a_partition():
    from cafexp_amd import synth
    pb, _ = synth.make_problem(n_taxa=20, n_families=600, max_count=90)
    for W in (1, 2, 3, 8):
        parts = P.shard_families_by_pattern_cost(pb, W)
        assert len(parts) == W and all(len(p) > 0 for p in parts)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(pb.n_families))
    a, b = P.shard_families_by_pattern_cost(pb, 4), P.shard_families_by_pattern_cost(pb, 4)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))            # every rank derives the same partition


def test_synthetic_generator_is_deterministic():
    from cafexp_amd import synth
    a, ta = synth.make_problem(n_taxa=16, n_families=200, max_count=80, seed=5, root_cap=50)
    b, tb = synth.make_problem(n_taxa=16, n_families=200, max_count=80, seed=5, root_cap=50)
    assert np.array_equal(a.counts, b.counts) and np.array_equal(a.branch_length, b.branch_length)
    assert synth.to_newick(ta) == synth.to_newick(tb)
    assert a.counts.max() == 80 and a.max_family_size == 130 and a.max_root_family_size == 100
    full, _ = synth.make_problem(n_families=300)
    assert (full.n_taxa, full.max_family_size, full.max_root_family_size, full.matrix_size) == (100, 720, 750, 751)
    assert (full.branch_length[:-1] >= 0.05).all()              # no branch may quantize to t_q = 0


def test_gamma_rates_plumbing_agrees_with_paml(oracle):
    from cafexp_amd.gamma_rates import discrete_gamma
    for K, a in [(4, 2.0), (8, 2.0), (3, 0.425), (4, 0.25)]:
        probs, mult = discrete_gamma(K, a)
        assert np.abs(mult / oracle.discrete_gamma(K, a)[1] - 1).max() < 1e-7
        assert abs(mult.mean() - 1) < 1e-12 and np.allclose(probs, 1.0 / K)


def test_c_abi_library_exports_every_declared_symbol():
    """The shared library must load here (hipcc cross-compiled it, no GPU needed) and export exactly the
    entry points the header declares.  No compute call is made."""
    header = open(os.path.join(ROOT, "include", "cafe_mi355x.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(cafe_[a-z0-9_]+)\s*\(", header))
    from cafexp_amd import capi
    lib = capi.load()
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.cafe_abi_version() == 3


def test_product_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing under cafexp_amd/ or include/ may import, link or call it."""
    for base in ("cafexp_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "cafe_oracle" not in text and "from oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)
