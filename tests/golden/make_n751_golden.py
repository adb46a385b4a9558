#!/usr/bin/env python3
"""Generates tests/golden/ref_n751.json from the REAL reference: fixtures at the matrix order of BASELINE.json's
configs 4 and 5 (max count 600 -> M = 720, R = 750, N = 751; user_data.cpp:45-46, base_model.cpp:77).

Run in the build container only (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_n751_golden.py [matrices] [scores] [inf]
Every value is printed by oracle/_ref/ref_harness (our driver around the reference's own functions, compiled in
place from /root/reference/src) with 17 significant digits; the fixture holds inputs + expected outputs only.

  matrices  rows 0, 1, 2, 100, 375, 720, 750 + the diagonal of three N = 751 matrices at config-4-like (lambda * m_k, t):
            the smallest and the largest multiplier of K = 8 at alpha = 2 and a middle one (matrix_cache.cpp:121-171)
  scores    big12_gamma_k8         12 taxa, 16 families, one count of 600, gamma K = 8 (gamma_core.cpp:169-244), per family
            big12_multilambda_err  the same table, base model, two lambdas (lambda tree) + a 3-tap error model whose
                                   last row (maxcnt = 600, where a family sits) differs from the others (config 5's shape;
                                   probability.cpp:182-193, error_model.cpp:52-57)
  inf       bench100_*             four families on the bench's own 100-taxon tree (picked on the GPU:
                                   tools/find_zero_categories.py) at the bench's scoring point (finite) and at a point where
                                   the reference returns +inf for one of them
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import numpy as np  # noqa: E402
from cafexp_amd import synth  # noqa: E402

D = os.path.join(HERE, "data")
OUT = os.path.join(HERE, "ref_n751.json")
ROWS = [0, 1, 2, 100, 375, 720, 750]


def write_table(name, tree, counts):
    species = [l.name for l in tree.leaves()]
    with open(os.path.join(D, name + "_tree.txt"), "w") as f:
        f.write(synth.to_newick(tree) + "\n")
    with open(os.path.join(D, name + "_families.txt"), "w") as f:
        f.write("Desc\tFamily ID\t" + "\t".join(species) + "\n")
        for i, row in enumerate(counts):
            f.write("(null)\tfam%04d\t" % i + "\t".join(str(int(x)) for x in row) + "\n")


def write_big12():
    """12 taxa, 16 families from the bench generator's parameters; family 0 is a large, slowly evolving family with one
    tip at exactly 600 (as synth.make_problem does), so that M = 720, R = 750, N = 751."""
    rng = np.random.default_rng(751)
    tree = synth.yule_tree(12, rng)
    counts = synth.simulate_families(tree, 16, 0.002, rng, max_count=600, root_cap=300)
    big = synth.simulate_families(tree, 1, 0.002 / 20.0, rng, max_count=600, root_cap=600, root_p=1e-9)
    counts[0] = big[0]
    counts[0, int(np.argmax(counts[0]))] = 600
    # a second large family and one with an outlying tip (a jump to 400 would have likelihood 0 in unscaled fp64)
    counts[1] = np.maximum(1, (counts[0] * 0.55).astype(np.int64))
    counts[2, 0] = 130
    write_table("big12", tree, counts)
    cands = [n for n in tree.postorder() if not n.is_leaf and n.parent is not None and len(n.leaves()) >= 3]
    pick = min(cands, key=lambda n: len(n.leaves()))
    marked = {id(x) for x in pick.postorder()}

    def rec(n):
        idx = 2 if id(n) in marked else 1
        if n.is_leaf:
            return "%s:%d" % (n.name, idx)
        return "(" + ",".join(rec(c) for c in n.children) + ")" + (":%d" % idx if n.parent is not None else "")
    with open(os.path.join(D, "big12_lambda_tree.txt"), "w") as f:
        f.write(rec(tree) + ";\n")
    with open(os.path.join(D, "errormodel_600.txt"), "w") as f:
        f.write("maxcnt:600\ncntdiff -1 0 1\n0 0.00 0.95 0.05\n1 0.05 0.9 0.05\n300 0.08 0.84 0.08\n600 0.1 0.8 0.1\n")


def load():
    if os.path.exists(OUT):
        with open(OUT) as f:
            return json.load(f)
    return {"generator": "tests/golden/make_n751_golden.py", "source": "oracle/_ref/ref_harness (real reference, g++ -O3 -fopenmp, no BLAS)"}


def save(g):
    with open(OUT, "w") as f:
        json.dump(g, f, indent=0, separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes", flush=True)


def make_matrices(g):
    _, mult = O.discrete_gamma(8, 2.0)
    mats = []
    # (lambda * m_k, t): smallest multiplier on a long branch, largest on a short one, a middle one in between
    for lam, t in [(0.002 * float(mult[0]), 61.337), (0.002 * float(mult[7]), 7.25), (0.002 * float(mult[4]), 23.904)]:
        print("ref matrix n=751 lambda=%r t=%r" % (lam, t), flush=True)
        r = O.ref("matrix", n=751, **{"lambda": repr(lam), "t": repr(t)})
        m = np.array(r["values"]).reshape(751, 751)
        mats.append({"n": 751, "lambda": lam, "t": t, "diag": m.diagonal().tolist(), "rows": {str(i): m[i].tolist() for i in ROWS},
                     "cols": {str(i): m[:, i].tolist() for i in (1, 375, 720)}})
    g["matrices"] = mats


def entry(kv, r):
    e = {"args": {k: (os.path.basename(v) if isinstance(v, str) and os.sep in v else v) for k, v in kv.items()}}
    e.update(r)
    return e


def make_scores(g):
    write_big12()
    data = lambda n: os.path.join(D, n)  # noqa: E731
    jobs = {
        "big12_gamma_k8": dict(tree=data("big12_tree.txt"), families=data("big12_families.txt"), per_family=1, model="gamma", k=8, alpha=2.0,
                               **{"lambda": 0.002}),
        "big12_multilambda_err": dict(tree=data("big12_tree.txt"), families=data("big12_families.txt"), per_family=1, lambdas="0.002,0.0035",
                                      lambda_tree=data("big12_lambda_tree.txt"), errfile=data("errormodel_600.txt")),
        "big12_base": dict(tree=data("big12_tree.txt"), families=data("big12_families.txt"), per_family=1, **{"lambda": 0.002}),
    }
    sc = g.setdefault("scores", {})
    for name, kv in jobs.items():
        print("ref score:", name, flush=True)
        sc[name] = entry(kv, O.ref("score", **kv))
        print("   -lnL", sc[name]["neg_lnl"], "M", sc[name]["max_family_size"], "R", sc[name]["max_root_family_size"], "seconds", sc[name]["seconds"], flush=True)
        save(g)


def make_inf(g):
    """100 taxa at N = 751 -- the bench's own tree -- scored by the reference.  Families 0 (the one holding the 600), 16666 and
    33333 of the headline table, and family 44630 of the table SURVEY 8d's generator parameters give (lambda_sim 0.003, root
    sizes capped at 480; same seed, hence the same tree): tools/find_zero_categories.py, run on the GPU over all 50 000
    families of both tables at four (lambda, alpha) points (profiles/r03_zero_categories.json), found every point finite
    except that one family at lambda 0.002 / alpha 1.5, whose slowest category has an all-zero root vector -- the reference
    then rejects the whole call (gamma_core.cpp:152, :227).  Two jobs: the bench's scoring point (finite, per family and
    category) and that +inf point.  About 1 350 matrices of order 751 each: half an hour per job on 8 threads."""
    from cafexp_amd import synth as S
    pb, tree = S.make_problem(n_taxa=100, n_families=50000, max_count=600)
    pb2, tree2 = S.make_problem(n_taxa=100, n_families=50000, max_count=600, lam_sim=0.003, root_cap=480)
    assert S.to_newick(tree) == S.to_newick(tree2) and pb.taxa == pb2.taxa
    rows = [pb.counts[0], pb.counts[16666], pb.counts[33333], pb2.counts[44630]]
    species = pb.taxa
    with open(os.path.join(D, "bench100_tree.txt"), "w") as f:
        f.write(S.to_newick(tree) + "\n")
    with open(os.path.join(D, "bench100_families.txt"), "w") as f:
        f.write("Desc\tFamily ID\t" + "\t".join(species) + "\n")
        for name, row in zip(["fam000000", "fam016666", "fam033333", "s8d_fam044630"], rows):
            f.write("(null)\t%s\t" % name + "\t".join(str(int(x)) for x in row) + "\n")
    data = lambda n: os.path.join(D, n)  # noqa: E731
    sc = g.setdefault("scores", {})
    for name, lam, alpha in [("bench100_gamma_k8_l0.002_a2", 0.002, 2.0), ("bench100_gamma_k8_l0.002_a1.5_inf", 0.002, 1.5)]:
        kv = dict(tree=data("bench100_tree.txt"), families=data("bench100_families.txt"), per_family=1, model="gamma", k=8, alpha=alpha, **{"lambda": lam})
        print("ref score:", name, flush=True)
        sc[name] = entry(kv, O.ref("score", **kv))
        print("   -lnL", sc[name]["neg_lnl"], "seconds", sc[name]["seconds"], flush=True)
        save(g)


def main():
    if not O.have_ref():
        raise SystemExit("oracle/_ref/ref_harness missing: run `make -C oracle ref` in the build container")
    what = set(sys.argv[1:]) or {"matrices", "scores"}
    g = load()
    if "matrices" in what:
        make_matrices(g)
        save(g)
    if "scores" in what:
        make_scores(g)
    if "inf" in what:
        make_inf(g)
    save(g)


if __name__ == "__main__":
    main()
