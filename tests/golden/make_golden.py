#!/usr/bin/env python3
"""Generates tests/golden/ref_golden.json from the REAL reference.

Run in the build container only (needs /root/reference and `make -C oracle ref`):
    python tests/golden/make_golden.py
Every value is printed by oracle/_ref/ref_harness (our driver around the reference's own
functions, compiled in place from /root/reference/src) with 17 significant digits.  The fixture
holds inputs + expected outputs only; no reference source is stored.
Synthetic inputs (tests/golden/data/synth*.txt) come from cafexp_amd/synth.py with fixed seeds.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
import numpy as np  # noqa: E402
from cafexp_amd import synth, problem as P  # noqa: E402

D = os.path.join(HERE, "data")


def write_synth(name, n_taxa, n_families, seed, max_count, lam_sim, root_cap, lambda_clade_min=0):
    """Writes <name>_tree.txt, <name>_families.txt (CAFE format) and optionally <name>_lambda_tree.txt."""
    rng = np.random.default_rng(seed)
    tree = synth.yule_tree(n_taxa, rng)
    counts = synth.simulate_families(tree, n_families, lam_sim, rng, max_count=max_count, root_cap=root_cap)
    species = [l.name for l in tree.leaves()]
    with open(os.path.join(D, name + "_tree.txt"), "w") as f:
        f.write(synth.to_newick(tree) + "\n")
    with open(os.path.join(D, name + "_families.txt"), "w") as f:
        f.write("Desc\tFamily ID\t" + "\t".join(species) + "\n")
        for i, row in enumerate(counts):
            f.write("(null)\tfam%04d\t" % i + "\t".join(str(int(x)) for x in row) + "\n")
    if lambda_clade_min:
        cands = [n for n in tree.postorder() if not n.is_leaf and n.parent is not None and len(n.leaves()) >= lambda_clade_min]
        pick = min(cands, key=lambda n: len(n.leaves()))
        marked = {id(x) for x in pick.postorder()}

        def rec(n):
            idx = 2 if id(n) in marked else 1
            if n.is_leaf:
                return "%s:%d" % (n.name, idx)
            return "(" + ",".join(rec(c) for c in n.children) + ")" + (":%d" % idx if n.parent is not None else "")
        with open(os.path.join(D, name + "_lambda_tree.txt"), "w") as f:
            f.write(rec(tree) + ";\n")


def make_pvalues():
    """tests/golden/ref_pvalues.json: compute_pvalues / get_random_probabilities of the real reference at a fixed
    seed of its global engine.  `cond` = sorted conditional distributions of root sizes 0..ncond-1."""
    data = lambda n: os.path.join(D, n)
    jobs = {
        # the reference's own known-answer case test.cpp:2229 (expects 0.666667)
        "test2229": dict(tree=data("ab_tree.txt"), families=data("ab_families.txt"), m=10, r=8, nsim=3, seed=10, ncond=8, rootfilter=0, **{"lambda": 0.05}),
        "mammals": dict(tree=data("mammals_tree.txt"), families=data("mammal_gene_families.txt"), nsim=1000, seed=10, ncond=4, limit=600,
                        **{"lambda": 0.0018174300635539}),
        "mammals_lambda_tree": dict(tree=data("mammals_tree.txt"), families=data("mammal_gene_families.txt"), nsim=100, seed=7, ncond=3, limit=300,
                                    lambdas="0.01,0.05", lambda_tree=data("chimphuman_separate_lambda.txt")),
        "synth20": dict(tree=data("synth20_tree.txt"), families=data("synth20_families.txt"), nsim=50, seed=3, ncond=3, **{"lambda": 0.004}),
    }
    out = {"generator": "tests/golden/make_golden.py pvalues", "source": "oracle/_ref/ref_harness pvalues (real reference, std::mt19937 seeded as given)", "cases": {}}
    for name, kv in jobs.items():
        print("ref pvalues:", name, flush=True)
        r = O.ref("pvalues", **kv)
        e = {"args": {k: (os.path.basename(v) if isinstance(v, str) and os.sep in v else v) for k, v in kv.items()}}
        e.update(r)
        out["cases"][name] = e
    path = os.path.join(HERE, "ref_pvalues.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


def make_reconstruct():
    """tests/golden/ref_reconstruct.json: the tail of estimator::execute on the real reference: Pupko
    reconstruction, Viterbi branch probabilities, and (files=1) the text of every report write_results produces."""
    data = lambda n: os.path.join(D, n)
    T, F = data("mammals_tree.txt"), data("mammal_gene_families.txt")
    jobs = {
        "mammals_base": dict(tree=T, families=F, limit=150, nsim=100, seed=10, files=1, **{"lambda": 0.0018174300635539}),
        "mammals_gamma_k3": dict(tree=T, families=F, limit=80, nsim=50, seed=4, files=1, model="gamma", k=3, alpha=0.6, **{"lambda": 0.002}),
        "mammals_lambda_tree": dict(tree=T, families=F, limit=100, lambdas="0.01,0.05", lambda_tree=data("chimphuman_separate_lambda.txt")),
        "mammals_poisson": dict(tree=T, families=F, limit=60, prior="poisson:10", **{"lambda": 0.01}),
        "synth20": dict(tree=data("synth20_tree.txt"), families=data("synth20_families.txt"), **{"lambda": 0.004}),
        # lambda * t > 1 on the longest branch (96.4): that matrix is saturated, rows s >= 1 all zero (matrix_cache.cpp:153)
        "mammals_saturated": dict(tree=T, families=F, limit=60, **{"lambda": 0.012}),
    }
    out = {"generator": "tests/golden/make_golden.py reconstruct", "source": "oracle/_ref/ref_harness reconstruct (real reference)", "cases": {}}
    for name, kv in jobs.items():
        print("ref reconstruct:", name, flush=True)
        r = O.ref("reconstruct", **kv)
        e = {"args": {k: (os.path.basename(v) if isinstance(v, str) and os.sep in v else v) for k, v in kv.items()}}
        e.update(r)
        out["cases"][name] = e
    path = os.path.join(HERE, "ref_reconstruct.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


def make_runs():
    """tests/golden/ref_runs.json: the reference PROGRAM itself (cafexp(), src/cafexp.cpp:175, through `ref_harness cafexp`)
    searching lambda+alpha, two lambdas, and lambda+epsilon on the first 1500 rows of its example family table."""
    import shutil
    import tempfile
    data = lambda n: os.path.join(D, n)
    with open(data("mammal_gene_families.txt")) as f:
        lines = f.readlines()[:1501]
    with open(data("mammals_1500.txt"), "w") as f:
        f.writelines(lines)
    runs = {}
    for name, model, args in [("g3", "Gamma", ["-k", "3"]), ("ml", "Base", ["-y", "chimphuman_separate_lambda.txt"]), ("em", "Base", ["-e"])]:
        print("ref program:", name, flush=True)
        out_dir = tempfile.mkdtemp(prefix="ref_run_")
        argv = [O.REF_HARNESS, "cafexp", "-t", data("mammals_tree.txt"), "-i", data("mammals_1500.txt")]
        argv += [data(a) if a.endswith(".txt") else a for a in args] + ["-o", out_dir]
        import subprocess
        log = subprocess.run(argv, check=True, capture_output=True, text=True).stdout
        e = {"args": args, "model": model, "results_txt": open(os.path.join(out_dir, model + "_results.txt")).read(),
             "harness": json.loads(log.strip().splitlines()[-1])}
        if name == "em":
            e["error_model_txt"] = open(os.path.join(out_dir, "Base_error_model.txt")).read()
        runs[name] = e
        shutil.rmtree(out_dir)
    out = {"generator": "tests/golden/make_golden.py runs  (oracle/_ref/ref_harness cafexp -t mammals_tree.txt -i mammals_1500.txt <args> -o DIR: "
                        "the reference program itself, its own random initial guesses; mammals_1500.txt = header + first 1500 rows of the "
                        "reference's example table)", "runs": runs}
    path = os.path.join(HERE, "ref_runs.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


def main():
    if sys.argv[1:] == ["runs"]:
        if not O.have_ref():
            raise SystemExit("oracle/_ref/ref_harness missing: run `make -C oracle ref` in the build container")
        return make_runs()
    if sys.argv[1:] == ["reconstruct"]:
        if not O.have_ref():
            raise SystemExit("oracle/_ref/ref_harness missing: run `make -C oracle ref` in the build container")
        return make_reconstruct()
    if sys.argv[1:] == ["pvalues"]:
        if not O.have_ref():
            raise SystemExit("oracle/_ref/ref_harness missing: run `make -C oracle ref` in the build container")
        return make_pvalues()
    if not O.have_ref():
        raise SystemExit("oracle/_ref/ref_harness missing: run `make -C oracle ref` in the build container")
    g = {"generator": "tests/golden/make_golden.py", "source": "oracle/_ref/ref_harness (real reference, g++ -O3 -fopenmp, no BLAS)"}

    # --- single transition probabilities (inputs of test.cpp:601-612, :641-644 and a spread of our own)
    bd = []
    for lam, t, s, c in [(0.05, 5, 5, 9), (0.05, 5, 10, 9), (0.05, 5, 10, 10), (0.05, 1, 10, 10), (0.006335, 68.7105, 5, 5),
                         (0.01, 96.435, 100, 120), (0.01, 4.566, 140, 140), (0.003, 50.0, 600, 580), (0.003, 50.0, 1, 0),
                         (0.003, 50.0, 0, 0), (0.003, 50.0, 0, 3), (0.02, 30.0, 7, 7), (0.5, 1.0, 3, 3), (1e-5, 0.01, 20, 21)]:
        bd.append({"lambda": lam, "t": t, "s": s, "c": c, "value": O.ref("bd", **{"lambda": lam, "t": t, "s": s, "c": c})["value"]})
    g["bd"] = bd
    bdlog = []
    for s, c, la, co in [(46, 45, -3.672556, 0.949177), (44, 46, -2.617970, 0.854098), (43, 43, -1.686354, 0.629613),
                         (43, 44, -1.686354, 0.629613), (13, 14, -2.617970, 0.854098), (40, 42, -1.37, 0.5), (41, 34, -1.262, 0.4),
                         (5, 5, -1.1931291703283662, 0.39345841643135504)]:     # test.cpp:1287-1300
        bdlog.append({"s": s, "c": c, "log_alpha": la, "coeff": co, "value": O.ref("bdlog", s=s, c=c, log_alpha=la, coeff=co)["value"]})
    g["bdlog"] = bdlog

    # --- key quantization (matrix_cache.h:42-61)
    keys = []
    for lam, t in [(0.006335, 68.7105), (0.01, 68.710507), (0.01, 4.566782), (0.01, 96.435575), (0.002, 1.001), (0.002, 1.003),
                   (0.0018174300635539, 36.302445), (0.005 * 3.4297126146760544, 20.722711), (0.003, 0.0005), (1.23456789123e-3, 7.0)]:
        r = O.ref("key", **{"lambda": repr(lam), "t": repr(t)})
        keys.append({"lambda": lam, "t": t, "lambda_q": r["lambda_q"], "t_q": r["t_q"]})
    g["keys"] = keys

    # --- whole matrices
    mats = []
    for n, lam, t, rows in [(5, 0.05, 5.0, None), (141, 0.006335, 68.7105, [0, 1, 2, 5, 70, 140]), (141, 0.006335, 68.0, [5, 139]),
                            (10, 0.05, 25.0, None), (12, 0.02, 25.0, None), (21, 0.045, 3.0, None), (64, 0.002, 0.0004, [0, 1, 63])]:
        r = O.ref("matrix", n=n, **{"lambda": lam, "t": t})
        m = np.array(r["values"]).reshape(n, n)
        ent = {"n": n, "lambda": lam, "t": t, "diag": m.diagonal().tolist()}
        if rows is None:
            ent["full"] = m.tolist()
        else:
            ent["rows"] = {str(i): m[i].tolist() for i in rows}
        mats.append(ent)
    g["matrices"] = mats

    # --- PAML discrete gamma
    gam = []
    for k, a in [(4, 0.25), (4, 2.0), (4, 4.0), (3, 0.425), (8, 2.0), (8, 1.5), (2, 0.5), (5, 10.0), (4, 0.05), (3, 0.7), (6, 1.0), (4, 0.5)]:
        r = O.ref("gamma", k=k, alpha=a)
        gam.append({"k": k, "alpha": a, "multipliers": r["multipliers"], "cat_probs": r["cat_probs"]})
    g["gamma"] = gam

    # --- inference_prune root vectors (test.cpp:1642, :1709, :1745 inputs)
    pr = []
    for kv in [dict(newick="(A:1,B:3):7", counts="A:3,B:6", mult=1.5, m=20, r=20, **{"lambda": 0.03}),
               dict(newick="(A:1,B:3):7", counts="A:3,B:6", mult=1.0, m=20, r=20, **{"lambda": 0.03}),
               dict(newick="(A:1,B:3):7", counts="A:3,B:6", mult=1.5, m=20, r=20, errfile=os.path.join(D, "errormodel_small.txt"), **{"lambda": 0.03}),
               dict(newick="((A:1,B:1):2,(C:3,D:0.5):1,E:4)", counts="A:2,B:0,C:5,D:1,E:3", mult=1.0, m=30, r=25, **{"lambda": 0.02}),
               dict(newick="(A:1,B:1)", counts="A:0,B:0", mult=1.0, m=10, r=10, **{"lambda": 0.01})]:
        r = O.ref("prune", **kv)
        e = {k: (os.path.basename(v) if k == "errfile" else v) for k, v in kv.items()}
        e["root"] = r["root"]
        pr.append(e)
    g["prune"] = pr

    # --- whole scorer calls
    data = lambda f: os.path.join(D, f)  # noqa: E731
    mt, mf = data("mammals_tree.txt"), data("mammal_gene_families.txt")
    write_synth("synth20", 20, 96, 7, 90, 0.004, 60, lambda_clade_min=4)
    write_synth("synth100", 100, 6, 11, 600, 0.002, 300)
    jobs = {
        "mammals_base_l0.01": dict(tree=mt, families=mf, **{"lambda": 0.01}),
        "mammals_base_l0.002": dict(tree=mt, families=mf, **{"lambda": 0.002}),
        "mammals_base_nofilter": dict(tree=mt, families=mf, rootfilter=0, **{"lambda": 0.005}),
        "mammals_gamma_k4_a2": dict(tree=mt, families=mf, model="gamma", k=4, alpha=2.0, **{"lambda": 0.005}),
        "mammals_gamma_k4_a4": dict(tree=mt, families=mf, model="gamma", k=4, alpha=4.0, **{"lambda": 0.005}),
        "mammals_gamma_k4_inf": dict(tree=mt, families=mf, model="gamma", k=4, alpha=0.5, **{"lambda": 0.002}),
        "mammals_gamma_k3_a0.425": dict(tree=mt, families=mf, model="gamma", k=3, alpha=0.425, **{"lambda": 0.002}),
        "mammals_multilambda": dict(tree=mt, families=mf, lambdas="0.01,0.05", lambda_tree=data("chimphuman_separate_lambda.txt")),
        "mammals_multilambda_err": dict(tree=mt, families=mf, lambdas="0.01,0.05", lambda_tree=data("chimphuman_separate_lambda.txt"), errfile=data("errormodel_0.1.txt")),
        "mammals_err_poisson10": dict(tree=mt, families=mf, errfile=data("errormodel_0.1.txt"), prior="poisson:10", **{"lambda": 0.01}),
        "mammals_rootdist": dict(tree=mt, families=mf, rootdist=data("poisson_root_dist_1000.txt"), **{"lambda": 0.01}),
        "mammals_first200_base": dict(tree=mt, families=mf, limit=200, per_family=1, **{"lambda": 0.0018}),
        "mammals_first200_gamma": dict(tree=mt, families=mf, limit=200, per_family=1, model="gamma", k=4, alpha=2.0, **{"lambda": 0.005}),
        "mammals_first200_err": dict(tree=mt, families=mf, limit=200, per_family=1, errfile=data("errormodel_0.1.txt"), **{"lambda": 0.0018}),
        "mammals_first200_base_files": dict(tree=mt, families=mf, limit=200, files=1, **{"lambda": 0.0018}),
        "mammals_first200_gamma_files": dict(tree=mt, families=mf, limit=200, files=1, model="gamma", k=3, alpha=2.5, **{"lambda": 0.005}),
        "mammals_first200_err_files": dict(tree=mt, families=mf, limit=200, files=1, errfile=data("errormodel_0.1.txt"), **{"lambda": 0.0018}),
        "synth20_base": dict(tree=data("synth20_tree.txt"), families=data("synth20_families.txt"), per_family=1, **{"lambda": 0.004}),
        "synth20_gamma_k8": dict(tree=data("synth20_tree.txt"), families=data("synth20_families.txt"), per_family=1, model="gamma", k=8, alpha=2.0, **{"lambda": 0.004}),
        "synth20_multilambda_err": dict(tree=data("synth20_tree.txt"), families=data("synth20_families.txt"), per_family=1, lambdas="0.004,0.008",
                                        lambda_tree=data("synth20_lambda_tree.txt"), errfile=data("errormodel_0.1.txt")),
        "synth100_base": dict(tree=data("synth100_tree.txt"), families=data("synth100_families.txt"), per_family=1, **{"lambda": 0.002}),
    }
    only = set(sys.argv[1:])
    scores = {}
    prev = {}
    out_path = os.path.join(HERE, "ref_golden.json")
    if only and os.path.exists(out_path):
        prev = json.load(open(out_path)).get("scores", {})
    for name, kv in jobs.items():
        if only and name not in only:
            if name in prev:
                scores[name] = prev[name]
            continue
        print("ref score:", name, flush=True)
        r = O.ref("score", **kv)
        e = {"args": {k: (os.path.basename(v) if isinstance(v, str) and os.sep in v else v) for k, v in kv.items()}}
        e.update(r)
        scores[name] = e
    g["scores"] = scores
    with open(out_path, "w") as f:
        json.dump(g, f, indent=0, separators=(",", ":"))
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
