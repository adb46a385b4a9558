"""The C++ host adapter (cafexp_amd/host): reference-shaped model / scorer classes above the C ABI.
CPU part: its own unit tests (mock models, reference test cases).  GPU part: the cafexp_hip driver on the
reference's example data against golden values of the compiled reference, and a full lambda search."""
import json
import math
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cafexp_amd", "host")
DATA = os.path.join(ROOT, "tests", "golden", "data")


def test_host_unit_tests_pass():
    mk = subprocess.run(["make", "-s", "-C", HOST, "host_tests"], capture_output=True, text=True)
    assert mk.returncode == 0, mk.stdout + mk.stderr
    out = subprocess.run([os.path.join(HOST, "host_tests")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failures" in out.stdout


def _run(*args, env=None):
    exe = os.path.join(HOST, "cafexp_hip")
    assert os.path.exists(exe), "cafexp_hip missing: run __graft_entry__.build()"
    out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600,
                         env=None if env is None else dict(os.environ, **env))
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    for k, v in list(d.items()):
        if v in ("inf", "-inf", "nan"):
            d[k] = float(v)
    return d


T, F = os.path.join(DATA, "mammals_tree.txt"), os.path.join(DATA, "mammal_gene_families.txt")
LT, EM, RD = (os.path.join(DATA, x) for x in ("chimphuman_separate_lambda.txt", "errormodel_0.1.txt", "poisson_root_dist_1000.txt"))


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", [
    ("mammals_base_l0.01", ["-l", 0.01]),
    ("mammals_base_nofilter", ["-l", 0.005, "-z"]),
    ("mammals_gamma_k4_a2", ["-l", 0.005, "-k", 4, "-a", 2.0]),
    ("mammals_gamma_k4_inf", ["-l", 0.002, "-k", 4, "-a", 0.5]),
    ("mammals_gamma_k3_a0.425", ["-l", 0.002, "-k", 3, "-a", 0.425]),
    ("mammals_multilambda", ["-m", "0.01,0.05", "-y", LT]),
    ("mammals_multilambda_err", ["-m", "0.01,0.05", "-y", LT, "-e", EM]),
    ("mammals_err_poisson10", ["-l", 0.01, "-e", EM, "-p", 10]),
    ("mammals_rootdist", ["-l", 0.01, "-f", RD]),
])
def test_driver_matches_compiled_reference(golden, name, args):
    e = golden["scores"][name]
    d = _run("-t", T, "-i", F, *args)
    assert (d["n_families"], d["max_family_size"], d["max_root_family_size"]) == (e["n_families"], e["max_family_size"], e["max_root_family_size"])
    if math.isinf(e["neg_lnl"]):
        assert d["neg_lnl"] == e["neg_lnl"]
    else:
        assert abs(d["neg_lnl"] - e["neg_lnl"]) / e["neg_lnl"] <= 1e-10
    if "multipliers" in e:      # PAML discrete gamma restated in C++: must quantize lambda*m_k like the reference
        assert max(abs(a / b - 1) for a, b in zip(d["multipliers"], e["multipliers"])) <= 1e-13


@pytest.mark.gpu
def test_lambda_search_reaches_the_reference_optimum():
    """`cafexp -t mammals_tree.txt -i mammal_gene_families.txt` with the compiled reference:
    lambda-hat = 0.0018174300635539, -lnL = 164769.22040624; the similarity-cutoff stop makes the optimum
    reproducible to ~1e-3 in -lnL, not to 1e-6, and initial guesses are RNG-driven."""
    d = _run("-t", T, "-i", F, "-s", 10)
    assert d["search"]["iterations"] >= 5 and d["search"]["scorer_calls"] >= 10
    assert abs(d["neg_lnl"] - 164769.22040624) < 0.05
    assert abs(d["lambda"][0] - 0.0018174300635539) < 2e-6


@pytest.mark.gpu
def test_lambda_epsilon_search_runs():
    """`-e` without a file: default error model + lambda_epsilon_optimizer (core.cpp:39-44, base_model.cpp:133)."""
    d = _run("-t", T, "-i", F, "-e", "-s", 3, "-I", 40)
    assert math.isfinite(d["neg_lnl"]) and 0 <= d["epsilon"] < 0.5 and d["lambda"][0] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name,args,model", [
    ("mammals_first200_base_files", ["-l", 0.0018], "Base"),
    ("mammals_first200_gamma_files", ["-l", 0.005, "-k", 3, "-a", 2.5], "Gamma"),
    ("mammals_first200_err_files", ["-l", 0.0018, "-e", EM], "Base"),
])
def test_output_files_match_the_reference_writers(golden, tmp_path, name, args, model):
    """<Model>_results.txt and <Model>_family_likelihoods.txt (estimator::compute, execute.cpp:49-54;
    write_vital_statistics core.cpp:96; write_family_likelihoods base_model.cpp:114 / gamma_core.cpp:49) against the
    text the compiled reference wrote for the same inputs.  Numbers are printed with the default stream precision
    (6 significant digits), so fields must agree to a unit in the last printed digit."""
    e = golden["scores"][name]
    _run("-t", T, "-i", F, "--limit", 200, "-o", str(tmp_path), *args)
    for fname, key in ((model + "_results.txt", "results_txt"), (model + "_family_likelihoods.txt", "family_likelihoods_txt")):
        got = open(os.path.join(str(tmp_path), fname)).read().splitlines()
        want = e[key].splitlines()
        assert len(got) == len(want), fname
        for g_line, w_line in zip(got, want):
            gf, wf = g_line.split("\t"), w_line.split("\t")
            assert len(gf) == len(wf), (g_line, w_line)
            for a, b in zip(gf, wf):
                if a == b:
                    continue
                pa, pb = a.split(), b.split()          # "Lambda:   0.005" style lines: compare token-wise
                assert len(pa) == len(pb), (g_line, w_line)
                for x, y in zip(pa, pb):
                    if x != y:
                        assert abs(float(x) - float(y)) <= 2e-5 * abs(float(y)), (g_line, w_line)


# ------------------------------------------------------------------ whole searches against the reference PROGRAM
def _results(text):
    out = {}
    for line in text.splitlines():
        if "Final Likelihood" in line:
            out["neg_lnl"] = float(line.split(":")[1])
        elif line.startswith("Lambda:"):
            out["lambda"] = [float(x) for x in line.split(":")[1].split(",")]
        elif line.startswith("Alpha:"):
            out["alpha"] = float(line.split(":")[1])
        elif line.startswith("Epsilon:"):
            out["epsilon"] = float(line.split(":")[1])
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g3", "ml", "em"])
def test_searches_reach_the_reference_programs_optimum(name, tmp_path):
    """tests/golden/ref_runs.json: `cafexp -t mammals_tree.txt -i mammals_1500.txt {-k 3 | -y lambda_tree | -e} -o DIR` run by the
    compiled reference itself (gamma lambda+alpha search, two-lambda search, lambda+epsilon search).  Initial guesses are
    random and both programs stop on a 1e-3 similarity cutoff, so optima are compared, not trajectories; *_results.txt
    prints 6 significant digits.  (The reference's gamma run ended against the saturation wall -- 26% of its values
    rejected -- at -lnL 38062.5; the same search here ends in the interior at about 37820.)"""
    with open(os.path.join(ROOT, "tests", "golden", "ref_runs.json")) as f:
        e = json.load(f)["runs"][name]
    want = _results(e["results_txt"])
    args = [os.path.join(DATA, a) if a.endswith(".txt") else a for a in e["args"]]
    best = None
    for seed in (1, 2, 3):            # a Nelder-Mead start can stall in a flat corner (the reference retries by hand too)
        d = _run("-t", T, "-i", os.path.join(DATA, "mammals_1500.txt"), "-s", seed, "-o", str(tmp_path), *args)
        if best is None or d["neg_lnl"] < best["neg_lnl"]:
            best = d
        if abs(d["neg_lnl"] - want["neg_lnl"]) <= 2e-5 * want["neg_lnl"]:
            break
    d = best
    # the same objective, an optimum at least as good as the reference program's own run ...
    assert d["neg_lnl"] <= want["neg_lnl"] * (1 + 2e-5), (d["neg_lnl"], want["neg_lnl"])
    # ... and that value is what the CPU oracle computes at the parameters the search returned
    import numpy as np
    from helpers import case_from_args
    from oracle import oracle as O
    ja = {"tree": "mammals_tree.txt", "families": "mammals_1500.txt"}
    if name == "ml":
        ja.update(lambdas=",".join(repr(x) for x in d["lambda"]), lambda_tree="chimphuman_separate_lambda.txt")
    else:
        ja["lambda"] = d["lambda"][0]
    if name == "g3":
        ja.update(model="gamma", k=3, alpha=d["alpha"])
    pb, pr, _ = case_from_args(ja, O)
    if name == "em":
        from cafexp_amd import problem as P
        eps = d["epsilon"]
        pb.n_deviations = 3
        pr.error_model = P.error_model_table([[0.0, 1 - eps, eps], [eps, 1 - 2 * eps, eps]], pb.max_family_size)
    assert pb.n_families == d["n_families"]
    assert abs(O.score(pb, pr) - d["neg_lnl"]) <= 1e-9 * d["neg_lnl"]
    if abs(d["neg_lnl"] - want["neg_lnl"]) <= 2e-5 * want["neg_lnl"]:       # same basin: the parameters agree as well
        got = _results(open(os.path.join(str(tmp_path), e["model"] + "_results.txt")).read())
        for a, b in zip(got["lambda"], want["lambda"]):
            assert abs(a - b) <= 0.03 * b
        if "epsilon" in want:
            assert abs(got["epsilon"] - want["epsilon"]) <= 0.03 * want["epsilon"] + 1e-3
    if name == "em":
        em = open(os.path.join(str(tmp_path), "Base_error_model.txt")).read().splitlines()
        assert em[:2] == e["error_model_txt"].splitlines()[:2]              # maxcnt / cntdiff header lines


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", [("mammals_gamma_k4_a2", ["-l", 0.005, "-k", 4, "-a", 2.0]), ("mammals_multilambda_err", ["-m", "0.01,0.05", "-y", LT, "-e", EM])])
def test_driver_through_the_multi_gpu_scorer_on_one_device(golden, name, args):
    """cafexp_hip --gpus N sends its scorer calls through cafe_create_sharded (family shards, a host thread per device, one
    RCCL all-reduce per call).  With CAFE_FORCE_SHARDED the same code runs with a world of the one GPU this box has: the
    plan, the worker thread, ncclCommInitAll / ncclAllReduce and the gather of per-family results back into table order."""
    e = golden["scores"][name]
    d = _run("-t", T, "-i", F, "--gpus", 1, *args, env={"CAFE_FORCE_SHARDED": "1"})
    assert abs(d["neg_lnl"] - e["neg_lnl"]) / e["neg_lnl"] <= 1e-10
    plain = _run("-t", T, "-i", F, *args)
    assert abs(d["neg_lnl"] - plain["neg_lnl"]) <= 1e-13 * abs(plain["neg_lnl"])      # another family order inside the shard
