"""The drop-in boundary, exercised from the REFERENCE's side.

oracle/_ref/ref_hip_harness is the real reference (compiled in place from /root/reference/src by `make -C oracle
ref_hip`, container only) linked with integration/hip_models.cpp -- the translation unit INTEGRATION.md tells a
maintainer to add -- and cafexp_amd/libcafe_mi355x.so.  With device=hip its models are hip_base_model / hip_gamma_model
(derived from the reference's base_model / gamma_model), so the reference's own optimizer (optimizer.cpp), scorers
(optimizer_scorer.cpp), file readers and report writers run against the MI355X library.  Expected values:
tests/golden/ref_binding.json, the same jobs run by the unmodified reference on the CPU (make_binding_golden.py).
"""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from binding_cases import CASES, FILE_KEYS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "data")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_hip_harness")


@pytest.fixture(scope="module")
def expected():
    with open(os.path.join(ROOT, "tests", "golden", "ref_binding.json")) as f:
        return json.load(f)["cases"]


def _fix(v):
    if isinstance(v, str) and v in ("inf", "-inf", "nan"):
        return float(v)
    if isinstance(v, list):
        return [_fix(x) for x in v]
    return v


def run_hip(name, env=None):
    case = CASES[name]
    args = [HARNESS, case["job"], "device=hip"]
    args += ["%s=%s" % (k, os.path.join(DATA, v) if k in FILE_KEYS else v) for k, v in case.items() if k != "job"]
    p = subprocess.run(args, capture_output=True, text=True, timeout=600, env=None if env is None else dict(os.environ, **env))
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return {k: _fix(v) for k, v in json.loads(line).items()}


def test_binding_sources_use_no_private_reference_members():
    """CPU check of the claim in integration/hip_models.h: the binding names none of the members the reference keeps
    private (gamma_core.h:45-54, lambda.h:69) -- the compile against the reference's headers (make -C oracle ref_hip,
    run by __graft_entry__.build() in the container) is what enforces it; this keeps the claim visible on the GPU box."""
    import re
    src = open(os.path.join(ROOT, "integration", "hip_models.cpp")).read()
    code = re.sub(r"//[^\n]*", "", src)
    for private in ("_lambda_multipliers", "_gamma_cat_probs", "_node_name_to_lambda_index", "_alpha"):
        assert not re.search(r"(?<![A-Za-z0-9_])%s\b" % private, code), private
    # _category_likelihoods: only as gamma_model_reconstruction's public field of that name (gamma_core.h:37)
    assert all(m.group(0).startswith("r.") for m in re.finditer(r"\S*(?<![A-Za-z0-9])_category_likelihoods\b", code))
    assert "get_lambda_index(" not in code


def test_integration_md_quotes_the_compiled_binding():
    """INTEGRATION.md section 2 shows the reference-side binding: every ```cpp block there must be text of
    integration/hip_models.{h,cpp}, the files `make -C oracle ref_hip` compiles against the reference's headers."""
    import re
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    src = open(os.path.join(ROOT, "integration", "hip_models.cpp")).read() + open(os.path.join(ROOT, "integration", "hip_models.h")).read()
    norm = lambda t: re.sub(r"\s+", " ", t).strip()
    blocks = re.findall(r"```cpp\n(.*?)```", md, flags=re.S)
    assert len(blocks) >= 4
    src_n = norm(src)
    for b in blocks:
        assert norm(b) in src_n, b[:200]


needs_harness = pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_hip_harness is built in the container only (make -C oracle ref_hip)")


@pytest.mark.gpu
@needs_harness
@pytest.mark.parametrize("name", [n for n in CASES if CASES[n]["job"] == "score"])
def test_score_through_the_reference_classes(name, expected):
    want, got = expected[name], run_hip(name)
    assert got["n_families"] == want["n_families"]
    if math.isinf(want["neg_lnl"]):
        assert got["neg_lnl"] == want["neg_lnl"]
        return
    assert abs(got["neg_lnl"] - want["neg_lnl"]) <= 1e-10 * abs(want["neg_lnl"])
    for key in ("family_lnl", "category_likelihood", "family_likelihood", "posterior", "multipliers"):
        if key in want:
            np.testing.assert_allclose(got[key], want[key], rtol=1e-9, atol=0, err_msg=key)
    for key in ("results_txt", "family_likelihoods_txt"):      # the reference's writers at their default precision
        if key in want:
            assert got[key] == want[key], key


@pytest.mark.gpu
@needs_harness
@pytest.mark.parametrize("name", [n for n in CASES if CASES[n]["job"] == "search"])
def test_reference_optimizer_drives_the_gpu_scorer(name, expected):
    """Same seed, same optimizer code, scores equal to ~1e-13: the Nelder-Mead trajectory is the reference's own, so
    the optimum is reproduced far below the search's own tolerance (a comparison flipping on a last-bit difference
    would show up as a different iteration count; the optimum must then still be at least as good)."""
    want, got = expected[name], run_hip(name)
    assert len(got["values"]) == len(want["values"])
    if got["iterations"] == want["iterations"]:
        np.testing.assert_allclose(got["values"], want["values"], rtol=1e-7)
        assert abs(got["score"] - want["score"]) <= 1e-9 * abs(want["score"])
    else:
        assert got["score"] <= want["score"] * (1 + 1e-6)


@pytest.mark.gpu
@needs_harness
@pytest.mark.parametrize("name", [n for n in CASES if CASES[n]["job"] == "reconstruct"])
def test_reconstruction_and_reports_through_the_reference_writers(name, expected):
    want, got = expected[name], run_hip(name)
    assert got["nodes"] == want["nodes"]
    assert abs(got["neg_lnl"] - want["neg_lnl"]) <= 1e-10 * abs(want["neg_lnl"])
    assert got["states"] == want["states"]
    if "category_states" in want:
        assert got["category_states"] == want["category_states"]
        np.testing.assert_allclose(got["averages"], want["averages"], rtol=1e-12)
    np.testing.assert_array_equal(got["pvalues"], want["pvalues"])          # host draws of the reference in both runs
    np.testing.assert_allclose(np.array(got["branch_probabilities"], dtype=float), np.array(want["branch_probabilities"], dtype=float),
                               rtol=1e-9, equal_nan=True)
    for key in ("asr_tre", "count_tab", "change_tab", "family_results_txt", "branch_probabilities_tab", "category_likelihoods_txt"):
        if key in want:
            assert got[key] == want[key], key
    assert sorted(got["clade_results_txt"].splitlines()) == sorted(want["clade_results_txt"].splitlines())   # ordered by heap address


@pytest.mark.gpu
@needs_harness
@pytest.mark.parametrize("name", ["score_gamma4", "score_lambda_tree_error", "search_lambda"])
def test_binding_through_the_multi_gpu_scorer_on_one_device(name, expected):
    """hip_base_model / hip_gamma_model with n_gpus > 1 send their scorer calls through cafe_create_sharded; CAFE_FORCE_SHARDED
    runs that path with a world of the one GPU of this box (plan, worker thread, ncclCommInitAll / ncclAllReduce, gather of
    the per-family results into the reference's family order)."""
    want, got = expected[name], run_hip(name, env={"CAFE_FORCE_SHARDED": "1"})
    if "neg_lnl" in want:
        assert abs(got["neg_lnl"] - want["neg_lnl"]) <= 1e-10 * abs(want["neg_lnl"])
        for key in ("family_lnl", "category_likelihood", "family_likelihood"):
            if key in want:
                np.testing.assert_allclose(got[key], want[key], rtol=1e-9, atol=0, err_msg=key)
    else:
        assert got["iterations"] == want["iterations"]
        np.testing.assert_allclose(got["values"], want["values"], rtol=1e-7)
