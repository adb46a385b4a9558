"""Cases shared by tests/golden/make_binding_golden.py (CPU reference -> fixture) and tests/test_reference_binding.py
(the same jobs through oracle/_ref/ref_hip_harness with device=hip on the GPU box).  Keys are ref_harness job arguments;
file names are relative to tests/golden/data."""

MAMMALS = dict(tree="mammals_tree.txt", families="mammals_1500.txt")

CASES = {
    # ---- one model::infer_family_likelihoods call through the reference's own classes
    "score_base": dict(job="score", limit=600, per_family=1, files=1, **MAMMALS, **{"lambda": 0.01}),
    "score_base_poisson_error": dict(job="score", limit=400, per_family=1, errfile="errormodel_0.1.txt", prior="poisson:10", **MAMMALS, **{"lambda": 0.01}),
    "score_lambda_tree": dict(job="score", limit=400, per_family=1, lambdas="0.01,0.05", lambda_tree="chimphuman_separate_lambda.txt", **MAMMALS),
    "score_lambda_tree_error": dict(job="score", limit=300, per_family=1, lambdas="0.004,0.02", lambda_tree="chimphuman_separate_lambda.txt",
                                    errfile="errormodel_0.1.txt", **MAMMALS),
    "score_gamma4": dict(job="score", model="gamma", k=4, alpha=2.0, limit=400, per_family=1, files=1, **MAMMALS, **{"lambda": 0.005}),
    "score_gamma_rejected": dict(job="score", model="gamma", k=4, alpha=0.5, limit=200, **MAMMALS, **{"lambda": 0.002}),
    "score_synth20_lambda_tree": dict(job="score", tree="synth20_tree.txt", families="synth20_families.txt", per_family=1,
                                      lambdas="0.004,0.008", lambda_tree="synth20_lambda_tree.txt"),
    # matrix order 751 (a count of 600: M 720, R 750); the error model's last row (maxcnt = 600) differs from the others and
    # family 0 sits on it: error_model::get_probs at its own maximum (error_model.cpp:52-57) through the binding's table
    "score_big12_lambda_tree_error_at_maxcnt": dict(job="score", tree="big12_tree.txt", families="big12_families.txt", per_family=1,
                                                    lambdas="0.002,0.0035", lambda_tree="big12_lambda_tree.txt", errfile="errormodel_600.txt"),
    # ---- estimator::estimate_missing_variables: the reference's optimizer and scorers, fixed seed
    "search_lambda": dict(job="search", limit=400, seed=10, **MAMMALS),
    "search_two_lambdas": dict(job="search", limit=300, seed=11, lambda_tree="chimphuman_separate_lambda.txt", **MAMMALS),
    "search_lambda_epsilon": dict(job="search", limit=300, seed=12, estimate_error=1, **MAMMALS),
    "search_gamma_lambda_alpha": dict(job="search", model="gamma", k=3, limit=150, seed=13, **MAMMALS),
    "search_gamma_alpha": dict(job="search", model="gamma", k=3, limit=150, seed=14, **MAMMALS, **{"lambda": 0.004}),
    # ---- the tail of estimator::execute: reconstruction, Viterbi branch probabilities, the report writers
    "reconstruct_base": dict(job="reconstruct", limit=150, nsim=50, seed=10, files=1, **MAMMALS, **{"lambda": 0.0018174300635539}),
    "reconstruct_gamma": dict(job="reconstruct", model="gamma", k=3, alpha=1.5, limit=100, nsim=20, seed=5, files=1, **MAMMALS, **{"lambda": 0.004}),
}

FILE_KEYS = ("tree", "families", "lambda_tree", "errfile", "rootdist")
