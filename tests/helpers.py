"""Shared test plumbing: rebuild a (Problem, Params) pair from a golden fixture's argument record."""
import os

import numpy as np

from cafexp_amd import problem as P

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")


def read(name):
    with open(os.path.join(DATA, name)) as f:
        return f.read()


_cache = {}


def table(name):
    if name not in _cache:
        _cache[name] = P.read_family_table(read(name))
    return _cache[name]


def case_from_args(args, oracle):
    """args: the `args` dict of a tests/golden/ref_golden.json score entry (ref_harness job_score keys)."""
    tree = P.parse_newick(read(args["tree"]))
    species, ids, counts = table(args["families"])
    lam_tree = P.parse_newick(read(args["lambda_tree"]), lambda_tree=True) if "lambda_tree" in args else None
    n_dev, dists = 0, None
    if "errfile" in args:
        _, dev, dists = P.read_error_model(read(args["errfile"]))
        n_dev = len(dev)
    pb = P.build_problem(tree, species, ids, counts, lambda_tree=lam_tree, root_filter=bool(int(args.get("rootfilter", 1))),
                         n_deviations=n_dev)
    if "limit" in args:
        lim = int(args["limit"])
        pb.counts = np.ascontiguousarray(pb.counts[:lim])
        pb.family_ids = pb.family_ids[:lim]
    if "lambdas" in args:
        lambdas = np.array([float(x) for x in str(args["lambdas"]).split(",")])
    else:
        lambdas = np.array([float(args["lambda"])])
    R = pb.max_root_family_size
    prior_spec = args.get("prior", "uniform")
    if "rootdist" in args:
        rd = {}
        for line in read(args["rootdist"]).splitlines():
            tk = line.split()
            if len(tk) >= 2:
                rd[int(tk[0])] = int(tk[1])
        prior = P.prior_rootdist(R, rd)
    elif prior_spec == "uniform":
        prior = P.prior_uniform(R)
    else:
        prior = P.prior_poisson(R, float(prior_spec.split(":")[1]))
    pr = P.Params(lambdas=lambdas, prior=prior)
    alpha = 1.0
    if args.get("model", "base") == "gamma":
        alpha = float(args["alpha"])
        probs, mult = oracle.discrete_gamma(int(args["k"]), alpha)
        pr.multipliers, pr.cat_probs = mult, probs
    if dists is not None:
        pr.error_model = P.error_model_table(dists, pb.max_family_size)
    return pb, pr, alpha


def rel_err(a, b):
    if np.isinf(a) or np.isinf(b):
        return 0.0 if a == b else np.inf
    return abs(a - b) / max(abs(b), 1e-300)
