#!/usr/bin/env python3
"""GPU-box helper for tests/golden/make_n751_golden.py `inf`: which families of the 100-taxon bench tables have a gamma
category whose root vector is exactly zero at SURVEY 8d's scoring point (lambda 0.003, alpha 1.5, K = 8)?  The reference
rejects the whole call then (gamma_core.cpp:152, :227).  Writes a small table -- a few such families and a few that are fine
-- in CAFE format under gpurun_out/bench100/, to be copied into tests/golden/data/ and scored by the REAL reference in the
container (hours of CPU for the whole table, minutes for six families)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cafexp_amd import capi, problem as P, synth  # noqa: E402
from cafexp_amd.gamma_rates import discrete_gamma  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "bench100")


def main():
    os.makedirs(OUT, exist_ok=True)
    report = {}
    for label, kw in [("headline_generator", {}), ("survey_8d_generator", dict(lam_sim=0.003, root_cap=480))]:
        pb, tree = synth.make_problem(n_taxa=100, n_families=50000, max_count=600, **kw)
        ctx = capi.Context(pb, max_categories=8)
        rep = {}
        for lam, alpha in [(0.003, 1.5), (0.002, 2.0), (0.003, 2.0), (0.002, 1.5)]:
            probs, mult = discrete_gamma(8, alpha)
            pr = P.Params(lambdas=np.array([lam]), prior=P.prior_uniform(pb.max_root_family_size), multipliers=mult, cat_probs=probs)
            v, res = ctx.score(pr, alpha=alpha, per_family=True) if True else (None, None)
            failed = np.nonzero(res["failed"])[0] if np.isinf(v) or True else []
            zero_cat = (res["category_likelihood"] == 0).sum(axis=0).tolist()
            rep["lambda_%g_alpha_%g" % (lam, alpha)] = {"neg_lnl": v, "families_with_a_zero_category": int(len(failed)),
                                                        "zero_entries_per_category": zero_cat, "first_failed": failed[:20].tolist()}
            if label == "headline_generator" and (lam, alpha) == (0.003, 1.5):
                bad = failed
        report[label] = rep
        if label == "headline_generator":
            good = np.setdiff1d(np.arange(pb.n_families), bad)
            # family 0 carries the table's 600 (M = 720, R = 750); three failing families of different sizes, two ordinary ones
            order = np.argsort(pb.counts[bad].max(axis=1))
            pick_bad = [int(bad[order[i]]) for i in (0, len(order) // 2, len(order) - 1)] if len(bad) >= 3 else [int(x) for x in bad]
            pick = [0] + pick_bad + [int(good[len(good) // 3]), int(good[2 * len(good) // 3])]
            pick = list(dict.fromkeys(pick))
            species = pb.taxa
            with open(os.path.join(OUT, "bench100_tree.txt"), "w") as f:
                f.write(synth.to_newick(tree) + "\n")
            with open(os.path.join(OUT, "bench100_families.txt"), "w") as f:
                f.write("Desc\tFamily ID\t" + "\t".join(species) + "\n")
                for i in pick:
                    f.write("(null)\t%s\t" % pb.family_ids[i] + "\t".join(str(int(x)) for x in pb.counts[i]) + "\n")
            report["picked"] = {"families": pick, "failing_at_0.003_1.5": pick_bad, "n_failing_total": int(len(bad))}
        ctx.close()
    with open(os.path.join(OUT, "report.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report)[:3000])


if __name__ == "__main__":
    main()
