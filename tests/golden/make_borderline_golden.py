#!/usr/bin/env python3
"""tests/golden/ref_borderline.json: the gamma model's failure rule at the edge of fp64 (gamma_core.cpp:152: a category whose
root vector sums to exactly 0 rejects the whole call).  Family `border` of tests/golden/data/borderline_families.txt
(cat = 120, every other taxon 1) underflows in the slowest category as lambda falls: the REAL reference
(oracle/_ref/ref_harness, container only) is run at lambdas on both sides of the edge; at 0.00085 the category's largest
root entry is 5 denormal units.  Inputs + expected outputs only.
    make -C oracle ref && python tests/golden/make_borderline_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

D = os.path.join(HERE, "data")


def main():
    out = {"generator": "tests/golden/make_borderline_golden.py", "source": "oracle/_ref/ref_harness score (real reference)",
           "args": {"tree": "mammals_tree.txt", "families": "borderline_families.txt", "model": "gamma", "k": 4, "alpha": 0.5}, "cases": {}}
    for lam in ("0.0008", "0.00085", "0.0009", "0.001", "0.002"):
        r = O.ref("score", tree=os.path.join(D, "mammals_tree.txt"), families=os.path.join(D, "borderline_families.txt"),
                  model="gamma", k=4, alpha=0.5, per_family=1, **{"lambda": lam})
        r.pop("seconds", None); r.pop("threads", None)
        out["cases"][lam] = r
        print(lam, r["neg_lnl"], r.get("category_likelihood", [])[:4])
    with open(os.path.join(HERE, "ref_borderline.json"), "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))


if __name__ == "__main__":
    main()
