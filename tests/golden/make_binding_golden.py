#!/usr/bin/env python3
"""Generates tests/golden/ref_binding.json: the jobs of tests/binding_cases.py run by the UNMODIFIED reference
(oracle/_ref/ref_harness, device=cpu) in the build container.  tests/test_reference_binding.py runs the same jobs on
the GPU box through the reference-side binding (oracle/_ref/ref_hip_harness device=hip) and compares.
    make -C oracle ref && python tests/golden/make_binding_golden.py [case ...]     (named cases only: the others are kept)
Inputs + expected outputs only; no reference source is stored."""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from binding_cases import CASES, FILE_KEYS  # noqa: E402

D = os.path.join(HERE, "data")


def main():
    out = {"generator": "tests/golden/make_binding_golden.py", "source": "oracle/_ref/ref_harness (real reference, CPU)", "cases": {}}
    only = set(sys.argv[1:])
    if only:
        with open(os.path.join(HERE, "ref_binding.json")) as f:
            out["cases"] = json.load(f)["cases"]
    for name, case in CASES.items():
        if only and name not in only:
            continue
        kv = {k: (os.path.join(D, v) if k in FILE_KEYS else v) for k, v in case.items() if k != "job"}
        t0 = time.time()
        r = O.ref(case["job"], **kv)
        r.pop("threads", None)
        print("%-28s %.1f s" % (name, time.time() - t0), flush=True)
        out["cases"][name] = r
    path = os.path.join(HERE, "ref_binding.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
