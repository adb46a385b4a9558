"""Seeded synthetic workload of BASELINE.json's configs 4/5: a 100-taxon ultrametric
binary tree and gene families simulated down it with the linear birth-death process.

The reference's own simulator (src/simulator.cpp) is RNG-driven and out of scope; this generator is
ours.  It is deterministic for a given seed (numpy PCG64), so tests, bench.py and the CPU baseline
see identical inputs on every machine.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from .problem import Node, Problem, build_problem

DEFAULT_SEED = 20251004


def yule_tree(n_taxa: int, rng: np.random.Generator, height: float = 100.0, min_branch: float = 0.05) -> Node:
    """Ultrametric binary tree: pure-birth waiting times going back from the tips, random joins,
    rescaled to `height`; branch lengths are printed with 3 decimals like a newick file."""
    nodes: List[Tuple[Node, float]] = []          # (node, age of node)
    for i in range(n_taxa):
        n = Node()
        n.name = "t%03d" % i
        nodes.append((n, 0.0))
    age = 0.0
    joins = []
    while len(nodes) > 1:
        k = len(nodes)
        age += rng.exponential(1.0 / k) + 1e-3
        i, j = sorted(rng.choice(k, size=2, replace=False))
        (a, age_a), (b, age_b) = nodes[i], nodes[j]
        p = Node()
        a.parent = p
        b.parent = p
        p.children = [a, b]
        joins.append((p, a, age_a, b, age_b, age))
        nodes.pop(j)
        nodes.pop(i)
        nodes.append((p, age))
    scale = height / age
    for p, a, age_a, b, age_b, ag in joins:
        a.length = max(min_branch, round((ag - age_a) * scale, 3))
        b.length = max(min_branch, round((ag - age_b) * scale, 3))
    return nodes[0][0]


def to_newick(node: Node) -> str:
    def rec(n: Node) -> str:
        if n.is_leaf:
            return "%s:%.3f" % (n.name, n.length)
        inner = ",".join(rec(c) for c in n.children)
        return "(%s)" % inner + (":%.3f" % n.length if n.parent is not None else "")
    return rec(node) + ";"


def _bd_step(sizes: np.ndarray, lam: float, t: float, rng: np.random.Generator) -> np.ndarray:
    """Child sizes after time t for the critical linear birth-death process (lambda = mu): every
    lineage dies out with probability a = lt/(1+lt), otherwise leaves 1 + Geometric(1-a) copies."""
    a = lam * t / (1.0 + lam * t)
    surv = rng.binomial(sizes, 1.0 - a)
    extra = np.zeros_like(surv)
    pos = surv > 0
    extra[pos] = rng.negative_binomial(surv[pos], 1.0 - a)
    return surv + extra


def simulate_families(tree: Node, n_families: int, lam: float, rng: np.random.Generator,
                      max_count: int = 600, root_cap: int = 480, root_p: float = 0.02,
                      rate_shape: Optional[float] = None) -> np.ndarray:
    """counts[F, T] in tree-leaf order; families absent from a root child subtree are redrawn
    (the reference drops them: gene_family::exists_at_root, src/gene_family.cpp:60)."""
    leaves = tree.leaves()
    col = {id(l): j for j, l in enumerate(leaves)}
    out = np.zeros((n_families, len(leaves)), dtype=np.int32)
    todo = np.arange(n_families)
    while todo.size:
        nf = todo.size
        root_sizes = np.minimum(rng.geometric(root_p, size=nf), root_cap).astype(np.int64)   # >= 1, mean 1/root_p
        rates = np.full(nf, lam)
        if rate_shape is not None:
            rates = lam * rng.gamma(rate_shape, 1.0 / rate_shape, size=nf)
        cur = {id(tree): root_sizes}
        block = np.zeros((nf, len(leaves)), dtype=np.int64)
        stack = [tree]
        while stack:
            n = stack.pop()
            for c in n.children:
                if rate_shape is None:
                    cs = _bd_step(cur[id(n)], lam, c.length, rng)
                else:
                    a = rates * c.length / (1.0 + rates * c.length)
                    surv = rng.binomial(cur[id(n)], 1.0 - a)
                    extra = np.zeros_like(surv)
                    pos = surv > 0
                    extra[pos] = rng.negative_binomial(surv[pos], 1.0 - a[pos])
                    cs = surv + extra
                cur[id(c)] = cs
                if c.is_leaf:
                    block[:, col[id(c)]] = cs
                else:
                    stack.append(c)
        block = np.minimum(block, max_count)
        ok = np.ones(nf, dtype=bool)
        for child in tree.children:
            cols = [col[id(l)] for l in child.leaves()]
            ok &= (block[:, cols] > 0).any(axis=1)
        out[todo[ok]] = block[ok]
        todo = todo[~ok]
    return out


def make_problem(n_taxa: int = 100, n_families: int = 50000, max_count: int = 600, lam_sim: float = 0.002,
                 seed: int = DEFAULT_SEED, root_cap: int = 300, rate_shape: Optional[float] = None,
                 lambda_clade_min: int = 0, n_deviations: int = 0) -> Problem:
    """Config 4 (and, with lambda_clade_min > 0 / n_deviations = 3, config 5) of BASELINE.json.
    Family 0 is forced to hold a count of `max_count` so that M and R follow user_data.cpp:45-46
    (max 600 -> M = 720, R = 750, matrix order 751)."""
    rng = np.random.default_rng(seed)
    tree = yule_tree(n_taxa, rng)
    counts = simulate_families(tree, n_families, lam_sim, rng, max_count=max_count, root_cap=root_cap, rate_shape=rate_shape)
    # family 0 carries the table's maximum: a large, slowly evolving family (root max_count, rate
    # lam_sim/20) with one tip at exactly max_count.  A jump to max_count inside an ordinary family
    # would have likelihood 0 in fp64 (the reference does not rescale) and make every score +inf.
    big = simulate_families(tree, 1, lam_sim / 20.0, rng, max_count=max_count, root_cap=max_count, root_p=1e-9)
    counts[0] = big[0]
    counts[0, int(np.argmax(counts[0]))] = max_count
    species = [l.name for l in tree.leaves()]
    ids = ["fam%06d" % i for i in range(n_families)]
    lam_tree = None
    if lambda_clade_min > 0:
        # chimphuman_separate_lambda-style: one clade of >= lambda_clade_min taxa gets lambda index 2
        cands = [n for n in tree.postorder() if not n.is_leaf and n.parent is not None and len(n.leaves()) >= lambda_clade_min]
        pick = min(cands, key=lambda n: len(n.leaves()))
        marked = {id(x) for x in pick.postorder()}
        lam_tree = _clone_with_lambda(tree, marked)
    return build_problem(tree, species, ids, counts, lambda_tree=lam_tree, root_filter=True, n_deviations=n_deviations), tree


def _clone_with_lambda(tree: Node, marked: set) -> Node:
    def rec(n: Node, parent: Optional[Node]) -> Node:
        c = Node(parent)
        c.name = n.name
        c.lambda_index = 2 if id(n) in marked else 1
        c.children = [rec(x, c) for x in n.children]
        return c
    return rec(tree, None)
