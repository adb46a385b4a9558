"""Input flattening for the likelihood path (host-side plumbing for tests and bench.py).

Turns the reference's input files (newick tree, lambda tree, family table, error model,
root distribution) into the flat arrays `include/cafe_mi355x.h`'s `cafe_problem` /
`cafe_params` take.  The C++ host adapter (cafexp_amd/host/) has its own loaders for the
drop-in build; this module exists so that Python tests and the bench can feed the C ABI.

Reference behaviour mirrored here (file:line relative to the reference root):
  * newick grammar and interior-node naming       src/clade.cpp:282-405, :125-139
  * lambda-tree index map (index - 1)             src/clade.cpp:154-164
  * family table, CAFE format                     src/io.cpp:134-215
  * max sizes M, R                                src/user_data.cpp:45-46
  * exists_at_root filter                         src/gene_family.cpp:60-89, src/cafexp.cpp:189-199
  * error-model file and row fill rule            src/io.cpp:226-272, src/error_model.cpp:31-57
  * root priors (float!)                          src/root_equilibrium_distribution.cpp:20-50
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

__all__ = [
    "Node", "parse_newick", "Problem", "Params", "read_family_table", "max_sizes", "exists_at_root",
    "read_error_model", "default_error_model", "error_model_table", "prior_uniform", "prior_poisson",
    "prior_rootdist", "build_problem", "shard_families",
]

_TOKEN = re.compile(r"\(|\)|[^\s\(\)\:\;\,]+|\:[+-]?[0-9]*\.?[0-9]+(?:[eE][+-]?[0-9]+)?|\,|\;")


class Node:
    __slots__ = ("name", "length", "lambda_index", "children", "parent")

    def __init__(self, parent: Optional["Node"] = None):
        self.name = ""
        self.length = 0.0
        self.lambda_index = 0
        self.children: List[Node] = []
        self.parent = parent

    @property
    def is_leaf(self) -> bool:
        return not self.children

    def leaves(self) -> List["Node"]:
        if self.is_leaf:
            return [self]
        out: List[Node] = []
        for c in self.children:
            out.extend(c.leaves())
        return out

    def key(self) -> str:
        """Interior nodes are named by their sorted, concatenated leaf names (clade.cpp:125-139)."""
        if self.is_leaf:
            return self.name
        return "".join(sorted(l.name for l in self.leaves()))

    def postorder(self) -> List["Node"]:
        out: List[Node] = []
        stack: List[Tuple[Node, int]] = [(self, 0)]
        while stack:
            node, i = stack.pop()
            if i < len(node.children):
                stack.append((node, i + 1))
                stack.append((node.children[i], 0))
            else:
                out.append(node)
        return out


def parse_newick(text: str, lambda_tree: bool = False) -> Node:
    """Same token grammar as the reference's regex tokenizer (clade.cpp:284); `:x` after a node is a
    branch length, or a 1-based lambda index when `lambda_tree`."""
    root = Node()
    cur = root
    for m in _TOKEN.finditer(text):
        tok = m.group(0)
        if tok == "(":
            child = Node(cur)
            cur.children.append(child)
            cur = child
        elif tok == ",":
            if cur is root:                       # newick without the outer parentheses
                new_root = Node()
                cur.parent = new_root
                new_root.children.append(cur)
                root = new_root
            sib = Node(cur.parent)
            cur.parent.children.append(sib)
            cur = sib
        elif tok == ")":
            cur = cur.parent
        elif tok == ";":
            break
        elif tok[0] == ":":
            if lambda_tree:
                cur.lambda_index = _atoi(tok[1:])       # strtol(..., 0): leading integer part
            else:
                cur.length = float(tok[1:])
        else:
            cur.name = tok
    if lambda_tree:
        if root.lambda_index == 0:
            root.lambda_index = 1
        for n in root.postorder():
            if n.lambda_index < 1:
                raise ValueError("Invalid lambda index set for " + n.key())
    else:
        for n in root.postorder():
            if n is not root and n.length <= 0:
                raise ValueError("Invalid branch length set for " + n.key())
    if root.is_leaf:
        raise ValueError("not a valid tree")
    return root


@dataclass
class Problem:
    """Flat arrays for cafe_problem (nodes in post-order: children before parents, root last)."""
    parent: np.ndarray            # int32 [n_nodes]
    branch_length: np.ndarray     # float64 [n_nodes]
    lambda_index: np.ndarray      # int32 [n_nodes] (0-based)
    leaf_taxon: np.ndarray        # int32 [n_nodes], -1 for interior nodes
    counts: np.ndarray            # int32 [n_families, n_taxa]
    max_family_size: int          # M
    max_root_family_size: int     # R
    n_lambdas: int = 1
    single_lambda: bool = True
    n_deviations: int = 0
    taxa: List[str] = field(default_factory=list)
    family_ids: List[str] = field(default_factory=list)
    node_names: List[str] = field(default_factory=list)

    @property
    def n_nodes(self) -> int:
        return int(self.parent.shape[0])

    @property
    def n_taxa(self) -> int:
        return int(self.counts.shape[1])

    @property
    def n_families(self) -> int:
        return int(self.counts.shape[0])

    @property
    def matrix_size(self) -> int:
        return max(self.max_family_size, self.max_root_family_size) + 1      # base_model.cpp:77


@dataclass
class Params:
    """Per scorer call values for cafe_params."""
    lambdas: np.ndarray                       # float64 [n_lambdas]
    prior: np.ndarray                         # float32 [R]
    multipliers: Optional[np.ndarray] = None  # float64 [K]; None => base model
    cat_probs: Optional[np.ndarray] = None    # float64 [K]
    error_model: Optional[np.ndarray] = None  # float64 [M+1, n_deviations]

    @property
    def is_gamma(self) -> bool:
        return self.multipliers is not None


def read_family_table(text: str) -> Tuple[List[str], List[str], np.ndarray]:
    """CAFE-format table (io.cpp:164-176, :189-197): header `Desc<TAB>Family ID<TAB>sp...`, then rows.
    Returns (species names, family ids, counts[F, S])."""
    lines = [ln.rstrip("\r") for ln in text.split("\n")]
    lines = [ln for ln in lines if ln != ""]
    if not lines:
        raise ValueError("No families found")
    header = lines[0].split("\t")
    species = header[2:]
    ids: List[str] = []
    rows: List[List[int]] = []
    for ln in lines[1:]:
        tk = ln.split("\t")
        ids.append(tk[1] if len(tk) > 1 else "")
        vals = [_atoi(x) for x in tk[2:2 + len(species)]]
        vals += [0] * (len(species) - len(vals))
        rows.append(vals)
    if not rows:
        raise ValueError("No families found")
    return species, ids, np.asarray(rows, dtype=np.int32).reshape(len(rows), len(species))


def _atoi(s: str) -> int:
    m = re.match(r"\s*[+-]?\d+", s)
    return int(m.group(0)) if m else 0


def max_sizes(counts: np.ndarray) -> Tuple[int, int]:
    """(M, R) from the largest observed count (user_data.cpp:45-46)."""
    mx = int(counts.max()) if counts.size else 0
    r = max(30, int(round(mx * 1.25)))          # std::rint: ties to even, like Python's round()
    m = mx + max(50, mx // 5)
    return m, r


def exists_at_root(root: Node, col_of_leaf: Dict[int, int], counts: np.ndarray) -> np.ndarray:
    """Boolean mask [F]: every child subtree of the root holds a leaf with count > 0 (gene_family.cpp:60-89)."""
    ok = np.ones(counts.shape[0], dtype=bool)
    for child in root.children:
        cols = [col_of_leaf[id(l)] for l in child.leaves()]
        ok &= (counts[:, cols] > 0).any(axis=1)
    return ok


def read_error_model(text: str) -> Tuple[int, List[int], List[List[float]]]:
    """Error-model file (io.cpp:226-272) -> (maxcnt, deviations, rows) where rows follows
    error_model::set_probabilities' fill rule (error_model.cpp:31-50): a skipped size repeats the
    previous row."""
    maxcnt = 0
    deviations = [-1, 0, 1]
    dists: List[List[float]] = []
    for line in text.split("\n"):
        line = line.rstrip("\r")
        if line.startswith("max"):
            maxcnt = int("".join(line.split(":")[1].split()))
        elif line.startswith("cnt"):
            tk = [t for t in line.split(" ")]
            if len(tk) % 2 != 0:
                raise ValueError("Number of different count differences in the error model (including 0) is not an odd number")
            deviations = [int(t) for t in tk[1:]]
        else:
            tk = line.split(" ")
            if len(tk) > 0 and tk[0] != "":
                sz = int(tk[0])
                probs = [float(t) for t in tk[1:] if t != ""]
                _set_probabilities(dists, sz, probs)
    return maxcnt, deviations, dists


def _nearly_equal(x: float, y: float) -> bool:
    return abs(x - y) <= 0.01 * abs(x)


def _set_probabilities(dists: List[List[float]], fam_size: int, probs: List[float]) -> None:
    if (fam_size == 0 or not dists) and not _nearly_equal(probs[0], 0.0):
        raise ValueError("Cannot have a non-zero probability for family size 0 for negative deviation")
    if not _nearly_equal(sum(probs), 1.0):
        raise ValueError("Sum of probabilities must be equal to one")
    if not dists:
        dists.append(list(probs))
    if len(dists) <= fam_size:
        dists.extend([list(dists[-1]) for _ in range(fam_size + 1 - len(dists))])
    dists[fam_size] = list(probs)


def default_error_model(max_family_size: int) -> List[List[float]]:
    """`-e` without a file (core.cpp:39-44): set_probabilities(0,{0,.95,.05}) then (M,{.05,.9,.05});
    the fill rule makes rows 1..M-1 copies of row 0."""
    dists: List[List[float]] = []
    _set_probabilities(dists, 0, [0, .95, 0.05])
    _set_probabilities(dists, max_family_size, [0.05, .9, 0.05])
    return dists


def error_model_table(dists: Sequence[Sequence[float]], max_family_size: int) -> np.ndarray:
    """[M+1, n_dev] table of error_model::get_probs(x) (error_model.cpp:52-57): sizes past the last
    row reuse the last row."""
    nd = len(dists[0])
    tab = np.zeros((max_family_size + 1, nd), dtype=np.float64)
    for x in range(max_family_size + 1):
        tab[x] = dists[x] if x < len(dists) else dists[-1]
    return tab


def prior_uniform(R: int) -> np.ndarray:
    """uniform_distribution with no rootdist file: float(1)/float(R) (root_equilibrium_distribution.cpp:26)."""
    return np.full(R, np.float32(1.0) / np.float32(R), dtype=np.float32)


def prior_poisson(R: int, poisson_lambda: float) -> np.ndarray:
    """poisson_distribution::compute: exp(i ln l - lgamma(i+1) - l) narrowed to float (poisson.cpp:19-36)."""
    out = np.empty(R, dtype=np.float32)
    ll = math.log(poisson_lambda)
    for i in range(R):
        out[i] = np.float32(math.exp(i * ll - math.lgamma(i + 1) - poisson_lambda))
    return out


def prior_rootdist(R: int, rootdist: Dict[int, int]) -> np.ndarray:
    """uniform_distribution over a vectorized (size -> count) map (root_distribution.cpp:15-23)."""
    lst: List[int] = []
    for size in sorted(rootdist):
        lst.extend([size] * rootdist[size])
    total = np.float32(sum(lst))
    out = np.zeros(R, dtype=np.float32)
    for j in range(min(R, len(lst))):
        out[j] = np.float32(lst[j]) / total
    return out


def build_problem(tree: Node, species: Sequence[str], family_ids: Sequence[str], counts: np.ndarray,
                  lambda_tree: Optional[Node] = None, root_filter: bool = True,
                  max_family_size: Optional[int] = None, max_root_family_size: Optional[int] = None,
                  n_deviations: int = 0) -> Problem:
    """Flatten (tree, table) into a Problem.  M and R come from the unfiltered table, then families that
    do not exist at the root are dropped -- the order the reference uses (user_data.cpp:118 then
    cafexp.cpp:189)."""
    nodes = tree.postorder()
    index = {id(n): i for i, n in enumerate(nodes)}
    lower = {s.lower(): j for j, s in enumerate(species)}     # species lookup is case-insensitive (gene_family.h:10-25)
    leaves = [n for n in nodes if n.is_leaf]
    col_in_table: List[int] = []
    for l in leaves:
        if l.name.lower() not in lower:
            raise KeyError(l.name + " was not found in gene family table")
        col_in_table.append(lower[l.name.lower()])
    taxa = [l.name for l in leaves]
    leaf_col = {id(l): j for j, l in enumerate(leaves)}
    cnt = np.ascontiguousarray(counts[:, col_in_table], dtype=np.int32)

    m, r = max_sizes(counts)
    if max_family_size is not None:
        m = max_family_size
    if max_root_family_size is not None:
        r = max_root_family_size

    ids = list(family_ids)
    if root_filter:
        keep = exists_at_root(tree, leaf_col, cnt)
        cnt = np.ascontiguousarray(cnt[keep])
        ids = [i for i, k in zip(ids, keep) if k]

    parent = np.array([index[id(n.parent)] if n.parent is not None else -1 for n in nodes], dtype=np.int32)
    length = np.array([n.length for n in nodes], dtype=np.float64)
    leaf_taxon = np.array([leaf_col.get(id(n), -1) for n in nodes], dtype=np.int32)
    lam_idx = np.zeros(len(nodes), dtype=np.int32)
    n_lambdas, single = 1, True
    if lambda_tree is not None:
        lmap = {n.key(): n.lambda_index - 1 for n in lambda_tree.postorder()}
        if set(lmap) != {n.key() for n in nodes}:
            raise ValueError("The lambda tree structure does not match that of the tree")
        lam_idx = np.array([lmap[n.key()] for n in nodes], dtype=np.int32)
        n_lambdas = len({n.lambda_index for n in lambda_tree.postorder()})
        single = False
    return Problem(parent=parent, branch_length=length, lambda_index=lam_idx, leaf_taxon=leaf_taxon, counts=cnt,
                   max_family_size=m, max_root_family_size=r, n_lambdas=n_lambdas, single_lambda=single,
                   n_deviations=n_deviations, taxa=taxa, family_ids=ids, node_names=[n.key() for n in nodes])


def shard_families(n_families: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous family block [lo, hi) owned by `rank`."""
    base, rem = divmod(n_families, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_families_by_pattern_cost(pb: "Problem", world_size: int) -> List[np.ndarray]:
    """Family indices of every rank for the multi-GPU path when the device shares likelihood columns between families
    that agree on a whole subtree (csrc/cafe_ctx.hip, compute_patterns): a shard's work is the number of DISTINCT
    leaf-count patterns under every interior node, not its number of families.  Families are ordered by total size and
    then lexicographically, so that look-alikes land on the same rank; a family's cost is the number of interior nodes at
    which it is the first of its shard-order neighbours to show its pattern, and the ranks get consecutive runs of equal
    cumulative cost.  Any partition gives the same -lnL (a sum over families); this one balances the ranks."""
    C = np.ascontiguousarray(pb.counts)
    F = C.shape[0]
    if world_size <= 1:
        return [np.arange(F)]
    n = pb.n_nodes
    children: List[List[int]] = [[] for _ in range(n)]
    for v in range(n):
        if pb.parent[v] >= 0:
            children[int(pb.parent[v])].append(v)
    leafset: List[List[int]] = [[] for _ in range(n)]
    for v in range(n):                                   # children before parents
        leafset[v] = [int(pb.leaf_taxon[v])] if pb.leaf_taxon[v] >= 0 else [t for c in children[v] for t in leafset[c]]
    order = np.lexsort(tuple(C[:, ::-1].T) + (C.sum(axis=1),))
    Cs = C[order]
    cost = np.zeros(F, dtype=np.float64)
    for v in range(n):
        if pb.leaf_taxon[v] >= 0 or pb.parent[v] < 0:
            continue
        cols = np.ascontiguousarray(Cs[:, leafset[v]])
        _, first = np.unique(cols.view([("", cols.dtype)] * cols.shape[1]), return_index=True)
        cost[first] += 1.0
    cum = np.cumsum(cost)
    bounds = [0] + [int(np.searchsorted(cum, cum[-1] * r / world_size)) for r in range(1, world_size)] + [F]
    for r in range(1, world_size + 1):                   # never an empty shard
        bounds[r] = max(bounds[r], bounds[r - 1] + 1) if r < world_size else F
    return [order[bounds[r]:bounds[r + 1]] for r in range(world_size)]
