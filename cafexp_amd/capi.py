"""ctypes binding of libcafe_mi355x.so (the C ABI declared in include/cafe_mi355x.h).

Plumbing only: builds the C structs from `problem.Problem` / `problem.Params`, calls the library
and raises on error codes.  There is no fallback: if the HIP library is missing or no GPU is
usable, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .problem import Params, Problem

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcafe_mi355x.so")

CAFE_MAX_CATEGORIES = 32
CAFE_FLAG_NO_DEDUP = 1
CAFE_FLAG_NO_SUBTREE_DEDUP = 2
CAFE_MODEL_BASE, CAFE_MODEL_GAMMA = 0, 1

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)


class CafeProblem(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("parent", _i32p), ("branch_length", _f64p), ("lambda_index", _i32p),
        ("leaf_taxon", _i32p), ("n_taxa", C.c_int32), ("n_families", C.c_int64), ("counts", _i32p),
        ("max_family_size", C.c_int32), ("max_root_family_size", C.c_int32), ("n_lambdas", C.c_int32),
        ("single_lambda", C.c_int32), ("max_categories", C.c_int32), ("n_deviations", C.c_int32),
        ("device", C.c_int32), ("flags", C.c_int32), ("workspace_limit", C.c_size_t),
    ]


class CafeParams(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("lambdas", _f64p), ("n_categories", C.c_int32), ("multipliers", _f64p),
        ("cat_probs", _f64p), ("alpha", C.c_double), ("prior", _f32p), ("error_model", _f64p),
    ]


class CafeFamilyOut(C.Structure):
    _fields_ = [("family_lnl", _f64p), ("category_likelihood", _f64p), ("family_likelihood", _f64p), ("failed", _i32p)]


class CafeStats(C.Structure):
    _fields_ = [
        ("ms_total", C.c_double), ("ms_matrices", C.c_double), ("ms_prune", C.c_double), ("ms_gemm", C.c_double),
        ("ms_reduce", C.c_double), ("gemm_flops", C.c_double), ("gemm_bytes", C.c_double), ("gemm_flops_per_family", C.c_double),
        ("gemm_launches", C.c_int64), ("n_matrices", C.c_int64), ("n_unique_families", C.c_int64),
        ("n_chunks", C.c_int64), ("matrix_bytes", C.c_int64), ("panel_bytes", C.c_int64),
        ("n_assemble_passes", C.c_int64), ("n_gather_epilogues", C.c_int64), ("n_leaf_passes", C.c_int64),
        ("gemm_flops_dense", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


EXPORTS = [
    "cafe_abi_version", "cafe_create", "cafe_destroy", "cafe_last_error", "cafe_score", "cafe_score_partial",
    "cafe_finish_partial", "cafe_family_results", "cafe_get_matrix", "cafe_get_root_likelihoods", "cafe_get_stats",
    "cafe_matrix_size", "cafe_build_matrices", "cafe_probe_fp64_mfma", "cafe_set_profiling", "cafe_debug_stamps",
    "cafe_root_max", "cafe_reconstruct", "cafe_branch_probabilities", "cafe_pvalues", "cafe_debug_force_tile",
    "cafe_comm_unique_id", "cafe_comm_attach", "cafe_comm_detach", "cafe_shard_plan", "cafe_shard_plan_scaled", "cafe_create_sharded",
    "cafe_sharded_destroy", "cafe_sharded_last_error", "cafe_sharded_score", "cafe_sharded_family_results",
    "cafe_sharded_size", "cafe_sharded_context", "cafe_set_graphs", "cafe_executed_flops", "cafe_debug_tile_range_flops", "cafe_get_extents", "cafe_debug_launch_flops", "cafe_debug_launch_ms", "cafe_debug_plan_check",
    "cafe_debug_fail_next", "cafe_debug_column_extents", "cafe_debug_leaf_transposes",
]
CAFE_COMM_ID_BYTES = 128

_lib = None


class CafeError(RuntimeError):
    pass


def load():
    """dlopen the HIP library; raises CafeError when it was not built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own libamdhip64.so.7; two HIP runtimes in one process cannot both own
    # the GPU.  Importing torch first makes this library bind to the runtime torch already loaded
    # (same SONAME), so torch tensors / streams / torch.distributed (RCCL) and these kernels share it.
    if os.environ.get("CAFE_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise CafeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.cafe_abi_version.restype = C.c_int
    L.cafe_create.restype = C.c_void_p
    L.cafe_create.argtypes = [C.POINTER(CafeProblem), C.c_char_p, C.c_size_t]
    L.cafe_destroy.argtypes = [C.c_void_p]
    L.cafe_destroy.restype = None
    L.cafe_last_error.restype = C.c_char_p
    L.cafe_last_error.argtypes = [C.c_void_p]
    L.cafe_score.restype = C.c_int
    L.cafe_score.argtypes = [C.c_void_p, C.POINTER(CafeParams), _f64p, C.POINTER(CafeFamilyOut)]
    L.cafe_score_partial.restype = C.c_int
    L.cafe_score_partial.argtypes = [C.c_void_p, C.POINTER(CafeParams), C.c_void_p, C.c_void_p]
    L.cafe_finish_partial.restype = C.c_double
    L.cafe_finish_partial.argtypes = [_f64p]
    L.cafe_family_results.restype = C.c_int
    L.cafe_family_results.argtypes = [C.c_void_p, C.POINTER(CafeFamilyOut)]
    L.cafe_get_matrix.restype = C.c_int
    L.cafe_get_matrix.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _f64p, C.c_size_t]
    L.cafe_get_root_likelihoods.restype = C.c_int
    L.cafe_get_root_likelihoods.argtypes = [C.c_void_p, C.c_int64, C.c_int32, _f64p, C.c_size_t]
    L.cafe_root_max.restype = C.c_int
    L.cafe_root_max.argtypes = [C.c_void_p, C.POINTER(CafeParams), _f64p]
    L.cafe_pvalues.restype = C.c_int
    L.cafe_pvalues.argtypes = [C.c_void_p, C.POINTER(CafeParams), C.c_int32, C.c_uint64, _f64p]
    L.cafe_reconstruct.restype = C.c_int
    L.cafe_reconstruct.argtypes = [C.c_void_p, C.POINTER(CafeParams), _f32p, _i32p]
    L.cafe_branch_probabilities.restype = C.c_int
    L.cafe_branch_probabilities.argtypes = [C.c_void_p, C.POINTER(CafeParams), _i32p, _f64p]
    L.cafe_get_stats.restype = C.c_int
    L.cafe_get_stats.argtypes = [C.c_void_p, C.POINTER(CafeStats)]
    L.cafe_matrix_size.restype = C.c_int
    L.cafe_matrix_size.argtypes = [C.c_void_p]
    L.cafe_build_matrices.restype = C.c_int
    L.cafe_build_matrices.argtypes = [C.c_int32, C.c_int32, C.c_int32, _f64p, _f64p, C.c_int32, _f64p]
    L.cafe_probe_fp64_mfma.restype = C.c_int
    L.cafe_probe_fp64_mfma.argtypes = [C.c_int32, _f64p]
    L.cafe_debug_stamps.restype = C.c_int
    L.cafe_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    L.cafe_debug_force_tile.restype = C.c_int
    L.cafe_debug_force_tile.argtypes = [C.c_void_p, C.c_int]
    L.cafe_set_profiling.restype = C.c_int
    L.cafe_set_profiling.argtypes = [C.c_void_p, C.c_int]
    L.cafe_set_graphs.restype = C.c_int
    L.cafe_set_graphs.argtypes = [C.c_void_p, C.c_int]
    L.cafe_get_extents.restype = C.c_int
    L.cafe_get_extents.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, C.c_size_t, _i32p, C.c_size_t, _i32p]
    L.cafe_debug_launch_flops.restype = C.c_int
    L.cafe_debug_launch_flops.argtypes = [C.c_void_p, _f64p, _f64p, _i32p, C.c_size_t]
    L.cafe_executed_flops.restype = C.c_int
    L.cafe_executed_flops.argtypes = [C.c_void_p, _f64p]
    L.cafe_comm_unique_id.restype = C.c_int
    L.cafe_comm_unique_id.argtypes = [C.c_char_p]
    L.cafe_comm_attach.restype = C.c_int
    L.cafe_comm_attach.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]
    L.cafe_comm_detach.restype = C.c_int
    L.cafe_comm_detach.argtypes = [C.c_void_p]
    L.cafe_shard_plan.restype = C.c_int
    L.cafe_shard_plan.argtypes = [C.POINTER(CafeProblem), C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.cafe_create_sharded.restype = C.c_void_p
    L.cafe_create_sharded.argtypes = [C.POINTER(CafeProblem), _i32p, C.c_int32, C.c_char_p, C.c_size_t]
    L.cafe_sharded_destroy.restype = None
    L.cafe_sharded_destroy.argtypes = [C.c_void_p]
    L.cafe_sharded_last_error.restype = C.c_char_p
    L.cafe_sharded_last_error.argtypes = [C.c_void_p]
    L.cafe_sharded_score.restype = C.c_int
    L.cafe_sharded_score.argtypes = [C.c_void_p, C.POINTER(CafeParams), _f64p, C.POINTER(CafeFamilyOut)]
    L.cafe_sharded_family_results.restype = C.c_int
    L.cafe_sharded_family_results.argtypes = [C.c_void_p, C.POINTER(CafeFamilyOut)]
    L.cafe_sharded_size.restype = C.c_int32
    L.cafe_sharded_size.argtypes = [C.c_void_p]
    L.cafe_sharded_context.restype = C.c_void_p
    L.cafe_sharded_context.argtypes = [C.c_void_p, C.c_int32]
    L.cafe_debug_fail_next.restype = C.c_int
    L.cafe_debug_fail_next.argtypes = [C.c_void_p, C.c_int]
    _lib = L
    return L


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _c_problem(pb: Problem, keep: list, max_categories: int = 1, device: int = 0, flags: int = 0, workspace_limit: int = 0) -> CafeProblem:
    def k(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a
    cp = CafeProblem()
    cp.n_nodes = pb.n_nodes
    cp.parent = _p(k(pb.parent, np.int32), _i32p)
    cp.branch_length = _p(k(pb.branch_length, np.float64), _f64p)
    cp.lambda_index = _p(k(pb.lambda_index, np.int32), _i32p)
    cp.leaf_taxon = _p(k(pb.leaf_taxon, np.int32), _i32p)
    cp.n_taxa = pb.n_taxa
    cp.n_families = pb.n_families
    cp.counts = _p(k(pb.counts, np.int32), _i32p)
    cp.max_family_size = pb.max_family_size
    cp.max_root_family_size = pb.max_root_family_size
    cp.n_lambdas = pb.n_lambdas
    cp.single_lambda = 1 if pb.single_lambda else 0
    cp.max_categories = max_categories
    cp.n_deviations = pb.n_deviations
    cp.device = device
    cp.flags = flags
    cp.workspace_limit = workspace_limit
    return cp


def _c_params(pr: Params, alpha: float = 1.0):
    keep = []

    def k(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a
    cp = CafeParams()
    cp.model = CAFE_MODEL_GAMMA if pr.multipliers is not None else CAFE_MODEL_BASE
    cp.lambdas = _p(k(pr.lambdas, np.float64), _f64p)
    if pr.multipliers is not None:
        cp.n_categories = len(pr.multipliers)
        cp.multipliers = _p(k(pr.multipliers, np.float64), _f64p)
        cp.cat_probs = _p(k(pr.cat_probs, np.float64), _f64p)
    else:
        cp.n_categories = 1
    cp.alpha = alpha
    cp.prior = _p(k(pr.prior, np.float32), _f32p)
    cp.error_model = _p(k(pr.error_model, np.float64), _f64p) if pr.error_model is not None else None
    return cp, keep


def shard_plan(pb: Problem, n_shards: int, max_categories: int = 1, family_scale=None):
    """cafe_shard_plan / cafe_shard_plan_scaled: the library's balanced family partition (host code, no GPU needed).
    family_scale: per family (table order), measured time of its shard under an earlier plan / mean over that plan's shards.
    Returns the family indices of every shard (a list of n_shards int64 arrays)."""
    keep = []
    cp = _c_problem(pb, keep, max_categories)
    order = np.empty(pb.n_families, dtype=np.int64)
    bounds = np.empty(n_shards + 1, dtype=np.int64)
    L = load()
    if family_scale is None:
        rc = L.cafe_shard_plan(C.byref(cp), n_shards, order.ctypes.data_as(C.POINTER(C.c_int64)), bounds.ctypes.data_as(C.POINTER(C.c_int64)))
    else:
        fs = np.ascontiguousarray(family_scale, dtype=np.float64)
        assert fs.shape == (pb.n_families,)
        L.cafe_shard_plan_scaled.restype = C.c_int
        L.cafe_shard_plan_scaled.argtypes = [C.POINTER(CafeProblem), C.c_int32, _f64p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        rc = L.cafe_shard_plan_scaled(C.byref(cp), n_shards, _p(fs, _f64p), order.ctypes.data_as(C.POINTER(C.c_int64)), bounds.ctypes.data_as(C.POINTER(C.c_int64)))
    if rc:
        raise CafeError("cafe_shard_plan failed with code %d" % rc)
    return [order[bounds[r]:bounds[r + 1]].copy() for r in range(n_shards)]


def rebalanced_plan(pb: Problem, plan, times, max_categories: int = 1, scale=None, return_scale: bool = False):
    """One step of measured rebalancing: `times[r]` is what shard r of `plan` took; the new plan (same number of shards) is
    made with every family scaled by its shard's time over the mean -- on top of `scale` (per family, table order) when
    `plan` itself came from an earlier step."""
    t = np.asarray(times, dtype=np.float64)
    scale = np.ones(pb.n_families) if scale is None else np.array(scale, dtype=np.float64)
    for r, fam in enumerate(plan):
        scale[fam] *= t[r] / t.mean()
    scale = np.clip(scale, 0.2, 5.0)
    new = shard_plan(pb, len(plan), max_categories, family_scale=scale)
    return (new, scale) if return_scale else new


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(CAFE_COMM_ID_BYTES)
    rc = load().cafe_comm_unique_id(buf)
    if rc:
        raise CafeError("cafe_comm_unique_id failed with code %d" % rc)
    return buf.raw


class Context:
    """One cafe_ctx: a family shard resident on one GPU (model state of the reference's scorer)."""

    def __init__(self, pb: Problem, max_categories: int = 1, device: int = 0, dedup: bool = True,
                 workspace_limit: int = 0, subtree_dedup: bool = True):
        self._lib = load()
        self._keep = []
        cp = _c_problem(pb, self._keep, max_categories, device,
                        (0 if dedup else CAFE_FLAG_NO_DEDUP) | (0 if subtree_dedup else CAFE_FLAG_NO_SUBTREE_DEDUP), workspace_limit)
        err = C.create_string_buffer(512)
        self._h = self._lib.cafe_create(C.byref(cp), err, 512)
        if not self._h:
            raise CafeError(err.value.decode() or "cafe_create failed")
        self.problem = pb
        self.R = pb.max_root_family_size
        self.M = pb.max_family_size
        self.n_nodes = pb.n_nodes
        self.n_families = pb.n_families
        self._keep = []          # cafe_create copied everything

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cafe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise CafeError("code %d: %s" % (rc, self._lib.cafe_last_error(self._h).decode()))

    def _params(self, pr: Params, alpha: float = 1.0):
        return _c_params(pr, alpha)

    def comm_attach(self, comm_id: bytes, world_size: int, rank: int):
        """Join the RCCL communicator of the ranks holding the other family shards: from now on score() all-reduces
        {sum lnL, rejects} and returns the whole table's value on every rank (collective call)."""
        self._check(self._lib.cafe_comm_attach(self._h, comm_id, world_size, rank))

    def comm_detach(self):
        self._check(self._lib.cafe_comm_detach(self._h))

    def score(self, pr: Params, alpha: float = 1.0, per_family: bool = False):
        """One infer_family_likelihoods call -> -lnL (float, may be inf / nan)."""
        cp, keep = self._params(pr, alpha)
        out = C.c_double()
        self._check(self._lib.cafe_score(self._h, C.byref(cp), C.byref(out), None))
        if not per_family:
            return out.value
        return out.value, self.family_results(len(pr.multipliers) if pr.multipliers is not None else 0)

    def family_results(self, K: int = 0):
        F = self.n_families
        fo = CafeFamilyOut()
        res = {"family_lnl": np.empty(F), "failed": np.empty(F, dtype=np.int32)}
        fo.family_lnl = _p(res["family_lnl"], _f64p)
        fo.failed = _p(res["failed"], _i32p)
        if K > 0:
            res["category_likelihood"] = np.empty((F, K))
            res["family_likelihood"] = np.empty(F)
            fo.category_likelihood = _p(res["category_likelihood"], _f64p)
            fo.family_likelihood = _p(res["family_likelihood"], _f64p)
        self._check(self._lib.cafe_family_results(self._h, C.byref(fo)))
        return res

    def root_max(self, lambdas) -> np.ndarray:
        """max_j L_root[j] per family under the plain lambdas (p-value path, probability.cpp:313, :399)."""
        lam = np.ascontiguousarray(lambdas, dtype=np.float64)
        cp = CafeParams()
        cp.model = CAFE_MODEL_BASE
        cp.lambdas = _p(lam, _f64p)
        cp.n_categories = 1
        out = np.empty(self.n_families)
        self._check(self._lib.cafe_root_max(self._h, C.byref(cp), _p(out, _f64p)))
        return out

    def pvalues(self, lambdas, n_simulations: int = 1000, seed: int = 1) -> np.ndarray:
        """compute_pvalues with the simulation on the device (probability.cpp:418): statistical, not draw-for-draw."""
        lam = np.ascontiguousarray(lambdas, dtype=np.float64)
        cp = CafeParams()
        cp.model = CAFE_MODEL_BASE
        cp.lambdas = _p(lam, _f64p)
        cp.n_categories = 1
        out = np.empty(self.n_families)
        self._check(self._lib.cafe_pvalues(self._h, C.byref(cp), n_simulations, seed, _p(out, _f64p)))
        return out

    def reconstruct(self, lambdas, root_prior, multipliers=None) -> np.ndarray:
        """Pupko joint reconstruction -> int32 [K][n_families][n_nodes] (gene_family_reconstructor.cpp:13-165).
        root_prior[j] = compute(j), j = 0..min(M, R)."""
        lam = np.ascontiguousarray(lambdas, dtype=np.float64)
        rp = np.ascontiguousarray(root_prior, dtype=np.float32)
        if len(rp) < min(self.M, self.R) + 1:
            raise CafeError("root_prior needs min(M, R) + 1 entries")
        cp = CafeParams()
        cp.lambdas = _p(lam, _f64p)
        K = 1
        mult = None
        if multipliers is not None:
            mult = np.ascontiguousarray(multipliers, dtype=np.float64)
            K = len(mult)
            cp.model = CAFE_MODEL_GAMMA
            cp.multipliers = _p(mult, _f64p)
        else:
            cp.model = CAFE_MODEL_BASE
        cp.n_categories = K
        out = np.empty((K, self.n_families, self.n_nodes), dtype=np.int32)
        self._check(self._lib.cafe_reconstruct(self._h, C.byref(cp), _p(rp, _f32p), _p(out, _i32p)))
        return out

    def branch_probabilities(self, lambdas, sizes) -> np.ndarray:
        """compute_viterbi_sum for every (family, node); NaN = invalid (gene_family_reconstructor.cpp:361-400)."""
        lam = np.ascontiguousarray(lambdas, dtype=np.float64)
        sz = np.ascontiguousarray(sizes, dtype=np.int32)
        assert sz.shape == (self.n_families, self.n_nodes)
        cp = CafeParams()
        cp.model = CAFE_MODEL_BASE
        cp.lambdas = _p(lam, _f64p)
        cp.n_categories = 1
        out = np.empty((self.n_families, self.n_nodes))
        self._check(self._lib.cafe_branch_probabilities(self._h, C.byref(cp), _p(sz, _i32p), _p(out, _f64p)))
        return out

    def score_partial(self, pr: Params, device_ptr: int, stream: int = 0, alpha: float = 1.0):
        """Enqueue the shard's work; {sum lnL, rejects} lands in 2 doubles of device memory."""
        cp, keep = self._params(pr, alpha)
        self._check(self._lib.cafe_score_partial(self._h, C.byref(cp), C.c_void_p(device_ptr), C.c_void_p(stream)))

    def finish(self, host_pair) -> float:
        a = np.ascontiguousarray(host_pair, dtype=np.float64)
        return self._lib.cafe_finish_partial(_p(a, _f64p))

    def matrix(self, node: int, category: int = 0) -> np.ndarray:
        n = self._lib.cafe_matrix_size(self._h)
        out = np.empty((n, n))
        self._check(self._lib.cafe_get_matrix(self._h, node, category, _p(out, _f64p), out.size))
        return out

    def root_likelihoods(self, family: int, category: int = 0) -> np.ndarray:
        out = np.empty(self.R)
        self._check(self._lib.cafe_get_root_likelihoods(self._h, family, category, _p(out, _f64p), out.size))
        return out

    def stats(self) -> dict:
        st = CafeStats()
        self._check(self._lib.cafe_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def extents(self, node: int, category: int = 0):
        """(matrix extents [blocks or N][2], panel tile extents [tiles][2] or None) of the last call (cafe_get_extents)."""
        n = self._lib.cafe_matrix_size(self._h)
        m = np.zeros((max(n, (n + 14) // 16), 2), dtype=np.int32)
        pt = np.zeros((self.n_families // 128 + 2, 2), dtype=np.int32)
        nt = C.c_int32()
        self._check(self._lib.cafe_get_extents(self._h, node, category, _p(m, _i32p), m.size, _p(pt, _i32p), pt.size, C.byref(nt)))
        leaf = self.problem.leaf_taxon[node] >= 0
        return (m[:n] if leaf else m[:(n - 1 + 15) // 16]), (pt[:nt.value] if nt.value else None)

    def column_extents(self, node: int, category: int = 0) -> np.ndarray:
        """Per-column zero extents [columns][2] of an interior non-root node's panel in the last call (diagnostic)."""
        self._lib.cafe_debug_column_extents.restype = C.c_int
        self._lib.cafe_debug_column_extents.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, C.c_size_t, C.POINTER(C.c_int64)]
        out = np.zeros((self.n_families + 256, 2), dtype=np.int32)
        n = C.c_int64()
        self._check(self._lib.cafe_debug_column_extents(self._h, node, category, _p(out, _i32p), out.size, C.byref(n)))
        return out[:n.value]

    def leaf_transposes(self):
        """(leaf branches with a transposed copy of their matrix for the assemble passes, whether the last call used them)."""
        self._lib.cafe_debug_leaf_transposes.restype = C.c_int
        self._lib.cafe_debug_leaf_transposes.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        n, used = C.c_int32(), C.c_int32()
        self._check(self._lib.cafe_debug_leaf_transposes(self._h, C.byref(n), C.byref(used)))
        return n.value, bool(used.value)

    def launch_flops(self):
        """Per K2 launch of the last call: (executed flops, flops over all K tiles, tile height in 16-row blocks)."""
        n = int(self.stats()["gemm_launches"])
        ex, al, mi = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)
        self._check(self._lib.cafe_debug_launch_flops(self._h, _p(ex, _f64p), _p(al, _f64p), _p(mi, _i32p), n))
        return ex, al, mi

    def plan_check(self):
        """(launches of the last call that ran from planned tile lists, worst modelled workgroup load / mean); raises when a
        list does not cover its launch's tiles exactly once with the K ranges the extents give."""
        n, w = C.c_int32(), C.c_double()
        self._lib.cafe_debug_plan_check.restype = C.c_int
        self._lib.cafe_debug_plan_check.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        self._check(self._lib.cafe_debug_plan_check(self._h, C.byref(n), C.byref(w)))
        return n.value, w.value

    def executed_flops(self) -> float:
        """Flops the K2 launches of the last call really ran (K tiles outside matrix extent x panel extent are skipped)."""
        v = C.c_double()
        self._check(self._lib.cafe_executed_flops(self._h, C.byref(v)))
        return v.value

    def tile_range_flops(self) -> float:
        """The same with every row block of a tile counted over the tile's whole K range (cafe_debug_tile_range_flops)."""
        self._lib.cafe_debug_tile_range_flops.restype = C.c_int
        self._lib.cafe_debug_tile_range_flops.argtypes = [C.c_void_p, _f64p]
        v = C.c_double()
        self._check(self._lib.cafe_debug_tile_range_flops(self._h, C.byref(v)))
        return v.value

    def debug_stamps(self, words: int) -> np.ndarray:
        out = np.zeros(words, dtype=np.uint64)
        self._check(self._lib.cafe_debug_stamps(self._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), words))
        return out

    def force_tile(self, mi: int):
        """Diagnostic: K2 row tile of 16*mi rows for every launch (0: chosen per launch)."""
        self._check(self._lib.cafe_debug_force_tile(self._h, mi))

    def set_profiling(self, on: bool):
        self._check(self._lib.cafe_set_profiling(self._h, 1 if on else 0))

    def debug_fail_next(self, n: int = 1):
        """Test hook: the n-th next call of this context fails with CAFE_ERR_DEVICE in the middle of its enqueue."""
        self._check(self._lib.cafe_debug_fail_next(self._h, n))

    def set_graphs(self, on: bool):
        """False: enqueue every call launch by launch instead of replaying its captured hipGraph."""
        self._check(self._lib.cafe_set_graphs(self._h, 1 if on else 0))


class _Borrowed(Context):
    """A shard's cafe_ctx owned by a Sharded object: statistics and introspection only."""

    def __init__(self, lib, handle, n_families):
        self._lib, self._h, self.n_families = lib, handle, n_families

    def close(self):
        self._h = None


class Sharded:
    """cafe_create_sharded: one process, several GPUs -- family shards, one host thread and stream per device, one
    RCCL all-reduce per scorer call."""

    def __init__(self, pb: Problem, devices, max_categories: int = 1, dedup: bool = True, subtree_dedup: bool = True):
        self._lib = load()
        keep = []
        cp = _c_problem(pb, keep, max_categories, 0, (0 if dedup else CAFE_FLAG_NO_DEDUP) | (0 if subtree_dedup else CAFE_FLAG_NO_SUBTREE_DEDUP))
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        err = C.create_string_buffer(512)
        self._h = self._lib.cafe_create_sharded(C.byref(cp), _p(dev, _i32p), len(dev), err, 512)
        if not self._h:
            raise CafeError(err.value.decode() or "cafe_create_sharded failed")
        self.n_families = pb.n_families
        self.size = self._lib.cafe_sharded_size(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cafe_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise CafeError("code %d: %s" % (rc, self._lib.cafe_sharded_last_error(self._h).decode()))

    def score(self, pr: Params, alpha: float = 1.0, per_family: bool = False):
        cp, keep = _c_params(pr, alpha)
        out = C.c_double()
        self._check(self._lib.cafe_sharded_score(self._h, C.byref(cp), C.byref(out), None))
        if not per_family:
            return out.value
        return out.value, self.family_results(len(pr.multipliers) if pr.multipliers is not None else 0)

    def family_results(self, K: int = 0):
        F = self.n_families
        fo = CafeFamilyOut()
        res = {"family_lnl": np.empty(F), "failed": np.empty(F, dtype=np.int32)}
        fo.family_lnl = _p(res["family_lnl"], _f64p)
        fo.failed = _p(res["failed"], _i32p)
        if K > 0:
            res["category_likelihood"] = np.empty((F, K))
            res["family_likelihood"] = np.empty(F)
            fo.category_likelihood = _p(res["category_likelihood"], _f64p)
            fo.family_likelihood = _p(res["family_likelihood"], _f64p)
        self._check(self._lib.cafe_sharded_family_results(self._h, C.byref(fo)))
        return res

    def shard(self, r: int) -> Context:
        h = self._lib.cafe_sharded_context(self._h, r)
        if not h:
            raise CafeError("no shard %d" % r)
        return _Borrowed(self._lib, h, 0)


def build_matrices(n: int, lambdas, ts, device: int = 0, layout: int = 0) -> np.ndarray:
    lam = np.ascontiguousarray(lambdas, dtype=np.float64)
    t = np.ascontiguousarray(ts, dtype=np.float64)
    out = np.empty((len(lam), n, n))
    rc = load().cafe_build_matrices(device, n, len(lam), _p(lam, _f64p), _p(t, _f64p), layout, _p(out, _f64p))
    if rc:
        raise CafeError("cafe_build_matrices failed with code %d" % rc)
    return out


def probe_fp64_mfma(device: int = 0) -> float:
    v = C.c_double()
    rc = load().cafe_probe_fp64_mfma(device, C.byref(v))
    if rc:
        raise CafeError("cafe_probe_fp64_mfma failed with code %d" % rc)
    return v.value
