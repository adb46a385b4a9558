"""TEST INFRASTRUCTURE -- ctypes binding of oracle/libcafe_oracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcafe_oracle.so")
REF_HARNESS = os.path.join(_HERE, "_ref", "ref_harness")

_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)


class _Tree(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("parent", _i32p), ("branch_length", _f64p),
                ("lambda_index", _i32p), ("leaf_taxon", _i32p)]


class _Problem(C.Structure):
    _fields_ = [("tree", _Tree), ("n_taxa", C.c_int32), ("n_families", C.c_int64), ("counts", _i32p),
                ("max_family_size", C.c_int32), ("max_root_family_size", C.c_int32), ("n_lambdas", C.c_int32),
                ("single_lambda", C.c_int32), ("n_deviations", C.c_int32)]


class _Params(C.Structure):
    _fields_ = [("lambdas", _f64p), ("n_categories", C.c_int32), ("multipliers", _f64p), ("cat_probs", _f64p),
                ("prior", _f32p), ("error_model", _f64p), ("fast_matrices", C.c_int32)]


def build(force: bool = False) -> None:
    """Compile the C restatement (and the real reference when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cafe_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_chooseln.restype = C.c_double
        L.orc_chooseln.argtypes = [C.c_double, C.c_double]
        L.orc_bd_log_alpha.restype = C.c_double
        L.orc_bd_log_alpha.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_bd_prob.restype = C.c_double
        L.orc_bd_prob.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_quantize.argtypes = [C.c_double, C.c_double, _f64p, _f64p]
        L.orc_is_saturated.restype = C.c_int
        L.orc_is_saturated.argtypes = [C.c_double, C.c_double]
        for fn in (L.orc_build_matrix, L.orc_build_matrix_conv):
            fn.argtypes = [C.c_int, C.c_double, C.c_double, _f64p]
        L.orc_matvec.argtypes = [_f64p, C.c_int, _f64p, C.c_int, C.c_int, C.c_int, C.c_int, _f64p]
        L.orc_point_normal.restype = C.c_double
        L.orc_point_normal.argtypes = [C.c_double]
        L.orc_incomplete_gamma.restype = C.c_double
        L.orc_incomplete_gamma.argtypes = [C.c_double] * 3
        L.orc_point_chi2.restype = C.c_double
        L.orc_point_chi2.argtypes = [C.c_double] * 2
        L.orc_discrete_gamma.argtypes = [C.c_int, C.c_double, _f64p, _f64p]
        L.orc_prior_uniform.argtypes = [C.c_int, _f32p]
        L.orc_prior_poisson.argtypes = [C.c_int, C.c_double, _f32p]
        L.orc_prior_rootdist.argtypes = [C.c_int, _i32p, _i32p, C.c_int, _f32p]
        L.orc_prune.restype = C.c_int
        L.orc_prune.argtypes = [C.POINTER(_Problem), C.POINTER(_Params), C.c_int64, C.c_double, _f64p]
        L.orc_score_base.restype = C.c_double
        L.orc_score_base.argtypes = [C.POINTER(_Problem), C.POINTER(_Params), _f64p]
        L.orc_score_gamma.restype = C.c_double
        L.orc_score_gamma.argtypes = [C.POINTER(_Problem), C.POINTER(_Params), _f64p, _f64p]
        L.orc_time_matrices.restype = C.c_double
        L.orc_time_matrices.argtypes = [C.c_int, C.c_double, _f64p, C.c_int, C.c_int]
        L.orc_num_threads.restype = C.c_int
        L.orc_last_timings.argtypes = [_f64p, _f64p]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_reconstruct.restype = C.c_int
        L.orc_reconstruct.argtypes = [C.POINTER(_Problem), C.POINTER(_Params), _f32p, _i32p]
        L.orc_branch_probabilities.restype = C.c_int
        L.orc_branch_probabilities.argtypes = [C.POINTER(_Problem), C.POINTER(_Params), _i32p, _f64p]
        L.orc_root_max.restype = C.c_int
        L.orc_root_max.argtypes = [C.POINTER(_Problem), C.POINTER(_Params), _f64p]
        L.orc_pvalue.restype = C.c_double
        L.orc_pvalue.argtypes = [C.c_double, _f64p, C.c_int]
        L.orc_tree_pvalues.restype = None
        L.orc_tree_pvalues.argtypes = [_f64p, C.c_int64, _f64p, C.c_int, C.c_int, _f64p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class _Bound:
    """Keeps the numpy buffers alive next to the ctypes structs that point into them."""

    def __init__(self, pb, pr=None, fast: bool = False):
        self.keep = []
        k = lambda a, dt: self._c(a, dt)
        self.pb = _Problem()
        self.pb.tree = _Tree(pb.n_nodes, _p(k(pb.parent, np.int32), _i32p), _p(k(pb.branch_length, np.float64), _f64p),
                             _p(k(pb.lambda_index, np.int32), _i32p), _p(k(pb.leaf_taxon, np.int32), _i32p))
        self.pb.n_taxa = pb.n_taxa
        self.pb.n_families = pb.n_families
        self.pb.counts = _p(k(pb.counts, np.int32), _i32p)
        self.pb.max_family_size = pb.max_family_size
        self.pb.max_root_family_size = pb.max_root_family_size
        self.pb.n_lambdas = pb.n_lambdas
        self.pb.single_lambda = 1 if pb.single_lambda else 0
        self.pb.n_deviations = pb.n_deviations
        self.pr = None
        if pr is not None:
            K = 1 if pr.multipliers is None else len(pr.multipliers)
            mult = pr.multipliers if pr.multipliers is not None else np.ones(1)
            cp = pr.cat_probs if pr.cat_probs is not None else np.ones(1)
            self.pr = _Params(_p(k(pr.lambdas, np.float64), _f64p), K, _p(k(mult, np.float64), _f64p),
                              _p(k(cp, np.float64), _f64p), _p(k(pr.prior, np.float32), _f32p),
                              _p(k(pr.error_model, np.float64), _f64p) if pr.error_model is not None else None,
                              1 if fast else 0)

    def _c(self, a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        self.keep.append(a)
        return a


def bd_prob(lam: float, t: float, s: int, c: int) -> float:
    return lib().orc_bd_prob(lam, t, s, c)


def bd_log_alpha(s: int, c: int, log_alpha: float, coeff: float) -> float:
    return lib().orc_bd_log_alpha(s, c, log_alpha, coeff)


def quantize(lam: float, t: float):
    a, b = C.c_double(), C.c_double()
    lib().orc_quantize(lam, t, C.byref(a), C.byref(b))
    return a.value, b.value


def is_saturated(t: float, lam: float) -> bool:
    return bool(lib().orc_is_saturated(t, lam))


def build_matrix(n: int, lam: float, t: float, fast: bool = False) -> np.ndarray:
    out = np.empty((n, n), dtype=np.float64)
    (lib().orc_build_matrix_conv if fast else lib().orc_build_matrix)(n, lam, t, _p(out, _f64p))
    return out


def matvec(mat: np.ndarray, v: np.ndarray, s_min: int, s_max: int, c_min: int, c_max: int) -> np.ndarray:
    mat = np.ascontiguousarray(mat, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.empty(s_max - s_min + 1, dtype=np.float64)
    lib().orc_matvec(_p(mat, _f64p), mat.shape[0], _p(v, _f64p), s_min, s_max, c_min, c_max, _p(out, _f64p))
    return out


def discrete_gamma(K: int, alpha: float):
    probs = np.empty(K, dtype=np.float64)
    mult = np.empty(K, dtype=np.float64)
    lib().orc_discrete_gamma(K, alpha, _p(probs, _f64p), _p(mult, _f64p))
    return probs, mult


def prior_uniform(R: int) -> np.ndarray:
    out = np.empty(R, dtype=np.float32)
    lib().orc_prior_uniform(R, _p(out, _f32p))
    return out


def prior_poisson(R: int, pl: float) -> np.ndarray:
    out = np.empty(R, dtype=np.float32)
    lib().orc_prior_poisson(R, pl, _p(out, _f32p))
    return out


def prior_rootdist(R: int, rootdist: dict) -> np.ndarray:
    sizes = np.array(sorted(rootdist), dtype=np.int32)
    counts = np.array([rootdist[int(s)] for s in sizes], dtype=np.int32)
    out = np.empty(R, dtype=np.float32)
    lib().orc_prior_rootdist(R, _p(sizes, _i32p), _p(counts, _i32p), len(sizes), _p(out, _f32p))
    return out


def prune(pb, pr, family: int, mult: float = 1.0, fast: bool = False) -> np.ndarray:
    b = _Bound(pb, pr, fast)
    out = np.empty(pb.max_root_family_size, dtype=np.float64)
    rc = lib().orc_prune(C.byref(b.pb), C.byref(b.pr), family, mult, _p(out, _f64p))
    if rc:
        raise RuntimeError("oracle: matrix missing (non-positive branch length?)")
    return out


def score_base(pb, pr, fast: bool = False, per_family: bool = False):
    b = _Bound(pb, pr, fast)
    fam = np.empty(pb.n_families, dtype=np.float64) if per_family else None
    v = lib().orc_score_base(C.byref(b.pb), C.byref(b.pr), _p(fam, _f64p))
    return (v, fam) if per_family else v


def score_gamma(pb, pr, fast: bool = False, per_family: bool = False):
    b = _Bound(pb, pr, fast)
    K = len(pr.multipliers)
    cat = np.zeros((pb.n_families, K), dtype=np.float64) if per_family else None
    fam = np.zeros(pb.n_families, dtype=np.float64) if per_family else None
    v = lib().orc_score_gamma(C.byref(b.pb), C.byref(b.pr), _p(cat, _f64p), _p(fam, _f64p))
    return (v, cat, fam) if per_family else v


def root_max(pb, lambdas, fast: bool = False) -> np.ndarray:
    """max_j L_root[j] per family: plain lambda, no error model, no prior (probability.cpp:313, :399)."""
    from cafexp_amd.problem import Params
    pr = Params(lambdas=np.asarray(lambdas, dtype=np.float64), prior=np.ones(pb.max_root_family_size, dtype=np.float32))
    b = _Bound(pb, pr, fast)
    out = np.empty(pb.n_families, dtype=np.float64)
    rc = lib().orc_root_max(C.byref(b.pb), C.byref(b.pr), _p(out, _f64p))
    if rc:
        raise RuntimeError("oracle: root_max failed (%d)" % rc)
    return out


def reconstruct(pb, lambdas, root_prior, multipliers=None, fast: bool = False) -> np.ndarray:
    """Pupko joint reconstruction -> int32 [K][F][n_nodes] (gene_family_reconstructor.cpp:13-165)."""
    from cafexp_amd.problem import Params
    pr = Params(lambdas=np.asarray(lambdas, dtype=np.float64), prior=np.ones(pb.max_root_family_size, dtype=np.float32))
    K = 1
    if multipliers is not None:
        pr.multipliers = np.asarray(multipliers, dtype=np.float64)
        pr.cat_probs = np.full(len(pr.multipliers), 1.0 / len(pr.multipliers))
        K = len(pr.multipliers)
    b = _Bound(pb, pr, fast)
    rp = np.ascontiguousarray(root_prior, dtype=np.float32)
    assert len(rp) >= min(pb.max_family_size, pb.max_root_family_size) + 1
    out = np.empty((K, pb.n_families, pb.n_nodes), dtype=np.int32)
    rc = lib().orc_reconstruct(C.byref(b.pb), C.byref(b.pr), _p(rp, _f32p), _p(out, _i32p))
    if rc:
        raise RuntimeError("oracle: reconstruct failed (%d)" % rc)
    return out


def branch_probabilities(pb, lambdas, sizes, fast: bool = False) -> np.ndarray:
    """compute_viterbi_sum for every (family, node); NaN = invalid (gene_family_reconstructor.cpp:361-400)."""
    from cafexp_amd.problem import Params
    pr = Params(lambdas=np.asarray(lambdas, dtype=np.float64), prior=np.ones(pb.max_root_family_size, dtype=np.float32))
    b = _Bound(pb, pr, fast)
    sz = np.ascontiguousarray(sizes, dtype=np.int32)
    out = np.empty((pb.n_families, pb.n_nodes), dtype=np.float64)
    rc = lib().orc_branch_probabilities(C.byref(b.pb), C.byref(b.pr), _p(sz, _i32p), _p(out, _f64p))
    if rc:
        raise RuntimeError("oracle: branch_probabilities failed (%d)" % rc)
    return out


def pvalue(v: float, conddist) -> float:
    c = np.ascontiguousarray(conddist, dtype=np.float64)
    return lib().orc_pvalue(float(v), _p(c, _f64p), len(c))


def tree_pvalues(observed, cond) -> np.ndarray:
    """cond: [R][nsim], rows sorted ascending -> per-family max over root sizes (probability.cpp:401-407)."""
    obs = np.ascontiguousarray(observed, dtype=np.float64)
    c = np.ascontiguousarray(cond, dtype=np.float64)
    out = np.empty(len(obs), dtype=np.float64)
    lib().orc_tree_pvalues(_p(obs, _f64p), len(obs), _p(c, _f64p), c.shape[0], c.shape[1], _p(out, _f64p))
    return out


def score(pb, pr, fast: bool = False):
    return score_gamma(pb, pr, fast) if pr.multipliers is not None else score_base(pb, pr, fast)


def time_matrices(n: int, lam: float, ts, fast: bool = False) -> float:
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    return lib().orc_time_matrices(n, lam, _p(ts, _f64p), len(ts), 1 if fast else 0)


def last_timings():
    a, b = C.c_double(), C.c_double()
    lib().orc_last_timings(C.byref(a), C.byref(b))
    return a.value, b.value


def host_cpu_share() -> int:
    """CPUs this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


def num_threads() -> int:
    return lib().orc_num_threads()


# --- the real reference (only where oracle/_ref/ref_harness was built) ---------------------------
def have_ref() -> bool:
    return os.path.exists(REF_HARNESS) and os.access(REF_HARNESS, os.X_OK)


def ref(job: str, **kv) -> dict:
    """Run one job of the real reference (oracle/ref_harness.cpp) and parse its JSON line."""
    args = [REF_HARNESS, job] + ["%s=%s" % (k, v) for k, v in kv.items()]
    out = subprocess.run(args, check=True, capture_output=True, text=True).stdout
    line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)

    def fix(v):
        if isinstance(v, str) and v in ("inf", "-inf", "nan"):
            return float(v)
        if isinstance(v, list):
            return [fix(x) for x in v]
        return v
    return {k: fix(v) for k, v in d.items()}
