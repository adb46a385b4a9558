"""Discrete-gamma rate categories for test/bench plumbing (mean rate of K equal-probability
categories of Gamma(alpha, alpha)), computed from scipy's exact quantiles.

The reference uses PAML's AS91/AS32 approximations (src/gamma.cpp:15-240); the C++ host adapter
(cafexp_amd/host/discrete_gamma.cpp) restates those so that lambda*m_k quantizes exactly like the
reference's scorer.  This scipy version agrees with them to ~1e-9 relative and is only used to make
INPUTS (multipliers are a parameter of the C ABI), never to judge parity.
"""
from __future__ import annotations

import numpy as np


def discrete_gamma(K: int, alpha: float):
    from scipy.special import gammainc
    from scipy.stats import gamma as gdist
    cuts = gdist.ppf(np.arange(1, K) / K, a=alpha, scale=1.0 / alpha)
    upper = np.concatenate([gammainc(alpha + 1.0, cuts * alpha), [1.0]])
    lower = np.concatenate([[0.0], upper[:-1]])
    return np.full(K, 1.0 / K), (upper - lower) * K
