"""Deterministic synthetic inputs for the BASELINE.json configs.

Same recipe as oracle/synth.js (which feeds the reference sampler when the golden
fixtures are generated): Philox4x32-10 uniforms combined with +,-,* only, so the
arrays are bit-identical in JavaScript, Python and C.  tests/test_synth.py checks
that against data stored in tests/golden/.

    z = (u1 + u2 + ... + u12) - 6            Irwin-Hall stand-in for N(0,1)
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
DATA_CHAIN0 = 4294967295  # data streams count down from 2^32-1; sampler chains count up from 0


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10; counters are uint64 arrays holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0, k1 = int(k0), int(k1)
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, p1 & _MASK, n2, p0 & _MASK
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniforms(seed, chain, n, start=0):
    """The first n uniforms (from index `start`, which must be even) of stream (seed, chain)."""
    assert start % 2 == 0
    nb = (n + 1) // 2
    b = np.arange(start // 2, start // 2 + nb, dtype=np.uint64)
    z = np.zeros(nb, dtype=np.uint64)
    r0, r1, r2, r3 = philox4x32_10(b & _MASK, b >> np.uint64(32), z + np.uint64(chain & 0xFFFFFFFF),
                                   z + np.uint64(chain >> 32), seed & 0xFFFFFFFF, seed >> 32)
    u = np.empty(nb * 2, dtype=np.float64)
    u[0::2] = ((r0 << np.uint64(21)) | (r1 >> np.uint64(11))).astype(np.float64) * 2.0 ** -53
    u[1::2] = ((r2 << np.uint64(21)) | (r3 >> np.uint64(11))).astype(np.float64) * 2.0 ** -53
    return u[:n]


def _z(seed, chain, n):
    u = uniforms(seed, chain, 12 * n).reshape(n, 12)
    s = u[:, 0].copy()
    for j in range(1, 12):          # sequential adds, same order as synth.js
        s = s + u[:, j]
    return s - 6.0


def normal(n_obs, data_seed):
    """cfg2: x_i ~ N(3, 2)."""
    return {"x": 3.0 + 2.0 * _z(data_seed, DATA_CHAIN0, n_obs)}


def bern(n_obs, data_seed):
    """cfg3: x_i ~ Bernoulli(0.3), returned as float64 0/1 like the JS array."""
    return {"x": (uniforms(data_seed, DATA_CHAIN0, n_obs) < 0.3).astype(np.float64)}


def hier(n_obs, n_groups, data_seed):
    """cfg4: theta_g ~ N(5,3); y_i ~ N(theta[g_i], 2); g_i = i mod G."""
    theta = 5.0 + 3.0 * _z(data_seed, DATA_CHAIN0 - 1, n_groups)
    g = (np.arange(n_obs) % n_groups).astype(np.int32)
    y = theta[g] + 2.0 * _z(data_seed, DATA_CHAIN0, n_obs)
    return {"y": y, "g": g, "G": n_groups, "theta_true": theta}


def glm(n_obs, data_seed, exp=None):
    """cfg5: Poisson GLM, 7 real columns (first = 1) + change-point shift beta[7].

    `exp` must be the bit-exact fdlibm exp when the counts have to match synth.js
    (tests pass the library's amwg_exp); defaults to numpy's exp.
    """
    K = 7
    beta = [0.5, 0.2, -0.1, 0.05, 0.1, -0.2, 0.15, 0.3]
    cp = int(np.floor(0.4 * n_obs))
    X = np.empty((n_obs, K), dtype=np.float64)
    X[:, 0] = 1.0
    X[:, 1:] = 0.5 * _z(data_seed, DATA_CHAIN0, n_obs * (K - 1)).reshape(n_obs, K - 1)
    eta = np.zeros(n_obs)
    for k in range(K):
        eta = eta + X[:, k] * beta[k]
    eta = np.where(np.arange(n_obs) >= cp, eta + beta[7], eta)
    ex = np.exp if exp is None else np.vectorize(exp, otypes=[np.float64])
    lam = ex(eta)
    u = uniforms(data_seed, DATA_CHAIN0 - 1, n_obs)
    p = ex(-lam)
    F = p.copy()
    y = np.zeros(n_obs)
    for n in range(1, 1001):        # inversion by sequential search, same order as synth.js
        live = u > F
        if not live.any():
            break
        p = np.where(live, p * lam / n, p)
        F = np.where(live, F + p, F)
        y = np.where(live, float(n), y)
    return {"X": X, "y": y, "K": K, "cp_true": cp, "beta_true": np.array(beta)}
