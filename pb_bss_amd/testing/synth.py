"""Deterministic synthetic multi-microphone STFT generator.

Test/bench infrastructure: shared by the oracle, the parity tests and bench.py
so that every leg sees bit-identical inputs.  Pure NumPy.
"""
import numpy as np


def _cn(rng, shape):
    """Circular complex normal CN(0, 1)."""
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)


def make_stft(F, T, D, K, seed=0, snr_db=20.0, dtype=np.complex64):
    """Synthetic observation Y[f, t, d] and an affiliation initialisation.

    steering a[f,k,:] ~ CN(0, I) unit-norm; sources s[k,f,t] = CN(0,1)*|N(0,1)|^2
    (heavy tailed => time-frequency sparse); x = sum_k a s; white noise at
    ``snr_db``.  Returns (Y complex64 (F,T,D), init float64 (F,K,T) normalised
    over k).
    """
    rng = np.random.default_rng(seed)
    a = _cn(rng, (F, K, D))
    a /= np.linalg.norm(a, axis=-1, keepdims=True)
    s = _cn(rng, (K, F, T)) * rng.standard_normal((K, F, T)) ** 2
    x = np.einsum('fkd,kft->ftd', a, s)
    n = _cn(rng, (F, T, D))
    px = np.mean(np.abs(x) ** 2)
    pn = np.mean(np.abs(n) ** 2)
    n *= np.sqrt(px / pn / (10.0 ** (snr_db / 10.0)))
    Y = (x + n).astype(dtype)
    init = rng.uniform(size=(F, K, T))
    init /= init.sum(axis=1, keepdims=True)
    return Y, init


def make_white(F, T, D, K, seed=1, dtype=np.complex64):
    """Worst-conditioning case: pure CN(0, I) observations."""
    rng = np.random.default_rng(seed)
    Y = _cn(rng, (F, T, D)).astype(dtype)
    init = rng.uniform(size=(F, K, T))
    init /= init.sum(axis=1, keepdims=True)
    return Y, init


def make_rank_deficient(F, T, D, K, rank, seed=2, dtype=np.complex64):
    """Near-singular case (rank < D) that exercises the 1e-10 eigenvalue floor."""
    rng = np.random.default_rng(seed)
    basis = _cn(rng, (F, rank, D))
    coef = _cn(rng, (F, T, rank))
    Y = np.einsum('ftr,frd->ftd', coef, basis).astype(dtype)
    init = rng.uniform(size=(F, K, T))
    init /= init.sum(axis=1, keepdims=True)
    return Y, init


def make_joint(F, T, D, K, E, seed=0, snr_db=20.0, spread=0.35, dtype=np.complex64,
               embedding_dtype=np.float32):
    """STFT plus a Deep-Clustering-style embedding per time-frequency point
    (BASELINE config 5): e[f,t,:] = unit(mu_k* + spread * N(0, I_E)) where k* is the
    dominant source of the point and mu_k are random unit vectors.
    Returns (Y (F,T,D), embedding (F,T,E), init (F,K,T))."""
    rng = np.random.default_rng(seed)
    a = _cn(rng, (F, K, D))
    a /= np.linalg.norm(a, axis=-1, keepdims=True)
    s = _cn(rng, (K, F, T)) * rng.standard_normal((K, F, T)) ** 2
    x = np.einsum('fkd,kft->ftd', a, s)
    n = _cn(rng, (F, T, D))
    n *= np.sqrt(np.mean(np.abs(x) ** 2) / np.mean(np.abs(n) ** 2) / (10.0 ** (snr_db / 10.0)))
    Y = (x + n).astype(dtype)
    init = rng.uniform(size=(F, K, T))
    init /= init.sum(axis=1, keepdims=True)
    mu = rng.standard_normal((K, E))
    mu /= np.linalg.norm(mu, axis=-1, keepdims=True)
    dominant = np.argmax(np.abs(s), axis=0)                       # (F, T)
    e = mu[dominant] + spread * rng.standard_normal((F, T, E))
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    return Y, e.astype(embedding_dtype), init
