"""Synthetic descriptors with the value conventions of the reference's data.

SIFT-128 u8: non-negative components, L2-normalised to 512 and rounded to uint8
(FeaturePoints.cpp:680-684, ConvertA2B).  SuperPoint-256 f32: unit L2 norm
(superpoint.py:192-197).  A share of the second image's descriptors are noisy copies of the
first image's, so ratio and mutual tests pass for some rows and fail for others.
"""
import numpy as np


def sift_like(n, rng):
    x = np.abs(rng.standard_normal((n, 128))).astype(np.float32)
    x *= 512.0 / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-6)
    return np.clip(np.rint(x), 0, 255).astype(np.uint8)


def sift_pair(n0, n1, seed, planted=0.3, noise=6.0):
    """Two u8 sets; about `planted` of set 1 are perturbed copies of random rows of set 0."""
    rng = np.random.default_rng(seed)
    a = sift_like(n0, rng)
    b = sift_like(n1, rng)
    k = int(min(n0, n1) * planted)
    if k > 0:
        src = rng.choice(n0, size=k, replace=False)
        dst = rng.choice(n1, size=k, replace=False)
        pert = a[src].astype(np.float32) + noise * rng.standard_normal((k, 128)).astype(np.float32)
        b[dst] = np.clip(np.rint(pert), 0, 255).astype(np.uint8)
    return a, b


def unit_f32(n, dim, rng):
    x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-12).astype(np.float32)
    return x.astype(np.float32)


def superpoint_pair(n0, n1, dim, seed, planted=0.25, sigma=0.05):
    rng = np.random.default_rng(seed)
    a = unit_f32(n0, dim, rng)
    b = unit_f32(n1, dim, rng)
    k = int(min(n0, n1) * planted)
    if k > 0:
        src = rng.choice(n0, size=k, replace=False)
        dst = rng.choice(n1, size=k, replace=False)
        pert = a[src] + sigma * rng.standard_normal((k, dim)).astype(np.float32)
        pert /= np.linalg.norm(pert, axis=1, keepdims=True).astype(np.float32)
        b[dst] = pert.astype(np.float32)
    return a, b
