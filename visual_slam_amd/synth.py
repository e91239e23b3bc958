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


def two_view(n, seed, inlier_fraction=0.6, noise=0.5, width=1600.0, height=1200.0, with_pose=False):
    """Keypoints of two views of a random 3-D scene and n putative matches, the first
    round(inlier_fraction * n) of them true correspondences (pixel noise `noise`), the rest random
    pairings, then shuffled.  Returns (xy0 [n, 2], xy1 [n, 2], idx0, idx1, is_true, F) with
    x1^T F x0 = 0 for the true ones; with_pose adds (K, R, t): X1 = R X0 + t in camera coordinates."""
    rng = np.random.default_rng(seed)
    f = 1200.0
    K = np.array([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1.0]])
    ang = rng.uniform(-0.25, 0.25, 3)
    cx, cy, cz = np.cos(ang)
    sx, sy, sz = np.sin(ang)
    R = (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1.0]]) @ np.array([[cy, 0, sy], [0, 1.0, 0], [-sy, 0, cy]])
         @ np.array([[1.0, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    t = rng.uniform(-1.0, 1.0, 3)
    t[2] *= 0.3
    X = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(4, 12, n)], 1)
    p0 = (K @ X.T).T
    p1 = (K @ (R @ X.T + t[:, None])).T
    xy0 = p0[:, :2] / p0[:, 2:3]
    xy1 = p1[:, :2] / p1[:, 2:3]
    ntrue = int(round(inlier_fraction * n))
    xy0 = xy0 + noise * rng.standard_normal(xy0.shape)
    xy1 = xy1 + noise * rng.standard_normal(xy1.shape)
    idx0 = np.arange(n)
    idx1 = np.arange(n)
    idx1[ntrue:] = rng.permutation(n)[: n - ntrue]          # wrong partners for the rest
    is_true = np.arange(n) < ntrue
    is_true[ntrue:] = idx1[ntrue:] == idx0[ntrue:]
    order = rng.permutation(n)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ tx @ R @ Kinv
    out = (xy0.astype(np.float32), xy1.astype(np.float32), idx0[order].astype(np.int32),
           idx1[order].astype(np.int32), is_true[order], F / np.linalg.norm(F))
    return out + (K, R, t) if with_pose else out


def multi_view(nimages, n, seed, shared=0.6, desc_noise=4, pixel_noise=0.4):
    """`nimages` views of one random 3-D scene for end-to-end tests: per image a keypoint array
    [n, 2] and SIFT-like descriptors [n, 128].  round(shared * n) features of every image show scene
    points (descriptor = the point's descriptor plus noise, keypoint = its projection plus noise, in a
    per-image random order); the rest are unrelated.  Returns (keypoints, descriptors) lists."""
    rng = np.random.default_rng(seed)
    npts = int(round(shared * n))
    X = np.stack([rng.uniform(-4, 4, npts), rng.uniform(-3, 3, npts), rng.uniform(5, 12, npts)], 1)
    base = sift_like(npts, rng)
    f, w, h = 1200.0, 1600.0, 1200.0
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1.0]])
    kps, dess = [], []
    for _ in range(nimages):
        a = rng.uniform(-0.2, 0.2, 3)
        ca, sa = np.cos(a), np.sin(a)
        R = (np.array([[ca[2], -sa[2], 0], [sa[2], ca[2], 0], [0, 0, 1.0]]) @ np.array([[ca[1], 0, sa[1]], [0, 1.0, 0], [-sa[1], 0, ca[1]]])
             @ np.array([[1.0, 0, 0], [0, ca[0], -sa[0]], [0, sa[0], ca[0]]]))
        t = rng.uniform(-1.0, 1.0, 3) * np.array([1.0, 1.0, 0.3])
        p = (K @ (R @ X.T + t[:, None])).T
        xy = p[:, :2] / p[:, 2:3] + pixel_noise * rng.standard_normal((npts, 2))
        kp = np.empty((n, 2), np.float32)
        des = sift_like(n, rng)
        slots = rng.permutation(n)[:npts]
        kp[:] = rng.uniform(0, [w, h], (n, 2))
        kp[slots] = xy
        des[slots] = np.clip(base.astype(np.int32) + rng.integers(-desc_noise, desc_noise + 1, (npts, 128)), 0, 255).astype(np.uint8)
        kps.append(kp)
        dess.append(des)
    return kps, dess


def image_groups(ngroups, per_group, n, seed, shared=0.5, noise=4.0, dtype="u8", dim=128):
    """`ngroups * per_group` descriptor sets; the images of one group show one scene: round(shared * n)
    rows of each are perturbed copies of rows of the group's pool, the rest is clutter.  Groups share
    nothing.  Returns (list of arrays, group index per image).  dtype "u8" (SIFT-like, 128) or "f32"."""
    rng = np.random.default_rng(seed)
    ns = int(round(shared * n))
    images, group = [], []
    for g in range(ngroups):
        pool = sift_like(2 * ns, rng) if dtype == "u8" else unit_f32(2 * ns, dim, rng)
        for _ in range(per_group):
            pick = rng.permutation(2 * ns)[:ns]
            if dtype == "u8":
                seen = np.clip(pool[pick].astype(np.float32) + rng.normal(0.0, noise, (ns, 128)), 0, 255)
                img = np.concatenate([np.rint(seen).astype(np.uint8), sift_like(n - ns, rng)])
            else:
                seen = pool[pick] + rng.normal(0.0, noise, (ns, dim)).astype(np.float32)
                seen /= np.linalg.norm(seen, axis=1, keepdims=True).astype(np.float32)
                img = np.concatenate([seen.astype(np.float32), unit_f32(n - ns, dim, rng)])
            images.append(np.ascontiguousarray(img[rng.permutation(n)]))
            group.append(g)
    return images, np.asarray(group, np.int32)


def sample_vocabulary(images, nwords, seed):
    """`nwords` rows drawn without replacement from the rows of all images: a cheap visual vocabulary."""
    rng = np.random.default_rng(seed)
    allrows = np.concatenate(images)
    return np.ascontiguousarray(allrows[rng.permutation(len(allrows))[:nwords]])
