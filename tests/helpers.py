"""Shared helpers for the parity tests (tests only)."""
import numpy as np


def random_placement(rng, B, nb, ny, half_len, half_wid, min_d, spread=1.0):
    """Non-overlapping random poses: ball [B,4], blue [B,nb,3], yellow [B,ny,3]."""
    N = nb + ny
    ball = np.zeros((B, 4))
    rob = np.zeros((B, N, 3))
    for e in range(B):
        pts = []
        while len(pts) < N + 1:
            p = np.array([rng.uniform(-half_len, half_len) * spread,
                          rng.uniform(-half_wid, half_wid) * spread])
            if all(np.hypot(*(p - q)) >= min_d for q in pts):
                pts.append(p)
        ball[e, :2] = pts[0]
        ball[e, 2:] = rng.uniform(-1.0, 1.0, 2)
        for k in range(N):
            rob[e, k, :2] = pts[k + 1]
            rob[e, k, 2] = rng.uniform(-180, 180)
    return ball, rob[:, :nb].copy(), rob[:, nb:].copy()


def f32_equal(a, b):
    """Bit-exact comparison of float32 values carried in float64 arrays (NaN == NaN)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or np.array_equal(a, b, equal_nan=True)


def mismatch_report(a, b, name=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b)
    idx = np.unravel_index(np.nanargmax(d), d.shape)
    return f"{name}: max |diff| {np.nanmax(d):.3e} at {idx}: {a[idx]!r} vs {b[idx]!r}; mismatches {np.count_nonzero(a != b)}/{a.size}"
