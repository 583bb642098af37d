"""Deterministic generators of banded-DP jobs (query/target pairs) for the parity tests."""
import numpy as np


def mutate(rng, t, sub, ins, dele, n_frac=0.0):
    out = []
    for b in t:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            b = (b + rng.integers(1, 4)) & 3
        out.append(b)
        while rng.random() < ins:
            out.append(rng.integers(0, 4))
    q = np.array(out, dtype=np.uint8)
    if n_frac > 0 and len(q):
        q[rng.random(len(q)) < n_frac] = 4
    return q


def make_jobs(seed, n, max_len=300, err=(0.04, 0.04, 0.04), tail_noise=True):
    """Return a list of (query, target) uint8 arrays with ragged sizes, including empty ones."""
    rng = np.random.default_rng(seed)
    jobs = []
    for i in range(n):
        kind = rng.integers(0, 10)
        tl = int(rng.integers(0, max_len + 1)) if kind else int(rng.integers(0, 4))
        t = rng.integers(0, 4, size=tl, dtype=np.uint8)
        scale = [0.0, 0.3, 1.0, 1.0, 1.0, 2.0, 1.0, 1.0, 4.0, 1.0][kind]
        q = mutate(rng, t, err[0] * scale, err[1] * scale, err[2] * scale, 0.01 if kind == 6 else 0.0)
        if kind == 7 and tail_noise:      # unrelated tails: exercises z-drop / local end / mid-fix
            cut = int(rng.integers(0, len(q) + 1))
            q = np.concatenate([q[:cut], rng.integers(0, 4, size=int(rng.integers(0, 120)), dtype=np.uint8)])
        if kind == 9:                     # length imbalance
            t = np.concatenate([t, rng.integers(0, 4, size=int(rng.integers(0, 60)), dtype=np.uint8)])
        if kind == 3 and len(q) > 10:     # unrelated middle
            a = len(q) // 3
            q = np.concatenate([q[:a], rng.integers(0, 4, size=int(rng.integers(1, 150)), dtype=np.uint8), q[a:]])
        jobs.append((np.ascontiguousarray(q, dtype=np.uint8), np.ascontiguousarray(t, dtype=np.uint8)))
    return jobs
