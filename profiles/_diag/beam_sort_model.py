"""Offline model: how many wavefront probe rounds a scan costs under different beam->lane assignments.
Counts the probes of each beam with a NumPy sphere march on the exact distance field (same rule as the kernel:
t += max(0.999 d, 1), stop at d == 0 or beyond the range), then sums, over the wavefronts of a 256-thread
workgroup, the maximum count of each wavefront.  python profiles/_diag/beam_sort_model.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from nav_gym_amd import world
import ref

E, SIZE, B = 24, 500, 1081
occ = world.make_maps(E, SIZE, 1)
ang = np.linspace(-0.75 * np.pi, 0.75 * np.pi, B)
rng = np.random.default_rng(0)


def counts(field, i0, j0, th, max_range=500.0):
    d = np.sqrt(field.astype(np.float64))
    dx, dy = np.cos(th + ang), np.sin(th + ang)
    t = np.zeros(B); n = np.zeros(B, int); act = np.ones(B, bool)
    while act.any():
        x = np.floor(i0 + t * dx).astype(int); y = np.floor(j0 + t * dy).astype(int)
        inside = (x >= 0) & (x < SIZE) & (y >= 0) & (y < SIZE) & (t <= max_range)
        act &= inside
        dd = np.where(act, d[np.clip(x, 0, SIZE - 1), np.clip(y, 0, SIZE - 1)], 0.0)
        n += act
        act &= dd > 0
        t = np.where(act, t + np.maximum(0.999 * dd, 1.0), t)
    return n


def rounds(order, n, block=256):
    """sum over generations and wavefronts of the max count; order = beam index per slot"""
    tot = 0
    for g in range(0, B, block):
        for w in range(g, min(g + block, B), 64):
            tot += n[order[w:min(w + 64, B, g + block)]].max()
    return tot


res = {k: 0 for k in ("adjacent", "ideal sort", "prev-step sort", "prev-step sort, shifted", "prev shifted, groups of 8",
                      "prev shifted, groups of 16", "sum/64 (perfect packing)")}
for e in range(E):
    f = ref.build_dt(occ[e][None])[0] if hasattr(ref, "build_dt") else None
    free = np.argwhere(f > 36)
    for trial in range(4):
        i0, j0 = free[rng.integers(len(free))] + 0.5
        th = rng.uniform(-np.pi, np.pi)
        n0 = counts(f, i0, j0, th)
        v, w = rng.uniform(0, 0.5), rng.uniform(-0.64, 0.64)
        th1 = th + w * 0.2
        i1, j1 = i0 + v * 0.2 / 0.05 * np.cos(th1), j0 + v * 0.2 / 0.05 * np.sin(th1)
        if f[int(i1), int(j1)] == 0: continue
        n1 = counts(f, i1, j1, th1)
        ident = np.arange(B)
        res["adjacent"] += rounds(ident, n1)
        res["ideal sort"] += rounds(np.argsort(-n1, kind="stable"), n1)
        res["prev-step sort"] += rounds(np.argsort(-n0, kind="stable"), n1)
        shift = int(round(w * 0.2 / (ang[1] - ang[0])))
        pred = n0[np.clip(ident + shift, 0, B - 1)]
        res["prev-step sort, shifted"] += rounds(np.argsort(-pred, kind="stable"), n1)
        for G in (8, 16):
            ng = (B + G - 1) // G
            key = np.array([pred[g * G:(g + 1) * G].max() for g in range(ng)])
            og = np.argsort(-key, kind="stable")
            order = np.concatenate([np.arange(g * G, min((g + 1) * G, B)) for g in og])
            res["prev shifted, groups of %d" % G] += rounds(order, n1)
        res["sum/64 (perfect packing)"] += n1.sum() / 64.0
base = res["adjacent"]
for k, v in res.items():
    print("%-34s %8.0f  %.3f" % (k, v, v / base))
