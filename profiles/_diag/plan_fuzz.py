"""navsim_plan (breadth-first search on bitmaps, round 4) against the oracle on random costmaps of many shapes -- square and not,
widths around the 64-bit word boundaries, sizes on both sides of the word-per-thread limit (128 x 128), sparse, dense and maze-like
obstacle patterns, unreachable goals, start == goal.  Every output (waypoints, count, path cells, path length) must be identical."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for p in ("nav-gym_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import ref
from nav_gym_amd import sim
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
shapes = [(20, 20), (63, 64), (64, 65), (100, 100), (97, 131), (128, 128), (129, 128), (128, 129), (150, 90), (200, 200), (180, 222), (60, 250)]
total = mism = reached = 0
for (Hc, Wc) in shapes:
    for kind in ("sparse", "dense", "walls"):
        n_maps, Q = 4, 40
        cost = np.zeros((n_maps, Hc, Wc), np.uint8)
        for m in range(n_maps):
            if kind == "sparse":
                cost[m] = rng.random((Hc, Wc)) < 0.08
            elif kind == "dense":
                cost[m] = rng.random((Hc, Wc)) < 0.33
            else:
                for _ in range(max(Hc, Wc) // 6):               # walls with one gap each
                    if rng.random() < 0.5:
                        r0 = rng.integers(1, Hc - 1); cost[m, r0, :] = 1; cost[m, r0, rng.integers(0, Wc)] = 0
                    else:
                        c0 = rng.integers(1, Wc - 1); cost[m, :, c0] = 1; cost[m, rng.integers(0, Hc), c0] = 0
        mi = rng.integers(0, n_maps, n_maps * Q).astype(np.int32)
        res = 0.25
        start = np.stack([rng.uniform(0, Wc * res, n_maps * Q), rng.uniform(0, Hc * res, n_maps * Q)], 1)
        goal = np.stack([rng.uniform(0, Wc * res, n_maps * Q), rng.uniform(0, Hc * res, n_maps * Q)], 1)
        goal[::17] = start[::17]
        for interval, P in ((2.0, 64), (0.6, 16)):
            exp = ref.plan(cost, start, goal, interval, max_wp=P, res_c=res, map_index=mi)
            got = sim.plan(torch.from_numpy(cost).cuda(), torch.from_numpy(start).cuda(), torch.from_numpy(goal).cuda(), interval,
                           max_wp=P, res_c=res, map_index=torch.from_numpy(mi).cuda())
            got = [g.cpu().numpy() for g in got]
            n_wp = exp[1]
            live = np.arange(P)[None, :] < n_wp[:, None]
            ok = (np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]) and np.array_equal(got[3], exp[3])
                  and np.array_equal(got[0][live], exp[0][live]))
            total += len(mi); mism += 0 if ok else 1; reached += int((n_wp > 0).sum())
            if not ok:
                print("MISMATCH", Hc, Wc, kind, interval)
print("%d queries on %d shapes x 3 obstacle patterns x 2 intervals: %d joined by a path, %d mismatching batches" % (total, len(shapes), reached, mism))
sys.exit(1 if mism else 0)
