"""What does a breadth-first level of navsim_plan cost?  An empty 100 x 100 costmap (0.25 m cells), 64 identical queries whose start
lies L cells from the goal along a row: the search runs L levels, the walk L cells.  Time per call vs L -> us per level + walk cell."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch
from nav_gym_amd import sim
dev = "cuda:0"
cost = torch.zeros((1, 100, 100), dtype=torch.uint8, device=dev)
mi = torch.zeros(64, dtype=torch.int32, device=dev)
rows = []
for L in (5, 20, 40, 60, 80, 98):
    start = torch.tensor([[0.125 + 0.25 * 0, 12.625]] * 64, dtype=torch.float64, device=dev)
    goal = torch.tensor([[0.125 + 0.25 * L, 12.625]] * 64, dtype=torch.float64, device=dev)
    for _ in range(3):
        out = sim.plan(cost, start, goal, 2.0, max_wp=64, map_index=mi)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        out = sim.plan(cost, start, goal, 2.0, max_wp=64, map_index=mi)
    e1.record(); torch.cuda.synchronize()
    rows.append((L, e0.elapsed_time(e1) / 20 * 1e3, int(out[2][0])))
    print("L = %3d cells: %.1f us per call (path cells %d)" % rows[-1])
(l0, t0, _), (l1, t1, _) = rows[1], rows[-1]
print("slope %.2f us per level-and-walk-cell" % ((t1 - t0) / (l1 - l0)))
