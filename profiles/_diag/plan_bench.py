import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nav-gym_amd"))
import numpy as np, torch
from nav_gym_amd import sim, world
n_maps, size = 64, 500
occ = torch.from_numpy(world.make_maps(n_maps, size, 5)).cuda()
cost = sim.costmap(occ)
rng = np.random.default_rng(0)
for nq in (64, 256, 1024, 4096):
    c = cost.cpu().numpy()
    mi = rng.integers(0, n_maps, nq).astype(np.int32)
    st, go = np.zeros((nq, 2)), np.zeros((nq, 2))
    for q in range(nq):
        free = np.argwhere(c[mi[q]] == 0)
        while True:
            a, b = free[rng.integers(len(free))], free[rng.integers(len(free))]
            if np.hypot(*(a - b)) * 0.25 > 10: break
        st[q] = (a[1] + .5) * .25, (a[0] + .5) * .25; go[q] = (b[1] + .5) * .25, (b[0] + .5) * .25
    S, G, M = torch.from_numpy(st).cuda(), torch.from_numpy(go).cuda(), torch.from_numpy(mi).cuda()
    import ctypes as C
    L = sim.load()
    wp = torch.zeros((nq, 16, 2), dtype=torch.float64, device="cuda"); nw = torch.zeros(nq, dtype=torch.int32, device="cuda")
    cells = torch.zeros(nq, dtype=torch.int32, device="cuda"); plen = torch.zeros(nq, dtype=torch.float64, device="cuda")
    def go():
        rc = L.navsim_plan(cost.data_ptr(), M.data_ptr(), nq, 100, 100, 0.25, 0.0, 0.0, S.data_ptr(), G.data_ptr(), 2.0, 16,
                           wp.data_ptr(), nw.data_ptr(), cells.data_ptr(), plen.data_ptr(), None)
        assert rc == 0
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): go()
    e1.record(); torch.cuda.synchronize()
    print("queries %5d  %.1f us per launch (events)  mean path cells %.0f  max %d  found %.2f" % (
        nq, e0.elapsed_time(e1) * 1e3 / 20, cells.float().mean().item(), cells.max().item(), (nw > 0).float().mean().item()), flush=True)
