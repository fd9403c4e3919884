import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "nav-gym_amd"))
import numpy as np, torch
from nav_gym_amd import sim
n = 1 << 22
x = torch.arange(n, dtype=torch.float64, device="cuda:0")
exact = np.sqrt(np.arange(n, dtype=np.float64)).astype(np.float32)       # correctly rounded: sqrt in f64 of an exact int, then round (double rounding safe for 24-bit ints? check vs f32 sqrt)
exact32 = np.sqrt(np.arange(n, dtype=np.float32))
print("numpy f32 sqrt == f64->f32:", np.array_equal(exact, exact32))
for fn in (6, 7, 8, 9, 10):
    got = sim.debug_math(fn, x).cpu().numpy().astype(np.float32)
    bad = np.where(got != exact32)[0]
    print("fn", fn, "mismatches below 65536:", int((bad < 65536).sum()), " below 2^22:", len(bad), "first", bad[:8])

a = sim.debug_math(11, x).cpu().numpy(); b = sim.debug_math(12, x).cpu().numpy()
bad = np.where(a != b)[0]
d = np.sqrt(np.arange(n, dtype=np.float32))
ref = np.maximum((d.astype(np.float64) * 0.999).astype(np.float32), np.float32(1.0)).astype(np.float64)
print("march step: device F64 rule vs numpy:", int((a != ref).sum()), " f32-only candidate vs F64 rule: mismatches", len(bad), bad[:8])
