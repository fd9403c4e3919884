"""Is navsim_regen bound by COLD INSTRUCTION FETCH?  After every step of a c5 world the call is issued twice in a row on
the same done flags (same arenas, same work): the first call's kernels follow the step kernel (whose ~100 KB of code has
just gone through the instruction caches), the second call's kernels follow themselves.  Run under
rocprofv3 --kernel-trace; profiles/_diag/regen_twice.sh prints the mean duration of each kernel by position."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench
wl = dict(bench.WORKLOADS["c5"]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
E = cfg.n_envs
if os.environ.get("NAVSIM_NO_CHECK"):             # without reset()'s first-scan test of the robot (env.py:776-781)
    sim.cfg.regen_check_discomfort = 0
if os.environ.get("NAVSIM_NO_PEDS"):              # no pedestrians to place
    sim.t["n_peds"].zero_()
if os.environ.get("NAVSIM_REGEN_CAP"):            # slots of a navsim_regen call (workgroups beyond the finished arenas exit at once)
    sim.cfg.regen_cap = int(os.environ["NAVSIM_REGEN_CAP"])
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
acts = torch.rand((120, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 1.0; acts[..., 1] = acts[..., 1] * 4.0 - 2.0
for t in range(120):
    sim.io.action = acts[t].data_ptr()
    sim.launch_step()
    sim.regen()
    sim.regen()
torch.cuda.synchronize()
