"""How much of ped_update_kernel would a side stream hide behind the step kernel (c3)?  Diagnostic build
-DNAVSIM_EXPERIMENT_PED_OVERLAP with profiles/_diag/tried/ped_overlap_experiment.patch.txt applied (the two halves of a
split step as separate calls; the overlapped run is NOT a valid
step -- it only times the schedule):   NAVSIM_LIB=build/libnavsim_exp.so python profiles/_diag/ped_overlap.py"""
import ctypes as C, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd"))
import torch, bench
from nav_gym_amd import abi, lib
wl = dict(bench.WORKLOADS["c3"]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
L = lib.load()
cfgp, stp, iop = C.POINTER(abi.NavsimConfig), C.POINTER(abi.NavsimState), C.POINTER(abi.NavsimStepIO)
L.navsim_exp_ped_update.argtypes = [cfgp, stp, C.c_void_p]
L.navsim_exp_step_peds_done.argtypes = [cfgp, stp, iop, C.c_void_p]
E = cfg.n_envs
g = torch.Generator(device="cuda:0"); g.manual_seed(5)
acts = torch.rand((64, E, 2), generator=g, device="cuda:0", dtype=torch.float64); acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
sim.t["scan_noise_std"].fill_(0.02); sim.cfg.add_scan_noise = 1
side = torch.cuda.Stream(priority=-1)
main = torch.cuda.current_stream()

def run(mode, n=200):
    ev = torch.cuda.Event()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(n):
        sim.io.action = acts[t % 64].data_ptr(); sim._reorder(); sim._flip()
        if mode == "serial":
            L.navsim_exp_ped_update(C.byref(sim.cfg), C.byref(sim.st), C.c_void_p(main.cuda_stream))
            L.navsim_exp_step_peds_done(C.byref(sim.cfg), C.byref(sim.st), C.byref(sim.io), C.c_void_p(main.cuda_stream))
        elif mode == "overlap":           # ped update of the NEXT step beside this step's kernel
            L.navsim_exp_ped_update(C.byref(sim.cfg), C.byref(sim.st), C.c_void_p(side.cuda_stream))
            L.navsim_exp_step_peds_done(C.byref(sim.cfg), C.byref(sim.st), C.byref(sim.io), C.c_void_p(main.cuda_stream))
            ev.record(side); main.wait_event(ev)
            ev2 = torch.cuda.Event(); ev2.record(main); side.wait_event(ev2)
        else:                             # the step kernel alone
            L.navsim_exp_step_peds_done(C.byref(sim.cfg), C.byref(sim.st), C.byref(sim.io), C.c_void_p(main.cuda_stream))
        sim.cur = 1 - sim.cur
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

for m in ("serial", "overlap", "step only", "serial", "overlap"):
    run(m, 30)
    print("%-10s %.1f us per step" % (m, run(m)))
