"""Cost of reference-faithful pedestrians: navsim_ped_scans -> navsim_ped_policy ->
navsim_step(NAVSIM_PED_EXTERNAL) on the c3 world (4096 arenas x 20 pedestrians), random actor weights.
Run on the GPU box: python profiles/policy_cost.py   (writes gpurun_out/policy_cost.json)"""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "nav-gym_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nav_gym_amd import abi

wl = dict(bench.WORKLOADS["c3"]); wl["field"] = "u16t"
cfg, sim, arrays, _ = bench.build_sim(wl, 0, wl["envs"])
sim.cfg.ped_model = abi.PED_EXTERNAL
rng = np.random.default_rng(0)
fan = {"cv1": 15, "cv2": 96, "fc1": 4096, "fc2": 260, "a1": 128, "a2": 128}
sim.set_policy({k: rng.uniform(-1, 1, s).astype(np.float32) / np.sqrt(fan[k.split("_")[0]]) for k, s in abi.POLICY_SHAPES.items()})
E = cfg.n_envs
act = torch.rand((E, 2), dtype=torch.float64, device="cuda:0"); act[:, 0] *= 0.5; act[:, 1] = act[:, 1] * 1.28 - 0.64


def timed(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


scans = sim.ped_scans()
rows = {
    "ped_scans_ms": timed(lambda: sim.ped_scans()),
    "ped_policy_ms": timed(lambda: sim.ped_policy(scans)),
    "step_external_ms": timed(lambda: sim.step(act)),
}
def two_calls():
    sim.ped_policy(sim.ped_scans(), pipeline=False); sim.step(act)
rows["full_step_two_calls_ms"] = timed(two_calls)          # navsim_ped_scans -> navsim_ped_policy: the scans through HBM
# round 5: the scans of the next slice of arenas beside the network of the current one (NavSim._ped_policy_pipelined)
for n_slices in (3, 4, 6, 8, 12):
    rows["ped_scans_policy_pipelined_%d_ms" % n_slices] = timed(lambda: sim.ped_policy(pipeline=n_slices))
def pipelined():
    sim.ped_policy(pipeline=3); sim.step(act)
rows["full_step_pipelined_3_ms"] = timed(pipelined)
rows["full_step_ms"] = rows["full_step_two_calls_ms"]
rows["ped_scan_policy_fused_ms"] = timed(lambda: sim.ped_policy(fused=True))   # round 4: navsim_ped_scan_policy (scan -> conv in one workgroup)
def fused():
    sim.ped_policy(fused=True); sim.step(act)
rows["full_step_fused_ms"] = timed(fused)
P = E * cfg.max_peds
rows["pedestrians"] = P
rows["env_steps_per_s"] = E / (rows["full_step_ms"] * 1e-3)
print(json.dumps(rows, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "policy_cost.json"), "w"), indent=1)
