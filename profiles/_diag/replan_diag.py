import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "nav-gym_amd"))
import torch
import nav_gym_env
env = nav_gym_env.make('NavGym-v0', num_envs=4096, n_beams=1081, map_size=500, pedestrian_model='sfm',
                       num_humans=20, device='cuda:0', seed=0)
env.reset()
t = env.sim.t
print("initial: planned peds %.3f" % (t["ped_n_waypoints"] > 1).float().mean().item())
act = torch.zeros((4096, 2), dtype=torch.float64, device='cuda:0'); act[:, 0] = 0.5
for s in range(260):
    env.step(act)
    ws = env.sim.t["replan_ws_1024"]
    cnt = ws[:4].view(torch.int32)[0].item()
    if s % 20 == 19:
        lst = ws[256:256 + 4 * cnt].view(torch.int32).long()
        e, i = lst // 20, lst % 20
        pp = t["ped_pose"][e, i, :2]
        Hc = 100
        ci = (pp / 0.25).long().clamp(0, Hc - 1)
        blocked = t["costmap"][e, ci[:, 1], ci[:, 0]]
        nw = t["ped_n_waypoints"][e, i]
        nw0 = nw.clone()
        env.step(act)
        nw1 = t["ped_n_waypoints"][e, i]
        print(s, "due", cnt, "served next step", int((nw1 > 1).sum()),  "start blocked", int(blocked.sum()), "still single", int((nw == 1).sum()), "v_pref<0.05", int((t["ped_v_pref"][e, i] < 0.05).sum()))
