"""CrowdSim-v0 `step()` for E envs at once (nav_gym/src/crowd_sim/envs/crowd_sim.py:724-997), composed from the
device entry points of include/navsim.h:

    pedestrians' actions      navsim_crowd_orca        ORCA.predict per pedestrian (policy/orca.py:85-135)
    collision / goal / reward navsim_crowd_check       crowd_sim.py:808-949
    agent update              navsim_crowd_agent_step  Agent.step (utils/agent.py:108-141), crowd_sim.py:951-958
    robot's local map         navsim_crowd_angular_map / navsim_crowd_local_map   crowd_sim.py:999-1186

Host side is torch plumbing only (building each pedestrian's query: itself first, then the agents it sees).  The
reference's learning policies (SARL / CADRL / SDOADRL) are out of scope: the robot's action comes from the caller.
Pedestrians see everything (the default field of view of orca.py:67-68 is the full circle).
"""
import numpy as np

from . import sim


class CrowdSimStepper(object):
    """State: humans [E,H,9] and robot [E,9] = px, py, vx, vy, radius, v_pref, gx, gy, theta (float64 CUDA tensors);
    obstacle polygons verts [E,O,4,2] (counter-clockwise, crowd_sim.py:250-258) with n_obst [E]; free_map [E,G,G]
    uint8 (1 = free, indexed [x][y] like CrowdSim.map); global_time [E]."""

    def __init__(self, humans, robot, verts, n_obst, free_map, params, orca_params=None, map_params=None,
                 safety_space=0.0, robot_visible=True, use_grid_map=False):
        import torch
        self.torch = torch
        self.h = humans.to(torch.float64).contiguous().clone()
        self.r = robot.to(torch.float64).contiguous().clone()
        self.verts = verts.to(torch.float64).contiguous()
        self.n_obst = n_obst.to(torch.int32).contiguous()
        self.free_map = free_map.to(torch.uint8).contiguous()
        self.params = dict(params)                      # navsim_crowd_params (time_step, penalties, map size, time limit)
        self.orca = dict(orca_params or dict(neighbor_dist=10, time_horizon=5, time_horizon_obst=5, max_neighbors=10))
        self.orca["time_step"] = self.params["time_step"]
        self.map_params = map_params
        self.safety_space = float(safety_space)
        self.robot_visible = bool(robot_visible)
        self.use_grid_map = bool(use_grid_map)
        E, H = self.h.shape[0], self.h.shape[1]
        self.global_time = torch.zeros(E, dtype=torch.float64, device=self.h.device)
        # query (e, h) lists pedestrian h first, then the other pedestrians in index order, then the robot
        idx = torch.arange(H, device=self.h.device)
        others = torch.stack([torch.cat([idx[:k], idx[k + 1:]]) for k in range(H)]) if H > 1 else idx.new_zeros((H, 0))
        self.order = torch.cat([idx[:, None], others], dim=1)                       # [H, H]

    def human_queries(self):
        """agents [E*H, A, 6] and pref_vel [E*H, 2] exactly as ORCA.predict hands them to rvo2 (orca.py:101-124)."""
        torch = self.torch
        E, H = self.h.shape[0], self.h.shape[1]
        hs = self.h[:, self.order]                                                  # [E, H(query), H(agent), 9]
        ag = torch.stack([hs[..., 0], hs[..., 1], hs[..., 2], hs[..., 3],
                          hs[..., 4] + 0.01 + self.safety_space,                    # orca.py:103, 111
                          hs[:, :, :1, 5].expand(E, H, H)], dim=-1)                 # every agent gets agent 0's v_pref
        if self.robot_visible:
            rb = torch.stack([self.r[:, 0], self.r[:, 1], self.r[:, 2], self.r[:, 3],
                              self.r[:, 4] + 0.01 + self.safety_space, self.r[:, 5]], dim=-1)
            rb = rb[:, None, None, :].expand(E, H, 1, 6).clone()
            rb[..., 5] = ag[:, :, :1, 5]
            ag = torch.cat([ag, rb], dim=2)
        vel = self.h[..., 6:8] - self.h[..., 0:2]                                   # orca.py:116-120
        speed = torch.sqrt(vel[..., :1] * vel[..., :1] + vel[..., 1:] * vel[..., 1:])     # separate ops: no fused multiply-add
        pref = torch.where(speed > 1, vel / speed, vel)
        return ag.reshape(E * H, ag.shape[2], 6).contiguous(), pref.reshape(E * H, 2).contiguous()

    def step(self, action, compute_local_map=True):
        """action [E,2] = ActionRot (v, r) of the robot.  -> dict(reward, done, info, min_dist, local_map)."""
        torch = self.torch
        E, H = self.h.shape[0], self.h.shape[1]
        dt = float(self.params["time_step"])
        action = action.to(device=self.h.device, dtype=torch.float64).reshape(E, 2)
        ag, pref = self.human_queries()
        obst_set = torch.arange(E, device=self.h.device, dtype=torch.int32).repeat_interleave(H)
        _, h_act = sim.crowd_orca(self.orca, ag, pref, self.verts, None, self.n_obst, obst_set,
                                  self.h[..., 8].reshape(-1))
        rpose = torch.stack([self.r[:, 0], self.r[:, 1], self.r[:, 8]], dim=1)
        npose, nvel = sim.crowd_agent_step(rpose, action, dt)                       # compute_position / compute_velocity
        robot10 = torch.stack([self.r[:, 0], self.r[:, 1], npose[:, 0], npose[:, 1], nvel[:, 0], nvel[:, 1],
                               self.r[:, 6], self.r[:, 7], self.r[:, 4], action[:, 1]], dim=1)
        agents5 = self.h[..., 0:5].contiguous()
        reward, done, info, min_dist = sim.crowd_check(self.params, self.free_map, robot10, agents5, self.global_time)
        # crowd_sim.py:951-958: update all agents
        self.r[:, 0:2] = npose[:, 0:2]; self.r[:, 2:4] = nvel; self.r[:, 8] = npose[:, 2]
        hpose = torch.stack([self.h[..., 0], self.h[..., 1], self.h[..., 8]], dim=-1).reshape(E * H, 3)
        hp, hv = sim.crowd_agent_step(hpose, h_act, dt)
        self.h[..., 0:2] = hp[:, 0:2].reshape(E, H, 2); self.h[..., 2:4] = hv.reshape(E, H, 2)
        self.h[..., 8] = hp[:, 2].reshape(E, H)
        self.global_time += dt
        local_map = None
        if compute_local_map and self.map_params is not None:
            rb4 = torch.stack([self.r[:, 0], self.r[:, 1], self.r[:, 8], self.r[:, 4]], dim=1)
            local_map = (sim.crowd_local_map(self.map_params, self.free_map, rb4) if self.use_grid_map
                         else sim.crowd_angular_map(self.map_params, rb4, self.verts, self.n_obst))
        return dict(reward=reward, done=done, info=info, min_dist=min_dist, local_map=local_map,
                    human_actions=h_act.reshape(E, H, 2))
