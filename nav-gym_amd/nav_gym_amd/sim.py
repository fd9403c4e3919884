"""Device-resident state of E arenas and the calls into the C ABI (include/navsim.h).

torch is plumbing here: it owns the HBM buffers and the HIP stream; every computation of the hot
path happens in libnavsim_hip.so.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .lib import check, load, require_gpu

_TORCH_DTYPE = None


def _dtype(name):
    global _TORCH_DTYPE
    import torch
    if _TORCH_DTYPE is None:
        _TORCH_DTYPE = {"float32": torch.float32, "float64": torch.float64, "int32": torch.int32,
                        "int64": torch.int64, "uint8": torch.uint8}
    return _TORCH_DTYPE[name]


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- thin functional wrappers over the mirror primitives (used by tests and by NavSim) ----------
def build_rect_index(table, H, W):
    """navsim_build_rect_index: the index form of a record table (int32 CUDA [E, T, 4] from build_rects) -> (uint8 [E, R] rows,
    int32 [E] distinct rectangles per map).  What the fused step stages in LDS (navsim_state.rect_index)."""
    torch = require_gpu()
    L = load()
    E = table.shape[0]
    rows = torch.empty((E, L.navsim_rect_index_bytes(1, H, W)), dtype=torch.uint8, device=table.device)
    n_rects = torch.zeros(E, dtype=torch.int32, device=table.device)
    check(L.navsim_build_rect_index(_ptr(table), E, H, W, _ptr(rows), _ptr(n_rects), _stream()), "navsim_build_rect_index")
    return rows, n_rects


def maps_closed(occ):
    """navsim_maps_closed: occ uint8 CUDA [E,H,W] -> int32 [E], 1 where the map's outer ring of 3 cells is fully occupied.
    cfg.closed_maps may be set when every map of a world is closed (the LDS form of the march has no bounds test)."""
    torch = require_gpu()
    occ = occ.to(torch.uint8).contiguous()
    E, H, W = occ.shape
    closed = torch.zeros(E, dtype=torch.int32, device=occ.device)
    check(load().navsim_maps_closed(_ptr(occ), E, H, W, _ptr(closed), _stream()), "navsim_maps_closed")
    return closed


def build_dt(occ):
    """occ: uint8 CUDA tensor [E,H,W] (nonzero = occupied) -> float32 distance field [E,H,W]."""
    torch = require_gpu()
    L = load()
    occ = occ.contiguous()
    E, H, W = occ.shape
    field = torch.empty((E, H, W), dtype=torch.float32, device=occ.device)
    per_map = L.navsim_build_dt_workspace_bytes(1, H, W)
    chunk = max(1, min(E, (512 << 20) // max(per_map, 1)))       # <= 512 MiB of scratch
    ws = torch.empty(per_map * chunk, dtype=torch.uint8, device=occ.device)
    check(L.navsim_build_dt(_ptr(occ), E, H, W, _ptr(field), _ptr(ws), ws.numel(), _stream()), "navsim_build_dt")
    return field


def build_field(occ, fmt=abi.FIELD_U16T, keep_overflow=None):
    """occ: uint8 CUDA tensor [E,H,W] -> (field, overflow, n_saturated).

    fmt FIELD_F32: field is the float32 [E,H,W] distance field, overflow None.
    fmt FIELD_U16T: field is the packed uint16 tile blob the fused step streams; the float32 plane is
    computed alongside (the reset path samples spawn cells from it) and returned as `overflow`;
    it is only REQUIRED by the step when n_saturated > 0."""
    torch = require_gpu()
    L = load()
    occ = occ.contiguous()
    E, H, W = occ.shape
    if fmt == abi.FIELD_F32:
        return build_dt(occ), None, 0
    nbytes = L.navsim_field_bytes(E, H, W, fmt)
    field = torch.empty(nbytes // 2, dtype=torch.int16, device=occ.device)     # opaque blob
    overflow = torch.empty((E, H, W), dtype=torch.float32, device=occ.device)
    nsat = torch.zeros(1, dtype=torch.int32, device=occ.device)
    per_map = L.navsim_build_dt_workspace_bytes(1, H, W)
    chunk = max(1, min(E, (512 << 20) // max(per_map, 1)))
    ws = torch.empty(per_map * chunk, dtype=torch.uint8, device=occ.device)
    check(L.navsim_build_field(_ptr(occ), E, H, W, fmt, _ptr(field), _ptr(overflow), _ptr(nsat), _ptr(ws),
                               ws.numel(), _stream()), "navsim_build_field")
    return field, overflow, int(nsat.item())          # reset path: a sync is fine here


def build_rects(occ, field, fmt, overflow=None):
    """navsim_build_rects: occ uint8 CUDA [E,H,W] + the distance field built from it (build_field's `field`, and
    its float32 plane when cells are saturated) -> int32 [E, T, 4] two-rectangle records of the 8x8 tiles."""
    torch = require_gpu()
    L = load()
    occ = occ.contiguous()
    E, H, W = occ.shape
    T = ((H + 7) // 8) * ((W + 7) // 8)
    table = torch.empty((E, T, 4), dtype=torch.int32, device=occ.device)
    per = L.navsim_build_rects_workspace_bytes(1, H, W) + 256
    chunk = max(1, min(E, (1 << 30) // per))                      # <= 1 GiB of scratch
    ws = torch.empty(per * chunk, dtype=torch.uint8, device=occ.device)
    check(L.navsim_build_rects(_ptr(occ), E, H, W, _ptr(field), int(fmt), _ptr(overflow), _ptr(table), _ptr(ws),
                               ws.numel(), _stream()), "navsim_build_rects")
    return table


def cast_static(field, queries, max_range, march_rule=abi.MARCH_F32):
    torch = require_gpu()
    E, H, W = field.shape
    q = queries.contiguous()
    out = torch.empty(q.shape[:2], dtype=torch.float32, device=field.device)
    check(load().navsim_cast_static(_ptr(field), E, H, W, _ptr(q), q.shape[1], float(max_range), int(march_rule),
                                    _ptr(out), _stream()), "navsim_cast_static")
    return out


def render_polys(ranges, angles, verts, n_verts, origin):
    require_gpu()
    E, B = ranges.shape
    check(load().navsim_render_polys(_ptr(ranges), _ptr(angles), E, B, _ptr(verts), _ptr(n_verts),
                                     verts.shape[1], _ptr(origin), _stream()), "navsim_render_polys")
    return ranges


def render_legs(ranges, angles, agents, n_agents, origin):
    require_gpu()
    E, B = ranges.shape
    check(load().navsim_render_legs(_ptr(ranges), _ptr(angles), E, B, _ptr(agents), _ptr(n_agents),
                                    agents.shape[1], _ptr(origin), _stream()), "navsim_render_legs")
    return ranges


def integrate(pose, cmd, time_step, axle_offset, vel_out=None):
    require_gpu()
    check(load().navsim_integrate(_ptr(pose), _ptr(cmd), _ptr(vel_out), pose.shape[0], float(time_step),
                                  float(axle_offset), _stream()), "navsim_integrate")
    return pose


def scan_threshold(cfg, footprint):
    torch = require_gpu()
    fp = footprint.contiguous()
    out = torch.empty(cfg.n_beams, dtype=torch.float32, device=fp.device)
    check(load().navsim_scan_threshold(C.byref(cfg), _ptr(fp), fp.shape[0], _ptr(out), _stream()),
          "navsim_scan_threshold")
    return out


def reward_done(cfg, obs, goals, thr, dthr):
    torch = require_gpu()
    obs = obs.contiguous()
    is64 = obs.dtype == torch.float64
    goals = goals.to(obs.dtype).contiguous()
    n = obs.shape[0]
    dev = obs.device
    out = dict(reward=torch.empty(n, dtype=torch.float64, device=dev),
               done=torch.empty(n, dtype=torch.uint8, device=dev),
               is_success=torch.empty(n, dtype=torch.float32, device=dev),
               is_crash=torch.empty(n, dtype=torch.float32, device=dev),
               distance=torch.empty(n, dtype=torch.float64, device=dev))
    check(load().navsim_reward_done(C.byref(cfg), _ptr(obs), _ptr(goals), int(is64), n, _ptr(thr), _ptr(dthr),
                                    _ptr(out["reward"]), _ptr(out["done"]), _ptr(out["is_success"]),
                                    _ptr(out["is_crash"]), _ptr(out["distance"]), _stream()), "navsim_reward_done")
    return out


def costmap(occ):
    """occ uint8 CUDA [n,H,W] -> uint8 [n,H/5,W/5] (env.py:312-332)."""
    torch = require_gpu()
    occ = occ.contiguous()
    n, H, W = occ.shape
    out = torch.zeros((n, H // 5, W // 5), dtype=torch.uint8, device=occ.device)
    check(load().navsim_costmap(_ptr(occ), n, H, W, _ptr(out), _stream()), "navsim_costmap")
    return out


def plan(cost, start, goal, interval, max_wp=8, res_c=0.25, origin=(0.0, 0.0), map_index=None):
    """Shortest 4-connected paths + waypoints for n queries (env.py:343-354, 1261-1277)."""
    torch = require_gpu()
    L = load()
    cost = cost.contiguous()
    _, Hc, Wc = cost.shape
    start = start.to(torch.float64).contiguous().reshape(-1, 2)
    n = start.shape[0]
    goal = goal.to(torch.float64).contiguous().reshape(n, 2)
    dev = cost.device
    mi = None if map_index is None else map_index.to(device=dev, dtype=torch.int32).contiguous()
    wp = torch.zeros((n, max_wp, 2), dtype=torch.float64, device=dev)
    n_wp = torch.zeros(n, dtype=torch.int32, device=dev)
    cells = torch.zeros(n, dtype=torch.int32, device=dev)
    plen = torch.zeros(n, dtype=torch.float64, device=dev)
    check(L.navsim_plan(_ptr(cost), _ptr(mi), n, Hc, Wc, float(res_c), float(origin[0]), float(origin[1]), _ptr(start),
                        _ptr(goal), float(interval), max_wp, _ptr(wp), _ptr(n_wp), _ptr(cells), _ptr(plen), _stream()),
          "navsim_plan")
    return wp, n_wp, cells, plen


def crowd_check(params, free_map, robot, agents, global_time, n_agents=None):
    """navsim_crowd_check: CrowdSim.step's collision / goal / reward block (crowd_sim.py:808-949) for E envs.
    params: dict with the fields of navsim_crowd_params; tensors on the GPU.  -> reward, done, info, min_dist."""
    torch = require_gpu()
    L = load()
    p = abi.NavsimCrowdParams(**{k: float(v) for k, v in params.items()})
    free_map = free_map.to(torch.uint8).contiguous()
    E, G = free_map.shape[0], free_map.shape[1]
    dev = free_map.device
    robot = robot.to(torch.float64).contiguous().reshape(E, 10)
    agents = agents.to(torch.float64).contiguous().reshape(E, -1, 5)
    A = agents.shape[1]
    gt = global_time.to(torch.float64).contiguous().reshape(E)
    na = None if n_agents is None else n_agents.to(device=dev, dtype=torch.int32).contiguous()
    reward = torch.zeros(E, dtype=torch.float64, device=dev); done = torch.zeros(E, dtype=torch.uint8, device=dev)
    info = torch.zeros(E, dtype=torch.int32, device=dev); md = torch.zeros(E, dtype=torch.float64, device=dev)
    check(L.navsim_crowd_check(C.byref(p), E, A, G, _ptr(free_map), _ptr(robot), _ptr(agents), _ptr(na), _ptr(gt),
                               _ptr(reward), _ptr(done), _ptr(info), _ptr(md), _stream()), "navsim_crowd_check")
    return reward, done, info, md


def _crowd_map_params(params):
    p = abi.NavsimCrowdMapParams()
    for k, v in params.items():
        setattr(p, k, int(v) if k in ("angular_dim", "normalize") else float(v))
    return p


def crowd_angular_map(params, robot, verts, n_obst=None):
    """navsim_crowd_angular_map: CrowdSim.get_local_map_angular (crowd_sim.py:1055-1102) for E envs.
    robot [E,4] px, py, theta, radius; verts [E,O,V,2]; -> float64 [E, angular_dim]."""
    torch = require_gpu()
    p = _crowd_map_params(params)
    robot = robot.to(torch.float64).contiguous()
    E = robot.shape[0]
    verts = verts.to(torch.float64).contiguous()
    O, V = (verts.shape[1], verts.shape[2]) if verts.numel() else (0, 4)
    no = None if n_obst is None else n_obst.to(device=robot.device, dtype=torch.int32).contiguous()
    out = torch.empty((E, p.angular_dim), dtype=torch.float64, device=robot.device)
    check(load().navsim_crowd_angular_map(C.byref(p), E, O, V, _ptr(robot), _ptr(verts) if O else None, _ptr(no), _ptr(out),
                                          _stream()), "navsim_crowd_angular_map")
    return out


def crowd_local_map(params, free_map, robot, rotate=True):
    """navsim_crowd_local_map: CrowdSim.get_local_map (crowd_sim.py:1104-1186) for E envs -> uint8 [E,S,S]."""
    torch = require_gpu()
    p = _crowd_map_params(params)
    free_map = free_map.to(torch.uint8).contiguous()
    E, G = free_map.shape[0], free_map.shape[1]
    robot = robot.to(torch.float64).contiguous()
    S = int(round(p.submap_size_m / p.map_resolution))
    out = torch.empty((E, S, S), dtype=torch.uint8, device=free_map.device)
    check(load().navsim_crowd_local_map(C.byref(p), E, G, _ptr(free_map), _ptr(robot), int(bool(rotate)), _ptr(out),
                                        _stream()), "navsim_crowd_local_map")
    return out


def crowd_orca(params, agents, pref_vel, verts=None, n_agents=None, n_obst=None, obst_set=None, theta=None):
    """navsim_crowd_orca: ORCA.predict (orca.py:85-135) for Q pedestrians: agents [Q,A,6] (agent 0 = the pedestrian),
    pref_vel [Q,2], verts [S,O,V,2] CCW polygons -> (new velocity [Q,2], ActionRot (v, r) [Q,2]), float64 tensors."""
    torch = require_gpu()
    p = abi.NavsimOrcaParams(**{k: (int(v) if k == "max_neighbors" else float(v)) for k, v in params.items()})
    agents = agents.to(torch.float64).contiguous()
    Q, A = agents.shape[0], agents.shape[1]
    dev = agents.device
    pref_vel = pref_vel.to(device=dev, dtype=torch.float64).contiguous().reshape(Q, 2)
    if verts is None or verts.numel() == 0:
        verts, O, V = None, 0, 4
    else:
        verts = verts.to(device=dev, dtype=torch.float64).contiguous()
        O, V = verts.shape[1], verts.shape[2]
    i32 = lambda a: None if a is None else a.to(device=dev, dtype=torch.int32).contiguous()
    na, no, os_ = i32(n_agents), i32(n_obst), i32(obst_set)
    th = None if theta is None else theta.to(device=dev, dtype=torch.float64).contiguous()
    vel = torch.empty((Q, 2), dtype=torch.float64, device=dev)
    act = torch.empty((Q, 2), dtype=torch.float64, device=dev)
    check(load().navsim_crowd_orca(C.byref(p), Q, A, _ptr(agents), _ptr(na), _ptr(pref_vel), O, V, _ptr(verts), _ptr(no),
                                   _ptr(os_), _ptr(th), _ptr(vel), _ptr(act), _stream()), "navsim_crowd_orca")
    return vel, act


def crowd_agent_step(pose, action, time_step):
    """navsim_crowd_agent_step: Agent.step with an ActionRot (agent.py:108-141) -> (pose [n,3], vel [n,2])."""
    torch = require_gpu()
    pose = pose.to(torch.float64).contiguous().clone()
    action = action.to(device=pose.device, dtype=torch.float64).contiguous()
    vel = torch.empty((pose.shape[0], 2), dtype=torch.float64, device=pose.device)
    check(load().navsim_crowd_agent_step(_ptr(pose), _ptr(action), _ptr(vel), pose.shape[0], float(time_step), _stream()),
          "navsim_crowd_agent_step")
    return pose, vel


def debug_kernarg_layout():
    """navsim_debug_kernarg_layout: the step kernels' views of their arguments (references into the kernarg segment) against the
    by-value copies, in one probe launch.  Raises when they differ."""
    require_gpu()
    check(load().navsim_debug_kernarg_layout(_stream()), "navsim_debug_kernarg_layout")


def debug_xy_to_ij(cfg, xy, as_f32):
    """Device batch_xy_to_ij (env.py:1228-1253): xy float64 CUDA [n,2] -> int32 [n,2] (i, j)."""
    torch = require_gpu()
    xy = xy.to(torch.float64).contiguous()
    out = torch.empty((xy.shape[0], 2), dtype=torch.int32, device=xy.device)
    check(load().navsim_debug_xy_to_ij(C.byref(cfg), _ptr(xy), int(bool(as_f32)), _ptr(out), xy.shape[0], _stream()),
          "navsim_debug_xy_to_ij")
    return out


def debug_spawn_decisions(cfg, cost, kind, start, goal, robot=None):
    """navsim_debug_spawn_decisions: the spawn loops' acceptance rules on supplied candidates (tests only).
    cost uint8 CUDA [Hc,Wc]; kind int32 [n]; start / goal / robot float64 [n,2] -> int32 codes [n]."""
    torch = require_gpu()
    dev = cost.device
    cost = cost.to(torch.uint8).contiguous()
    Hc, Wc = cost.shape
    kind = torch.as_tensor(kind).to(device=dev, dtype=torch.int32).contiguous()
    n = kind.shape[0]
    f64 = lambda a: None if a is None else torch.as_tensor(a).to(device=dev, dtype=torch.float64).contiguous().reshape(n, 2)
    start, goal, robot = f64(start), f64(goal), f64(robot)
    ws = torch.zeros((n, cfg.max_waypoints, 2), dtype=torch.float64, device=dev)
    code = torch.full((n,), -1, dtype=torch.int32, device=dev)
    L = load()
    L.navsim_debug_spawn_decisions.argtypes = [C.POINTER(abi.NavsimConfig), C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
    check(L.navsim_debug_spawn_decisions(C.byref(cfg), _ptr(cost), Hc, Wc, n, _ptr(kind), _ptr(start), _ptr(goal),
                                         _ptr(robot), _ptr(ws), _ptr(code), _stream()), "navsim_debug_spawn_decisions")
    return code


def debug_math(fn, x, x2=None):
    torch = require_gpu()
    out = torch.empty_like(x)
    check(load().navsim_debug_math(fn, _ptr(x), _ptr(x2), _ptr(out), x.numel(), _stream()), "navsim_debug_math")
    return out


def concurrent_stream(device, priority=0, tries=8, beside=None):
    """A stream whose kernels really run BESIDE those of the current stream.  HIP spreads a process's streams over a handful of
    hardware queues (four by default) in creation order; two streams that land on the same queue take turns, whatever the
    events between them say -- a process that has created a few streams before (torch hands its pool out round-robin) got a
    staging stream on the main stream's queue one time in four, and the pipelined reset path then ran at 1.6 M env-steps/s
    instead of 2.4 M (profiles/_diag/bench_bisect.py).  So: try a few candidates, time two ~50 us spin kernels queued on the two
    streams at once, and keep the first candidate on which they overlap (else the best seen)."""
    import torch
    if not hasattr(torch.cuda, "_sleep"):                       # (no spin kernel to time with: any stream)
        return torch.cuda.Stream(device=device, priority=priority)
    others = list(beside) if beside else [torch.cuda.current_stream(device)]     # beside: the streams it must not share a queue with
    best, best_ratio = None, None
    spin = 100_000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(tries):
        cand = torch.cuda.Stream(device=device, priority=priority)
        with torch.cuda.stream(cand):                           # one spin alone: what "beside" is measured against (twice: warm-up)
            for rep in range(2):
                torch.cuda.synchronize(device)
                e0.record(cand); torch.cuda._sleep(spin); e1.record(cand)
                torch.cuda.synchronize(device)
        alone = e0.elapsed_time(e1)
        worst = 0.0
        for other in others:
            for rep in range(2):
                torch.cuda.synchronize(device)
                e0.record(other)
                cand.wait_event(e0)
                with torch.cuda.stream(cand):
                    torch.cuda._sleep(spin)
                with torch.cuda.stream(other):
                    torch.cuda._sleep(spin)
                other.wait_stream(cand)
                e1.record(other)
                torch.cuda.synchronize(device)
            worst = max(worst, e0.elapsed_time(e1) / max(alone, 1e-6))
        if best_ratio is None or worst < best_ratio:
            best, best_ratio = cand, worst
        if worst < 1.5:
            return cand
    return best


class NavSim(object):
    """E arenas resident on one GPU.  `arrays` maps navsim_state field names to numpy arrays or
    torch tensors (host or device); missing optional fields stay NULL."""

    def __init__(self, cfg, arrays, device="cuda:0", launch_order=None, final_obs=False):
        """launch_order: None = longest-first launch order when a launch runs several generations of
        workgroups, True / False = force it on / off (a scheduling hint: results never depend on it).
        final_obs: keep the terminal observation of arenas that restart inside the step that ends their episode
        (cfg.auto_reset = AUTORESET_SAME_STEP; include/navsim.h navsim_step_io.final_obs) -> self.final.
        cfg.auto_reset = AUTORESET_NEXT_STEP: step() hands the done flags of the previous step to the launch as its reset mask
        (the arenas that finished are reset by the next step; regen() then serves those arenas)."""
        torch = require_gpu()
        self.lib = load()
        self.cfg = cfg.copy()
        self.device = torch.device(device)
        self.t = {}
        self.st = abi.NavsimState()
        for name, (dtype, shape) in abi.STATE_LAYOUT.items():
            a = arrays.get(name)
            if a is None:
                setattr(self.st, name, None)
                continue
            if name == "field" and self.cfg.field_format != abi.FIELD_F32:
                t = a.to(self.device).contiguous()              # packed blob, kept as is
                want = self.lib.navsim_field_bytes(self.cfg.n_envs, self.cfg.map_h, self.cfg.map_w,
                                                   self.cfg.field_format)
                if t.numel() * t.element_size() != want:
                    raise ValueError("field blob has %d bytes, expected %d" % (t.numel() * t.element_size(), want))
                self.t[name] = t
                setattr(self.st, name, t.data_ptr())
                continue
            if isinstance(a, np.ndarray):
                a = torch.from_numpy(np.ascontiguousarray(a))
            t = a.to(device=self.device, dtype=_dtype(dtype)).contiguous()
            if t.data_ptr() == a.data_ptr():
                t = t.clone()
            want = abi.resolve_shape(shape, self.cfg)
            if tuple(t.shape) != want:
                raise ValueError("%s: shape %s, expected %s" % (name, tuple(t.shape), want))
            self.t[name] = t
            setattr(self.st, name, t.data_ptr())
        if self.cfg.closed_maps and "rect_index" in self.t:
            # cfg.closed_maps is an assertion about THIS world that the LDS form of the march relies on without a bounds
            # test (include/navsim.h): verify it once, against the distance fields as they stand (round-4 advisor)
            n_open = torch.zeros(1, dtype=torch.int32, device=self.device)
            check(self.lib.navsim_world_closed(C.byref(self.cfg), C.byref(self.st), _ptr(n_open), _stream()), "navsim_world_closed")
            if int(n_open.item()):
                raise ValueError("cfg.closed_maps = 1, but %d arenas have a map without a closed ring of 3 occupied cells "
                                 "(navsim_maps_closed / navsim_world_closed): clear closed_maps or close the maps" % int(n_open.item()))
        self.due = None
        if "ped_waypoints" in self.t:
            # ABI 5: the index of every pedestrian's current waypoint (a route starts at its first one), and the two buffers
            # of "waits for navsim_replan" flags the step writes / navsim_step_part reads.  The flags of the LATEST step
            # live in due[self.cur] -- they flip with the observation buffers (_flip)
            if "ped_wp_head" not in self.t:
                self.t["ped_wp_head"] = torch.zeros((self.cfg.n_envs, self.cfg.max_peds), dtype=torch.int32, device=self.device)
                self.st.ped_wp_head = self.t["ped_wp_head"].data_ptr()
            self.due = [torch.zeros(self.cfg.n_envs, dtype=torch.int64, device=self.device) for _ in range(2)]
            self.st.ped_due = self.due[0].data_ptr()
            self.st.ped_due_prev = self.due[1].data_ptr()
        if "done_steps" not in self.t and self.cfg.auto_reset:  # length of every arena's last episode (cfg.regen_min_steps)
            self.t["done_steps"] = torch.zeros(self.cfg.n_envs, dtype=torch.int32, device=self.device)
            self.st.done_steps = self.t["done_steps"].data_ptr()
        if "counters" not in self.t:                            # what the library's caps left unserved (include/navsim.h)
            self.t["counters"] = torch.zeros(abi.N_COUNTERS, dtype=torch.int64, device=self.device)
            self.st.counters = self.t["counters"].data_ptr()
        if "beam_table" not in self.t:                          # accelerator table, built on device
            tab = torch.empty((self.cfg.n_beams, 2), dtype=torch.float64, device=self.device)
            check(self.lib.navsim_beam_table(C.byref(self.cfg), _ptr(tab), _stream()), "navsim_beam_table")
            self.t["beam_table"] = tab
            self.st.beam_table = tab.data_ptr()
        # longest-first launch order (navsim_launch_order): the step measures every arena's workgroup, every
        # few steps the arenas are re-sorted so that the slow ones start first (lpt_period = 0 disables).  8 measured
        # against 4 / 16 / 32 / 64 in profiles/r03_lpt/: the costs drift slowly, the sort is 13 us
        self.lpt_period = 8
        self._steps_launched = 0
        # only when a launch runs several generations of workgroups: with one generation everything starts
        # at once and the order is irrelevant (threads per arena as dispatch_step picks them)
        n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count or 256
        block = self.cfg.step_block or (                    # navsim_kernels.hip pick_step_block
            64 if self.cfg.n_beams <= 64 else (256 if self.cfg.n_envs >= 12 * n_cu or self.cfg.n_beams <= 256 else
                                               (512 if self.cfg.n_envs > 2 * n_cu or self.cfg.n_beams <= 512 else 1024)))
        generations = self.cfg.n_envs * (block // 64) / float(32 * n_cu)
        if (launch_order is True or (launch_order is None and generations > 1.25)) and "arena_cost" not in self.t:
            self.t["arena_cost"] = torch.zeros(self.cfg.n_envs, dtype=torch.int32, device=self.device)
            self.t["launch_order"] = torch.arange(self.cfg.n_envs, dtype=torch.int32, device=self.device)
            self.st.arena_cost = self.t["arena_cost"].data_ptr()
            self.st.launch_order = self.t["launch_order"].data_ptr()
        E = self.cfg.n_envs
        D = self.cfg.n_scan_stack * self.cfg.n_beams + abi.OBS_TAIL
        self.obs_buf = [torch.zeros((E, D), dtype=torch.float32, device=self.device) for _ in range(2)]
        self.cur = 0
        # what step() returns beside the observation (reward, done, info, the two goal arrays): two sets, flipped with the
        # observation buffers, so that the arrays a step returned stay intact through the NEXT step (round-4 advisor: the
        # env handed out views of single buffers that the next step overwrote -- `obs = next_obs`, `dones.append(done)`)
        self.out_buf = [{k: torch.zeros(abi.resolve_shape(s, self.cfg), dtype=_dtype(d), device=self.device)
                         for k, (d, s) in abi.IO_LAYOUT.items() if k not in ("obs", "obs_prev", "action")} for _ in range(2)]
        self.action = torch.zeros((E, 2), dtype=torch.float64, device=self.device)
        self.io = abi.NavsimStepIO()
        self.io.action = self.action.data_ptr()
        for k, v in self.out_buf[0].items():
            setattr(self.io, k, v.data_ptr())
        # ABI 6: terminal observations (same-step restarts), kept in pairs like everything a step returns
        self.final_buf = None
        if final_obs:
            self.final_buf = [{k: torch.zeros(abi.resolve_shape(s, self.cfg), dtype=_dtype(d), device=self.device)
                               for k, (d, s) in abi.FINAL_LAYOUT.items()} for _ in range(2)]
        self.next_step = self.cfg.auto_reset == abi.AUTORESET_NEXT_STEP

    @property
    def obs(self):
        return self.obs_buf[self.cur]

    @property
    def out(self):
        """reward / done / is_success / is_crash / distance / achieved_goal / desired_goal of the latest step."""
        return self.out_buf[self.cur]

    @property
    def final(self):
        """final_obs / final_goals of the latest step: rows of the arenas whose done flag that step set (others: stale)."""
        return None if self.final_buf is None else self.final_buf[self.cur]

    @property
    def reset_flags(self):
        """AUTORESET_NEXT_STEP: the arenas the latest step RESET (= the done flags of the step before it)."""
        return self.out_buf[1 - self.cur]["done"]

    def _flip(self):
        self.io.obs_prev = self.obs_buf[self.cur].data_ptr()
        self.io.obs = self.obs_buf[1 - self.cur].data_ptr()
        # NEXT_STEP: whoever finished in the latest step (its done flags, out_buf[cur]) is reset by the launch that follows
        self.io.reset_mask = self.out_buf[self.cur]["done"].data_ptr() if self.next_step else None
        if self.final_buf is not None:
            for k, v in self.final_buf[1 - self.cur].items():
                setattr(self.io, k, v.data_ptr())
        for k, v in self.out_buf[1 - self.cur].items():
            setattr(self.io, k, v.data_ptr())
        if self.due is not None:                    # the launch reads the latest flags and writes the other buffer
            self.st.ped_due_prev = self.due[self.cur].data_ptr()
            self.st.ped_due = self.due[1 - self.cur].data_ptr()

    def _latest_flags(self):
        """st.ped_due -> the flags the latest step wrote (due[self.cur]): what navsim_replan / navsim_regen read and clear.
        (True after every step by construction; enable_graphs captures both parities from one state.)"""
        if self.due is not None:
            self.st.ped_due = self.due[self.cur].data_ptr()
            self.st.ped_due_prev = self.due[1 - self.cur].data_ptr()

    def reset_obs(self, mask=None):
        """First observation of an episode (reference reset(), env.py:808-831) for masked envs."""
        # a reset-only launch writes the goal arrays of the arenas it resets and nothing else: the other set starts as a copy
        for k, v in self.out_buf[1 - self.cur].items():
            v.copy_(self.out_buf[self.cur][k])
        m = None
        if mask is not None:
            import torch
            m = torch.as_tensor(mask).to(device=self.device, dtype=torch.uint8).contiguous()
        # an arena that is reset is not finished: its done flag (NEXT_STEP: the next launch's reset mask) is cleared
        if m is None:
            self.out_buf[1 - self.cur]["done"].zero_()
        else:
            self.out_buf[1 - self.cur]["done"].mul_(m == 0)
        self._flip()
        check(self.lib.navsim_reset_obs(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), _ptr(m), _stream()),
              "navsim_reset_obs")
        if self.due is not None:                    # a reset-only launch advances nobody: the latest flags stay the latest
            self.due[1 - self.cur].copy_(self.due[self.cur])
        self.cur = 1 - self.cur
        return self.obs

    def _reorder(self):
        # re-sort before steps 2..5 (short runs profit from the very first measured costs), then every lpt_period-th
        self._steps_launched += 1
        n = self._steps_launched
        if "launch_order" in self.t and self.lpt_period > 0 and (2 <= n <= 5 or n % self.lpt_period == 0):
            check(self.lib.navsim_launch_order(_ptr(self.t["arena_cost"]), _ptr(self.t["launch_order"]), self.cfg.n_envs,
                                               _stream()), "navsim_launch_order")

    def step(self, action=None):
        """One fused launch: NavGymEnv.step for all E arenas.  `action` [E,2] (v, omega)."""
        if action is not None:
            import torch
            if (isinstance(action, torch.Tensor) and action.is_cuda and action.dtype == torch.float64 and action.is_contiguous()
                    and action.device == self.device and action.numel() == self.action.numel()):
                self.io.action = action.data_ptr()         # resident actions are read where they are (stream-ordered: no copy)
                self._action_ref = action
            else:
                self.action.copy_(self._as(action, self.action))
                self.io.action = self.action.data_ptr()
        self._reorder()
        self._flip()
        self._launch()
        self.cur = 1 - self.cur
        return self.obs, self.out

    def _launch(self):
        """The step's launch on the current stream: navsim_step, or navsim_step_install behind the staging pass it may rest on."""
        if not getattr(self, "pg_install", False):
            rc = self.lib.navsim_step(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), _stream())
            if rc:
                check(rc, "navsim_step")
            return
        import torch                                            # enable_pregen(pipeline=P, install=True)
        main = torch.cuda.current_stream()
        P, k = self.pg_period, self.pg_k
        if k % P == 0 and k >= 2 * P:
            main.wait_event(self.pg_staged[(k // P - 2) % 3])   # the pass queued two periods ago; the later two may still run
        if getattr(self, "late2", None) is not None:
            return self._launch_next_step(main)
        # pg_replan_cap > 0 (worlds with planned pedestrian routes): navsim_replan of the previous step's flags inside this launch
        # where the search fits the arena's workgroup (navsim_step_install_replan); else the caller re-plans behind the step
        late = None if getattr(self, "lone", False) else self.late
        if getattr(self, "pg_replan_cap", 0) > 0 and self.pg_replan_in_step:
            rc = self.lib.navsim_step_install_replan(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), C.byref(self.stage_st),
                                                     _ptr(self.stage_obs), _ptr(self.mark), _ptr(self.ready), _ptr(late),
                                                     int(self.pg_replan_cap), C.c_void_p(main.cuda_stream))
            if rc == 0:
                return
            if rc != abi.E_UNSUPPORTED:
                check(rc, "navsim_step_install_replan")
            self.pg_replan_in_step = False
        rc = self.lib.navsim_step_install(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), C.byref(self.stage_st),
                                          _ptr(self.stage_obs), _ptr(self.mark), _ptr(self.ready), _ptr(late),
                                          C.c_void_p(main.cuda_stream))
        if rc == abi.E_UNSUPPORTED and late is None and self.late is not None:
            self.lone = False                                   # (fewer than 256 threads per arena: flags + navsim_regen behind the step)
            rc = self.lib.navsim_step_install(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), C.byref(self.stage_st),
                                              _ptr(self.stage_obs), _ptr(self.mark), _ptr(self.ready), _ptr(self.late),
                                              C.c_void_p(main.cuda_stream))
        check(rc, "navsim_step_install")

    def _launch_next_step(self, main):
        """NEXT_STEP with staged worlds: the step's launch (arenas that finished in the previous call install their staged worlds
        at its front) and, on the `urgent` stream at the same time, navsim_regen for the arenas that call flagged as late."""
        import torch
        prev, nxt = self.late2[self.cur], self.late2[1 - self.cur]      # (self.cur: the buffers of the PREVIOUS call)
        cap = int(getattr(self, "pg_replan_cap", 0)) if self.pg_replan_in_step else 0
        rc = self.lib.navsim_step_install_next(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), C.byref(self.stage_st),
                                               _ptr(self.stage_obs), _ptr(self.mark), _ptr(self.ready), _ptr(nxt), _ptr(prev),
                                               cap if cap > 0 else -1, C.c_void_p(main.cuda_stream))
        if rc == abi.E_UNSUPPORTED and cap > 0:                 # the search does not fit the arena's workgroup: re-plan behind the step
            self.pg_replan_in_step = False
            rc = self.lib.navsim_step_install_next(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), C.byref(self.stage_st),
                                                   _ptr(self.stage_obs), _ptr(self.mark), _ptr(self.ready), _ptr(nxt), _ptr(prev),
                                                   -1, C.c_void_p(main.cuda_stream))
        check(rc, "navsim_step_install_next")
        # beside it: the new worlds of the arenas the previous call found unstaged (mostly nobody: launches that find nothing to do)
        urgent = self.urgent
        urgent.wait_event(self.ev_stepped)                      # the previous call (its flags, its state) -- NOT this launch
        C.memmove(C.byref(self.late_cfg), C.byref(self.cfg), C.sizeof(self.cfg))
        self.late_cfg.regen_cap = self.late_cap
        self.late_cfg.defer_reset_scan = 1
        io = abi.NavsimStepIO()
        C.memmove(C.byref(io), C.byref(self.io), C.sizeof(io))  # (self.io: this call's buffers -- _flip() ran)
        io.done = prev.data_ptr()
        io.reset_mask = None
        helper = getattr(self, "_regen_helper", None)
        if helper is not None:                                  # no fork: the helper stream belongs to the staging passes
            self.lib.navsim_regen_helper(C.c_void_p(urgent.cuda_stream))
        check(self.lib.navsim_regen(C.byref(self.late_cfg), C.byref(self.st), C.byref(io), _ptr(self.late_ws), self.late_ws.numel(),
                                    C.c_void_p(urgent.cuda_stream)), "navsim_regen (arenas whose world was not staged, beside the step)")
        if helper is not None:
            self.lib.navsim_regen_helper(C.c_void_p(helper.cuda_stream))
        self.ev_urgent.record(urgent)
        main.wait_event(self.ev_urgent)                         # whatever follows this step on the caller's stream sees both
        self.ev_stepped.record(main)

    # arrays navsim_regen writes: the staged state of enable_pregen() owns a copy of each
    STAGED = ("field", "field_overflow", "rect_table", "rect_index", "costmap", "scan_noise_std", "robot_pose", "robot_goal", "prev_action",
              "prev_pose", "n_hist", "steps", "episode", "n_peds", "ped_pose", "ped_vel", "ped_prev_yaw", "ped_dist",
              "ped_v_pref", "ped_has_legs", "ped_waypoints", "ped_n_waypoints", "ped_wp_head", "ped_goal", "spawn_pose", "spawn_goal")

    MAPS = ("field", "field_overflow", "rect_table", "rect_index", "costmap")      # the per-map arrays (navsim_state.map_slot)

    def enable_pregen(self, scratch_bytes=4 << 30, pipeline=0, install=False, stage_cap=None, map_slots=True, fallback=None,
                      fallback_cap=None, late_beside=False, fallback_poll=None, stage_lanes=None):
        """navsim_regen off the step's critical path (include/navsim.h navsim_regen_swap): the world every arena will
        get at the end of its CURRENT episode -- a function of (seed, global arena, episode number) only -- is generated
        ahead of time into a second, staged state by the ordinary navsim_regen on a side stream; regen() then only
        installs the staged worlds of the finished arenas (one copy kernel on the caller's stream) and queues the next
        staging pass.  Same state, same observations as navsim_regen, bit for bit.  Needs cfg.auto_reset = 1.

        pipeline = P > 0 (round 5): a staging pass every P steps, and the steps wait only for the pass queued two periods
        earlier -- a pass has P .. 2 P steps of wall time to finish beside the step kernels instead of having to fit between
        two steps.  That needs the simulation's own guarantee that an arena does not want its next world sooner:
        cfg.regen_min_steps >= 4 P (shorter episodes restart in place, in navsim_regen and the oracle alike), which makes
        the outcome independent of timing; counters()['regen_late'] stays 0.

        install=True (with pipeline): no swap kernel either -- step() is navsim_step_install, in which a finished arena's own
        workgroup copies its staged world in place of the restart's second scan; regen() only queues the passes.  Every
        arena decides alone, so cfg.regen_cap must be >= n_envs (no cap in index order); `stage_cap` bounds what ONE pass
        stages (default: what 2 P steps finish at one arena in 64 per step) -- arenas beyond it wait for the next pass.
        fallback (with install; default: on when cfg.regen_min_steps < 4 P): no rule is needed -- an arena that finishes before
        its world is staged is regenerated on the spot by the ordinary navsim_regen (regen() launches it after every step, with
        the flags the step wrote: launches that find nothing to do when nobody was late, `fallback_cap` arenas at most -- more late arenas than that in
        ONE step restart on their old map like navsim_regen's own cap, counters()['regen_unserved']).  The
        rollout then equals step + navsim_regen whatever the passes' timing, also with cfg.regen_min_steps = 0: the reference's
        "a new map at every reset()" unchanged.
        fallback_poll (with fallback; round 6): the fallback's navsim_regen is launched only when the step flagged somebody -- regen()
        reads "any flag set?" back from the device (one byte, a host wait for the step's launch) instead of enqueueing the
        call's launches blind.  Default: on where that call is the long chain of worlds with corridor maps or planned starts
        (~18 launches, ~160 us of mostly empty launches behind every ~80 us step of the reference's configuration:
        profiles/r05_refdef/timeline_pipeline8_no_rule_cap16.txt), off where it is three short launches (c5) and under graph
        capture.  Same rollout either way.
        stage_lanes (with install and pipeline; round 6): 2 = the staging passes alternate between two side streams, each with its own
        share of the arenas (navsim_regen_stage_part: groups of four by mark word), workspace and helper stream, so that two
        passes run at the same time -- a pass for corridor maps with planned starts is a chain of ~25 latency-bound launches
        (the planner's four wait for their longest search each) that leaves the chip idle, and the passes' rate is what bounds the
        reference's configuration once the fallback's blind launches are gone.  Default: 2 for those worlds, else 1.
        map_slots (with install): the live and the staged state share the per-map arrays (t['field'], ... then hold 2 E slots)
        and each has a slot table (navsim_state.map_slot); an install exchanges two table entries instead of copying the
        map -- numpy_state() resolves the table, code that indexes t['field'] by arena must go through t['map_slot']."""
        import torch
        if not self.cfg.auto_reset:
            raise ValueError("enable_pregen needs cfg.auto_reset = 1 (the step advances episode[e] when an arena finishes)")
        E = self.cfg.n_envs
        P = int(pipeline)
        if fallback is None:
            fallback = bool(install) and P > 0 and self.cfg.regen_min_steps < 4 * P
        if fallback and not install:
            raise ValueError("enable_pregen(fallback=True) is a feature of install=True")
        if P < 0 or (P > 0 and self.cfg.regen_min_steps < 4 * P and not fallback):
            raise ValueError("enable_pregen(pipeline=%d) needs cfg.regen_min_steps >= %d (it is %d): an arena must not want its "
                             "next world before the pass that stages it was waited for -- or fallback=True (install=True)"
                             % (P, 4 * P, self.cfg.regen_min_steps))
        self.pg_period, self.pg_k = P, 0
        if install and (not P or self.cfg.regen_cap < E):
            raise ValueError("enable_pregen(install=True) needs pipeline >= 1 and cfg.regen_cap >= n_envs (every finished arena "
                             "decides alone inside the step: there is no cap in index order)")
        self.ready = torch.zeros(2 * E, dtype=torch.int64, device=self.device) if P else None
        # (the per-map arrays of a slot-table world are allocated ONCE at twice the size below instead of cloned and then concatenated:
        #  fresh device memory is what the first reset() of a large world waits for -- 50 GB at 4096 arenas of 1000 x 1000 cells)
        shared_maps = bool(install and map_slots)
        self.stage_t = {k: self.t[k].clone() for k in self.STAGED if k in self.t and not (shared_maps and k in self.MAPS)}
        self.stage_t["episode"] = self.t["episode"] + 1
        self.stage_st = abi.NavsimState()
        C.memmove(C.byref(self.stage_st), C.byref(self.st), C.sizeof(self.st))
        for k, v in self.stage_t.items():
            setattr(self.stage_st, k, v.data_ptr())
        if install and map_slots:
            for k in self.MAPS:
                if k in self.t:
                    live = self.t[k]
                    n = live.shape[0]
                    both = torch.empty((2 * n,) + tuple(live.shape[1:]), dtype=live.dtype, device=live.device)
                    both[:n].copy_(live); both[n:].copy_(live)          # (the staged half: any valid world; _stage_all() redraws it)
                    del live
                    self.t[k] = self.stage_t[k] = both
                    setattr(self.st, k, both.data_ptr()); setattr(self.stage_st, k, both.data_ptr())
            self.t["map_slot"] = torch.arange(E, dtype=torch.int32, device=self.device)
            self.stage_t["map_slot"] = torch.arange(E, 2 * E, dtype=torch.int32, device=self.device)
            self.st.map_slot = self.t["map_slot"].data_ptr()
            self.stage_st.map_slot = self.stage_t["map_slot"].data_ptr()
        self.stage_st.arena_cost = None
        self.stage_st.launch_order = None
        self.stage_st.counters = None          # staging ahead serves nobody yet: navsim_regen_swap counts the installs
        self.stage_st.ped_due = None           # the live arenas' flags: navsim_regen_swap clears them at the install
        self.stage_st.ped_due_prev = None
        self.stage_obs = torch.zeros_like(self.obs_buf[0])
        self.want = torch.ones(E, dtype=torch.uint8, device=self.device)
        self.mark = torch.zeros((E + 3) // 4 * 4, dtype=torch.uint8, device=self.device)   # consumed in 32-bit words
        self.stage_io = abi.NavsimStepIO()
        self.stage_io.obs = self.stage_obs.data_ptr()
        self.stage_io.done = self.want.data_ptr()
        self._stage_all(scratch_bytes)
        # a pipelined pass serves what the swaps of P steps marked: P caps' worth of arenas (twice that: what a pass leaves
        # waits for the next one and would be late)
        self.stage_cap = self.cfg.regen_cap
        if P:
            # (install: every launch of a pass is sized by the cap, not by what it serves -- 512 slots for the ~35 arenas a pass
            #  of the reference's configuration stages made its planner launches dispatch 30 000 empty workgroups, 367 us each)
            per_step = max(8, E // 64) if install else self.cfg.regen_cap
            self.stage_cap = int(min(E, stage_cap if stage_cap else 2 * P * per_step))
        self.stage_cfg = self.cfg.copy()
        self.stage_cfg.regen_cap = self.stage_cap
        nbytes = self.lib.navsim_regen_workspace_bytes(C.byref(self.stage_cfg))
        self.stage_ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        # a high-priority stream: its few small kernels go first whenever wave slots free up.  (Measured and dropped:
        # CU-masked streams -- hipExtStreamCreateWithCUMask, 8 / 16 / 32 CUs for the staging passes and the rest for the
        # steps -- did not make the passes overlap a step kernel that fills the chip: c5 3.0 M env-steps/s either way.
        # Pipelined passes: the stream's priority, either way round, changes nothing -- c5 62.8 us per step.)
        # (Its priority: high for the pass that must fit between two steps; ordinary for pipelined passes, which have 2 P steps
        # of slack -- and a process that has used a high-priority stream replays hipGraphs ~10 us per kernel slower from then
        # on: a reference-default environment built afterwards ran at 0.59 M env-steps/s instead of 0.87 M,
        # profiles/_diag/after_pregen.py.)
        self.side = concurrent_stream(self.device, priority=0 if P else -1)
        # navsim_regen's helper stream (the distance transform of new corridor maps beside the searches): beside the steps AND
        # beside the passes -- three streams, three hardware queues
        if self.cfg.regen_plan and self.cfg.regen_indoor_ratio > 0.0:
            self._regen_helper = concurrent_stream(self.device, beside=[torch.cuda.current_stream(self.device), self.side])
            check(self.lib.navsim_regen_helper(C.c_void_p(self._regen_helper.cuda_stream)), "navsim_regen_helper")
        heavy = bool(self.cfg.regen_plan) or self.cfg.regen_indoor_ratio > 0.0
        lanes = int(stage_lanes) if stage_lanes else (2 if (install and P and heavy) else 1)
        if lanes not in (1, 2) or (lanes == 2 and not (install and P)):
            raise ValueError("enable_pregen(stage_lanes=%r): 1, or 2 with install=True and pipeline > 0" % (stage_lanes,))
        # lane 0 = the objects above; lane 1: its own want[], io, workspace, stream and helper stream
        self.stage_lane = [dict(want=self.want, io=self.stage_io, ws=self.stage_ws, side=self.side,
                                helper=getattr(self, "_regen_helper", None))]
        if lanes == 2:
            want2 = torch.zeros(E, dtype=torch.uint8, device=self.device)
            io2 = abi.NavsimStepIO()
            C.memmove(C.byref(io2), C.byref(self.stage_io), C.sizeof(io2))
            io2.done = want2.data_ptr()
            side2 = concurrent_stream(self.device, beside=[torch.cuda.current_stream(self.device), self.side], priority=0)
            helper2 = None
            if getattr(self, "_regen_helper", None) is not None:
                helper2 = concurrent_stream(self.device, beside=[torch.cuda.current_stream(self.device), self.side, side2])
            self.stage_lane.append(dict(want=want2, io=io2, ws=torch.zeros(nbytes, dtype=torch.uint8, device=self.device), side=side2,
                                        helper=helper2))
        self.pg_passes = 0
        self.ev_swapped, self.ev_staged = torch.cuda.Event(), torch.cuda.Event()
        self.ev_staged.record(torch.cuda.current_stream())
        self.pg_swapped = [torch.cuda.Event() for _ in range(3)]      # pass j uses slot j % 3; step j P waits for pass j - 2
        self.pg_staged = [torch.cuda.Event() for _ in range(3)]
        self.pregen = True
        self.pg_install = bool(install)
        self.pg_replan_cap, self.pg_replan_in_step = 0, True       # (set pg_replan_cap to re-plan inside the step's launch, _launch)
        # the fallback: flags the step writes, and a navsim_regen of its own size for the arenas they name
        self.late = torch.zeros(E, dtype=torch.uint8, device=self.device) if fallback else None
        # NEXT_STEP (round 6): an arena learns when its episode ENDS whether the next world is staged; if not, the caller's
        # navsim_regen generates it BESIDE the next step's launch, on a stream of its own (include/navsim.h
        # navsim_step_install_next) -- nothing of the reset path is left on the steps' critical path, and no rule is needed
        # MEASURED AND NOT THE DEFAULT (late_beside=True asks for it): c5, 512 arenas, no rule: 5.4 M env-steps/s against 6.9 M
        # with the fallback's (mostly empty) launches behind the step on the same stream -- the two cross-stream waits per step
        # cost more than the three launches they take off the critical path (profiles/r06_c5/README.md; the same finding as
        # round 5's two-stream re-plan)
        # NEXT_STEP, worlds of outdoor maps without planning or costmap: NO fallback call at all -- an arena that finds nothing staged
        # regenerates its own world inside the step's launch, in place of the step it does not take (include/navsim.h
        # navsim_step_install with late = NULL; kernels_regen_dev.hpp regen_lone).  The library refuses (fewer than 256 threads per
        # arena): _launch falls back to the flags + navsim_regen form.
        self.lone = bool(fallback and self.next_step and not late_beside and not (self.cfg.regen_indoor_ratio > 0.0) and
                         not self.cfg.regen_plan and "costmap" not in self.t)
        self.late2 = None
        if fallback and self.next_step and late_beside:
            self.late2 = [torch.zeros(E, dtype=torch.uint8, device=self.device) for _ in range(2)]
            self.urgent = concurrent_stream(self.device, beside=[torch.cuda.current_stream(self.device), self.side])
            self.ev_stepped, self.ev_urgent = torch.cuda.Event(), torch.cuda.Event()
            self.ev_stepped.record(torch.cuda.current_stream())
        if fallback_poll is None:
            fallback_poll = bool(self.cfg.regen_plan) or self.cfg.regen_indoor_ratio > 0.0
        self.late_poll = bool(fallback and fallback_poll and not self.next_step)
        if fallback:
            self.late_cfg = self.cfg.copy()
            # (every launch of the fallback is sized by its cap whether anybody is late or not: with 16 slots the planner's
            #  four launches alone dispatched 2 x 960 empty 1024-thread workgroups per step, 0.25 ms of a 0.38 ms step of the
            #  reference's configuration, profiles/r05_refdef/timeline_pipeline8_no_rule_cap16.txt.  Late arenas are one in
            #  ~100 steps of that loop; more of them than the cap in ONE step would restart in place, counted as regen_unserved.)
            self.late_cap = int(min(E, fallback_cap if fallback_cap else max(8, E // 128)))
            self.late_cfg.regen_cap = self.late_cap
            if self.late2 is not None:
                self.late_cfg.defer_reset_scan = 1          # (the launch skips a flagged arena altogether: its row is navsim_regen's,
                                                            #  also when the arena lies beyond this call's cap and restarts in place)
            self.late_ws = torch.zeros(self.lib.navsim_regen_workspace_bytes(C.byref(self.late_cfg)), dtype=torch.uint8, device=self.device)

    def restage_all(self, scratch_bytes=4 << 30, slots_from_live=False):
        """After regenerate_all() on a world with enable_pregen(): every staged world is stale (reset() drew new worlds and,
        the second time, new episode numbers) -- stage the world behind each arena's current one again.
        slots_from_live: the live slot table was replaced (load_state_dict) -- the staged one is its complement."""
        import torch
        for ln in getattr(self, "stage_lane", [dict(side=self.side)]):
            ln["side"].synchronize()
        torch.cuda.current_stream().synchronize()
        for ln in getattr(self, "stage_lane", [])[1:]:
            ln["want"].zero_()
        if slots_from_live and "map_slot" in self.t:
            free = torch.ones(2 * self.cfg.n_envs, dtype=torch.bool, device=self.device)
            free[self.t["map_slot"].long()] = False
            self.stage_t["map_slot"].copy_(torch.nonzero(free).flatten().to(torch.int32))
        self.stage_t["episode"].copy_(self.t["episode"] + 1)
        self.want.fill_(1)
        self.mark.zero_()
        if getattr(self, "late2", None) is not None:
            self.urgent.synchronize()
            for b in self.late2:
                b.zero_()
            self.ev_stepped.record(torch.cuda.current_stream())
        self.pg_k, self.pg_open = 0, []
        self._stage_all(scratch_bytes)
        self.ev_staged.record(torch.cuda.current_stream())

    def _stage_all(self, scratch_bytes):
        """The (first) staging of every arena, in chunks of as many arenas as the scratch allows (reset path)."""
        import torch
        E = self.cfg.n_envs
        cfg = self.cfg.copy()
        cfg.regen_cap = 1
        per = self.lib.navsim_regen_workspace_bytes(C.byref(cfg))
        cfg.regen_cap = 2
        per = max(self.lib.navsim_regen_workspace_bytes(C.byref(cfg)) - per, 1)
        cfg.regen_cap = int(max(1, min(E, scratch_bytes // per)))
        ws = torch.empty(self.lib.navsim_regen_workspace_bytes(C.byref(cfg)), dtype=torch.uint8, device=self.device)
        for _ in range((E + cfg.regen_cap - 1) // cfg.regen_cap):
            check(self.lib.navsim_regen_stage(C.byref(cfg), C.byref(self.stage_st), C.byref(self.stage_io), _ptr(self.want),
                                              _ptr(self.mark), _ptr(self.ready), _ptr(ws), ws.numel(), _stream()),
                  "navsim_regen_stage (first staging)")
        torch.cuda.current_stream().synchronize()
        assert int(self.want.sum().item()) == 0

    def _regen_pregen(self):
        import torch
        main = torch.cuda.current_stream()
        P, k = self.pg_period, self.pg_k
        self.pg_k += 1
        j = k // P if P else 0
        if self.pg_install:                              # step() has installed; only the passes are left
            need = self.late is not None and getattr(self, "late2", None) is None and not getattr(self, "lone", False)
            if need and getattr(self, "late_poll", False) and not torch.cuda.is_current_stream_capturing():
                # (the host waits for the step's launch here; what it saves is the fallback's blind launches behind EVERY step)
                need = bool(self.late.any().item())
            if need:   # ... and whoever finished before its world was staged (rare): now
                C.memmove(C.byref(self.late_cfg), C.byref(self.cfg), C.sizeof(self.cfg))
                self.late_cfg.regen_cap = self.late_cap
                io = abi.NavsimStepIO()
                C.memmove(C.byref(io), C.byref(self.io), C.sizeof(io))
                io.obs = self.obs_buf[self.cur].data_ptr()
                io.done = self.late.data_ptr()
                self._latest_flags()
                # this call does not fork: navsim_regen's helper stream belongs to the staging passes, whose distance transforms the
                # fallback's (mostly empty) fork / join would queue behind -- the pass back on the step's critical path
                # (round-5 advisor).  A helper equal to the call's own stream means "no fork" (include/navsim.h).
                helper = getattr(self, "_regen_helper", None)
                if helper is not None:
                    self.lib.navsim_regen_helper(C.c_void_p(main.cuda_stream))
                check(self.lib.navsim_regen(C.byref(self.late_cfg), C.byref(self.st), C.byref(io), _ptr(self.late_ws),
                                            self.late_ws.numel(), C.c_void_p(main.cuda_stream)), "navsim_regen (arenas whose world was not staged)")
                if helper is not None:
                    self.lib.navsim_regen_helper(C.c_void_p(helper.cuda_stream))
            if k % P == 0:
                self._latest_flags()
                self._queue_pass(main, self.pg_swapped[j % 3], self.pg_staged[j % 3])
            return self.obs
        if not P:
            main.wait_event(self.ev_staged)             # the staging pass that served the arenas of the last swap
        elif k % P == 0 and j >= 2:
            main.wait_event(self.pg_staged[(j - 2) % 3])    # the pass queued two periods ago; the later two may still run
        self._latest_flags()
        io = abi.NavsimStepIO()
        C.memmove(C.byref(io), C.byref(self.io), C.sizeof(io))
        io.obs = self.obs_buf[self.cur].data_ptr()
        check(self.lib.navsim_regen_swap(C.byref(self.cfg), C.byref(self.st), C.byref(self.stage_st), C.byref(io),
                                         _ptr(self.stage_obs), _ptr(self.want), _ptr(self.mark), _ptr(self.ready),
                                         C.c_void_p(main.cuda_stream)),
              "navsim_regen_swap")
        if P and k % P != 0:
            return self.obs                             # the next pass takes these arenas' marks with it
        swapped, staged = (self.pg_swapped[j % 3], self.pg_staged[j % 3]) if P else (self.ev_swapped, self.ev_staged)
        self._queue_pass(main, swapped, staged)
        return self.obs

    def pregen_sync(self):
        """Pipelined pre-generation: wait (host) for every staging pass queued so far and start the period count anew -- the
        next 2 P steps then rest on no earlier pass.  Call it before capturing steps into a hipGraph: a captured wait must
        not refer to an event recorded outside the capture."""
        import torch
        for ln in getattr(self, "stage_lane", [dict(side=self.side)]):
            ln["side"].synchronize()
        torch.cuda.current_stream().synchronize()
        self.pg_k = 0
        self.pg_open = []

    def pregen_join(self):
        """Pipelined pre-generation: the current stream waits for the staging passes still open (the last two periods') --
        the join a hipGraph capture needs before it ends; a replay then starts like pregen_sync() left things."""
        import torch
        main = torch.cuda.current_stream()
        for ev in self.pg_open:
            main.wait_event(ev)
        self.pg_k = 0
        self.pg_open = []

    def _queue_pass(self, main, swapped, staged):
        """A staging pass on the side stream, behind everything `main` holds now."""
        if self.pg_period:
            self.pg_open = (getattr(self, "pg_open", []) + [staged])[-2:]
        lanes = getattr(self, "stage_lane", None) or [dict(want=self.want, io=self.stage_io, ws=self.stage_ws, side=self.side, helper=None)]
        n = len(lanes)
        ln = lanes[self.pg_passes % n] if n > 1 else lanes[0]
        part = self.pg_passes % n
        self.pg_passes = getattr(self, "pg_passes", 0) + 1
        side = ln["side"]
        swapped.record(main)
        side.wait_event(swapped)
        C.memmove(C.byref(self.stage_cfg), C.byref(self.cfg), C.sizeof(self.cfg))     # the configuration as it stands NOW
        self.stage_cfg.regen_cap = self.stage_cap
        ws = ln["ws"]
        if n > 1 and ln["helper"] is not None:                 # this lane's passes fork their distance transforms to their own helper
            self.lib.navsim_regen_helper(C.c_void_p(ln["helper"].cuda_stream))
        check(self.lib.navsim_regen_stage_part(C.byref(self.stage_cfg), C.byref(self.stage_st), C.byref(ln["io"]), _ptr(ln["want"]),
                                               _ptr(self.mark), _ptr(self.ready), _ptr(ws), ws.numel(), part, n,
                                               C.c_void_p(side.cuda_stream)),
              "navsim_regen_stage_part")
        if n > 1 and ln["helper"] is not None and getattr(self, "_regen_helper", None) is not None:
            self.lib.navsim_regen_helper(C.c_void_p(self._regen_helper.cuda_stream))
        staged.record(side)

    def close(self):
        """Waits for what this simulator has in flight on streams of its own (staging passes, the overlapped re-plan): their
        kernels write into arrays that are about to be released."""
        extra = [ln[k] for ln in getattr(self, "stage_lane", [])[1:] for k in ("side", "helper")]
        for st in [getattr(self, name, None) for name in ("side", "_side", "_scan_stream", "_regen_helper", "urgent")] + extra:
            if st is not None:
                try:
                    st.synchronize()
                except Exception:
                    pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _choose_regen_helper(self):
        """navsim_regen's helper stream for worlds whose reset forks (include/navsim.h navsim_regen_helper): one that really runs
        beside the current stream, chosen once (before any hipGraph capture of the call)."""
        if getattr(self, "_regen_helper", None) is None and self.cfg.regen_plan and self.cfg.regen_indoor_ratio > 0.0:
            self._regen_helper = concurrent_stream(self.device)
            check(self.lib.navsim_regen_helper(C.c_void_p(self._regen_helper.cuda_stream)), "navsim_regen_helper")

    def regen(self):
        """navsim_regen right after step(): finished arenas get a new map, tables, pedestrians, first obs."""
        import torch
        if getattr(self, "pregen", False):
            return self._regen_pregen()
        self._choose_regen_helper()
        if "regen_ws" not in self.t:
            nbytes = self.lib.navsim_regen_workspace_bytes(C.byref(self.cfg))
            self.t["regen_ws"] = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        io = abi.NavsimStepIO()
        C.memmove(C.byref(io), C.byref(self.io), C.sizeof(io))
        io.obs = self.obs_buf[self.cur].data_ptr()
        if self.next_step:                              # the arenas the latest step RESET: those that finished the step before
            io.done = self.reset_flags.data_ptr()
        self._latest_flags()
        ws = self.t["regen_ws"]
        check(self.lib.navsim_regen(C.byref(self.cfg), C.byref(self.st), C.byref(io), _ptr(ws), ws.numel(), _stream()),
              "navsim_regen")
        return self.obs

    def reset_arenas(self, mask, new_world=False, scratch_bytes=1 << 30):
        """reset() of the arenas of `mask` (uint8 / bool [E]) alone (env.py:730-831 is per environment): navsim_restart -- the
        next start / goal pair of the arena's table, the next episode number -- then their first observations
        (navsim_reset_obs) on the same map, or with new_world=True a new world each (navsim_regen in chunks of cfg.regen_cap
        arenas; worlds that draw a map per episode).  The other arenas keep their state and their rows."""
        import torch
        if getattr(self, "pregen", False):
            raise ValueError("reset_arenas is not available with enable_pregen (the staged worlds follow the episodes' own order)")
        m = torch.as_tensor(mask).to(device=self.device).ne(0).to(torch.uint8).contiguous()
        check(self.lib.navsim_restart(C.byref(self.cfg), C.byref(self.st), _ptr(m), _stream()), "navsim_restart")
        self.reset_obs(m)
        if self.due is not None:                        # nobody of a reset arena waits for navsim_replan
            self.due[self.cur].mul_(m == 0)
        if new_world:
            self._choose_regen_helper()
            cfg = self.cfg.copy()
            cfg.regen_min_steps = 0
            if "regen_ws" not in self.t:
                self.t["regen_ws"] = torch.zeros(self.lib.navsim_regen_workspace_bytes(C.byref(self.cfg)), dtype=torch.uint8, device=self.device)
            ws = self.t["regen_ws"]
            io = abi.NavsimStepIO()
            C.memmove(C.byref(io), C.byref(self.io), C.sizeof(io))
            io.obs = self.obs_buf[self.cur].data_ptr()
            self._latest_flags()
            idx = torch.nonzero(m).flatten()
            chunk = torch.zeros_like(m)
            io.done = chunk.data_ptr()
            for a in range(0, int(idx.numel()), int(self.cfg.regen_cap)):
                chunk.zero_()
                chunk[idx[a:a + int(self.cfg.regen_cap)]] = 1
                check(self.lib.navsim_regen(C.byref(cfg), C.byref(self.st), C.byref(io), _ptr(ws), ws.numel(), _stream()),
                      "navsim_regen (reset of some arenas)")
            torch.cuda.current_stream().synchronize()  # `chunk` is released on return
        return self.obs

    def regenerate_all(self, new_episode=False, scratch_bytes=4 << 30):
        """reset() of EVERY arena on the device (env.py:730-831 for the whole batch): navsim_regen with all arenas
        marked finished -- a new map per arena (indoor / outdoor by cfg.regen_indoor_ratio), its distance field
        (and rect records / costmap when those buffers exist), start / goal table, robot, pedestrians, the
        per-episode env_param draws and the first observation.  Runs in chunks of arenas so that the scratch
        stays below `scratch_bytes`.  new_episode: bump every arena's episode counter first (a second reset()
        must not reproduce the maps of the first)."""
        import torch
        E = self.cfg.n_envs
        self._latest_flags()
        self._choose_regen_helper()
        if new_episode:
            self.t["episode"] += 1
        cfg = self.cfg.copy()
        cfg.regen_min_steps = 0                 # a reset of everything: whatever the last episodes' lengths
        cfg.regen_cap = 1
        per = self.lib.navsim_regen_workspace_bytes(C.byref(cfg))
        cfg.regen_cap = 2
        per = max(self.lib.navsim_regen_workspace_bytes(C.byref(cfg)) - per, 1)
        chunk = int(max(1, min(E, scratch_bytes // per)))
        cfg.regen_cap = chunk
        ws = torch.empty(self.lib.navsim_regen_workspace_bytes(C.byref(cfg)), dtype=torch.uint8, device=self.device)
        done = torch.zeros(E, dtype=torch.uint8, device=self.device)
        io = abi.NavsimStepIO()
        C.memmove(C.byref(io), C.byref(self.io), C.sizeof(io))
        io.obs = self.obs_buf[self.cur].data_ptr()
        io.done = done.data_ptr()
        for a in range(0, E, chunk):
            done.zero_()
            done[a:a + chunk] = 1
            check(self.lib.navsim_regen(C.byref(cfg), C.byref(self.st), C.byref(io), _ptr(ws), ws.numel(), _stream()),
                  "navsim_regen (reset of all arenas)")
        for b in self.out_buf:                             # nobody is finished after a reset (NEXT_STEP: the next launch's reset mask)
            b["done"].zero_()
        torch.cuda.current_stream().synchronize()          # `ws` and `done` are released on return
        return self.obs

    def counters(self, reset=False):
        """navsim_state.counters as a dict (abi.COUNTERS): arenas / pedestrians the calls served and what their caps
        (cfg.regen_cap, replan's max_queries, cfg.max_waypoints) left waiting or cut.  One device -> host copy."""
        v = self.t["counters"].cpu().numpy()
        if reset:
            self.t["counters"].zero_()
        return {k: int(v[i]) for i, k in enumerate(abi.COUNTERS)}

    def occupancy(self, e=0):
        """uint8 [H, W] occupancy grid (1 = occupied) of arena e, read back from its distance field
        (a cell is occupied exactly when its distance is 0)."""
        H, W = self.cfg.map_h, self.cfg.map_w
        if "map_slot" in self.t:                        # where the arena's map lies (enable_pregen(install=True, map_slots=True))
            e = int(self.t["map_slot"][e].item())
        if self.cfg.field_format == abi.FIELD_F32:
            return (self.t["field"][e] == 0).to(_dtype("uint8")).cpu().numpy()
        tpr, tpc = (W + 7) // 8, (H + 7) // 8
        per = tpr * tpc * 64
        blob = self.t["field"].reshape(-1)[e * per:(e + 1) * per].cpu().numpy().view(np.uint16)
        tiles = blob.reshape(tpc, tpr, 8, 8).transpose(0, 2, 1, 3).reshape(tpc * 8, tpr * 8)
        return (tiles[:H, :W] == 0).astype(np.uint8)

    def replan(self, max_queries=1024, flags=True):
        """navsim_replan (env.py:667-680): pedestrians standing on their final waypoint get a new goal and
        the waypoints of a planned path.  Needs the resident costmap (world.make_world(plan_paths=True)).
        flags=True: the candidates are the pedestrians the last step flagged (navsim_state.ped_due); False: the call
        finds them by its own pass over the state (after the caller moved pedestrians by hand)."""
        import torch
        key = "replan_ws_%d" % max_queries
        if key not in self.t:
            nbytes = self.lib.navsim_replan_workspace_bytes(C.byref(self.cfg), max_queries)
            self.t[key] = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        ws = self.t[key]
        self._latest_flags()
        st = self.st
        if not flags:
            st = abi.NavsimState()
            C.memmove(C.byref(st), C.byref(self.st), C.sizeof(st))
            st.ped_due = None
        check(self.lib.navsim_replan(C.byref(self.cfg), C.byref(st), max_queries, _ptr(ws), ws.numel(), _stream()),
              "navsim_replan")

    # ---- navsim_replan beside the step (round 5).  env.py:667-680 plans a new route inside step() for the pedestrian that
    # reached its goal; here that is one breadth-first search per arrived pedestrian (~2 % of the arenas per step), a chain of
    # dependent levels that used to sit serially behind every step (c3 world through the gym API: 166 us of step + 98 us of
    # re-plan).  The arenas WITHOUT a waiting pedestrian do not need its result: navsim_step_part steps them on the caller's
    # stream while a side stream runs the re-plan of the previous step and then steps the arenas that waited for it.
    # Same kernels on the same per-arena inputs in the same per-arena order -- step(t), replan(t), step(t + 1) -- so every
    # result is what the serial sequence gives.
    replan_in_step = True             # planned routes: navsim_step_replan (one launch) instead of the two-stream overlap

    def launch_step_replan(self, replan_cap=1024, reorder=True):
        """navsim_step_replan: the re-plan of the PREVIOUS step's flags inside this step's launch (include/navsim.h)."""
        if reorder:
            self._reorder()
        self._flip()
        rc = self.lib.navsim_step_replan(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), int(replan_cap), _stream())
        if rc:
            check(rc, "navsim_step_replan")
        self.cur = 1 - self.cur

    overlap_big_first = False         # which stream takes the big launch (launch_step_overlapped); A/B: profiles/r05_replan/

    def _overlap_streams(self):
        import torch
        if not hasattr(self, "_side"):
            self._side = concurrent_stream(self.device, priority=-1 if self.overlap_big_first else 0)
        return torch.cuda.current_stream(self.device), self._side

    def launch_step_overlapped(self, replan_cap=1024, reorder=True):
        """navsim_step_part(NOT_DUE) beside [navsim_replan of the PREVIOUS step's flags -> navsim_step_part(DUE)], on two
        streams, joined at the end.  Needs the costmap (planned routes).
        Which stream takes what (overlap_big_first = False, measured: profiles/r05_replan/README.md): the re-plan chain runs on
        the caller's stream, where nothing has to be waited for, and the big launch on a side stream behind an event -- a few us
        later, so the searches' workgroups are resident before 4096 arena workgroups take every slot of the chip
        (c3 world through the gym API: 19.2-19.4 M env-steps/s; the other way round 18.6 M; serial, round 4: 15.5 M)."""
        if self.replan_in_step:
            if reorder:
                self._reorder()
            self._flip()
            rc = self.lib.navsim_step_replan(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), int(replan_cap), _stream())
            if rc == 0:
                self.cur = 1 - self.cur
                return
            if rc != abi.E_UNSUPPORTED:
                check(rc, "navsim_step_replan")
            self.replan_in_step = False             # a costmap of more words than the arena's workgroup has threads: two streams
            self._flip()                            # (idempotent: the same parity)
            reorder = False
        import torch
        main, side = self._overlap_streams()
        big, chain = (main, side) if self.overlap_big_first else (side, main)
        key = "replan_ws_%d" % replan_cap
        if key not in self.t:
            self.t[key] = torch.zeros(self.lib.navsim_replan_workspace_bytes(C.byref(self.cfg), replan_cap), dtype=torch.uint8,
                                      device=self.device)
        ws = self.t[key]
        if reorder:
            self._reorder()
        for s_ in {big, chain} - {main}:            # the previous step (both parts were joined on `main`), regen, the actions
            s_.wait_stream(main)
        # the re-plan reads st.ped_due = the flags the previous step wrote: launched BEFORE the buffers flip
        self._latest_flags()
        rc = self.lib.navsim_replan(C.byref(self.cfg), C.byref(self.st), replan_cap, _ptr(ws), ws.numel(), C.c_void_p(chain.cuda_stream))
        if rc:
            check(rc, "navsim_replan")
        self._flip()
        rc = self.lib.navsim_step_part(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), abi.STEP_NOT_DUE, C.c_void_p(big.cuda_stream))
        if rc:
            check(rc, "navsim_step_part (not due)")
        rc = self.lib.navsim_step_part(C.byref(self.cfg), C.byref(self.st), C.byref(self.io), abi.STEP_DUE, C.c_void_p(chain.cuda_stream))
        if rc:
            check(rc, "navsim_step_part (due)")
        for s_ in {big, chain} - {main}:
            main.wait_stream(s_)
        self.cur = 1 - self.cur

    def step_overlapped(self, action=None, replan_cap=1024):
        """step() of a world with planned pedestrian routes: replan (of the previous step) + step, overlapped
        (launch_step_overlapped).  The sequence of calls  step_overlapped, step_overlapped, ...  equals
        step, replan, step, replan, ...  shifted by one replan: finish a rollout with replan() to leave the same state."""
        if action is not None:
            import torch
            if (isinstance(action, torch.Tensor) and action.is_cuda and action.dtype == torch.float64 and action.is_contiguous()
                    and action.device == self.device and action.numel() == self.action.numel()):
                self.io.action = action.data_ptr()
                self._action_ref = action
            else:
                self.action.copy_(self._as(action, self.action))
                self.io.action = self.action.data_ptr()
        self.launch_step_overlapped(replan_cap)
        return self.obs, self.out

    def set_policy(self, weights):
        """HumanPolicy actor weights: dict of arrays / tensors named like abi.POLICY_FIELDS, or a reference
        state_dict (human_policy.py names, e.g. torch.load('human_policy.pth'))."""
        import torch
        named = {abi.POLICY_STATE_DICT.get(k, k): v for k, v in weights.items()}
        self.policy_t = {}
        self.policy_w = abi.NavsimPolicyWeights()
        for k in abi.POLICY_FIELDS:
            t = torch.as_tensor(named[k]).detach().to(device=self.device, dtype=torch.float32).reshape(abi.POLICY_SHAPES[k]).contiguous()
            self.policy_t[k] = t
            setattr(self.policy_w, k, t.data_ptr())
        nbytes = self.lib.navsim_ped_policy_workspace_bytes(C.byref(self.cfg))
        self.t["policy_ws"] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        if "policy_prev_actions" not in self.t:
            self.t["policy_prev_actions"] = torch.zeros((self.cfg.n_envs, self.cfg.max_peds, 2), dtype=torch.float32,
                                                        device=self.device)

    def _ped_policy_pipelined(self, n_slices):
        """scans -> network in slices of arenas on two streams (round 5; MEASURED AND NOT THE DEFAULT): the scans of slice
        k + 1 beside the network of slice k, same kernels on the same rows.  The round-4 verdict's idea -- a latency-bound
        march beside a layer at 0.8 of the f32 MFMA peak -- loses: 3.26 / 3.54 / 5.5 ms for 3 / 6 / 12 slices against 3.0 ms for
        the two calls at c3 size (profiles/r05_policy/): the march's waves and the layer's share the CUs' issue slots and LDS,
        smaller slices run the layer below its peak, and every slice costs two cross-stream waits."""
        import torch
        main = torch.cuda.current_stream(self.device)
        if not hasattr(self, "_scan_stream"):
            self._scan_stream = torch.cuda.Stream(device=self.device)
            self.t["ped_scan_rows"] = torch.zeros((self.cfg.n_envs, self.cfg.max_peds, self.cfg.ped_n_beams), dtype=torch.float32,
                                                  device=self.device)
        side, rows, ws, E = self._scan_stream, self.t["ped_scan_rows"], self.t["policy_ws"], self.cfg.n_envs
        bounds = [E * k // n_slices for k in range(n_slices + 1)]
        side.wait_stream(main)                      # the state the scans read
        events = []
        for k in range(n_slices):
            e0, n_e = bounds[k], bounds[k + 1] - bounds[k]
            check(self.lib.navsim_ped_scans_part(C.byref(self.cfg), C.byref(self.st), _ptr(rows), e0, n_e, C.c_void_p(side.cuda_stream)),
                  "navsim_ped_scans_part")
            ev = torch.cuda.Event()
            ev.record(side)
            events.append(ev)
        for k in range(n_slices):
            e0, n_e = bounds[k], bounds[k + 1] - bounds[k]
            main.wait_event(events[k])
            check(self.lib.navsim_ped_policy_part(C.byref(self.cfg), C.byref(self.st), C.byref(self.policy_w), _ptr(rows),
                                                  _ptr(self.t["policy_prev_actions"]), _ptr(self.t["ped_cmd"]), _ptr(ws), ws.numel(),
                                                  e0, n_e, C.c_void_p(main.cuda_stream)), "navsim_ped_policy_part")
        return self.t["ped_cmd"], self.t["policy_prev_actions"]

    def ped_policy(self, scans=None, fused=False, scans_out=None, pipeline=None):
        """navsim_ped_policy (env.py:617-662): pedestrian scans -> HumanPolicy actor -> ped_cmd for a
        NAVSIM_PED_EXTERNAL step.  Returns (ped_cmd [E,N,2] float64, clip(mean) [E,N,2] float32).
        scans=None: the scans of the current state (navsim_ped_scans first; pipeline=n: taken and consumed in n slices of
        arenas on two streams, _ped_policy_pipelined -- measured slower, kept as an option).
        fused=True: navsim_ped_scan_policy -- every
        scan is taken inside the pass by the workgroup that convolves it and only written to HBM when scans_out (float32
        [E,N,512]) is given; bit-identical, saves the [E,N,512] buffer, measured 1 % slower than the two calls."""
        ws = self.t["policy_ws"]
        if scans is None and not fused and pipeline:
            return self._ped_policy_pipelined(int(pipeline) if int(pipeline) > 1 else 3)
        if fused:
            check(self.lib.navsim_ped_scan_policy(C.byref(self.cfg), C.byref(self.st), C.byref(self.policy_w),
                                                  _ptr(scans_out) if scans_out is not None else None,
                                                  _ptr(self.t["policy_prev_actions"]), _ptr(self.t["ped_cmd"]), _ptr(ws),
                                                  ws.numel(), _stream()), "navsim_ped_scan_policy")
            return self.t["ped_cmd"], self.t["policy_prev_actions"]
        if scans is None:
            scans = self.ped_scans()
        check(self.lib.navsim_ped_policy(C.byref(self.cfg), C.byref(self.st), C.byref(self.policy_w), _ptr(scans),
                                         _ptr(self.t["policy_prev_actions"]), _ptr(self.t["ped_cmd"]), _ptr(ws),
                                         ws.numel(), _stream()), "navsim_ped_policy")
        return self.t["ped_cmd"], self.t["policy_prev_actions"]

    def ped_scans(self):
        """Scan of every pedestrian (env.py:685-693) from the current state -> float32 [E, N, 512]."""
        import torch
        out = torch.zeros((self.cfg.n_envs, self.cfg.max_peds, self.cfg.ped_n_beams), dtype=torch.float32,
                          device=self.device)
        check(self.lib.navsim_ped_scans(C.byref(self.cfg), C.byref(self.st), _ptr(out), _stream()), "navsim_ped_scans")
        return out

    # ---- hipGraph replay of a whole step (round 4).  A step of a world that draws new maps is four or five launches
    # (navsim_step, navsim_regen's three, navsim_replan's three) whose gaps are launch-bound: c5 4.77 -> 5.08 M env-steps/s.
    def enable_graphs(self, regen=False, replan_cap=0, overlap=True):
        """Capture a whole step once per observation-buffer parity and replay it in step_graphed():
        [navsim_step, navsim_regen (regen=True)], and with replan_cap > 0 the re-plan -- overlap=True (default): the re-plan
        of the PREVIOUS step beside this step's launch (launch_step_overlapped: a fork and a join inside the graph), False:
        navsim_replan behind the step as in round 4.  Actions go through the simulator's own action buffer (a graph's
        arguments are frozen), the longest-first launch order is not re-sorted (graphs are for the launches of one
        generation, where it does not matter).  Same kernels, same arguments, same per-arena order: same results.
        The configuration is captured BY VALUE: step_graphed() re-captures when self.cfg has changed since."""
        import torch
        if getattr(self, "pregen", False):
            raise ValueError("enable_graphs and enable_pregen are alternatives")
        if regen:
            self._choose_regen_helper()                   # (timed with host synchronisation: before the capture)
        io = abi.NavsimStepIO()
        C.memmove(C.byref(io), C.byref(self.io), C.sizeof(io))
        io.obs = self.obs_buf[self.cur].data_ptr()
        check(self.lib.navsim_prepare(C.byref(self.cfg), C.byref(self.st), C.byref(io)), "navsim_prepare")
        if regen and "regen_ws" not in self.t:
            nbytes = self.lib.navsim_regen_workspace_bytes(C.byref(self.cfg))
            self.t["regen_ws"] = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        if replan_cap:
            key = "replan_ws_%d" % replan_cap
            if key not in self.t:
                self.t[key] = torch.zeros(self.lib.navsim_replan_workspace_bytes(C.byref(self.cfg), replan_cap), dtype=torch.uint8,
                                          device=self.device)
        self.io.action = self.action.data_ptr()
        self._graph_args = (bool(regen), int(replan_cap), bool(overlap))
        overlap = bool(overlap) and replan_cap > 0
        if overlap:
            self._overlap_streams()
        torch.cuda.synchronize(self.device)
        cur0, self._graphs = self.cur, {}
        for p in (0, 1):
            self.cur = p
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                if overlap:
                    self.launch_step_overlapped(replan_cap, reorder=False)
                else:
                    self.launch_step(reorder=False)
                if regen:
                    self.regen()
                if replan_cap and not overlap:
                    self.replan(replan_cap)
            self._graphs[p] = g
        self.cur = cur0
        # a captured launch holds navsim_config BY VALUE (round-4 advisor: _override_reward_factor after the capture changed
        # compute_rewards() but not the step): remember what was captured, step_graphed() compares
        self._graph_cfg = bytes(self.cfg)
        torch.cuda.synchronize(self.device)

    def step_graphed(self, action=None):
        """step() (+ regen + replan, as captured by enable_graphs) as ONE graph launch."""
        if bytes(self.cfg) != self._graph_cfg:      # the configuration changed since the capture: capture again
            self.enable_graphs(*self._graph_args)
        if action is not None:
            self.action.copy_(self._as(action, self.action))
        self._graphs[self.cur].replay()
        self.cur = 1 - self.cur
        return self.obs, self.out

    def launch_step(self, reorder=True):
        """step() without the action copy: inputs already resident (bench inner loop).  reorder=False: the
        caller has already called _reorder() (bench.py keeps it outside its per-kernel events)."""
        if reorder:
            self._reorder()
        self._flip()
        self._launch()
        self.cur = 1 - self.cur

    def set_ped_cmd(self, cmd):
        self.t["ped_cmd"].copy_(self._as(cmd, self.t["ped_cmd"]))

    def _as(self, a, like):
        import torch
        if not isinstance(a, torch.Tensor):          # numpy >= 2 arrays also carry a `.device`
            a = torch.as_tensor(np.asarray(a))
        return a.to(device=self.device, dtype=like.dtype).reshape(like.shape)

    def by_arena(self, name):
        """t[name] with one entry per arena: the per-map arrays of a world with slot tables (2 E slots, a packed field being one
        flat blob) gathered through t['map_slot']; every other array as it is."""
        v = self.t[name]
        slots = self.t.get("map_slot")
        if slots is None or name not in self.MAPS:
            return v
        n_slots = 2 * self.cfg.n_envs
        rows = v.reshape(n_slots, -1)[slots.long()]
        return rows.reshape(-1) if v.dim() == 1 else rows.reshape((self.cfg.n_envs,) + tuple(v.shape[1:]))

    def numpy_state(self, *names):
        """State arrays on the host.  "ped_due": the latest step's "waits for navsim_replan" flags (they flip, so not in t)."""
        out = {n: self.by_arena(n).detach().cpu().numpy() for n in (names or self.t.keys()) if n != "ped_due"}
        if self.due is not None and (not names or "ped_due" in names):
            out["ped_due"] = self.due[self.cur].detach().cpu().numpy()
        return out
