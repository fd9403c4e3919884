"""Reset-path world generation (host + torch plumbing; NOT the hot path).

Restates what NavGymEnv.reset() needs before the first step (env.py:730-831) for E arenas at once:
occupancy maps, the distance field (navsim_build_field on the GPU), start / goal tables and
pedestrians; with plan_paths the planning costmap (env.py:312-332) and planned first paths
(navsim_plan, env.py:342-383).

Everything is keyed by the GLOBAL env index (seed + env_index_base + e), so a sharded run over
G GPUs builds bit-identical arenas to a single-GPU run.
"""
import numpy as np

from . import abi


def outdoor_map(rng, size, n_obstacles=10, width_range=(0.3, 1.0)):
    """create_outdoor_map (map_generator.py:126-143) scaled to size x size cells: 5-cell border
    wall and `n_obstacles` squares of half-width int(10*U[width_range]) cells; uint8 {0,1}."""
    m = np.ones((size, size), np.uint8)
    m[5:size - 5, 5:size - 5] = 0
    hw = int(10 * rng.uniform(*width_range))
    for _ in range(n_obstacles):
        cx = rng.integers(hw + 2, size - hw - 1)
        cy = rng.integers(hw + 2, size - hw - 1)
        m[cx - hw:cx + hw + 1, cy - hw:cy + hw + 1] = 1
    return np.flipud(m)


def indoor_map(rng, size, corridor_width=3, iterations=100):
    """create_indoor_map (map_generator.py:97-123): a random tree of corridors carved on a coarse grid
    (L1-nearest node, L-shaped paths), nearest-neighbour upscaled to `size`.  The reference carves a
    100x100 grid for its 1000-cell maps (0.5 m per grid cell); other sizes keep that METRIC scale --
    size/10 grid cells, iterations scaled with the area -- so corridors stay 3.5-4.5 m wide and survive
    the planner's 1 m inflation."""
    r = int(corridor_width)
    G = max(int(size) // 10, 2 * r + 8)
    n_it = max(4, int(round(int(iterations) * (G / 100.0) ** 2)))
    g = np.ones((G, G), np.uint8)
    tree = [(G // 2, G // 2)]
    g[G // 2, G // 2] = 0
    for _ in range(n_it):
        p = (int(rng.integers(r + 2, G - r - 1)), int(rng.integers(r + 2, G - r - 1)))
        d = [abs(p[0] - q[0]) + abs(p[1] - q[1]) for q in tree]
        q = tree[int(np.argmin(d))]
        tree.append(p)
        x1, x2 = sorted((p[0], q[0]))
        y1, y2 = sorted((p[1], q[1]))
        if rng.random() >= 0.5:
            g[x1 - r:x1 + r + 1, y1 - r:y2 + r + 1] = 0
            corner_y = y1 if ((p[0] > q[0]) != (p[1] < q[1])) else y2
            g[x1 - r:x2 + r + 1, corner_y - r:corner_y + r + 1] = 0
        else:
            g[x2 - r:x2 + r + 1, y1 - r:y2 + r + 1] = 0
            corner_y = y2 if ((p[0] > q[0]) != (p[1] < q[1])) else y1
            g[x1 - r:x2 + r + 1, corner_y - r:corner_y + r + 1] = 0
    idx = np.minimum((np.arange(size) * (float(G) / size)).astype(int), G - 1)
    return np.flipud(g[np.ix_(idx, idx)])


def make_maps(n_envs, size, seed, env_index_base=0, indoor_ratio=0.0, n_obstacles=10):
    """uint8 [E, size, size], nonzero = occupied (map_info['data'] >= 0.1, env.py:339)."""
    occ = np.empty((n_envs, size, size), np.uint8)
    for e in range(n_envs):
        rng = np.random.default_rng(seed + env_index_base + e)
        if rng.random() < indoor_ratio:
            occ[e] = indoor_map(rng, size, rng.integers(3, 5), rng.integers(80, 151))
        else:
            occ[e] = outdoor_map(rng, size, n_obstacles)
    return occ


# ---- counter-based uniforms in torch (shard-invariant) -------------------------------------------
def _s64(c):
    return c - (1 << 64) if c >= (1 << 63) else c


def _mix64(z):
    """splitmix64 finaliser on int64 tensors (wrapping multiply, logical shifts via masks)."""
    m34, m37, m33 = (1 << 34) - 1, (1 << 37) - 1, (1 << 33) - 1
    z = z + _s64(0x9E3779B97F4A7C15)
    z = (z ^ ((z >> 30) & m34)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ ((z >> 27) & m37)) * _s64(0x94D049BB133111EB)
    return z ^ ((z >> 31) & m33)


def _uniform(seed, genv, stream, shape_tail, device):
    """float64 uniforms in [0,1) of shape [E, *shape_tail], keyed by (seed, global env, stream, index)."""
    import torch
    n = int(np.prod(shape_tail)) if len(shape_tail) else 1
    idx = torch.arange(n, dtype=torch.int64, device=device)[None, :]
    key = _mix64(genv.to(torch.int64)[:, None] * 1000003 + seed * 7919 + stream)
    h = _mix64(key ^ _mix64(idx))
    u = ((h >> 11) & ((1 << 53) - 1)).to(torch.float64) / float(1 << 53)
    return u.reshape((genv.shape[0],) + tuple(shape_tail))


def sample_free_cells(field, clearance_cells, k, seed, genv, stream, chunk=128):
    """k cells per env with field >= clearance, uniformly (hash scores + top-k).  -> int64 [E,k] (j*W+i).
    An arena too cramped for `clearance_cells` uses 60 % of its own largest clearance instead."""
    import torch
    E, H, W = field.shape
    out = torch.empty((E, k), dtype=torch.int64, device=field.device)
    for e0 in range(0, E, chunk):
        f = field[e0:e0 + chunk].reshape(-1, H * W)
        u = _uniform(seed, genv[e0:e0 + chunk], stream, (H * W,), field.device).to(torch.float32)
        thr = torch.clamp(0.6 * f.max(dim=1, keepdim=True).values, max=float(clearance_cells))
        u = torch.where(f >= thr, u, torch.full_like(u, -1.0))
        out[e0:e0 + chunk] = torch.topk(u, k, dim=1).indices
    return out


def cells_to_xy(cells, W, resolution, origin):
    """ij_to_xy (env.py:1214-1220): cell centres."""
    import torch
    i = (cells % W).to(torch.float64)
    j = (cells // W).to(torch.float64)
    return torch.stack([(i + 0.5) * resolution + origin[0], (j + 0.5) * resolution + origin[1]], dim=-1)


def make_world(cfg, occ, seed=None, n_peds=0, min_goal_dist=10.0, max_goal_dist=20.0,
               robot_clearance=1.2, ped_clearance=0.5, noise_std_range=(0.0, 0.0),
               has_legs_ratio=0.5, v_pref_range=(0.0, 0.6), device="cuda:0", field=None, plan_paths=False,
               rect_table=None):
    """Builds every navsim_state array for cfg.n_envs arenas on `device`.

    occ: uint8 numpy/torch [E,H,W].  Returns dict name -> torch tensor (see abi.STATE_LAYOUT).
    plan_paths: also keep the planning costmap of every arena resident (env.py:312-332) and give the
    pedestrians the waypoints of a planned path to their first goal (env.py:788-804); pedestrians
    then re-plan through NavSim.replan() (env.py:667-680).
    rect_table: build the two-rectangle tile records of the packed field (navsim_build_rects), the march's
    exact shortcut around most field reads; None = whenever the format allows it (FIELD_U16T, side <= 1024)."""
    import torch
    from . import sim
    seed = int(cfg.seed if seed is None else seed)
    dev = torch.device(device)
    E, N, K = cfg.n_envs, cfg.max_peds, max(cfg.n_spawn, 1)
    res, org = cfg.resolution, (cfg.origin_x, cfg.origin_y)
    a = {}
    if field is None:
        occ_t = torch.as_tensor(occ).to(dev)
        H0, W0 = occ_t.shape[1:]
        packed, f32, nsat = sim.build_field(occ_t, cfg.field_format)
        if cfg.field_format == abi.FIELD_F32:
            field = packed
            a["field"] = field
        else:
            field = f32                                   # float32 plane: spawn sampling below
            a["field"] = packed
            if nsat > 0:                                  # some cell >= 256 cells from every obstacle
                a["field_overflow"] = f32
            if rect_table is None:
                rect_table = max(H0, W0) <= 1024
            if rect_table and cfg.field_format == abi.FIELD_U16T:
                a["rect_table"] = sim.build_rects(occ_t, packed, cfg.field_format, f32)
                a["rect_index"], _ = sim.build_rect_index(a["rect_table"], H0, W0)    # what the step stages in LDS
                # the LDS form of the march needs closed maps (include/navsim.h closed_maps): asserted only when the library
                # has found every map of this world closed
                cfg.closed_maps = int(bool((sim.maps_closed(occ_t) == 1).all().item()))
        if plan_paths:
            a["costmap"] = sim.costmap(occ_t)
        del occ_t
    else:
        if cfg.field_format != abi.FIELD_F32:
            raise ValueError("a pre-built field must be float32 (cfg.field_format = FIELD_F32)")
        a["field"] = field
    H, W = field.shape[1:]
    genv = torch.arange(E, device=dev, dtype=torch.int64) + int(cfg.env_index_base)
    # ---- robot start / goal tables (env.py:748-783 without A*) -------------------------------------
    KK = max(2 * K, 8)
    if "costmap" in a:              # env.py:356-368: starts and goals are centres of free costmap cells
        cfield = (1.0 - a["costmap"].to(torch.float32)) * 1000.0
        Wc = cfield.shape[2]
        cells = sample_free_cells(cfield, 0.5, KK, seed, genv, 11)
        xy = cells_to_xy(cells, Wc, res * 5, org)
    else:
        cells = sample_free_cells(field, robot_clearance / res, KK, seed, genv, 11)
        xy = cells_to_xy(cells, W, res, org)                               # [E,KK,2]
    theta = _uniform(seed, genv, 12, (KK,), dev) * (2 * np.pi)
    d = torch.cdist(xy, xy)                                                # [E,KK,KK]
    ok = (d > min_goal_dist) & (d < max_goal_dist)
    # first acceptable partner in cyclic order, else the farthest candidate
    order = (torch.arange(KK, device=dev)[None, :] + torch.arange(KK, device=dev)[:, None] + 1) % KK   # [KK,KK]
    ok_c = torch.gather(ok, 2, order[None].expand(E, KK, KK))
    first = torch.argmax(ok_c.to(torch.int8), dim=2)
    partner = torch.gather(order[None].expand(E, KK, KK), 2, first[..., None])[..., 0]
    far = torch.argmax(d, dim=2)
    partner = torch.where(ok_c.any(dim=2), partner, far)
    goal = torch.gather(xy, 1, partner[..., None].expand(E, KK, 2))
    if "costmap" in a:              # env.py:756-762: keep pairs joined by a path <= 2x the straight line
        mi = torch.arange(E, device=dev, dtype=torch.int32).repeat_interleave(K)
        s_, g_ = xy[:, :K].reshape(-1, 2), goal[:, :K].reshape(-1, 2)
        _, pn, _, plen = sim.plan(a["costmap"], s_, g_, 5.0, max_wp=cfg.max_waypoints, res_c=res * 5, origin=org,
                                  map_index=mi)
        valid = ((pn > 0) & (plen <= 2.0 * (g_ - s_).norm(dim=1))).reshape(E, K)
        # an invalid slot borrows the next valid one in cyclic order (none valid: left as drawn)
        order_k = (torch.arange(K, device=dev)[None, :] + torch.arange(K, device=dev)[:, None]) % K      # [K,K]
        v_c = torch.gather(valid[:, None, :].expand(E, K, K), 2, order_k[None].expand(E, K, K))
        src = torch.gather(order_k[None].expand(E, K, K), 2, torch.argmax(v_c.to(torch.int8), dim=2)[..., None])[..., 0]
        xy = torch.cat([torch.gather(xy[:, :K], 1, src[..., None].expand(E, K, 2)), xy[:, K:]], dim=1)
        goal = torch.cat([torch.gather(goal[:, :K], 1, src[..., None].expand(E, K, 2)), goal[:, K:]], dim=1)
    a["spawn_pose"] = torch.cat([xy[:, :K], theta[:, :K, None]], dim=2).contiguous()
    a["spawn_goal"] = goal[:, :K].contiguous()
    a["robot_pose"] = a["spawn_pose"][:, 0].clone()
    a["robot_goal"] = a["spawn_goal"][:, 0].clone()
    a["prev_action"] = torch.zeros((E, 2), dtype=torch.float64, device=dev)
    a["prev_pose"] = torch.zeros((E, 3), dtype=torch.float64, device=dev)
    a["n_hist"] = torch.zeros(E, dtype=torch.int32, device=dev)
    a["episode"] = torch.zeros(E, dtype=torch.int64, device=dev)
    a["steps"] = torch.zeros(E, dtype=torch.int64, device=dev)
    lo, hi = noise_std_range
    a["scan_noise_std"] = (lo + (hi - lo) * _uniform(seed, genv, 13, (), dev)).to(torch.float32).reshape(E)
    # ---- pedestrians (env.py:786-806 without A*) -----------------------------------------------------
    if cfg.ped_model != abi.PED_NONE:
        n_peds_t = torch.as_tensor(n_peds, device=dev).to(torch.int32).expand(E).contiguous() \
            if not hasattr(n_peds, "shape") or len(getattr(n_peds, "shape", ())) == 0 \
            else torch.as_tensor(n_peds).to(device=dev, dtype=torch.int32)
        M = 4 * N
        if "costmap" in a:
            pc = sample_free_cells(cfield, 0.5, M, seed, genv, 21)
            pxy = cells_to_xy(pc, Wc, res * 5, org)
        else:
            pc = sample_free_cells(field, ped_clearance / res, M, seed, genv, 21)
            pxy = cells_to_xy(pc, W, res, org)                              # [E,M,2]
        far_enough = (pxy - a["robot_pose"][:, None, :2]).norm(dim=2) >= 4.0   # env.py:372
        rank = torch.argsort((~far_enough).to(torch.int8), dim=1, stable=True)  # acceptable starts first
        start = torch.gather(pxy, 1, rank[:, :N, None].expand(E, N, 2))
        gd = torch.cdist(start, pxy)                                        # [E,N,M]
        gok = gd > float(cfg.ped_min_goal_dist)                             # env.py:788-791 (10 m)
        gfirst = torch.argmax(gok.to(torch.int8), dim=2)
        gidx = torch.where(gok.any(dim=2), gfirst, torch.argmax(gd, dim=2))
        pgoal = torch.gather(pxy, 1, gidx[..., None].expand(E, N, 2))
        pth = _uniform(seed, genv, 22, (N,), dev) * (2 * np.pi)
        a["n_peds"] = n_peds_t
        a["ped_pose"] = torch.cat([start, pth[..., None]], dim=2).contiguous()
        a["ped_vel"] = torch.zeros((E, N, 2), dtype=torch.float64, device=dev)
        a["ped_prev_yaw"] = torch.zeros((E, N), dtype=torch.float64, device=dev)
        a["ped_dist"] = torch.zeros((E, N, 3), dtype=torch.float64, device=dev)
        a["ped_v_pref"] = v_pref_range[0] + (v_pref_range[1] - v_pref_range[0]) * _uniform(seed, genv, 23, (N,), dev)
        a["ped_has_legs"] = (_uniform(seed, genv, 24, (N,), dev) < has_legs_ratio).to(torch.uint8)
        P = cfg.max_waypoints
        wp = torch.zeros((E, N, P, 2), dtype=torch.float64, device=dev)
        wp[:, :, 0] = pgoal
        nwp = torch.ones((E, N), dtype=torch.int32, device=dev)
        if "costmap" in a:                                                  # env.py:788-804
            mi = torch.arange(E, device=dev, dtype=torch.int32).repeat_interleave(N)
            pw, pn, _, _ = sim.plan(a["costmap"], start.reshape(-1, 2), pgoal.reshape(-1, 2), 2.0,
                                    max_wp=P, res_c=res * 5, origin=org, map_index=mi)
            found = (pn > 0).reshape(E, N)
            wp = torch.where(found[..., None, None], pw.reshape(E, N, P, 2), wp)
            nwp = torch.where(found, pn.reshape(E, N), nwp)
        a["ped_waypoints"] = wp.contiguous()
        a["ped_n_waypoints"] = nwp.contiguous()
        a["ped_wp_head"] = torch.zeros((E, N), dtype=torch.int32, device=dev)     # every route starts at its first waypoint
        a["ped_goal"] = pgoal.contiguous()          # a route stored cut is continued to this goal (navsim_replan)
        a["ped_cmd"] = torch.zeros((E, N, 2), dtype=torch.float64, device=dev)
    return a


def empty_world(cfg, device="cuda:0", plan_paths=False, rect_table=False):
    """Zero-initialised navsim_state arrays for cfg.n_envs arenas: the buffers NavSim.regenerate_all() fills on the
    device (maps, distance fields, spawn tables, robots, pedestrians, first observations -- navsim_regen with
    every arena marked finished).  No map is generated on the host."""
    import torch
    from . import sim
    dev = torch.device(device)
    E, N, K, H, W = cfg.n_envs, cfg.max_peds, max(cfg.n_spawn, 1), cfg.map_h, cfg.map_w
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    a = {}
    if cfg.field_format == abi.FIELD_F32:
        a["field"] = z((E, H, W), torch.float32)
    else:
        a["field"] = z(sim.load().navsim_field_bytes(E, H, W, cfg.field_format) // 2, torch.int16)
        if max(H, W) > 520:             # cells of such maps can be >= 256 cells from every obstacle (d2 >= 65535): the
            a["field_overflow"] = z((E, H, W), torch.float32)      # packed field escapes to this exact float plane
        if rect_table:
            a["rect_table"] = z((E, ((H + 7) // 8) * ((W + 7) // 8), 4), torch.int32)
            a["rect_table"][:, :, 0] = 0x7FFF                  # "no valid record" until the builder has run
            a["rect_index"] = torch.full((E, abi.rect_index_row_bytes(H, W)), 255, dtype=torch.uint8, device=dev)   # no index either
    if plan_paths:
        a["costmap"] = z((E, H // 5, W // 5), torch.uint8)
    a["spawn_pose"] = z((E, K, 3), torch.float64)
    a["spawn_goal"] = z((E, K, 2), torch.float64)
    a["robot_pose"] = z((E, 3), torch.float64)
    a["robot_goal"] = z((E, 2), torch.float64)
    a["prev_action"] = z((E, 2), torch.float64)
    a["prev_pose"] = z((E, 3), torch.float64)
    a["n_hist"] = z(E, torch.int32)
    a["episode"] = z(E, torch.int64)
    a["steps"] = z(E, torch.int64)
    a["scan_noise_std"] = z(E, torch.float32)
    if cfg.ped_model != abi.PED_NONE:
        a["n_peds"] = z(E, torch.int32)
        a["ped_pose"] = z((E, N, 3), torch.float64)
        a["ped_vel"] = z((E, N, 2), torch.float64)
        a["ped_prev_yaw"] = z((E, N), torch.float64)
        a["ped_dist"] = z((E, N, 3), torch.float64)
        a["ped_v_pref"] = z((E, N), torch.float64)
        a["ped_has_legs"] = z((E, N), torch.uint8)
        a["ped_waypoints"] = z((E, N, cfg.max_waypoints, 2), torch.float64)
        a["ped_n_waypoints"] = torch.ones((E, N), dtype=torch.int32, device=dev)
        a["ped_wp_head"] = z((E, N), torch.int32)
        a["ped_goal"] = z((E, N, 2), torch.float64)
        a["ped_cmd"] = z((E, N, 2), torch.float64)
    return a


def lidar_1081(cfg):
    """270 deg / 0.25 deg planar lidar of the BASELINE configs: beams at -135 .. +135 deg inclusive."""
    cfg.n_beams = 1081
    cfg.angle_min = -0.75 * np.pi
    cfg.angle_last = 0.75 * np.pi
    return cfg


def lidar_full_circle(cfg, n_beams):
    """n beams over 2*pi, np.linspace(-pi, pi - 2*pi/n, n) (BASELINE config 1 uses 64)."""
    cfg.n_beams = n_beams
    cfg.angle_min = -np.pi
    cfg.angle_last = np.pi - 2 * np.pi / n_beams
    return cfg
