"""State of ONE arena of the batch in the layout RosEnv puts on the wire.

ros_env.py:65-185 fills a ResetMap request (nav_gym/srv/ResetMap.srv:1-6) and a StrictUpdate request
(nav_gym/srv/StrictUpdate.srv:1-9) from attributes of the wrapped env.  ROS is not part of this build;
these functions return the same fields as plain dicts / arrays so that a bridge process (or a debugging
view) can fill the messages without touching device memory itself.  Host side only.
"""
import numpy as np

from . import robots


def _quaternion_from_yaw(yaw):
    """tf.transformations.quaternion_from_euler(0, 0, yaw) -> (x, y, z, w)."""
    return (0.0, 0.0, float(np.sin(0.5 * yaw)), float(np.cos(0.5 * yaw)))


def _transform(footprint, x, y, yaw):
    """utils.transform_xys(translation(x, y), rotation(yaw), footprint) (utils.py:34-61)."""
    f = np.asarray(footprint, dtype=np.float64).reshape(-1, 2)
    c, s = np.cos(yaw), np.sin(yaw)
    return np.stack([c * f[:, 0] - s * f[:, 1] + x, s * f[:, 0] + c * f[:, 1] + y], axis=1)


def reset_map_fields(env, arena=0):
    """nav_msgs/OccupancyGrid of ResetMap.srv as ros_env.py:69-85 fills it (width <- height, as there)."""
    import torch
    cfg = env.cfg
    f = env.sim.t["field"]
    if "map_slot" in env.sim.t:                               # the arena's map may lie in another slot (pipelined reset path)
        arena = int(env.sim.t["map_slot"][arena].item())
    if f.dtype == torch.float32 and f.dim() == 3:
        occ = (f[arena] == 0)
    else:                                                     # packed field: zero squared distance = occupied
        from . import abi
        tpr = (cfg.map_w + 7) // 8
        raw = f.view(torch.int16).reshape(-1, (cfg.map_h + 7) // 8, tpr, 8, 8)[arena]
        occ = (raw.permute(0, 2, 1, 3).reshape(-1, tpr * 8)[: cfg.map_h, : cfg.map_w] == 0)
    data = (occ.to(torch.int8) * 100).cpu().numpy()
    return {"data": data, "resolution": cfg.resolution, "width": cfg.map_h, "height": cfg.map_w,
            "origin_position": (cfg.origin_x, cfg.origin_y, 0.0), "origin_orientation": (0.0, 0.0, 0.0, 1.0)}


def strict_update_fields(env, arena=0):
    """The six request fields of StrictUpdate.srv as ros_env.py:87-185 fills them."""
    t = env.sim.t
    spec = robots.ROBOTS[env.robot_type]
    x, y, yaw = [float(v) for v in t["robot_pose"][arena].cpu().numpy()]
    B = env.cfg.n_beams
    obs = env.sim.obs[arena].cpu().numpy()
    S = env.cfg.n_scan_stack
    out = {
        "pose": {"frame_id": "map", "position": (x, y, 0.0), "orientation": _quaternion_from_yaw(yaw)},
        "footprint": _transform(spec["footprint"], x, y, yaw),
        "threshold_footprint": _transform(spec["threshold_footprint"], x, y, yaw),
        "discomfort_threshold_footprint": _transform(spec["discomfort_threshold_footprint"], x, y, yaw),
        "scan": {"frame_id": "laser_link", "angle_min": spec["angle_min"], "angle_max": spec["angle_max"],
                 "angle_increment": spec["angle_increment"], "range_max": spec["range_max"],
                 "ranges": obs[(S - 1) * B: S * B].astype(np.float64)},      # observation_to_dict(...)['scan']
        "humans": [],
    }
    n = int(t["n_peds"][arena]) if "n_peds" in t and env.cfg.ped_model != 0 else 0
    if n:
        pose = t["ped_pose"][arena, :n].cpu().numpy()
        vel = t["ped_vel"][arena, :n].cpu().numpy()
        for i in range(n):
            out["humans"].append({"track_id": i, "detection_id": i, "position": (float(pose[i, 0]), float(pose[i, 1])),
                                  "orientation": _quaternion_from_yaw(float(pose[i, 2])),
                                  "linear": (float(vel[i, 0]), float(vel[i, 1]))})
    return out
