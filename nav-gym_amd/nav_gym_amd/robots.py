"""Robot / pedestrian constants of the reference entity models."""
import numpy as np

# KetiRobot (keti_robot.py:12-48)
KETI = dict(
    footprint=[[0.3, 0.4], [-0.70, 0.4], [-0.70, -0.4], [0.3, -0.4]],
    threshold_footprint=[[0.6, 0.6], [-0.7, 0.6], [-0.7, -0.6], [0.6, -0.6]],
    discomfort_threshold_footprint=[[0.6 + 0.5, 0.6 + 0.5], [-0.7, 0.6 + 0.5], [-0.7, -0.6 - 0.5], [0.6 + 0.5, -0.6 - 0.5]],
    axle_offset=0.14474,
    angle_min=-3.141592, angle_max=3.141592, angle_increment=0.0122718463, range_max=25.0, n_angles=512,
    linvel_range=[0.0, 0.5], rotvel_range=[-0.64, 0.64],
)

# BUILD-DEFINED (not in the reference: third_party/husky_description only ships URDF geometry,
# husky.urdf.xacro:61-67: wheelbase 0.512, track 0.5708, wheel radius 0.1651).  A centre-axle
# unicycle with the Husky's published limits and a footprint of its 0.99 x 0.67 m chassis.
HUSKY = dict(
    footprint=[[0.495, 0.335], [-0.495, 0.335], [-0.495, -0.335], [0.495, -0.335]],
    threshold_footprint=[[0.75, 0.55], [-0.75, 0.55], [-0.75, -0.55], [0.75, -0.55]],
    discomfort_threshold_footprint=[[1.25, 1.05], [-0.75, 1.05], [-0.75, -1.05], [1.25, -1.05]],
    axle_offset=0.0,
    angle_min=-3.141592, angle_max=3.141592, angle_increment=0.0122718463, range_max=25.0, n_angles=512,
    linvel_range=[0.0, 1.0], rotvel_range=[-2.0, 2.0],
    wheel_radius=0.1651, wheel_track=0.5708,      # husky.urdf.xacro:67, 62
)

# Human (human.py:5-16)
HUMAN = dict(
    footprint=[[0.22, 0.19], [-0.22, 0.19], [-0.22, -0.19], [0.22, -0.19]],
    angle_min=-1.57079632679, angle_max=1.57079632679, angle_increment=0.00613592315, range_max=6.0, n_angles=512,
)

ROBOTS = {"keti": KETI, "husky": HUSKY}

# Husky wheel geometry (third_party/husky_description/urdf/husky.urdf.xacro:61-67)
HUSKY_TRACK = 0.5708
HUSKY_WHEEL_RADIUS = 0.1651


def husky_twist_from_wheels(omega_left, omega_right):
    """Skid-steer form of the Husky command (BUILD-DEFINED like the model itself): wheel speeds of
    the left / right side in rad/s -> the (v, omega) the step integrates,
    v = r (w_l + w_r) / 2, omega = r (w_r - w_l) / track.  Arrays or torch tensors of any shape; returns the same kind,
    last axis (v, omega)."""
    v = HUSKY_WHEEL_RADIUS * (omega_left + omega_right) * 0.5
    w = HUSKY_WHEEL_RADIUS * (omega_right - omega_left) / HUSKY_TRACK
    if hasattr(v, "stack") or type(v).__module__.startswith("torch"):
        import torch
        return torch.stack([v, w], dim=-1)
    return np.stack([np.asarray(v, dtype=np.float64), np.asarray(w, dtype=np.float64)], axis=-1)


def husky_wheels_from_twist(v, omega):
    """Inverse of husky_twist_from_wheels: (v, omega) -> (omega_left, omega_right) in rad/s."""
    wl = (v - 0.5 * HUSKY_TRACK * omega) / HUSKY_WHEEL_RADIUS
    wr = (v + 0.5 * HUSKY_TRACK * omega) / HUSKY_WHEEL_RADIUS
    return wl, wr


def wheels_from_twist(v, omega, radius, track):
    """(omega_left, omega_right) in rad/s of a skid-steer base for the twist (v, omega): the inverse of what the step
    computes on the device under NAVSIM_ACTION_WHEELS (include/navsim.h)."""
    return (v - 0.5 * track * omega) / radius, (v + 0.5 * track * omega) / radius


def footprint_array(robot, key):
    return np.asarray(ROBOTS[robot][key], dtype=np.float32)
