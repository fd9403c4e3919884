"""render() for ONE arena of the batch (nav_gym_env/env.py:833-1050): a NumPy rasteriser, host side only.

The reference draws with OpenCV (not installed here) and shows a window; this restates the same picture --
occupancy map (free white, obstacles black), goal (blue square, 1 m), pedestrians' local goals (yellow, 0.2 m),
pedestrian / robot heading arrows and footprints, the robot's threshold and discomfort rectangles, the lidar
returns below range_max (green discs, 0.2 m), flipped to world orientation and resized to 800 x 800 -- as a
float32 BGR array in [0, 1] like the reference's `img`.  The debug text overlay (env.py:1035-1046) is drawn with a
built-in 5 x 7 stroke font at the reference's positions, colour and thickness (OpenCV's Hershey glyphs are not available).
Drawing is in cell space with the reference's xy_to_ij (truncation, clipped to the map).
"""
import numpy as np

HEIGHT, WIDTH = 800, 800                       # env.py:834


def xy_to_ij(xy, map_info, clip=True):
    """env.py:1255-1258 / batch_xy_to_ij env.py:1228-1253 (float32 division, clip, truncate)."""
    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    ij = np.zeros_like(xy, dtype=np.float32)
    ij[:, 0] = (xy[:, 0] - map_info["origin"][0]) / map_info["resolution"]
    ij[:, 1] = (xy[:, 1] - map_info["origin"][1]) / map_info["resolution"]
    if clip:
        ij[:, 0] = np.clip(ij[:, 0], 0, map_info["height"] - 1)
        ij[:, 1] = np.clip(ij[:, 1], 0, map_info["width"] - 1)
    return ij.astype(np.int64)


def _rect(img, i, j, r, color):
    H, W = img.shape[:2]                       # OpenCV points are (column = i, row = j)
    img[max(j - r, 0):min(j + r + 1, H), max(i - r, 0):min(i + r + 1, W)] = color


def _disc(img, i, j, r, color):
    H, W = img.shape[:2]
    y0, y1, x0, x1 = max(j - r, 0), min(j + r + 1, H), max(i - r, 0), min(i + r + 1, W)
    if y0 >= y1 or x0 >= x1:
        return
    yy, xx = np.mgrid[y0:y1, x0:x1]
    img[y0:y1, x0:x1][(yy - j) ** 2 + (xx - i) ** 2 <= r * r] = color


def _line(img, p, q, color, thickness=1):
    n = int(max(abs(q[0] - p[0]), abs(q[1] - p[1]))) + 1
    xs = np.rint(np.linspace(p[0], q[0], n)).astype(int)
    ys = np.rint(np.linspace(p[1], q[1], n)).astype(int)
    h = max(int(thickness) // 2, 0)
    for x, y in zip(xs, ys):
        _rect(img, x, y, h, color)


def _arrow(img, p, q, color, thickness):
    """cv2.arrowedLine: shaft + two head strokes of 10 % of the length at +-45 degrees."""
    _line(img, p, q, color, thickness)
    d = np.array([q[0] - p[0], q[1] - p[1]], dtype=np.float64)
    L = np.hypot(*d)
    if L < 1e-9:
        return
    tip = 0.1 * L
    ang = np.arctan2(p[1] - q[1], p[0] - q[0])
    for s in (np.pi / 4, -np.pi / 4):
        _line(img, q, (int(round(q[0] + tip * np.cos(ang + s))), int(round(q[1] + tip * np.sin(ang + s)))), color, thickness)


def _transform(fp, px, py, theta):
    fp = np.asarray(fp, dtype=np.float64)
    c, s = np.cos(theta), np.sin(theta)
    return np.stack([c * fp[:, 0] - s * fp[:, 1] + px, s * fp[:, 0] + c * fp[:, 1] + py], axis=1)


def _polygon(img, fp, px, py, theta, map_info, color=(0, 0, 0)):
    pts = _transform(np.concatenate([fp, fp[:1]]), px, py, theta)
    ij = xy_to_ij(pts, map_info)
    for a, b in zip(ij[:-1], ij[1:]):
        _line(img, tuple(a), tuple(b), color, 1)


def render_arena(map_info, robot, humans, scan, scan_yaw, lidar):
    """map_info: {'data' int8 [H,W] in {0,100}, origin, resolution, width, height}.
    robot: dict px, py, theta, gx, gy, footprint, threshold_footprint, discomfort_threshold_footprint.
    humans: list of dicts px, py, theta, gx, gy, footprint.  scan: float [B] (latest scan of prev_obs), scan_yaw its
    yaw (env.py:966-975); lidar: dict angle_min, angle_last, range_max.  -> float32 [800, 800, 3] BGR in [0, 1]."""
    data = np.asarray(map_info["data"])
    img = np.where(data == 0, 1.0, 0.0).astype(np.float32)               # env.py:838-840
    img = np.stack([img, img, img], axis=2)
    one_m = int(xy_to_ij([1, 0], map_info)[0, 0])
    r02 = int(xy_to_ij([0.2, 0], map_info)[0, 0])
    gi, gj = xy_to_ij([robot["gx"], robot["gy"]], map_info)[0]
    _rect(img, gi, gj, one_m, (0, 0, 1))                                   # goal (env.py:842-851)
    for h in humans:                                                       # local goals (env.py:853-863)
        i, j = xy_to_ij([h["gx"], h["gy"]], map_info)[0]
        _rect(img, i, j, r02, (1, 1, 0))
    for h in humans:                                                       # env.py:865-896
        i, j = xy_to_ij([h["px"], h["py"]], map_info)[0]
        di, dj = xy_to_ij([0.6 * np.cos(h["theta"]), 0.6 * np.sin(h["theta"])], map_info, clip=False)[0]
        _arrow(img, (i, j), (i + di, j + dj), (0, 0, 0), r02)
        _polygon(img, np.asarray(h["footprint"], dtype=np.float64), h["px"], h["py"], h["theta"], map_info)
    i, j = xy_to_ij([robot["px"], robot["py"]], map_info)[0]               # env.py:898-909
    di, dj = xy_to_ij([0.8 * np.cos(robot["theta"]), 0.8 * np.sin(robot["theta"])], map_info, clip=False)[0]
    _arrow(img, (i, j), (i + di, j + dj), (0, 0, 0), r02)
    for key in ("footprint", "threshold_footprint", "discomfort_threshold_footprint"):        # env.py:911-963
        _polygon(img, np.asarray(robot[key], dtype=np.float64), robot["px"], robot["py"], robot["theta"], map_info)
    scan = np.asarray(scan, dtype=np.float64)                              # env.py:965-992
    angles = np.linspace(lidar["angle_min"], lidar["angle_last"], len(scan)) + scan_yaw
    pts = np.stack([robot["px"] + scan * np.cos(angles), robot["py"] + scan * np.sin(angles)], axis=1)
    for (li, lj) in xy_to_ij(pts, map_info)[scan != lidar["range_max"]]:
        _disc(img, int(li), int(lj), r02, (0.0, 1.0, 0.0))
    img = np.flipud(img)                                                   # env.py:1026
    H, W = img.shape[:2]                                                   # cv2.resize -> nearest-neighbour here
    yy = np.minimum((np.arange(HEIGHT) * (H / float(HEIGHT))).astype(int), H - 1)
    xx = np.minimum((np.arange(WIDTH) * (W / float(WIDTH))).astype(int), W - 1)
    return np.ascontiguousarray(img[yy][:, xx])


# ---- the debug text of render() (env.py:182-217, 1035-1046) ------------------------------------------------------------
def obs_text(steps, prev_pose, pose, vel, yaw, goal):
    """_make_render_obs_txt (env.py:182-201): the six lines about the latest observation."""
    return ('t: {}\n'.format(steps)
            + 'prev_pose: ({:.2f} {:.2f})\n'.format(prev_pose[0], prev_pose[1])
            + 'pose: ({:.2f} {:.2f})\n'.format(pose[0], pose[1])
            + 'vel: ({:.2f} {:.2f})\n'.format(vel[0], vel[1])
            + 'yaw: {:.2f}\n'.format(yaw)
            + 'goal: ({:.2f} {:.2f})'.format(goal[0], goal[1]))


REWARD_TERMS = ("reward_success", "reward_crash", "reward_progress", "reward_forward", "reward_rotation", "reward_discomfort")


def reward_text(terms):
    """_make_render_reward_txt (env.py:203-217): one line per term of compute_rewards."""
    return "\n".join('{}: {:.5f}'.format(k, terms[k]) for k in REWARD_TERMS)


def reward_terms(scan, prev_pose, pose, vel, goal, scan_threshold, scan_discomfort_threshold, factors, distance_threshold):
    """The six terms of compute_rewards (env.py:513-573) for ONE observation, NumPy like the reference: scan [B] (the
    latest scan), poses / vel / goal float64; factors: dict reward_scale, reward_*_factor.  Their sum is the step's reward."""
    scan = np.asarray(scan, dtype=np.float64)             # the reference's observation is float64; the thresholds float32
    thr = np.asarray(scan_threshold, dtype=np.float32); dthr = np.asarray(scan_discomfort_threshold, dtype=np.float32)
    goal = np.asarray(goal, dtype=np.float64)
    distance = np.linalg.norm(goal - np.asarray(pose, dtype=np.float64)[:2])
    prev_distance = np.linalg.norm(goal - np.asarray(prev_pose, dtype=np.float64)[:2])
    success = bool(distance < distance_threshold)
    crash = bool(np.any(scan - thr < 0))
    discomfort = bool(np.any(scan - dthr < 0)) and not crash
    sc = factors["reward_scale"]
    t = dict.fromkeys(REWARD_TERMS, 0.0)
    if success:
        t["reward_success"] = 1.0 * factors["reward_success_factor"] * sc
    if crash:
        t["reward_crash"] = -1.0 * factors["reward_crash_factor"] * sc
    t["reward_progress"] = float((prev_distance - distance) * factors["reward_progress_factor"] * sc)
    t["reward_forward"] = float(vel[0] * factors["reward_forward_factor"] * sc)
    t["reward_rotation"] = float(-1.0 * (vel[1] ** 2) * factors["reward_rotation_factor"] * sc)
    if discomfort:
        ratio = np.min(np.divide(scan - thr, dthr - thr + 1e-6))
        t["reward_discomfort"] = float(-(1.0 - ratio) * factors["reward_discomfort_factor"] * sc)
    return t


# 5 x 7 glyphs, one string of 7 rows per character ('#' = stroke); unknown characters draw as a box
_GLYPHS = {
    "0": ".###.|#...#|#..##|#.#.#|##..#|#...#|.###.", "1": "..#..|.##..|..#..|..#..|..#..|..#..|.###.",
    "2": ".###.|#...#|....#|...#.|..#..|.#...|#####", "3": ".###.|#...#|....#|..##.|....#|#...#|.###.",
    "4": "...#.|..##.|.#.#.|#..#.|#####|...#.|...#.", "5": "#####|#....|####.|....#|....#|#...#|.###.",
    "6": "..##.|.#...|#....|####.|#...#|#...#|.###.", "7": "#####|....#|...#.|..#..|.#...|.#...|.#...",
    "8": ".###.|#...#|#...#|.###.|#...#|#...#|.###.", "9": ".###.|#...#|#...#|.####|....#|...#.|.##..",
    "a": ".....|.....|.###.|....#|.####|#...#|.####", "b": "#....|#....|####.|#...#|#...#|#...#|####.",
    "c": ".....|.....|.###.|#....|#....|#...#|.###.", "d": "....#|....#|.####|#...#|#...#|#...#|.####",
    "e": ".....|.....|.###.|#...#|#####|#....|.###.", "f": "..##.|.#..#|.#...|###..|.#...|.#...|.#...",
    "g": ".....|.####|#...#|#...#|.####|....#|.###.", "h": "#....|#....|#.##.|##..#|#...#|#...#|#...#",
    "i": "..#..|.....|.##..|..#..|..#..|..#..|.###.", "l": ".##..|..#..|..#..|..#..|..#..|..#..|.###.",
    "m": ".....|.....|##.#.|#.#.#|#.#.#|#...#|#...#", "n": ".....|.....|#.##.|##..#|#...#|#...#|#...#",
    "o": ".....|.....|.###.|#...#|#...#|#...#|.###.", "p": ".....|####.|#...#|#...#|####.|#....|#....",
    "r": ".....|.....|#.##.|##..#|#....|#....|#....", "s": ".....|.....|.####|#....|.###.|....#|####.",
    "t": ".#...|.#...|###..|.#...|.#...|.#..#|..##.", "u": ".....|.....|#...#|#...#|#...#|#..##|.##.#",
    "v": ".....|.....|#...#|#...#|#...#|.#.#.|..#..", "w": ".....|.....|#...#|#...#|#.#.#|#.#.#|.#.#.",
    "y": ".....|#...#|#...#|#...#|.####|....#|.###.", "_": ".....|.....|.....|.....|.....|.....|#####",
    ":": ".....|..#..|.....|.....|.....|..#..|.....", ".": ".....|.....|.....|.....|.....|.##..|.##..",
    "-": ".....|.....|.....|#####|.....|.....|.....", "(": "...#.|..#..|.#...|.#...|.#...|..#..|...#.",
    ")": ".#...|..#..|...#.|...#.|...#.|..#..|.#...", " ": ".....|.....|.....|.....|.....|.....|.....",
}


def put_text(img, txt, org, scale=2, color=(0, 0, 1)):
    """Draws txt with its BASELINE's left end at org = (x, y), like cv2.putText's origin (env.py:1037-1046); a glyph cell is
    6 x 7 stroke units of `scale` pixels (2 -> 14 px tall letters, the height of FONT_HERSHEY_SIMPLEX at fontScale 0.7)."""
    H, W = img.shape[:2]
    x0, ytop = int(org[0]), int(org[1]) - 7 * scale
    for ch in txt:
        rows = _GLYPHS.get(ch, _GLYPHS.get(ch.lower(), "#####|#...#|#...#|#...#|#...#|#...#|#####")).split("|")
        for ry, row in enumerate(rows):
            for rx, c in enumerate(row):
                if c == "#":
                    ya, xa = ytop + ry * scale, x0 + rx * scale
                    if ya < H and xa < W and ya + scale > 0 and xa + scale > 0:
                        img[max(ya, 0):ya + scale, max(xa, 0):xa + scale] = color
        x0 += 6 * scale
    return img


def overlay_text(img, obs_txt, reward_txt):
    """env.py:1035-1046: every line of the two texts at (50, 50 + 50 i), red (BGR (0, 0, 1))."""
    for i, txt in enumerate(obs_txt.split("\n") + reward_txt.split("\n")):
        put_text(img, txt, (50, 50 + i * 50))
    return img
