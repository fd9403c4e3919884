"""GPU: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Bar (BASELINE.json north_star): collision / done flags bit-exact, lidar ranges and poses within
1e-5.  Because oracle and kernels implement the same operation-by-operation specification
(DESIGN.md sections 3-4), the tests demand BIT-EXACT agreement everywhere and fall back to the
1e-5 tolerance only in the error message."""
import ctypes as C
import os

import numpy as np
import pytest

import ref
from helpers import load_trace, trace_setup
from nav_gym_amd import abi

pytestmark = pytest.mark.gpu

TOL = 1e-5   # north_star tolerance on ranges / pose (we assert equality, which is stricter)


@pytest.fixture(scope="module")
def gpu():
    import torch
    from nav_gym_amd import lib, sim, world
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    lib.load()
    return type("G", (), dict(torch=torch, lib=lib, sim=sim, world=world, dev=torch.device("cuda:0")))


def _t(gpu, a, dtype=None):
    t = gpu.torch.from_numpy(np.ascontiguousarray(a)).to(gpu.dev)
    return t if dtype is None else t.to(dtype)


def _eq(a, b, what):
    a = np.asarray(a); b = np.asarray(b)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        diff = np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))
        raise AssertionError("%s: %d mismatches, max |diff| %.3e (north_star tol %g), first at %s"
                             % (what, len(bad), diff, TOL, bad[0]))


def test_device_math_bit_exact(gpu):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-40, 40, 200000), [0.0, np.pi, -np.pi, 2 * np.pi, 1e-12, -3.141592]])
    for fn in (0, 1, 4, 5):
        _eq(gpu.sim.debug_math(fn, _t(gpu, x)).cpu().numpy(), ref.math_fn(fn, x), "math fn %d" % fn)
    y, xx = rng.normal(size=200000), rng.normal(size=200000)
    y[:4] = [0, 1, -1, 0]; xx[:4] = [1, 0, 0, -1]
    _eq(gpu.sim.debug_math(2, _t(gpu, y), _t(gpu, xx)).cpu().numpy(), ref.math_fn(2, y, xx), "atan2")
    e = -rng.uniform(0, 720, 200000)
    _eq(gpu.sim.debug_math(3, _t(gpu, e)).cpu().numpy(), ref.math_fn(3, e), "exp")


def test_social_force_pair_term_is_antisymmetric(gpu):
    """The fused step keeps HALF of an arena's social-force pair table: the term on pedestrian j from i is stored once and
    subtracted for the second reader, which is exact only while sfm_pair(i, j) == -sfm_pair(j, i) bit for bit (round-3
    advisor).  Debug function 14 evaluates both orders on 2e6 random agent pairs, degenerate ones included."""
    x = gpu.torch.arange(2_000_000, dtype=gpu.torch.float64, device=gpu.dev)
    bad = gpu.sim.debug_math(14, x)
    assert int(bad.sum()) == 0


def test_packed_field_sqrt_is_correctly_rounded(gpu):
    """nv::sqrt_small_int (v_rsq_f32 + one exact-residual step, 5 instructions) against IEEE sqrt for EVERY integer the
    packed field or a rect record can hold: all of [0, 2^22) -- the packed field's d2 < 65536, a record's d2 < 2^21 on
    maps up to 1024 cells per side."""
    n = 1 << 22
    x = gpu.torch.arange(n, dtype=gpu.torch.float64, device=gpu.dev)
    got = gpu.sim.debug_math(6, x).cpu().numpy()
    exp = np.sqrt(np.arange(n, dtype=np.float32)).astype(np.float64)
    _eq(got, exp, "exact integer sqrt")
    # the march step fl32(fl64(d) * 0.999) and its float32-only evaluation (kernels_field.hpp march_step) agree on
    # every d = sqrtf(n), n < 2^22, and both equal NumPy's float64 computation
    a, b = gpu.sim.debug_math(11, x).cpu().numpy(), gpu.sim.debug_math(12, x).cpu().numpy()
    ref_step = np.maximum((exp * 0.999).astype(np.float32), np.float32(1.0)).astype(np.float64)
    _eq(a, ref_step, "march step, float64 form")
    _eq(b, ref_step, "march step, float32-only form")


def test_beam_table_direction_is_the_full_evaluation(gpu):
    """The scan takes a beam's direction from the beam table by one angle addition and accepts it only when its
    float32 rounding is provably that of beam_dir() (navsim_device.hpp round_if_safe); otherwise it evaluates
    beam_dir().  Debug function 13 runs both on the device for (robot heading, robot-frame beam angle) pairs:
    0 = accepted and bit-identical, 1 = not accepted, 2 = accepted but different.  4e7 random pairs plus the
    places where the acceptance test has something to get wrong: directions along the axes (a component near 0),
    at 60 / 120 degrees (a component at a power of two), headings far outside [-pi, pi]."""
    torch = gpu.torch
    g = torch.Generator(device=gpu.dev); g.manual_seed(5)
    n = 10_000_000
    step = 1.5 * np.pi / 1080
    for rep in range(4):
        lth = (torch.rand(n, generator=g, device=gpu.dev, dtype=torch.float64) * 2 - 1) * (np.pi if rep < 3 else 50.0)
        k = torch.randint(0, 1081, (n,), generator=g, device=gpu.dev).to(torch.float64)
        lin = k * step - 0.75 * np.pi
        code = gpu.sim.debug_math(13, lth, lin)
        assert int((code == 2).sum()) == 0, "table direction differs from beam_dir()"
        assert int((code == 1).sum()) < n // 20000, "fallback rate %g" % (float((code == 1).sum()) / n)
    # special directions: heading + beam angle on the axes and at +-60 / 120 degrees, and float32 neighbours of them
    base = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 1 / 3, 2 / 3, 4 / 3, 5 / 3, -0.5, -1.0, -1 / 3, -2 / 3]) * np.pi
    lin = np.repeat(np.arange(0, 1081, 7) * step - 0.75 * np.pi, len(base) * 9)
    tgt = np.tile(np.repeat(base, 9), len(lin) // (len(base) * 9))
    lth32 = (tgt - lin).astype(np.float32)
    for j in range(9):                                                    # the 9 nearest float32 headings
        sl = slice(j, None, 9)
        for _ in range(abs(j - 4)):
            lth32[sl] = np.nextafter(lth32[sl], np.float32(np.inf if j > 4 else -np.inf))
    code = gpu.sim.debug_math(13, _t(gpu, lth32.astype(np.float64)), _t(gpu, lin)).cpu().numpy()
    assert (code != 2).all(), "special directions: %d wrong" % int((code == 2).sum())
    assert (code == 0).mean() > 0.5           # the fast path is not simply declining everything here


@pytest.mark.parametrize("size,n", [(100, 3), (400, 2), (500, 2), (1000, 1)])
def test_build_dt(gpu, size, n):
    occ = gpu.world.make_maps(n, size, 77 + size, indoor_ratio=0.5 if size == 1000 else 0.0)
    got = gpu.sim.build_dt(_t(gpu, occ)).cpu().numpy()
    _eq(got, ref.build_dt(occ), "distance field %d" % size)


def test_build_dt_ragged_and_empty(gpu):
    rng = np.random.default_rng(5)
    occ = (rng.random((4, 37, 53)) < 0.02).astype(np.uint8)
    occ[1] = 0; occ[1, 0, 0] = 1
    occ[2] = 0                                      # no obstacle at all
    occ[3] = 1                                      # everything occupied
    got = gpu.sim.build_dt(_t(gpu, occ)).cpu().numpy()
    _eq(got, ref.build_dt(occ), "ragged distance field")


def _queries(rng, occ, n):
    E = occ.shape[0]
    q = np.zeros((E, n, 3), np.float32)
    for e in range(E):
        free = np.argwhere(occ[e] == 0)
        pick = free[rng.integers(0, len(free), n)]
        q[e, :, 0] = pick[:, 1]; q[e, :, 1] = pick[:, 0]
    q[:, :, 2] = rng.uniform(-np.pi, 3 * np.pi, (E, n))
    return q


@pytest.mark.parametrize("size", [100, 400, 500, 1000])
def test_device_xy_to_ij_reference_goldens(gpu, golden_dir, size):
    """Row a15: the device's batch_xy_to_ij (nv::xy_to_ij / nv::xy_to_ij_f32, the functions the scan and the
    social force call) on the vectors recorded from the reference's own batch_xy_to_ij (env.py:1228-1253),
    float64 inputs and the float32 inputs of the lidar origin (env.py:386, 419): exact integers."""
    units = np.load(os.path.join(golden_dir, "golden_units.npz"))
    cfg = gpu.lib.default_config(map_h=size, map_w=size)
    got = gpu.sim.debug_xy_to_ij(cfg, _t(gpu, units["xy_%d" % size]), False).cpu().numpy()
    _eq(got, units["ij_%d" % size].astype(np.int32), "xy_to_ij float64 %d" % size)
    got = gpu.sim.debug_xy_to_ij(cfg, _t(gpu, units["xyf32_%d" % size].astype(np.float64)), True).cpu().numpy()
    _eq(got, units["ijf32_%d" % size].astype(np.int32), "xy_to_ij float32 %d" % size)
    _eq(got, ref.xy_to_ij_f32(units["xyf32_%d" % size], (0.0, 0.0), 0.05, size, size).astype(np.int32), "oracle")


@pytest.mark.parametrize("rule", abi.MARCH_RULES)
@pytest.mark.parametrize("size", [100, 500])
def test_cast_static(gpu, size, rule):
    rng = np.random.default_rng(size)
    occ = gpu.world.make_maps(3, size, 5)
    field = ref.build_dt(occ)
    q = _queries(rng, occ, 4096)
    q[0, :8, 0:2] = [[-3, 5], [size + 2, 5], [5, -1], [5, size], [0, 0], [size - 1, size - 1], [2.5, 2.5], [7, 7]]
    got = gpu.sim.cast_static(_t(gpu, field), _t(gpu, q), float(size * size), rule).cpu().numpy()
    _eq(got, ref.cast_static(field, q, float(size * size), rule), "cast_static %d rule %d" % (size, rule))
    assert gpu.sim.cast_static(_t(gpu, field), _t(gpu, q[:, :0]), 1.0).shape == (3, 0)


def test_march_rule_switch_full_step(gpu):
    """cfg.march_rule is the ONE switch between the candidate roundings of range_libc's march
    (include/navsim.h NAVSIM_MARCH_*): under each of them the fused step, the pedestrian scans and the
    mirror primitive equal the oracle bit for bit, and the rules do differ on some beams (so the
    switch is live on the device)."""
    E, size = 32, 240
    obs = {}
    for rule in abi.MARCH_RULES:
        cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=6, ped_model=abi.PED_SFM, n_spawn=8,
                                     auto_reset=1, seed=99, field_format=abi.FIELD_U16T, march_rule=rule)
        gpu.world.lidar_1081(cfg)
        occ = gpu.world.make_maps(E, size, 99)
        rows = []
        for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=5, steps=25, seed=3):
            _eq(go, ro, "obs at step %d (rule %d)" % (t, rule))
            for k in rout:
                _eq(gout[k], rout[k], "%s at step %d (rule %d)" % (k, t, rule))
            rows.append(go.copy())
        _eq(g.ped_scans().cpu().numpy(), r.ped_scans(), "pedestrian scans (rule %d)" % rule)
        obs[rule] = rows
    # liveness of the switch on the device: rays on which the two rules are KNOWN to differ (about 3 in 10^6 do;
    # tests/test_oracle_crosscheck.py wrote the fixture from 10^7) give the recorded answer under each rule
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "march_rule_cases.npz"))
    for m in range(int(d["n_maps"])):
        H, W = [int(x) for x in d["shape_%d" % m]]
        occ = np.unpackbits(d["occ_%d" % m])[: H * W].reshape(1, H, W)
        f = gpu.sim.build_dt(_t(gpu, occ))
        q = _t(gpu, d["q_%d" % m][None])
        _eq(gpu.sim.cast_static(f, q, float(H * W), abi.MARCH_F64).cpu().numpy()[0], d["r64_%d" % m], "rule F64 map %d" % m)
        _eq(gpu.sim.cast_static(f, q, float(H * W), abi.MARCH_F32).cpu().numpy()[0], d["r32_%d" % m], "rule F32 map %d" % m)
        _eq(gpu.sim.cast_static(f, q, float(H * W), abi.MARCH_F32_FMA).cpu().numpy()[0], d["r32fma_%d" % m], "rule F32_FMA map %d" % m)
        assert not np.array_equal(d["r64_%d" % m], d["r32_%d" % m])
        assert not np.array_equal(d["r32_%d" % m], d["r32fma_%d" % m])


def test_render_polys_and_legs(gpu):
    rng = np.random.default_rng(9)
    E, B, V, A = 5, 777, 24, 6
    ranges = rng.uniform(1, 25, (E, B)).astype(np.float32)
    angles = np.linspace(-np.pi, np.pi, B)[None] + rng.uniform(0, 6.28, (E, 1))
    origin = rng.uniform(4, 6, (E, 2)).astype(np.float32)
    verts = np.zeros((E, V, 3), np.float32)
    n_verts = np.array([24, 20, 0, 5, 4], np.int32)
    for e in range(E):
        for c in range(V // 4):
            cx, cy = origin[e] + rng.uniform(-4, 4, 2)
            th = rng.uniform(0, 6.28)
            fp = np.array([[0.22, 0.19], [-0.22, 0.19], [-0.22, -0.19], [0.22, -0.19]])
            R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            verts[e, 4 * c:4 * c + 4, 0] = c
            verts[e, 4 * c:4 * c + 4, 1:] = fp @ R.T + [cx, cy]
    verts[3, :5, 0] = 0                              # a closed 5-vertex contour (env.py:411)
    verts[3, 4, 1:] = verts[3, 0, 1:]
    got = gpu.sim.render_polys(_t(gpu, ranges), _t(gpu, angles), _t(gpu, verts), _t(gpu, n_verts), _t(gpu, origin))
    exp = ref.render_polys(ranges, angles, verts, n_verts, origin)
    _eq(got.cpu().numpy(), exp, "render_polys")
    assert (exp < ranges).any()
    agents = np.zeros((E, A, 8), np.float32)
    agents[:, :, 0:2] = origin[:, None, :] + rng.uniform(-3, 3, (E, A, 2))
    agents[:, :, 2] = rng.uniform(0, 6.28, (E, A))
    agents[:, :, 3:6] = rng.uniform(-2, 2, (E, A, 3))
    n_agents = np.array([6, 0, 3, 1, 6], np.int32)
    got = gpu.sim.render_legs(_t(gpu, ranges), _t(gpu, angles), _t(gpu, agents), _t(gpu, n_agents), _t(gpu, origin))
    exp = ref.render_legs(ranges, angles, agents, n_agents, origin)
    _eq(got.cpu().numpy(), exp, "render_legs")
    assert (exp < ranges).any()


def test_integrate_and_thresholds(gpu, golden_dir):
    import os
    u = np.load(os.path.join(golden_dir, "golden_units.npz"))
    inp = u["set_vel_in"]
    m = inp[:, 5] == 0.2
    for off, key in ((0.0, "human_set_vel_out"), (0.14474, "keti_set_vel_out")):
        pose = _t(gpu, inp[m, 0:3].copy()); vel = gpu.torch.zeros((int(m.sum()), 2), dtype=gpu.torch.float64, device=gpu.dev)
        gpu.sim.integrate(pose, _t(gpu, inp[m, 3:5].copy()), 0.2, off, vel)
        exp_pose, exp_vel = ref.integrate(inp[m, 0:3], inp[m, 3:5], 0.2, off)
        _eq(pose.cpu().numpy(), exp_pose, "set_vel pose"); _eq(vel.cpu().numpy(), exp_vel, "set_vel vel")
        np.testing.assert_allclose(pose.cpu().numpy(), u[key][m][:, 0:3], rtol=0, atol=1e-12)   # vs the reference
    for B in (512, 1081, 64):
        cfg = gpu.lib.default_config()
        if B == 1081: gpu.world.lidar_1081(cfg)
        if B == 64: gpu.world.lidar_full_circle(cfg, 64)
        for key in ("keti_threshold_footprint", "keti_discomfort_footprint"):
            got = gpu.sim.scan_threshold(cfg, _t(gpu, u[key].astype(np.float32))).cpu().numpy()
            _eq(got, ref.scan_threshold(cfg, u[key]), "scan_threshold B=%d" % B)


@pytest.mark.parametrize("S,B", [(1, 128), (3, 64)])
@pytest.mark.parametrize("f64", [True, False])
def test_reward_done(gpu, golden_dir, S, B, f64):
    import os
    u = np.load(os.path.join(golden_dir, "golden_units.npz"))
    tag = "rd_S%d_B%d_" % (S, B)
    obs = np.concatenate([u[tag + "scans"].astype(np.float64), u[tag + "tail"]], axis=1)
    goals = u[tag + "goals"]
    if not f64:
        obs = obs.astype(np.float32); goals = goals.astype(np.float32)
    cfg = gpu.lib.default_config(n_beams=B, n_scan_stack=S)
    got = gpu.sim.reward_done(cfg, _t(gpu, obs), _t(gpu, goals), _t(gpu, u[tag + "thr"]), _t(gpu, u[tag + "dthr"]))
    exp = ref.reward_done(cfg, obs, goals, u[tag + "thr"], u[tag + "dthr"])
    for k in exp:
        _eq(got[k].cpu().numpy(), exp[k], "reward_done " + k)
    if f64:                                          # and against the reference's own outputs
        np.testing.assert_allclose(got["reward"].cpu().numpy(), u[tag + "reward"], rtol=0, atol=1e-9)
        assert np.array_equal(got["done"].cpu().numpy().astype(bool), u[tag + "done"].astype(bool))
        assert np.array_equal(got["is_crash"].cpu().numpy(), u[tag + "is_crash"])
        assert np.array_equal(got["is_success"].cpu().numpy(), u[tag + "is_success"])


@pytest.mark.parametrize("name", ["random_S1", "peds_S1", "crash_S3", "success_S2", "corridor_S1"])
def test_step_golden_traces(gpu, name):
    """The fused HIP step reproduces the reference's own reset()/step() traces (corridor_S1: on the reference's own
    1000 x 1000 corridor map, recorded in round 4)."""
    tr = load_trace(name)
    cfg, arrays, occ = trace_setup(tr, gpu.lib.default_config, lambda o: gpu.sim.build_dt(_t(gpu, o)).cpu().numpy())
    _eq(arrays["field"], ref.build_dt(occ[None]), "field")
    sim = gpu.sim.NavSim(cfg, arrays)
    S, B = int(tr["S"]), int(tr["B"])
    first = sim.reset_obs().cpu().numpy()
    _eq(first[0, : S * B], tr["first_obs"][: S * B].astype(np.float32), "first scan")
    np.testing.assert_allclose(first[0, S * B:], tr["first_obs"][S * B:], rtol=0, atol=2e-6)
    for t in range(tr["actions"].shape[0]):
        sim.set_ped_cmd(tr["ped_cmd"][t][None])
        obs, out = sim.step(tr["actions"][t][None])
        obs = obs.cpu().numpy(); o = {k: v.cpu().numpy() for k, v in out.items()}
        assert o["done"][0] == tr["done"][t] and o["is_crash"][0] == tr["is_crash"][t] \
            and o["is_success"][0] == tr["is_success"][t], t
        assert abs(o["reward"][0] - tr["reward"][t]) < 1e-9, t
        _eq(obs[0, : S * B], tr["obs_scan"][t], "scan stack at step %d" % t)
        np.testing.assert_allclose(obs[0, S * B:], tr["obs_tail"][t], rtol=0, atol=2e-6)
        if "ped_scan_steps" in tr and t in tr["ped_scan_steps"]:          # env.py:685-693 on the GPU
            idx = int(np.where(tr["ped_scan_steps"] == t)[0][0])
            _eq(sim.ped_scans().cpu().numpy()[0, : tr["ped_scan"].shape[1]], tr["ped_scan"][idx], "pedestrian scans")
        stt = sim.numpy_state("robot_pose", "ped_pose", "ped_dist")
        np.testing.assert_allclose(stt["robot_pose"][0], tr["traj_robot_pose"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(stt["ped_pose"][0], tr["traj_ped_pose"][t], rtol=0, atol=1e-12)
        np.testing.assert_allclose(stt["ped_dist"][0], tr["traj_ped_dist"][t], rtol=0, atol=1e-10)


def _rollout_pair(gpu, cfg, occ, n_peds, steps, seed, noise=False, policy=None, **world_kw):
    """Runs the same world through the HIP step and the oracle; yields per-step comparisons.
    policy: HumanPolicy weights -> pedestrians are driven by navsim_ped_policy on both sides."""
    torch = gpu.torch
    world_kw_policy = policy
    arrays = gpu.world.make_world(cfg, occ, n_peds=n_peds, device=gpu.dev, **world_kw)
    key = "keti"
    from nav_gym_amd import robots
    arrays["scan_threshold"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array(key, "threshold_footprint")))
    arrays["scan_discomfort"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array(key, "discomfort_threshold_footprint")))
    host = {k: v.cpu().numpy() for k, v in arrays.items() if k not in ("field", "field_overflow", "rect_table", "rect_index")}
    host["field"] = ref.build_dt(occ)                 # the oracle always reads its own float32 field
    g = gpu.sim.NavSim(cfg, arrays)
    r = ref.RefSim(cfg, host)
    _eq(g.reset_obs().cpu().numpy(), r.reset_obs(), "reset obs")
    policy = world_kw_policy
    rng = np.random.default_rng(seed)
    E = cfg.n_envs
    for t in range(steps):
        act = np.stack([rng.uniform(0.0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        if t % 7 == 3:
            act[:, 0] = 0.5; act[:, 1] = 0.0          # bursts of straight driving provoke crashes
        if policy is not None:
            if t == 0:
                g.set_policy(policy)
            g.ped_policy(); r.ped_policy(policy)
        elif cfg.ped_model == abi.PED_EXTERNAL:
            cmd = np.stack([rng.uniform(0, 0.6, (E, cfg.max_peds)), rng.uniform(-0.6, 0.6, (E, cfg.max_peds))], axis=2)
            g.set_ped_cmd(cmd); r.set_ped_cmd(cmd)
        go, gout = g.step(torch.from_numpy(act).to(gpu.dev))
        ro, rout = r.step(act)
        yield t, go.cpu().numpy(), {k: v.cpu().numpy() for k, v in gout.items()}, ro, rout, g, r


@pytest.mark.parametrize("fmt", [abi.FIELD_F32, abi.FIELD_U16T, "u16t-no-rects", "u16t-rects-from-global-memory"])
@pytest.mark.parametrize("ped_model,S,auto_reset", [(abi.PED_NONE, 1, 1), (abi.PED_SFM, 2, 1), (abi.PED_EXTERNAL, 3, 0)])
def test_step_rollout_vs_oracle(gpu, ped_model, S, auto_reset, fmt):
    """48 arenas x 60 steps on 240x240 maps, 1081 beams: every output and every state array of the
    fused kernel equals the oracle's, bit for bit, including crash reverts and respawns -- for the
    float32 field, for the packed uint16 tile field with the two-rectangle tile records the bench marches
    through (navsim_build_rects), and for the packed field alone."""
    world_kw = {}
    rect_lds = 0                # a launch of 48 arenas stages the rect records in LDS (1024 threads per arena) ...
    if fmt == "u16t-no-rects":
        fmt, world_kw = abi.FIELD_U16T, {"rect_table": False}
    if fmt == "u16t-rects-from-global-memory":
        fmt, rect_lds = abi.FIELD_U16T, 1               # ... unless told to read them from global memory, like large launches do
    E, size, N = 48, 240, 8
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, n_scan_stack=S, ped_model=ped_model,
                                 auto_reset=auto_reset, n_spawn=8, seed=4242, field_format=fmt, rect_lds=rect_lds)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 4242)
    crashes = resets = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=6, steps=60, seed=1, **world_kw):
        assert ("rect_table" in g.t) == (fmt == abi.FIELD_U16T and not world_kw)
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        _eq(go, ro, "obs at step %d" % t)
        crashes += int(rout["is_crash"].sum()); resets += int(rout["done"].sum())
        if t % 10 == 9:
            gs = g.numpy_state()
            for k, v in r.a.items():
                if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
                    _eq(gs[k], v, "state %s at step %d" % (k, t))
    assert crashes > 0, "rollout never exercised the crash-revert branch"
    assert resets > 0


@pytest.mark.parametrize("fmt", [abi.FIELD_F32, abi.FIELD_U16T, "u16t-no-rects"])
@pytest.mark.parametrize("S", [1, 3])
def test_step_rollout_256_threads_parked_rays(gpu, fmt, S):
    """The 256-thread kernels without pedestrians (what a 4096-arena launch runs) leave a 64-beam chunk when at most 16
    of its rays are still marching, park those in LDS and march the parked rays 64 at a time afterwards
    (kernels_step.hpp "Parking").  Forced here on a small batch (cfg.step_block = 256): every output, observation and
    state array equals the oracle's over 80 steps with crash reverts and respawns (which scan a second time)."""
    world_kw = {}
    if fmt == "u16t-no-rects":
        fmt, world_kw = abi.FIELD_U16T, {"rect_table": False}
    E, size = 40, 260
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=1, n_scan_stack=S, ped_model=abi.PED_NONE,
                                 auto_reset=1, n_spawn=8, seed=77, field_format=fmt, step_block=256)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 77)
    crashes = resets = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=0, steps=80, seed=3, **world_kw):
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        _eq(go, ro, "obs at step %d" % t)
        crashes += int(rout["is_crash"].sum()); resets += int(rout["done"].sum())
    gs = g.numpy_state()
    for k, v in r.a.items():
        if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
            _eq(gs[k], v, "state %s at the end" % k)
    assert crashes > 0 and resets > 0


@pytest.mark.parametrize("ped_model,N,n_peds", [(abi.PED_SFM, 8, 7), (abi.PED_EXTERNAL, 8, 7), (abi.PED_SFM, 64, 40)])
def test_step_rollout_with_split_pedestrian_kernel(gpu, ped_model, N, n_peds):
    """cfg.ped_split = 2 advances the pedestrians in ped_update_kernel ahead of the fused step (a pack of arenas per
    workgroup; the default since round 3 is the fused form, wavefront 0 beside the scan): every output and state array equals the
    oracle's.  With max_peds = 64 an arena's scratch is 42 KB and the pack shrinks to ONE arena per workgroup (round-3
    advisor: two did not fit 64 KB of LDS and the request silently ran the fused form)."""
    E, size = 24, 240
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, n_scan_stack=2, ped_model=ped_model,
                                 auto_reset=1, n_spawn=8, seed=13, field_format=abi.FIELD_U16T, ped_split=2)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 13)
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=n_peds, steps=40 if N == 8 else 12, seed=5):
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        _eq(go, ro, "obs at step %d" % t)
    gs = g.numpy_state()
    for k, v in r.a.items():
        if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
            _eq(gs[k], v, "state %s" % k)


# NAVSIM_FUZZ_SEEDS=n widens the sweep for a one-off run (profiles/r03_soak/fuzz_4000.txt: 4000 seeds at the final kernels)
@pytest.mark.parametrize("seed", list(range(101, 101 + int(os.environ.get("NAVSIM_FUZZ_SEEDS", "16")))))
def test_step_fuzzed_configurations(gpu, seed):
    """Random launch shapes: map size (incl. odd), beam count and field of view, stack depth, pedestrian
    count and model, field format, arena count (which also moves the threads-per-arena heuristic), robot
    type, turning radius.  Eight steps each, every output and the final state against the oracle."""
    rng = np.random.default_rng(seed)
    size = int(rng.choice([97, 128, 200, 253, 320]))
    E = int(rng.choice([1, 3, 17, 40]))
    N = int(rng.choice([1, 4, 11]))
    ped_model = int(rng.choice([abi.PED_NONE, abi.PED_SFM, abi.PED_EXTERNAL]))
    fmt = int(rng.choice([abi.FIELD_F32, abi.FIELD_U16T]))
    S = int(rng.choice([1, 2, 5]))
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, n_scan_stack=S, ped_model=ped_model,
                                 auto_reset=int(rng.integers(0, 2)), n_spawn=int(rng.choice([1, 5])), seed=seed,
                                 field_format=fmt, min_turning_radius=float(rng.choice([0.0, 0.3])),
                                 lidar_legs=int(rng.integers(0, 2)), step_block=int(rng.choice([0, 0, 64, 256, 512, 1024])),
                                 ped_split=int(rng.integers(0, 3)), march_rule=int(rng.integers(0, 3)))
    nb = int(rng.choice([33, 64, 180, 512, 1081, 1300]))
    if nb == 1081:
        gpu.world.lidar_1081(cfg)
    else:
        cfg.n_beams = nb
        cfg.angle_min = float(rng.uniform(-3.1, -0.5)); cfg.angle_last = float(rng.uniform(0.5, 3.1))
    if rng.random() < 0.5:
        from nav_gym_amd import robots
        cfg.axle_offset = robots.ROBOTS["husky"]["axle_offset"]
    occ = gpu.world.make_maps(E, size, seed)
    n_peds = 0 if ped_model == abi.PED_NONE else int(rng.integers(0, N + 1))
    goal = (1.0, 3.0) if size < 200 else (2.0, 5.0)
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=n_peds, steps=8, seed=seed,
                                                     min_goal_dist=goal[0], max_goal_dist=goal[1], robot_clearance=0.5):
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        _eq(go, ro, "obs at step %d" % t)
    gs = g.numpy_state()
    for k, v in r.a.items():
        if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
            _eq(gs[k], v, "state %s" % k)


@pytest.mark.parametrize("step_block", [64, 256, 1024])
def test_step_rollout_with_the_maximum_pedestrian_count(gpu, step_block):
    """64 pedestrians per arena (NAVSIM_MAX_PEDS: every lane of the wavefront that runs the pedestrian phase is a
    pedestrian, 2080 pair terms, up to 256 lidar primitives) and a ragged count beside it: the fused step equals the oracle."""
    E, size, N = 6, 400, 64
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, n_scan_stack=2, ped_model=abi.PED_SFM,
                                 auto_reset=1, n_spawn=8, seed=77, field_format=abi.FIELD_U16T, step_block=step_block)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 77)
    steps = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=N, steps=6, seed=77,
                                                     min_goal_dist=4.0, max_goal_dist=9.0, robot_clearance=0.6):
        if t == 0:                                   # a ragged count in two arenas (n_peds is read every step)
            for sim_arrays in (g.t["n_peds"],):
                sim_arrays[1] = 37; sim_arrays[4] = 1
            r.a["n_peds"][1] = 37; r.a["n_peds"][4] = 1
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        _eq(go, ro, "obs at step %d" % t)
        steps += 1
    assert steps == 6
    gs = g.numpy_state()
    for k in ("ped_pose", "ped_vel", "ped_dist", "ped_prev_yaw", "robot_pose"):
        _eq(gs[k], r.a[k], "state %s" % k)


def test_packed_field_decodes_to_float_field(gpu):
    """uint16 tiles: sqrtf(d2) must be the float32 field bit for bit; odd sizes exercise edge tiles."""
    for size, n in ((100, 2), (253, 2), (500, 1)):
        occ = gpu.world.make_maps(n, size, 31 + size)
        packed, f32, nsat = gpu.sim.build_field(_t(gpu, occ), abi.FIELD_U16T)
        _eq(f32.cpu().numpy(), ref.build_dt(occ), "overflow plane %d" % size)
        assert nsat == 0
        tpr = (size + 7) // 8
        raw = packed.cpu().numpy().view(np.uint16).reshape(n, tpr, tpr, 8, 8)
        full = raw.transpose(0, 1, 3, 2, 4).reshape(n, tpr * 8, tpr * 8)[:, :size, :size]
        _eq(np.sqrt(full.astype(np.float32)), ref.build_dt(occ), "decoded tiles %d" % size)


def test_packed_field_overflow_path(gpu):
    """An arena with > 256 cells of free space around the robot: the packed field saturates and the
    step must read the exact float32 overflow plane (results still bit-identical to the oracle)."""
    E, size = 4, 720
    occ = np.zeros((E, size, size), np.uint8)
    occ[:, :5] = 1; occ[:, -5:] = 1; occ[:, :, :5] = 1; occ[:, :, -5:] = 1
    occ[:, 350:370, 100:120] = 1
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, n_spawn=4, auto_reset=1, seed=5,
                                 field_format=abi.FIELD_U16T)
    gpu.world.lidar_1081(cfg)
    packed, f32, nsat = gpu.sim.build_field(_t(gpu, occ), abi.FIELD_U16T)
    assert nsat > 0
    seen = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=0, steps=25, seed=3):
        assert "field_overflow" in g.t
        _eq(go, ro, "obs at step %d" % t)
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        seen += 1
    assert seen == 25


@pytest.mark.parametrize("fmt", [abi.FIELD_F32, abi.FIELD_U16T])
def test_ped_scans_vs_oracle(gpu, fmt):
    """navsim_ped_scans (env.py:685-693) for 20 pedestrians per arena: bit-exact vs the oracle."""
    E, size, N = 12, 240, 20
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=8,
                                 auto_reset=1, seed=31, field_format=fmt)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 31)
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=17, steps=6, seed=8):
        got = g.ped_scans().cpu().numpy()
        exp = r.ped_scans()
        _eq(got[:, :17], exp[:, :17], "pedestrian scans at step %d" % t)
        assert (exp[:, :17] < 6.0).any()


@pytest.mark.parametrize("beams", [16, 64, 100])
def test_ped_scans_few_beams_many_pedestrians(gpu, beams):
    """ped_scan_kernel with fewer beams than threads and more agents than one wavefront's share of primitives (round-3
    advisor finding: with ped_n_beams <= 64 the second wavefront marches nothing and used to read the side table and
    its count before the threads that write them had passed a barrier).  24 and 64 pedestrians (thread 64, second
    wavefront, then writes the robot's sides), repeated calls must agree with the oracle every time."""
    for N, n_peds, seed in ((24, 24, 41), (64, 64, 43)):
        E, size = 6, 240
        cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=8,
                                     auto_reset=1, seed=seed, field_format=abi.FIELD_U16T, ped_n_beams=beams)
        gpu.world.lidar_1081(cfg)
        occ = gpu.world.make_maps(E, size, seed)
        for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=n_peds, steps=3, seed=8):
            exp = r.ped_scans()
            for rep in range(3):
                _eq(g.ped_scans().cpu().numpy()[:, :n_peds], exp[:, :n_peds],
                    "pedestrian scans, %d beams, %d pedestrians, step %d" % (beams, n_peds, t))
            assert (exp[:, :n_peds] < 6.0).any()


@pytest.mark.parametrize("fmt,ped_model,plan,defer", [(abi.FIELD_F32, abi.PED_NONE, 0, 0), (abi.FIELD_U16T, abi.PED_SFM, 0, 0),
                                                      (abi.FIELD_F32, abi.PED_SFM, 1, 0), (abi.FIELD_U16T, abi.PED_NONE, 1, 0),
                                                      (abi.FIELD_U16T, abi.PED_SFM, 0, 1), (abi.FIELD_U16T, abi.PED_NONE, 1, 1),
                                                      (abi.FIELD_F32, abi.PED_SFM, 0, 1)])
def test_regen_vs_oracle(gpu, fmt, ped_model, plan, defer):
    """navsim_regen: finished arenas get a new map, field, start/goal table, robot,
    pedestrians and first observation on the device -- every array bit-identical to the oracle's.
    plan=1: candidates on the costmap, kept only when the planner joins them (env.py:342-383).
    defer=1: cfg.defer_reset_scan -- the step leaves every restart's first observation to navsim_regen, which scans the
    regenerated arenas AND those beyond regen_cap that restarted in place (regen_cap = 2 there: steps with more finished
    arenas occur); between the two calls only the rows of unfinished arenas are specified."""
    E, size, N = 40, 200 + 60 * plan, 6
    cap = 2 if defer else 5
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=ped_model, n_spawn=6,
                                 auto_reset=1, seed=17, field_format=fmt, regen_cap=cap, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.9, ped_min_robot_dist=2.0, ped_min_goal_dist=4.0,
                                 regen_plan=plan, regen_indoor_ratio=0.5 if fmt == abi.FIELD_U16T else 0.0,
                                 defer_reset_scan=defer)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 17)
    regenerated = capped = 0
    B = cfg.n_beams * cfg.n_scan_stack
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=5, steps=45, seed=6,
                                                     plan_paths=bool(plan) and ped_model != abi.PED_NONE):
        if defer:
            live = rout["done"] == 0
            _eq(go[live], ro[live], "obs of the unfinished arenas at step %d" % t)
            _eq(go[:, B:], ro[:, B:], "tails at step %d" % t)
            for k in rout:
                _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        else:
            _eq(go, ro, "obs at step %d" % t)
        n_done = int(rout["done"].sum())
        go2 = g.regen().cpu().numpy()
        ro2 = r.regen()
        _eq(go2, ro2, "obs after regen at step %d" % t)
        regenerated += min(n_done, cap); capped += n_done > cap
        if n_done:
            gs = g.numpy_state()
            for k, v in r.a.items():
                if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
                    _eq(gs[k], v, "state %s after regen at step %d" % (k, t))
            if fmt == abi.FIELD_F32:
                _eq(gs["field"], r.a["field"], "field after regen at step %d" % t)
    assert regenerated > 5 and (capped > 0 or not defer)
    if fmt == abi.FIELD_U16T:           # navsim_regen keeps the two-rectangle tile records of the new maps current
        d2, valid = _decode_rect_table(g.t["rect_table"].cpu().numpy(), size, size)
        exact = np.rint(r.a["field"].astype(np.float64) ** 2).astype(np.int64)
        assert np.array_equal(d2[valid], exact[valid]) and valid.mean() > 0.9
    if plan and ped_model != abi.PED_NONE:
        assert (r.a["ped_n_waypoints"] > 1).any(), "no pedestrian ever received a planned path"
    # a regenerated arena has a valid closed map: 5-cell border, obstacles inside
    f = r.a["field"]
    assert (f[:, :5] == 0).all() and (f[:, :, -5:] == 0).all()
    if fmt == abi.FIELD_U16T:          # half of the new maps are corridor maps (walls fill most of the arena)
        occupied = (f == 0).mean(axis=(1, 2))
        assert (occupied > 0.3).any() and (occupied < 0.2).any()


@pytest.mark.parametrize("fmt,ped_model,plan", [(abi.FIELD_U16T, abi.PED_SFM, 0), (abi.FIELD_F32, abi.PED_NONE, 0),
                                                (abi.FIELD_U16T, abi.PED_SFM, 1)])
def test_pregenerated_worlds_equal_navsim_regen(gpu, fmt, ped_model, plan):
    """NavSim.enable_pregen() (navsim_regen_swap + navsim_regen_stage): the next world of every arena is generated ahead
    of time on a side stream and only installed when the arena finishes.  The rollout -- observations after every
    regen and every state array -- equals the ORACLE's synchronous navsim_regen_cpu bit for bit (regen_cap is never
    exceeded here; arenas do finish twice, so staged worlds of staged worlds are used)."""
    E, size, N = 40, 200 + 60 * plan, 6
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=ped_model, n_spawn=6,
                                 auto_reset=1, seed=23, field_format=fmt, regen_cap=E, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.9, ped_min_robot_dist=2.0, ped_min_goal_dist=4.0,
                                 regen_plan=plan, regen_indoor_ratio=0.5 if plan else 0.0)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 23)
    regenerated = 0
    twice = np.zeros(E, int)
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=5, steps=70, seed=9,
                                                     plan_paths=bool(plan) and ped_model != abi.PED_NONE):
        if t == 0:
            g.enable_pregen()
        _eq(go, ro, "obs at step %d" % t)
        n_done = int(rout["done"].sum())
        twice += rout["done"].astype(int)
        go2 = g.regen().cpu().numpy()
        ro2 = r.regen()
        _eq(go2, ro2, "obs after the swap at step %d" % t)
        regenerated += n_done
        if n_done:
            gpu.torch.cuda.synchronize()
            gs = g.numpy_state()
            for k, v in r.a.items():
                if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
                    if k == "ped_waypoints":      # slots beyond n_waypoints keep whatever the buffer held before
                        live = np.arange(cfg.max_waypoints)[None, None, :] < r.a["ped_n_waypoints"][..., None]
                        _eq(gs[k][live], v[live], "state %s after the swap at step %d" % (k, t))
                    else:
                        _eq(gs[k], v, "state %s after the swap at step %d" % (k, t))
            if fmt == abi.FIELD_F32:
                _eq(gs["field"], r.a["field"], "field after the swap at step %d" % t)
    assert regenerated > 8 and (twice >= 2).any(), (regenerated, twice.max())


@pytest.mark.parametrize("period,min_steps,slow,install,E", [(1, 12, False, False, 48), (3, 12, False, False, 48), (2, 10, True, False, 48),
                                                            (1, 12, False, "copy", 48), (3, 12, False, "slots", 48), (2, 10, True, "slots", 48),
                                                            (1, 12, False, "slots", 48), (2, 8, False, "slots", 5), (1, 6, False, False, 1),
                                                            (2, 10, False, "slots-256", 48), (2, 10, False, "slots-64", 48),
                                                            (1, 0, False, "slots", 48), (8, 0, True, "slots", 48), (2, 0, True, "slots", 48),
                                                            (3, 5, True, "copy", 48)])
def test_pipelined_pregeneration_equals_navsim_regen(gpu, monkeypatch, period, min_steps, slow, install, E):
    """enable_pregen(pipeline=P) with cfg.regen_min_steps >= 4 P: staging passes every P steps, waited for two periods later.
    The rule -- an episode shorter than regen_min_steps restarts in place -- is the simulation's (the oracle's
    navsim_regen_cpu applies it from done_steps), so the rollout equals the oracle's synchronous one bit for bit whatever
    the passes' timing: `slow` delays every pass by a few ms on its stream (several steps' worth).  Both kinds of
    episode ends occur, and no arena ever finds its world unstaged (counters: regen_late 0).
    min_steps < 4 P (0: no rule at all, the reference's "a new map at every reset()"): the fallback -- an arena that finishes
    before its world is staged is regenerated on the spot by navsim_regen; still the oracle's rollout bit for bit, and the
    delayed passes now make arenas late (counters: regen_late > 0) instead of breaking anything.
    install: step() is navsim_step_install -- the finished arena's own workgroup copies the staged world, no swap kernel;
    "slots": the two states share the per-map arrays and exchange slot-table entries (navsim_state.map_slot), no map is copied."""
    size, N = 200, 6                                # (E = 5, 1: the flags are consumed in 32-bit words of four arenas)
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=6,
                                 auto_reset=1, seed=29, field_format=abi.FIELD_U16T, regen_cap=E, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.9, ped_min_robot_dist=2.0, ped_min_goal_dist=4.0,
                                 regen_plan=0, regen_indoor_ratio=0.0, regen_min_steps=min_steps)
    gpu.world.lidar_1081(cfg)
    if str(install).startswith("slots-"):               # the install instantiations of the other threads-per-arena families
        cfg.step_block = int(install.split("-")[1])
        if cfg.step_block == 64:
            gpu.world.lidar_full_circle(cfg, 64)
        install = "slots"
    occ = gpu.world.make_maps(E, size, 29)
    n_long = n_short = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=5, steps=110, seed=13):
        if t == 0:
            with pytest.raises(ValueError, match="regen_min_steps"):
                g.enable_pregen(pipeline=min_steps // 4 + 1)
            g.enable_pregen(pipeline=period, install=bool(install), map_slots=(install == "slots"))
            if slow:
                stage = g.lib.navsim_regen_stage
                def delayed(*a, _stage=stage, _g=g):
                    with gpu.torch.cuda.stream(_g.side):
                        # ~ 3 ms at 2 GHz: longer than a step of this world; without the rule ~ 100 ms: the passes then finish
                        # only when the steps wait for them, and an arena that ends two episodes within 2 P steps finds nothing staged
                        gpu.torch.cuda._sleep(6_000_000 if min_steps >= 4 * period else 200_000_000)
                    return _stage(*a)
                monkeypatch.setattr(g.lib, "navsim_regen_stage", delayed)
        done = rout["done"].astype(bool)
        for k in ("reward", "done", "is_success", "is_crash"):
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))
        keep = ~done if install else np.ones(E, bool)       # (navsim_step_install: the finished arenas' rows are already the new worlds')
        _eq(go[keep], ro[keep], "obs at step %d" % t)
        lng = r.a["done_steps"] >= cfg.regen_min_steps
        n_long += int((done & lng).sum()); n_short += int((done & ~lng).sum())
        _eq(g.t["done_steps"].cpu().numpy()[done], r.a["done_steps"][done], "episode lengths at step %d" % t)
        go2 = g.regen().cpu().numpy()
        ro2 = r.regen()
        _eq(go2, ro2, "obs after the swap at step %d" % t)
        if done.any():
            gpu.torch.cuda.synchronize()
            gs = g.numpy_state()
            for k, v in r.a.items():
                if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index", "ped_waypoints", "counters"):
                    _eq(gs[k], v, "state %s after the swap at step %d" % (k, t))
        if min_steps < 4 * period and E >= 8 and t in (30, 31, 60, 61):
            # two episode ends one step apart (the robot put on its goal twice): the second finds nothing staged
            idx = gpu.torch.arange(2, 8, device=gpu.dev)
            g.t["robot_goal"][idx] = g.t["robot_pose"][idx, :2]
            r.a["robot_goal"][2:8] = r.a["robot_pose"][2:8, :2]
    assert (n_long > 8 and (n_short > 0 or min_steps == 0)) or (E < 48 and n_long + n_short > 0), (n_long, n_short)
    cg, cr = g.counters(), r.counters()
    if min_steps >= 4 * period:
        assert cg["regen_late"] == 0 and g.late is None
    else:
        # (late = an arena that finished again within the passes' latency -- ~ 3 P steps -- of its last restart)
        # (unhurried passes are ready one step later; under a rule the forced one-step episodes are short, not late)
        assert g.late is not None and (cg["regen_late"] > 0 or not slow or E < 8 or min_steps > 0)
    assert cg["regen_unserved"] == 0
    assert cg["regen_short"] == cr["regen_short"] == n_short and cg["regen_served"] == cr["regen_served"] == n_long


def test_more_late_arenas_than_the_fallback_cap_are_counted(gpu):
    """The fallback of the pipelined reset path regenerates at most `fallback_cap` late arenas per step (its launches are
    sized by the cap even when nobody is late, so the default is small: max(8, E / 128)).  More than that in ONE step is
    navsim_regen's own cap rule: the rest restart on their old map, and counters()['regen_unserved'] says how many -- the
    only case in which the pipelined rollout is not step + navsim_regen's."""
    torch = gpu.torch
    E, size = 48, 200
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=6, ped_model=abi.PED_SFM, n_spawn=6,
                                 auto_reset=1, seed=29, field_format=abi.FIELD_U16T, regen_cap=E, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.9, ped_min_robot_dist=2.0, ped_min_goal_dist=4.0,
                                 regen_plan=0, regen_indoor_ratio=0.0, regen_min_steps=0)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 29)
    arrays = gpu.world.make_world(cfg, occ, n_peds=5, device=gpu.dev)
    from nav_gym_amd import robots
    arrays["scan_threshold"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "threshold_footprint")))
    arrays["scan_discomfort"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "discomfort_threshold_footprint")))
    g = gpu.sim.NavSim(cfg, arrays)
    g.reset_obs()
    g.enable_pregen(pipeline=2, install=True, fallback_cap=4)
    assert g.late_cap == 4
    act = torch.zeros((E, 2), dtype=torch.float64, device=gpu.dev)
    act[:, 0] = 0.2
    idx = torch.arange(2, 8, device=gpu.dev)
    for t in range(24):
        obs, out = g.step(act)
        g.regen()
        if t in (10, 11):                           # six arenas end two episodes one step apart: six late at once, four slots
            torch.cuda.synchronize()                # (the pass queued after step 10 has merged its requests: the worlds the
            g.t["robot_goal"][idx] = g.t["robot_pose"][idx, :2]     # six ask for at step 11 are staged after step 12 at the earliest)
    torch.cuda.synchronize()
    c = g.counters()
    assert c["regen_late"] >= 6 and c["regen_unserved"] >= 2, c
    assert torch.isfinite(obs).all()


def test_env_pipeline_falls_back_when_the_staged_copy_does_not_fit(gpu, monkeypatch):
    """The env's own choice of the pipelined reset path (pregen_pipeline=None) needs room for a second copy of the world: where
    the device has none it says so once and runs navsim_regen after every step -- the same rollout."""
    import warnings
    import nav_gym_amd
    from nav_gym_amd.env import NavGymEnv
    torch = gpu.torch
    kw = dict(nav_gym_amd.DEFAULT_KWARGS)
    E = 16
    mk = lambda: nav_gym_amd.NavGymEnv(num_envs=E, map_size=200, seed=21, num_humans=3, randomize_maps=True, **kw)
    a = mk(); a.reset()
    assert a.pregen_pipeline == 4 and a.sim.pg_install and len(a.sim.stage_lane) == 2
    NavGymEnv._warned.discard("pregen_memory")
    total = torch.cuda.mem_get_info(torch.device(gpu.dev))[1]
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *args, **kwargs: (1 << 20, total))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        b = mk(); b.reset()
    assert b.pregen_pipeline == 0 and not getattr(b.sim, "pregen", False) and b.sim.cfg.regen_cap == E
    assert any("pregen_pipeline falls back" in str(x.message) for x in w)
    g = torch.Generator(device=gpu.dev); g.manual_seed(2)
    acts = torch.rand((40, E, 2), generator=g, device=gpu.dev, dtype=torch.float64)
    acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    acts[::4, :, 0] = 0.5; acts[::4, :, 1] = 0.0
    for t in range(40):
        oa, ra, da, _ = a.step(acts[t]); ob, rb, db, _ = b.step(acts[t])
        assert torch.equal(oa["observation"], ob["observation"]) and torch.equal(ra, rb) and torch.equal(da, db), "step %d" % t
    assert a.counters()["regen_served"] > 2


def test_polled_fallback_equals_the_blind_one(gpu):
    """pregen_fallback_poll (round 6): the on-the-spot generation of an arena that finished before its world was staged is
    launched only when the step flagged somebody (one byte read back per step) -- the env's default for worlds of corridor maps /
    planned starts -- or enqueued blind behind every step: the same rollout, and late arenas do occur in it."""
    import nav_gym_amd
    torch = gpu.torch
    kw = dict(nav_gym_amd.DEFAULT_KWARGS)
    E = 48
    mk = lambda poll: nav_gym_amd.NavGymEnv(num_envs=E, map_size=200, seed=33, num_humans=3, randomize_maps=True, pregen_pipeline=4,
                                            pregen_fallback_poll=poll, **kw)
    a, b, d = mk(True), mk(False), mk(None)
    a.reset(); b.reset(); d.reset()
    assert a.sim.late_poll and not b.sim.late_poll and d.sim.late_poll        # (planned starts: the default is to poll)
    g = torch.Generator(device=gpu.dev); g.manual_seed(4)
    T = 120
    acts = torch.rand((T, E, 2), generator=g, device=gpu.dev, dtype=torch.float64)
    acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    acts[::3, :, 0] = 0.5; acts[::3, :, 1] = 0.0                              # (straight ahead every third step: episodes end often)
    for t in range(T):
        oa, ra, da, ia = a.step(acts[t]); ob, rb, db, ib = b.step(acts[t])
        assert torch.equal(oa["observation"], ob["observation"]) and torch.equal(ra, rb) and torch.equal(da, db), "step %d" % t
        assert torch.equal(ia["final_observation"]["observation"], ib["final_observation"]["observation"])
    ca, cb = a.counters(), b.counters()
    late = (ca.pop("regen_late"), cb.pop("regen_late"))     # (how many arenas were late depends on the passes' timing; nothing else does)
    assert ca == cb and ca["regen_served"] > 10 and ca["regen_unserved"] == 0 and late[0] > 0, (ca, cb, late)
    a.close(); b.close(); d.close()


@pytest.mark.parametrize("pipeline", [0, 2])
def test_env_state_dict_continues_the_rollout(gpu, pipeline):
    """NavGymEnv.state_dict() / load_state_dict(): a second environment made with the same arguments continues the first
    one's rollout bit for bit -- new maps per episode, planned pedestrians; with the pipelined reset path the staged worlds
    are not part of the snapshot and are staged again (they are functions of seed, arena and episode number)."""
    import nav_gym_amd
    torch = gpu.torch
    kw = dict(nav_gym_amd.DEFAULT_KWARGS)
    E = 24
    extra = dict(regen_min_steps=8, pregen_pipeline=2) if pipeline else {}
    make = lambda: nav_gym_amd.NavGymEnv(num_envs=E, map_size=200, seed=11, num_humans=4, randomize_maps=True, **kw, **extra)
    g = torch.Generator(device=gpu.dev); g.manual_seed(5)
    acts = torch.rand((60, E, 2), generator=g, device=gpu.dev, dtype=torch.float64)
    acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64
    acts[::5, :, 1] = 0.0; acts[::5, :, 0] = 0.5                       # straight bursts: crashes, episode ends
    a = make(); a.reset()
    for t in range(35):
        a.step(acts[t])
    sd = a.state_dict()
    want = []
    for t in range(35, 60):
        o, r, d, _ = a.step(acts[t])
        want.append((o["observation"].clone(), r.clone(), d.clone()))
    ended = int(a.sim.t["episode"].sum().item())
    b = make(); b.reset()
    b.load_state_dict(sd)
    for t in range(35, 60):
        o, r, d, _ = b.step(acts[t])
        wo, wr, wd = want[t - 35]
        assert torch.equal(o["observation"], wo) and torch.equal(r, wr) and torch.equal(d, wd), "step %d" % t
    assert int(b.sim.t["episode"].sum().item()) == ended
    assert ended - int(sd["t.episode"].sum().item()) > 3, "no episode ended after the snapshot"
    if pipeline:
        assert b.counters()["regen_late"] == 0


def test_env_reset_at_the_reference_map_size(gpu):
    """1000 x 1000 cells is the reference's own indoor map size (map_generator.py:108-122).  Such a packed world
    carries the float32 overflow plane (cells >= 256 cells from every obstacle), regenerated with the field, and
    its rect records (d2 up to 2 * 10^6): reset() and steps of sampled arenas equal the oracle bit for bit."""
    import nav_gym_amd
    kw = dict(nav_gym_amd.DEFAULT_KWARGS)
    E, size = 12, 1000
    env = nav_gym_amd.NavGymEnv(num_envs=E, map_size=size, seed=5, n_spawn=4, num_humans=6, **kw)
    env.reset()
    assert env.cfg.field_format == abi.FIELD_U16T and "field_overflow" in env.sim.t and "rect_table" in env.sim.t
    ovf = env.sim.t["field_overflow"].cpu().numpy()
    assert (ovf >= 256.0).any(), "no saturated cell: the overflow path was not exercised"
    env.sim.cfg.add_scan_noise = 0
    o = env.sim.reset_obs().cpu().numpy()
    cfg = env.sim.cfg
    refs = []
    for e in (0, 5, 11):
        c1 = cfg.copy(); c1.n_envs = 1; c1.env_index_base = int(e); c1.regen_cap = 1
        host = {k: v.cpu().numpy() for k, v in gpu.world.empty_world(c1, device="cpu", plan_paths=True).items()
                if k not in ("field", "field_overflow", "rect_table", "rect_index")}
        host["field"] = np.zeros((1, size, size), np.float32)
        host["scan_threshold"] = env.scan_threshold.cpu().numpy(); host["scan_discomfort"] = env.scan_discomfort_threshold.cpu().numpy()
        r = ref.RefSim(c1, host)
        r.out["done"][:] = 1
        _eq(o[e:e + 1], r.regen(), "first observation of arena %d" % e)
        _eq(ovf[e], r.a["field"][0], "float plane of arena %d" % e)
        refs.append((e, r))
    d2, valid = _decode_rect_table(env.sim.t["rect_table"].cpu().numpy()[:2], size, size)
    assert np.array_equal(d2[valid], np.rint(ovf[:2].astype(np.float64) ** 2).astype(np.int64)[valid]) and valid.mean() > 0.9
    rng = np.random.default_rng(3)
    for t in range(6):
        act = np.stack([rng.uniform(0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        obs, _, _, _ = env.step(act)
        og = obs["observation"].cpu().numpy()
        for e, r in refs:
            ro, _ = r.step(act[e:e + 1])
            r.replan(1024)
            _eq(og[e:e + 1], ro, "arena %d obs at step %d" % (e, t))


@pytest.mark.parametrize("pipeline", [0, 2, "graphs", "norule"])
def test_reference_default_configuration_sampled_oracle(gpu, pipeline):
    """Round-4 verdict: the configuration a user of the reference runs unmodified -- every registered default of NavGym-v0
    (__init__.py:4-40: indoor_ratio 0.5, 5-15 pedestrians on planned routes, per-episode env_param draws), KetiRobot's 512
    beams over 2 pi (keti_robot.py:44-48), 1000 x 1000 corridor maps and 400 x 400 outdoor maps (map_generator.py:97-143),
    a new map at every episode end -- batched over 1024 arenas through gym.make.  reset() and ten step() calls (graph replay:
    the re-plan of the previous step beside the step, then navsim_regen) against single-arena oracles of sampled arenas,
    bit for bit (scan noise off: its per-beam Gaussians are covered by the step tests); size-independent properties on all.
    pipeline = 2: the same world with regen_min_steps = 8 and pregen_pipeline = 2 -- the worlds are staged ahead on a side
    stream and installed inside the step (navsim_step_install, slot tables); the oracles apply the same rule."""
    import nav_gym_env
    torch = gpu.torch
    E = 1024
    kw = dict(regen_min_steps=8, pregen_pipeline=2) if pipeline == 2 else dict(pregen_pipeline=0)
    norule = pipeline == "norule"       # the env's DEFAULT for this world: the pipeline without a rule -- whoever finishes before
    if norule:                          # its world is staged is generated on the spot
        kw, pipeline = {}, 4
    if pipeline == "graphs":                                 # (the default is plain launches since navsim_regen forks: the captured form too)
        kw, pipeline = dict(use_graphs=True, pregen_pipeline=0), 0
    env = nav_gym_env.make("NavGym-v0", num_envs=E, map_size="reference", randomize_maps=True, seed=41, device=gpu.dev, **kw)
    assert env.use_graphs == bool(kw.get("use_graphs", False)) and env.pregen_pipeline == pipeline
    assert env.cfg.n_beams == 512 and env.cfg.map_h == 1000 and env.cfg.outdoor_map_size == 400 and env.plan_paths
    assert env.cfg.regen_indoor_ratio == 0.5 and (env.cfg.num_humans_lo, env.cfg.num_humans_hi) == (5, 15)
    obs = env.reset()
    env.sim.cfg.add_scan_noise = 0                       # (the captured graphs are re-captured: NavSim.step_graphed)
    env.cfg.add_scan_noise = 0
    if pipeline:
        env.sim.restage_all()                                # (the staged first observations were drawn with the noise on)
    o0 = env.sim.reset_obs().cpu().numpy()
    n_peds = env.sim.t["n_peds"].cpu().numpy()
    assert n_peds.min() >= 5 and n_peds.max() <= 15 and len(np.unique(n_peds)) > 5
    occupied = (env.sim.t["field_overflow"][:64] == 0).float().mean(dim=(1, 2)).cpu().numpy()
    assert (occupied > 0.8).any() and (occupied < 0.8).any(), "both map kinds among the first 64 arenas"     # outdoor: 400^2 of 1000^2 live
    sample = [0, 1, 17, 100, 511, 1023]
    on_goal = torch.as_tensor([1, 100], device=gpu.dev)      # these finish at step 0: new map, pedestrians, routes, first observation
    env.sim.t["robot_goal"][on_goal] = env.sim.t["robot_pose"][on_goal, :2]
    env.sim.t["steps"][on_goal] = 30                         # (long enough for cfg.regen_min_steps)
    if pipeline:                                             # ... and one that ends its episode at once: it restarts in place
        env.sim.t["robot_goal"][17] = env.sim.t["robot_pose"][17, :2]
    cfg = env.sim.cfg
    refs = []
    for e in sample:
        c1 = cfg.copy(); c1.n_envs = 1; c1.env_index_base = int(e); c1.regen_cap = 1
        host = {k: v.cpu().numpy() for k, v in gpu.world.empty_world(c1, device="cpu", plan_paths=True).items()
                if k not in ("field", "field_overflow", "rect_table", "rect_index")}
        host["field"] = np.zeros((1, 1000, 1000), np.float32)
        host["scan_threshold"] = env.scan_threshold.cpu().numpy(); host["scan_discomfort"] = env.scan_discomfort_threshold.cpu().numpy()
        r = ref.RefSim(c1, host)
        r.out["done"][:] = 1
        r.a["done_steps"][:] = 1 << 20                       # reset(): whatever cfg.regen_min_steps says about short episodes
        _eq(o0[e:e + 1], r.regen(), "first observation of arena %d" % e)
        if e in (1, 100):
            r.a["robot_goal"][0] = r.a["robot_pose"][0, :2]
            r.a["steps"][0] = 30
        if e == 17 and pipeline:
            r.a["robot_goal"][0] = r.a["robot_pose"][0, :2]
        refs.append(r)
    rng = np.random.default_rng(3)
    regenerated = 0
    for t in range(10):
        act = np.stack([rng.uniform(0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        obs, rew, done, info = env.step(torch.from_numpy(act).to(gpu.dev))
        og = obs["observation"].cpu().numpy()
        dn = done.cpu().numpy()
        assert int(dn.sum()) <= cfg.regen_cap, "more arenas finished than regen_cap: the single-arena oracles are not comparable"
        assert np.isfinite(og).all() and (og[:, :512] >= 0).all() and (og[:, :512] <= 25.0).all()
        for e, r in zip(sample, refs):
            if t > 0:
                r.replan(1024)                              # the device plans the previous step's arrivals beside this step
            ro, rout = r.step(act[e:e + 1])
            _eq(dn[e:e + 1].astype(np.uint8), rout["done"], "arena %d done at step %d" % (e, t))
            _eq(rew[e:e + 1].cpu().numpy(), rout["reward"], "arena %d reward at step %d" % (e, t))
            regenerated += int(rout["done"][0])
            _eq(og[e:e + 1], r.regen(), "arena %d obs at step %d" % (e, t))
    assert regenerated >= 2, "no sampled arena went through navsim_regen"
    c = env.counters()
    assert c["regen_served"] >= regenerated and c["regen_unserved"] == 0 and (c["regen_late"] == 0 or norule)
    if pipeline:
        assert env.sim.pg_install and "map_slot" in env.sim.t and (c["regen_short"] > 0) == (not norule)
        assert (env.sim.late is not None) == norule
        moved = (env.sim.t["map_slot"].cpu().numpy() != np.arange(E)).sum()
        assert moved >= regenerated                          # the installed maps came by exchange of slot-table entries


# NAVSIM_FUZZ_RESET_SEEDS=n widens the sweep for a one-off run (profiles/r04_soak/)
@pytest.mark.parametrize("seed", list(range(201, 201 + int(os.environ.get("NAVSIM_FUZZ_RESET_SEEDS", "12")))))
def test_reset_path_fuzzed(gpu, seed):
    """navsim_regen (+ planning, corridor maps, resident costmap) and navsim_replan at random sizes, formats,
    caps and pedestrian counts: state and observations stay bit-identical to the oracle."""
    rng = np.random.default_rng(seed)
    size = int(rng.choice([150, 205, 260, 300]))
    E = int(rng.choice([6, 20, 33]))
    N = int(rng.choice([2, 5, 9]))
    ped_model = int(rng.choice([abi.PED_NONE, abi.PED_SFM]))
    fmt = int(rng.choice([abi.FIELD_F32, abi.FIELD_U16T]))
    plan = int(rng.integers(0, 2))
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=ped_model,
                                 n_spawn=int(rng.choice([2, 6])), auto_reset=1, seed=seed, field_format=fmt,
                                 regen_cap=int(rng.choice([1, 3, 40])), min_goal_dist=2.0, max_goal_dist=6.0,
                                 spawn_clearance=0.7, ped_min_robot_dist=1.5, ped_min_goal_dist=3.0, regen_plan=plan,
                                 regen_indoor_ratio=float(rng.choice([0.0, 0.5, 1.0])), obstacle_number=int(rng.choice([3, 10])))
    gpu.world.lidar_full_circle(cfg, int(rng.choice([60, 180])))
    occ = gpu.world.make_maps(E, size, seed)
    with_costmap = bool(plan) and ped_model != abi.PED_NONE
    cfg.defer_reset_scan = int(seed % 3 == 0)          # (drawn from the seed, not from rng: the worlds of the old seeds stay)
    regenerated = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=N - 1, steps=22, seed=seed,
                                                     min_goal_dist=2.0, max_goal_dist=5.0, robot_clearance=0.6,
                                                     plan_paths=with_costmap):
        if cfg.defer_reset_scan:                       # the scan rows of finished arenas are navsim_regen's to write
            live = rout["done"] == 0
            _eq(go[live], ro[live], "obs of the unfinished arenas at step %d" % t)
        else:
            _eq(go, ro, "obs at step %d" % t)
        regenerated += int(rout["done"].sum())
        _eq(g.regen().cpu().numpy(), r.regen(), "obs after regen at step %d" % t)
        if with_costmap:
            g.replan(4); r.replan(4)
    gs = g.numpy_state()
    for k, v in r.a.items():
        if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
            if k == "ped_waypoints":
                live = np.arange(cfg.max_waypoints)[None, None, :] < r.a["ped_n_waypoints"][..., None]
                _eq(gs[k][live], v[live], "state %s" % k)
            else:
                _eq(gs[k], v, "state %s" % k)
    if fmt == abi.FIELD_F32:
        _eq(gs["field"], r.a["field"], "field")
    assert regenerated > 0 or seed > 212           # (the suite's twelve worlds all finish episodes; a wider sweep may draw one that does not)


@pytest.mark.parametrize("fmt", [abi.FIELD_F32, abi.FIELD_U16T])
def test_regen_odd_map_size(gpu, fmt):
    """navsim_regen on 205 x 205 maps: per-arena field sizes that are not multiples of 16 bytes (float32)
    and ragged edge tiles (uint16), corridor and outdoor maps mixed."""
    E, size = 12, 205
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=3, ped_model=abi.PED_SFM, n_spawn=4,
                                 auto_reset=1, seed=29, field_format=fmt, regen_cap=4, min_goal_dist=2.0,
                                 max_goal_dist=6.0, spawn_clearance=0.8, ped_min_robot_dist=1.5, ped_min_goal_dist=3.0,
                                 regen_indoor_ratio=0.5)
    gpu.world.lidar_full_circle(cfg, 90)
    occ = gpu.world.make_maps(E, size, 29)
    regenerated = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=3, steps=30, seed=3):
        regenerated += min(int(rout["done"].sum()), 4)
        _eq(g.regen().cpu().numpy(), r.regen(), "obs after regen at step %d" % t)
    gs = g.numpy_state()
    for k, v in r.a.items():
        if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
            _eq(gs[k], v, "state %s" % k)
    if fmt == abi.FIELD_F32:
        _eq(gs["field"], r.a["field"], "field")
    assert regenerated >= 2


@pytest.mark.parametrize("fmt", [abi.FIELD_F32, abi.FIELD_U16T])
def test_replan_vs_oracle(gpu, fmt):
    """navsim_replan (env.py:667-680): pedestrians that reach their final waypoint get a new planned path;
    waypoints, counts and everything downstream stay bit-identical to the oracle over a rollout, including
    calls that hit the per-call query cap."""
    E, size, N = 16, 300, 6
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=6,
                                 auto_reset=1, seed=23, field_format=fmt, ped_min_goal_dist=3.0, obstacle_number=6)
    gpu.world.lidar_full_circle(cfg, 180)
    occ = gpu.world.make_maps(E, size, 23, n_obstacles=6)
    replans = 0
    prev_n = None
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=5, steps=120, seed=4, plan_paths=True,
                                                     v_pref_range=(0.5, 0.6)):
        _eq(go, ro, "obs at step %d" % t)
        if prev_n is None:
            assert (r.a["ped_n_waypoints"] > 1).any(), "no pedestrian started with a planned path"
        cap = 2 if t % 2 else 64
        _eq(g.numpy_state("ped_due")["ped_due"], r.a["ped_due"], "who waits for a re-plan after step %d" % t)
        # the candidates: the step's own flags (ABI 5) / the call's pass over the state, alternating -- the same set
        g.replan(cap, flags=(t % 4 < 2)); r.replan(cap)
        gs = g.numpy_state("ped_waypoints", "ped_n_waypoints", "ped_wp_head", "costmap")
        n_now, head_now = r.a["ped_n_waypoints"].copy(), r.a["ped_wp_head"].copy()
        _eq(gs["ped_n_waypoints"], n_now, "waypoint counts at step %d" % t)
        _eq(gs["ped_wp_head"], head_now, "current waypoints at step %d" % t)
        live = np.arange(cfg.max_waypoints)[None, None, :] < n_now[..., None]
        _eq(gs["ped_waypoints"][live], r.a["ped_waypoints"][live], "waypoints at step %d" % t)
        if prev_n is not None:                  # a new route starts at its first waypoint
            replans += int(((head_now < prev_n[1]) | (n_now != prev_n[0])).sum())
        prev_n = (n_now, head_now)
    _eq(gs["costmap"], r.a["costmap"], "costmap")
    assert replans >= 5, replans


@pytest.mark.parametrize("mode,graphs,E", [("in_step", False, 24), ("in_step", True, 24), ("in_step", False, 600),
                                            ("two_streams", False, 24), ("two_streams", True, 24)])
def test_replan_beside_the_step_equals_the_serial_sequence(gpu, mode, graphs, E):
    """Round 5: the re-plan of step t beside step t + 1.  "in_step": navsim_step_replan -- ONE launch, the arena with a
    waiting pedestrian is re-planned by its own workgroup, and those workgroups go first (600 arenas: 37 front workgroups,
    more than one generation, a cap that is hit).  "two_streams": navsim_step_part -- the re-plan and then the waiting
    arenas on one stream, the other arenas on a second one (graphs=True: the fork and join captured in a hipGraph).  Per
    arena the order is still step, replan, step: observations, outputs and every state array equal the oracle's serial
    step, replan, step, ...  bit for bit, and every arena is stepped exactly once (steps[] advances by one everywhere)."""
    size, N = 300, 6
    cap = 64 if E < 100 else 7                      # (600 arenas: more pedestrians wait than a call serves)
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=6,
                                 auto_reset=0, seed=29, field_format=abi.FIELD_U16T, ped_min_goal_dist=3.0, obstacle_number=6)
    gpu.world.lidar_full_circle(cfg, 180)
    occ = gpu.world.make_maps(E, size, 29, n_obstacles=6)
    torch = gpu.torch
    from nav_gym_amd import robots
    arrays = gpu.world.make_world(cfg, occ, n_peds=5, device=gpu.dev, plan_paths=True, v_pref_range=(0.5, 0.6))
    arrays["scan_threshold"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "threshold_footprint")))
    arrays["scan_discomfort"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "discomfort_threshold_footprint")))
    host = {k: v.cpu().numpy() for k, v in arrays.items() if k not in ("field", "field_overflow", "rect_table", "rect_index")}
    host["field"] = ref.build_dt(occ)
    g = gpu.sim.NavSim(cfg, arrays)
    g.replan_in_step = mode == "in_step"
    r = ref.RefSim(cfg, host)
    _eq(g.reset_obs().cpu().numpy(), r.reset_obs(), "reset obs")
    if graphs:
        g.enable_graphs(regen=False, replan_cap=cap, overlap=True)
    rng = np.random.default_rng(3)
    waited = split = 0
    for t in range(100 if E < 100 else 40):
        act = np.stack([rng.uniform(0.0, 0.3, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        n_due = int((r.a["ped_due"] != 0).sum())           # arenas the side stream steps behind the re-plan
        waited += n_due; split += 0 < n_due < E
        if t > 0:
            r.replan(cap)                                    # the serial sequence: ... step, replan, step ...
        ro, rout = r.step(act)
        if graphs:
            go, gout = g.step_graphed(torch.from_numpy(act).to(gpu.dev))
        else:
            go, gout = g.step_overlapped(torch.from_numpy(act).to(gpu.dev), cap)
        _eq(go.cpu().numpy(), ro, "obs at step %d" % t)
        for k in rout:
            _eq(gout[k].cpu().numpy(), rout[k], "%s at step %d" % (k, t))
        assert int(g.t["steps"].min()) == t + 1 == int(g.t["steps"].max()), "an arena was stepped twice or not at all"
        if t % 10 == 9:
            _state_equal(g, r, cfg, "at step %d" % t)
    _state_equal(g, r, cfg, "at the end")
    assert waited >= 5 and split >= 5, (waited, split)
    assert g.counters() == r.counters() and r.counters()["replan_served"] >= 5
    if E >= 100:
        assert r.counters()["replan_unserved"] > 0, "the cap was never hit"


@pytest.mark.parametrize("E,slow", [(24, False), (24, True), (300, False)])
def test_pipelined_reset_with_the_replan_inside_the_step(gpu, monkeypatch, E, slow):
    """navsim_step_install_replan: ONE launch per step for a world with planned pedestrian routes and a new map per episode --
    the re-plan of the previous step's arrivals by the waiting arena's own workgroup, the step, and the install of the
    finished arenas' staged worlds (no rule: whoever finds nothing staged is regenerated by navsim_regen right behind the
    launch; `slow` delays the staging passes so that this happens).  Equals the oracle's serial  replan, step, regen, ...
    bit for bit: observations, outputs, every state array, the counters."""
    size, N = 300, 6
    cap = 64
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=4,
                                 auto_reset=1, seed=31, field_format=abi.FIELD_U16T, ped_min_goal_dist=3.0, obstacle_number=6,
                                 regen_cap=E, regen_plan=1, min_goal_dist=3.0, max_goal_dist=8.0, ped_min_robot_dist=2.0,
                                 spawn_clearance=0.9)
    gpu.world.lidar_full_circle(cfg, 180)
    occ = gpu.world.make_maps(E, size, 31, n_obstacles=6)
    torch = gpu.torch
    from nav_gym_amd import robots
    arrays = gpu.world.make_world(cfg, occ, n_peds=5, device=gpu.dev, plan_paths=True, v_pref_range=(0.5, 0.6))
    arrays["scan_threshold"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "threshold_footprint")))
    arrays["scan_discomfort"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "discomfort_threshold_footprint")))
    host = {k: v.cpu().numpy() for k, v in arrays.items() if k not in ("field", "field_overflow", "rect_table", "rect_index")}
    host["field"] = ref.build_dt(occ)
    g = gpu.sim.NavSim(cfg, arrays)
    r = ref.RefSim(cfg, host)
    _eq(g.reset_obs().cpu().numpy(), r.reset_obs(), "reset obs")
    g.enable_pregen(pipeline=2, install=True)
    assert g.late is not None
    g.pg_replan_cap = cap
    if slow:
        stage = g.lib.navsim_regen_stage
        def delayed(*a, _stage=stage, _g=g):
            with torch.cuda.stream(_g.side):
                torch.cuda._sleep(200_000_000)
            return _stage(*a)
        monkeypatch.setattr(g.lib, "navsim_regen_stage", delayed)
    rng = np.random.default_rng(5)
    ended = waited = 0
    for t in range(90 if E < 100 else 40):
        act = np.stack([rng.uniform(0.0, 0.5, E), rng.uniform(-0.64, 0.64, E)], axis=1)
        if t % 7 == 3:
            act[:, 0] = 0.5; act[:, 1] = 0.0            # bursts of straight driving: crashes, episode ends
        waited += int((r.a["ped_due"] != 0).sum())
        if t > 0:
            r.replan(cap)
        ro, rout = r.step(act)
        ro2 = r.regen()
        go, gout = g.step(torch.from_numpy(act).to(gpu.dev))
        go2 = g.regen().cpu().numpy()
        _eq(go2, ro2, "obs after step + regen at step %d" % t)
        for k in rout:                                   # (the goal arrays are the new worlds' on both sides by now)
            _eq(gout[k].cpu().numpy(), rout[k], "%s at step %d" % (k, t))
        ended += int(rout["done"].sum())
        if slow and t in (30, 31, 50, 51):              # two episode ends one step apart: the second finds nothing staged
            idx = torch.arange(2, 8, device=gpu.dev)
            g.t["robot_goal"][idx] = g.t["robot_pose"][idx, :2]
            r.a["robot_goal"][2:8] = r.a["robot_pose"][2:8, :2]
        if t % 10 == 9:
            gpu.torch.cuda.synchronize()
            gs = g.numpy_state()
            for k, v in r.a.items():
                if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index", "ped_waypoints", "counters"):
                    _eq(gs[k], v, "state %s at step %d" % (k, t))
    assert g.pg_replan_in_step, "the search did not fit the arena's workgroup: the launch fell back"
    assert ended > 8 and waited >= 5, (ended, waited)
    cg, cr = g.counters(), r.counters()
    assert cg["replan_served"] == cr["replan_served"] >= 5 and cg["regen_served"] == cr["regen_served"] == ended
    assert cg["regen_late"] > 0 or not slow


def test_long_routes_on_the_device(gpu, golden_dir):
    """Round 4, row a16: routes of full length.  On the costmap of the reference's own 1000 x 1000 corridor episode and
    the starts / goals its _sample_start_goal_path drew (tests/golden/golden_long_routes.npz), navsim_plan equals the
    oracle bit for bit at the default capacity (64 waypoints: nothing is cut, routes of up to 29 waypoints) and with a
    capacity of 8 (cut lists are prefixes of the full ones, path_distance still runs over every waypoint); the number of
    cells of every path is the one the reference's planner stand-in found."""
    d = np.load(os.path.join(golden_dir, "golden_long_routes.npz"))
    shape = tuple(int(x) for x in d["cost_shape"])
    cost = np.unpackbits(d["cost_packed"])[: shape[0] * shape[1]].reshape((1,) + shape)
    res_c = float(d["cost_resolution"])
    n = d["route_start"].shape[0]
    mi = np.zeros(n, np.int32)
    full = None
    for cap in (gpu.lib.default_config().max_waypoints, 8):
        gw, gn, gc, gl = gpu.sim.plan(_t(gpu, cost), _t(gpu, d["route_start"]), _t(gpu, d["route_goal"]), 2.0, max_wp=cap,
                                      res_c=res_c, map_index=_t(gpu, mi))
        rw, rn, rcells, rl = ref.plan(cost, d["route_start"], d["route_goal"], 2.0, max_wp=cap, res_c=res_c, map_index=mi)
        _eq(gn.cpu().numpy(), rn, "waypoint counts (capacity %d)" % cap)
        _eq(gc.cpu().numpy(), rcells, "path cells"); _eq(gc.cpu().numpy(), d["route_cells"].astype(np.int32), "path cells vs the reference's")
        _eq(gl.cpu().numpy(), rl, "path_distance")
        live = np.arange(cap)[None, :] < rn[:, None]
        _eq(gw.cpu().numpy()[live], rw[live], "waypoints (capacity %d)" % cap)
        if full is None:
            full = (rw, rn, rl)
            assert rn.max() > 16 and rn.max() < cap
        else:
            assert (rn == np.minimum(full[1], cap)).all() and np.array_equal(rl, full[2])
            assert np.array_equal(rw[live], full[0][:, :cap][live])


def _state_equal(g, r, cfg, what, skip=()):
    gs = g.numpy_state()
    for k, v in r.a.items():
        if k not in gs or k in ("field", "field_overflow", "rect_table", "rect_index") or k in skip:
            continue
        if k == "ped_waypoints":                  # slots beyond n_waypoints keep whatever the buffer held before
            live = np.arange(cfg.max_waypoints)[None, None, :] < r.a["ped_n_waypoints"][..., None]
            _eq(gs[k][live], v[live], "state %s %s" % (k, what))
        else:
            _eq(gs[k], v, "state %s %s" % (k, what))


def test_cut_routes_resume_to_their_goal(gpu):
    """cfg.max_waypoints = 2 with goals at least 5 m away: every planned route is stored CUT (the reference keeps every
    waypoint, env.py:788-804; the build keeps max_waypoints and counts the rest).  A pedestrian that reaches the end
    of its cut list is planned to the SAME goal by navsim_replan (navsim_state.ped_goal), not to a new one; device ==
    oracle bit for bit on every state array -- ped_goal and the counters included -- over the rollout."""
    E, size, N = 12, 300, 6
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=6,
                                 auto_reset=1, seed=37, field_format=abi.FIELD_U16T, ped_min_goal_dist=5.0, obstacle_number=6,
                                 max_waypoints=2)
    gpu.world.lidar_full_circle(cfg, 180)
    occ = gpu.world.make_maps(E, size, 37, n_obstacles=6)
    goal0 = None
    kept = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=5, steps=110, seed=4, plan_paths=True,
                                                     v_pref_range=(0.55, 0.6)):
        _eq(go, ro, "obs at step %d" % t)
        if goal0 is None:
            goal0 = r.a["ped_goal"].copy()
            last = r.a["ped_waypoints"][np.arange(E)[:, None], np.arange(N)[None, :], r.a["ped_n_waypoints"] - 1]
            cut0 = (r.a["ped_n_waypoints"][:, :5] == 2) & (last[:, :5] != goal0[:, :5]).any(axis=2)
            assert cut0.mean() > 0.8, "nearly every route of this world must start cut"
        before = r.counters()["routes_resumed"]
        g.replan(64); r.replan(64)
        if r.counters()["routes_resumed"] > before:          # a resumed route keeps its goal
            kept += 1
        if t % 10 == 9:
            _state_equal(g, r, cfg, "at step %d" % t)
    _state_equal(g, r, cfg, "at the end")
    c = r.counters()
    assert g.counters() == c
    # (the first routes of this world were cut by the host-side navsim_plan calls of make_world, which count nothing:
    # routes_cut counts the routes navsim_replan stored cut, i.e. resumed routes still more than two waypoints long)
    assert c["routes_resumed"] >= 3 and c["routes_cut"] >= 3 and kept >= 3, c
    # the pedestrians whose route was resumed and not yet completed still head for the goal they were given at the start
    same = (r.a["ped_goal"][:, :5] == goal0[:, :5]).all(axis=2)
    assert same.sum() >= 3


def test_restart_noise_does_not_depend_on_who_scans(gpu):
    """Round-4 advisor: an arena beyond cfg.regen_cap restarts in place; its first observation was scanned by the step
    (noise key of the old episode's step) or, with cfg.defer_reset_scan, by navsim_regen's masked launch (the new
    episode's reset key) -- the same arena saw different noise depending on the batch size that picks the mode.  Both
    paths draw from the reset key now: with scan noise ON, the observations after step + regen are identical."""
    from helpers import finished_world
    from nav_gym_amd import robots
    E, size, N = 24, 200, 4
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=4,
                                 auto_reset=1, seed=5, field_format=abi.FIELD_F32, regen_cap=5, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.8, ped_min_robot_dist=1.5, ped_min_goal_dist=3.0,
                                 add_scan_noise=1)
    gpu.world.lidar_full_circle(cfg, 90)
    occ = gpu.world.make_maps(E, size, 5)
    thr = ref.scan_threshold(cfg, robots.footprint_array("keti", "threshold_footprint"))
    dthr = ref.scan_threshold(cfg, robots.footprint_array("keti", "discomfort_threshold_footprint"))
    host = finished_world(cfg, occ, ref.build_dt(occ), 3, (thr, dthr))       # every robot already stands on its goal
    host["scan_noise_std"][:] = 0.03
    # spread the in-place restarts over free cells (finished_world's table holds the finish pose only)
    for e in range(E):
        free = np.argwhere(host["field"][e] > 30)
        pick = free[np.random.default_rng(e).integers(0, len(free), 4)]
        host["spawn_pose"][e, :, 0] = (pick[:, 1] + 0.5) * cfg.resolution; host["spawn_pose"][e, :, 1] = (pick[:, 0] + 0.5) * cfg.resolution
    obs = []
    for defer in (0, 1):
        c = cfg.copy(); c.defer_reset_scan = defer
        g = gpu.sim.NavSim(c, {k: v.copy() for k, v in host.items()})
        g.reset_obs()
        _, out = g.step(_t(gpu, np.zeros((E, 2))))
        assert bool(out["done"].all())
        obs.append(g.regen().cpu().numpy().copy())
        assert g.counters()["regen_unserved"] == E - 5
    _eq(obs[0], obs[1], "observations after step + regen, scan noise on, step-scanned vs regen-scanned restarts")
    assert len(np.unique(obs[0][5:, :90])) > 100             # the rows of the in-place restarts carry noise


def test_caps_are_counted(gpu):
    """Round-3 verdict: arenas beyond cfg.regen_cap "play on in place" and pedestrians beyond navsim_replan's
    max_queries wait -- silently.  navsim_state.counters makes both observable: device == oracle, and the numbers are
    the ones the done flags / the waiting pedestrians imply."""
    from helpers import finished_world
    from nav_gym_amd import robots
    E, size, N = 24, 200, 4
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=4,
                                 auto_reset=1, seed=5, field_format=abi.FIELD_F32, regen_cap=5, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.8, ped_min_robot_dist=1.5, ped_min_goal_dist=3.0)
    gpu.world.lidar_full_circle(cfg, 90)
    occ = gpu.world.make_maps(E, size, 5)
    thr = ref.scan_threshold(cfg, robots.footprint_array("keti", "threshold_footprint"))
    dthr = ref.scan_threshold(cfg, robots.footprint_array("keti", "discomfort_threshold_footprint"))
    host = finished_world(cfg, occ, ref.build_dt(occ), 3, (thr, dthr))       # every robot already stands on its goal
    host["costmap"] = ref.costmap(occ)
    host["ped_goal"] = np.zeros((E, N, 2))
    r = ref.RefSim(cfg, host)
    g = gpu.sim.NavSim(cfg, {k: v for k, v in host.items()})
    _eq(g.reset_obs().cpu().numpy(), r.reset_obs(), "reset obs")
    act = np.zeros((E, 2))
    go, gout = g.step(_t(gpu, act)); ro, rout = r.step(act)
    assert rout["done"].all()
    _eq(g.regen().cpu().numpy(), r.regen(), "obs after regen")
    c = r.counters()
    assert g.counters() == c
    assert c["regen_served"] == 5 and c["regen_unserved"] == E - 5, c
    # every pedestrian of the regenerated arenas is far from its goal; park three of them ON their final waypoint
    for sim_ in (g, r):
        a = sim_.t if sim_ is g else sim_.a
        for (e, i) in ((0, 0), (1, 2), (3, 1)):
            wp = a["ped_waypoints"][e, i, int(a["ped_n_waypoints"][e, i]) - 1]
            a["ped_pose"][e, i, 0] = wp[0]; a["ped_pose"][e, i, 1] = wp[1]
    due = int((np.linalg.norm(r.a["ped_pose"][:, :, :2] - r.a["ped_waypoints"][
        np.arange(E)[:, None], np.arange(N)[None, :], r.a["ped_n_waypoints"] - 1], axis=2) < 0.5)[
            np.arange(N)[None, :] < r.a["n_peds"][:, None]].sum())
    assert due >= 3
    # (the pedestrians were moved by hand: the step's flags know nothing of them, the calls look at the state themselves)
    g.replan(0, flags=False); r.replan(0)                 # a cap of zero serves nobody and counts everybody
    g.replan(2, flags=False); r.replan(2)
    c2 = r.counters()
    assert g.counters() == c2
    assert c2["replan_served"] == 2 and c2["replan_unserved"] == due + (due - 2), c2
    _state_equal(g, r, cfg, "after the capped calls")


@pytest.mark.parametrize("clamp", [0, 1])
def test_wheel_speed_actions(gpu, clamp):
    """NAVSIM_ACTION_WHEELS (round 4; build-defined like the Husky model): io.action = the angular speeds of a
    skid-steer base's left / right wheel pairs, converted on the device with the Husky's wheel radius and track
    (husky.urdf.xacro:61-67), optionally clamped to linvel_range x rotvel_range.  Device == oracle bit for bit, and
    the rollout equals the one driven by the converted (and clipped) twists."""
    from nav_gym_amd import robots
    E, size = 24, 200
    kw = dict(n_envs=E, map_h=size, map_w=size, max_peds=4, ped_model=abi.PED_SFM, n_spawn=6, auto_reset=1, seed=19,
              field_format=abi.FIELD_U16T, axle_offset=0.0, clamp_action=clamp, linvel_lo=0.0, linvel_hi=1.0,
              rotvel_lo=-2.0, rotvel_hi=2.0)
    cfg_w = gpu.lib.default_config(action_kind=abi.ACTION_WHEELS, **kw)
    cfg_t = gpu.lib.default_config(action_kind=abi.ACTION_TWIST, **kw)
    assert (cfg_w.wheel_radius, cfg_w.wheel_track) == (robots.HUSKY_WHEEL_RADIUS, robots.HUSKY_TRACK)
    for cfg in (cfg_w, cfg_t):
        gpu.world.lidar_full_circle(cfg, 128)
    occ = gpu.world.make_maps(E, size, 19)
    arrays = gpu.world.make_world(cfg_w, occ, n_peds=3, device=gpu.dev)
    arrays["scan_threshold"] = gpu.sim.scan_threshold(cfg_w, _t(gpu, robots.footprint_array("husky", "threshold_footprint")))
    arrays["scan_discomfort"] = gpu.sim.scan_threshold(cfg_w, _t(gpu, robots.footprint_array("husky", "discomfort_threshold_footprint")))
    host = {k: v.cpu().numpy() for k, v in arrays.items() if k not in ("field", "field_overflow", "rect_table", "rect_index")}
    host["field"] = ref.build_dt(occ)
    gw, gt, r = gpu.sim.NavSim(cfg_w, arrays), gpu.sim.NavSim(cfg_t, arrays), ref.RefSim(cfg_w, host)
    _eq(gw.reset_obs().cpu().numpy(), r.reset_obs(), "reset obs"); gt.reset_obs()
    rng = np.random.default_rng(2)
    clipped = 0
    for t in range(25):
        wheels = rng.uniform(-3.0, 9.0, (E, 2))                       # rad/s: up to 1.5 m/s, well beyond the limits
        twist = robots.husky_twist_from_wheels(wheels[:, 0], wheels[:, 1])
        if clamp:
            clipped += int((twist[:, 0] > 1.0).sum() + (twist[:, 0] < 0.0).sum() + (np.abs(twist[:, 1]) > 2.0).sum())
            twist = np.stack([np.clip(twist[:, 0], 0.0, 1.0), np.clip(twist[:, 1], -2.0, 2.0)], axis=1)
        go, gout = gw.step(_t(gpu, wheels)); ro, rout = r.step(wheels)
        go = go.cpu().numpy().copy()
        _eq(go, ro, "obs at step %d" % t)
        for k in rout:
            _eq(gout[k].cpu().numpy(), rout[k], "%s at step %d" % (k, t))
        go2, gout2 = gt.step(_t(gpu, twist))
        _eq(go2.cpu().numpy(), go, "twist-driven obs at step %d" % t)
    _state_equal(gw, r, cfg_w, "wheel-driven")
    _eq(gw.numpy_state("prev_action")["prev_action"], gt.numpy_state("prev_action")["prev_action"], "prev_action holds the twist")
    assert not clamp or clipped > 20


def _policy_weights_random(seed):
    rng = np.random.default_rng(seed)
    fan = {"cv1": 15, "cv2": 96, "fc1": 4096, "fc2": 260, "a1": 128, "a2": 128}
    w = {}
    for k, shape in abi.POLICY_SHAPES.items():
        b = 1.0 / np.sqrt(fan[k.split("_")[0]])
        w[k] = rng.uniform(-b, b, shape).astype(np.float32)
    return w


@pytest.mark.parametrize("fmt", [abi.FIELD_F32, abi.FIELD_U16T])
def test_policy_closed_loop_vs_oracle(gpu, fmt):
    """Row a10 on the device: pedestrian scans -> HumanPolicy actor (convolutions on the vector units, the
    4096 -> 256 layer on v_mfma_f32_32x32x2_f32) -> (v, omega) -> Human.set_vel, in closed loop for 17
    pedestrians per arena.  The float32 network is specified as fused-multiply-add chains in index order,
    which is what the MFMA computes, so commands, network outputs, observations and all state stay
    bit-identical to the oracle step after step."""
    E, size, N = 10, 240, 20
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_EXTERNAL, n_spawn=8,
                                 auto_reset=1, seed=41, field_format=fmt)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 41)
    w = _policy_weights_random(7)
    first = True
    rollout = _rollout_pair(gpu, cfg, occ, n_peds=17, steps=8, seed=9, policy=w)
    for t, go, gout, ro, rout, g, r in rollout:
        _eq(go, ro, "obs at step %d" % t)
        _eq(g.t["policy_prev_actions"].cpu().numpy(), r.prev_actions, "network output at step %d" % t)
        _eq(g.t["ped_cmd"].cpu().numpy(), r.a["ped_cmd"], "pedestrian commands at step %d" % t)
        if t % 4 == 3:
            gs = g.numpy_state()
            for k, v in r.a.items():
                if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
                    _eq(gs[k], v, "state %s at step %d" % (k, t))
    m = r.prev_actions[:, :17]
    assert (m[..., 0] > 0).all() and (m[..., 0] < 1).all() and np.abs(m[..., 1]).max() < 1 and m.std() > 1e-3


def test_policy_large_batch_is_chunk_invariant(gpu):
    """navsim_ped_policy processes the pedestrians in passes of 32 768 (feature scratch); 34 000 pedestrians
    built from a 10-arena world repeated 170 times: every copy gets the values of the original, whichever
    pass and tile it falls into, and the original equals the oracle."""
    torch = gpu.torch
    E0, rep, size, N = 10, 170, 120, 20
    cfg0 = gpu.lib.default_config(n_envs=E0, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_EXTERNAL, n_spawn=4,
                                  auto_reset=1, seed=3, field_format=abi.FIELD_F32)
    gpu.world.lidar_full_circle(cfg0, 64)
    occ = gpu.world.make_maps(E0, size, 3)
    base = gpu.world.make_world(cfg0, occ, n_peds=18, device=gpu.dev, min_goal_dist=2.0, max_goal_dist=4.0, robot_clearance=0.6)
    from nav_gym_amd import robots
    for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
        base[key] = gpu.sim.scan_threshold(cfg0, _t(gpu, robots.footprint_array("keti", name)))
    rng = np.random.default_rng(1)
    scans0 = rng.uniform(0.0, 7.0, (E0, N, 512)).astype(np.float32)
    w = _policy_weights_random(11)
    host = {k: v.cpu().numpy() for k, v in base.items()}
    r = ref.RefSim(cfg0, host)
    rc, rm = r.ped_policy(w, scans0)
    cfg = cfg0.copy(); cfg.n_envs = E0 * rep
    shared = ("scan_threshold", "scan_discomfort", "beam_table")
    big = {k: (v if k in shared else v.repeat((rep,) + (1,) * (v.dim() - 1))) for k, v in base.items()}
    g = gpu.sim.NavSim(cfg, big)
    g.set_policy(w)
    cmd, mean = g.ped_policy(_t(gpu, scans0).repeat(rep, 1, 1))
    cmd = cmd.cpu().numpy().reshape(rep, E0, N, 2); mean = mean.cpu().numpy().reshape(rep, E0, N, 2)
    _eq(cmd[0], rc, "commands of the first copy vs oracle")
    _eq(mean[0], rm, "network output of the first copy vs oracle")
    assert (cmd == cmd[:1]).all() and (mean == mean[:1]).all()
    assert np.abs(mean[0, :, :18]).max() > 0


@pytest.mark.parametrize("fmt,rects", [(abi.FIELD_U16T, True), (abi.FIELD_U16T, False), (abi.FIELD_F32, False)])
def test_fused_scan_policy_equals_the_two_calls(gpu, fmt, rects):
    """navsim_ped_scan_policy (round 4: every pedestrian's scan taken, clipped, scaled and convolved by ONE workgroup, the
    beam directions through the beam table) == navsim_ped_scans followed by navsim_ped_policy, bit for bit: the scans it
    can write out, the network output, the commands, the popped waypoints.  Ragged pedestrian counts (dead slots keep their
    rows), pedestrians close enough to see each other and the robot."""
    torch = gpu.torch
    E, size, N = 12, 200, 20
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_EXTERNAL, n_spawn=6,
                                 auto_reset=1, seed=23, field_format=fmt)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 23)
    arrays = gpu.world.make_world(cfg, occ, n_peds=17, device=gpu.dev, rect_table=rects)
    from nav_gym_amd import robots
    for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
        arrays[key] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", name)))
    arrays["n_peds"][::3] = 9
    arrays["n_peds"][1] = 0
    g = gpu.sim.NavSim(cfg, arrays)
    g.set_policy(_policy_weights_random(5))
    g.t["policy_prev_actions"].copy_(torch.rand((E, N, 2), device=gpu.dev) * 0.5)
    keep = {k: g.t[k].clone() for k in ("policy_prev_actions", "ped_waypoints", "ped_n_waypoints", "ped_wp_head")}
    scans = g.ped_scans()
    cmd_a, mean_a = [x.clone() for x in g.ped_policy(scans)]
    wp_a, nwp_a, head_a = g.t["ped_waypoints"].clone(), g.t["ped_n_waypoints"].clone(), g.t["ped_wp_head"].clone()
    for k, v in keep.items():
        g.t[k].copy_(v)
    out = torch.full_like(scans, -7.0)
    cmd_b, mean_b = g.ped_policy(fused=True, scans_out=out)
    assert torch.equal(cmd_a, cmd_b) and torch.equal(mean_a, mean_b)
    assert torch.equal(wp_a, g.t["ped_waypoints"]) and torch.equal(nwp_a, g.t["ped_n_waypoints"])
    assert torch.equal(head_a, g.t["ped_wp_head"])
    live = (torch.arange(N, device=gpu.dev)[None, :] < g.t["n_peds"][:, None].clamp(max=N))
    assert torch.equal(out[live], scans[live]) and bool((out[~live] == -7.0).all())
    assert float(mean_b[live].abs().max()) > 0 and float(scans[live].min()) < 5.9


@pytest.mark.parametrize("name", ["random_S1", "peds_S1"])
def test_policy_vs_reference_trace(gpu, name):
    """The same control block against the reference's own step(): with the weights the golden traces were
    recorded with, the device reproduces the (v, omega) the reference handed to Human.set_vel (1e-5)."""
    from helpers import policy_weights
    tr = load_trace(name)
    cfg, arrays, occ = trace_setup(tr, gpu.lib.default_config, lambda o: gpu.sim.build_dt(_t(gpu, o)).cpu().numpy())
    N = tr["init_ped_pose"].shape[0]
    wp = np.zeros((1, N, cfg.max_waypoints, 2)); wp[0, :, :tr["init_ped_waypoints"].shape[1]] = tr["init_ped_waypoints"]
    arrays["ped_waypoints"] = wp
    arrays["ped_n_waypoints"] = tr["init_ped_n_waypoints"][None].astype(np.int32)
    g = gpu.sim.NavSim(cfg, arrays)
    g.reset_obs()
    g.set_policy(policy_weights(int(tr["policy_seed"])))
    worst, checked = 0.0, 0
    for t in range(10):                                   # no crash and no re-plan in the first steps of these traces
        if t > 0:
            g.t["policy_prev_actions"][0] = _t(gpu, tr["ped_mean"][t - 1])
        cmd, mean = g.ped_policy()
        ok = np.ones(N, bool) if name == "random_S1" or t == 0 else np.arange(N) > 0   # peds_S1: pedestrian 0 re-plans
        worst = max(worst, np.abs(cmd[0].cpu().numpy()[ok] - tr["ped_cmd"][t][ok]).max(),
                    np.abs(mean[0].cpu().numpy()[ok] - tr["ped_mean"][t][ok]).max())
        checked += int(ok.sum())
        g.set_ped_cmd(_t(gpu, tr["ped_cmd"][t][None]))
        g.step(_t(gpu, tr["actions"][t][None]))
    assert checked > 50 and worst < 1e-5, (checked, worst)


def test_launch_order_is_result_neutral(gpu):
    """navsim_launch_order returns the arenas by descending cost, and stepping with an arbitrary launch order
    (here: reversed, then the measured longest-first order) gives bit-identical outputs and state to the
    identity order -- which workgroup takes an arena is a scheduling hint only."""
    torch = gpu.torch
    cost = torch.randint(0, 50000, (5000,), dtype=torch.int32, device=gpu.dev)
    order = torch.empty(5000, dtype=torch.int32, device=gpu.dev)
    gpu.lib.check(gpu.lib.load().navsim_launch_order(cost.data_ptr(), order.data_ptr(), 5000, None), "launch_order")
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(5000))
    c = cost.cpu().numpy()[o].astype(np.int64)
    assert (np.diff(c) <= c.max() // 1023 + 1).all()              # descending up to one histogram bucket
    cost[17] = 2_000_000_000                                      # one held-up workgroup must not flatten the rest
    gpu.lib.check(gpu.lib.load().navsim_launch_order(cost.data_ptr(), order.data_ptr(), 5000, None), "launch_order")
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(5000)) and 17 in o[:50]
    c = cost.cpu().numpy()[o].astype(np.int64)
    rest = c[c < 2_000_000_000]
    assert (np.diff(rest) <= 4 * int(cost.cpu().numpy().astype(np.int64).mean()) // 1023 + 2).all()
    E, size = 64, 240
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=6, ped_model=abi.PED_SFM, n_spawn=8,
                                 auto_reset=1, seed=5, field_format=abi.FIELD_U16T)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 5)
    arrays = gpu.world.make_world(cfg, occ, n_peds=5, device=gpu.dev)
    from nav_gym_amd import robots
    for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
        arrays[key] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", name)))
    plain = gpu.sim.NavSim(cfg, arrays, launch_order=False)
    lpt = gpu.sim.NavSim(cfg, arrays, launch_order=True)   # a 64-arena launch is one generation: off by default
    lpt.lpt_period = 3
    lpt.t["launch_order"].copy_(torch.arange(E - 1, -1, -1, dtype=torch.int32, device=gpu.dev))
    assert "launch_order" not in plain.t
    _eq(plain.reset_obs().cpu().numpy(), lpt.reset_obs().cpu().numpy(), "reset obs")
    g = torch.Generator(device=gpu.dev); g.manual_seed(2)
    for t in range(10):
        act = torch.rand((E, 2), generator=g, device=gpu.dev, dtype=torch.float64)
        act[:, 0] *= 0.5; act[:, 1] = act[:, 1] * 1.28 - 0.64
        o1, out1 = plain.step(act)
        o2, out2 = lpt.step(act)
        assert torch.equal(o1, o2), t
        for k in out1:
            assert torch.equal(out1[k], out2[k]), (k, t)
    s1, s2 = plain.numpy_state(), lpt.numpy_state()
    for k in s1:
        _eq(s1[k], s2[k], "state %s" % k)
    assert (lpt.t["arena_cost"] > 0).all()
    assert not np.array_equal(lpt.t["launch_order"].cpu().numpy(), np.arange(E - 1, -1, -1))      # re-sorted by now


def test_crowd_check_vs_reference_and_oracle(gpu):
    """CrowdSim-v0's collision / goal / reward block (crowd_sim.py:808-949) on the device:
    equal to the oracle bit for bit, and to the reference's own CrowdSim.step on the 600 golden situations;
    plus a larger random batch with ragged agent counts against the oracle."""
    from test_oracle_golden import _crowd_golden
    d, maps, params = _crowd_golden()
    got = gpu.sim.crowd_check(params, _t(gpu, maps), _t(gpu, d["robot"]), _t(gpu, d["agents"]), _t(gpu, d["global_time"]))
    reward, done, info, md = [x.cpu().numpy() for x in got]
    assert np.array_equal(info, d["info"]) and np.array_equal(done, d["done"])
    assert np.abs(reward - d["reward"]).max() < 1e-12
    exp = ref.crowd_check(params, maps, d["robot"], d["agents"], d["global_time"])
    for a, b, what in zip((reward, done, info, md), exp, ("reward", "done", "info", "min_dist")):
        _eq(a, b, what)
    rng = np.random.default_rng(8)
    E, A, G = 3000, 70, 128
    params2 = dict(params, map_size_m=12.8, map_resolution=0.1)
    fm = (rng.random((E, G, G)) > 0.01).astype(np.uint8)
    robot = np.zeros((E, 10)); robot[:, :2] = rng.uniform(-6.2, 6.2, (E, 2))
    robot[:, 4:6] = rng.uniform(-1, 1, (E, 2)); robot[:, 2:4] = robot[:, :2] + robot[:, 4:6] * 0.25
    robot[:, 6:8] = rng.uniform(-6, 6, (E, 2)); robot[:, 8] = rng.uniform(0.2, 0.5, E); robot[:, 9] = rng.uniform(-1, 1, E)
    agents = np.zeros((E, A, 5)); agents[..., :2] = robot[:, None, :2] + rng.uniform(-5, 5, (E, A, 2))
    agents[..., 2:4] = rng.uniform(-1, 1, (E, A, 2)); agents[..., 4] = rng.uniform(0.2, 0.4, (E, A))
    na = rng.integers(0, A + 1, E).astype(np.int32)
    gt = rng.uniform(0, 26, E)
    got = gpu.sim.crowd_check(params2, _t(gpu, fm), _t(gpu, robot), _t(gpu, agents), _t(gpu, gt), _t(gpu, na))
    exp = ref.crowd_check(params2, fm, robot, agents, gt, na)
    for a, b, what in zip(got, exp, ("reward", "done", "info", "min_dist")):
        _eq(a.cpu().numpy(), b, what + " (random batch)")
    assert len(np.unique(exp[2])) >= 4


def test_crowd_local_maps_vs_reference_and_oracle(gpu, golden_dir):
    """CrowdSim-v0 local maps on the device: get_local_map_angular equals the oracle bit for
    bit (same deterministic math, order-independent min) and the reference's recorded maps to 1e-12; get_local_map's
    window equals the reference's recorded windows exactly; with the rotation (restated from OpenCV, unpinned) the
    device equals the oracle on random headings, ragged obstacle counts and windows clipped by the map border."""
    d = np.load(os.path.join(golden_dir, "golden_crowd_maps.npz"))
    P = {str(k): float(v) for k, v in zip(d["param_names"], d["params"])}
    maps = np.unpackbits(d["maps"])[: int(np.prod(d["map_shape"]))].reshape(d["map_shape"])
    lmap = np.unpackbits(d["lmap"])[: int(np.prod(d["lmap_shape"]))].reshape(d["lmap_shape"])
    got = gpu.sim.crowd_angular_map(P, _t(gpu, d["robot"]), _t(gpu, d["verts"]), _t(gpu, d["n_obst"])).cpu().numpy()
    _eq(got, ref.crowd_angular_map(P, d["robot"], d["verts"], d["n_obst"]), "angular map vs oracle")
    np.testing.assert_allclose(got, d["amap"], rtol=0, atol=1e-12)
    w = gpu.sim.crowd_local_map(P, _t(gpu, maps), _t(gpu, d["robot2"]), rotate=False).cpu().numpy()
    _eq(w, lmap, "local-map window vs the reference")
    r = gpu.sim.crowd_local_map(P, _t(gpu, maps), _t(gpu, d["robot2"]), rotate=True).cpu().numpy()
    _eq(r, ref.crowd_local_map(P, maps, d["robot2"], rotate=True), "rotated local map vs oracle")
    assert not np.array_equal(r, w)
    # a larger random batch, other parameters: 5000 envs, 36 sectors over the front half plane, 8-vertex outlines
    rng = np.random.default_rng(9)
    E, O, V = 5000, 7, 8
    P2 = dict(P, angular_dim=36, angular_min=-np.pi / 2, angular_max=np.pi / 2, angular_max_range=4.0, normalize=0,
              submap_size_m=4.0, map_size_m=10.0)
    robot = np.concatenate([rng.uniform(-5, 5, (E, 2)), rng.uniform(-np.pi, np.pi, (E, 1)), rng.uniform(0.2, 0.5, (E, 1))], axis=1)
    ang = np.sort(rng.uniform(0, 2 * np.pi, (E, O, V)), axis=2)
    ctr = rng.uniform(-4, 4, (E, O, 1, 2)); rad = rng.uniform(0.2, 1.5, (E, O, V, 1))
    verts = ctr + rad * np.stack([np.cos(ang), np.sin(ang)], axis=3)
    n_obst = rng.integers(0, O + 1, E).astype(np.int32)
    _eq(gpu.sim.crowd_angular_map(P2, _t(gpu, robot), _t(gpu, verts), _t(gpu, n_obst)).cpu().numpy(),
        ref.crowd_angular_map(P2, robot, verts, n_obst), "angular map, random batch")
    G = 100
    fm = (rng.random((600, G, G)) > 0.08).astype(np.uint8)
    _eq(gpu.sim.crowd_local_map(P2, _t(gpu, fm), _t(gpu, robot[:600]), rotate=True).cpu().numpy(),
        ref.crowd_local_map(P2, fm, robot[:600], rotate=True), "rotated local map, random batch")


def test_crowd_orca_and_agent_step_vs_oracle(gpu, golden_dir):
    """CrowdSim-v0 pedestrians on the device: the restated RVO2 step (navsim_crowd_orca; rvo2 absent -> unpinned)
    equals the oracle bit for bit on 20 000 random queries -- ragged agent counts, crowded and colliding
    situations, box and triangle obstacles from several polygon sets -- and Agent.step equals the oracle bit for bit
    and the reference's recorded goldens to 1e-12."""
    rng = np.random.default_rng(17)
    Q, A, S, O, V = 20000, 12, 7, 6, 4
    P = dict(time_step=0.25, neighbor_dist=10, time_horizon=5, time_horizon_obst=5, max_neighbors=10)
    ag = np.zeros((Q, A, 6))
    ag[..., :2] = rng.uniform(-4, 4, (Q, A, 2)); ag[..., 2:4] = rng.uniform(-1, 1, (Q, A, 2))
    ag[..., 4] = rng.uniform(0.2, 0.5, (Q, A)); ag[..., 5] = rng.uniform(0.5, 1.5, (Q, 1))
    ag[::7, 1, :2] = ag[::7, 0, :2] + rng.uniform(-0.3, 0.3, (len(ag[::7]), 2))       # already overlapping neighbours
    n_agents = rng.integers(1, A + 1, Q).astype(np.int32)
    pv = rng.uniform(-1.2, 1.2, (Q, 2))
    ctr = rng.uniform(-5, 5, (S, O, 1, 2)); half = rng.uniform(0.2, 1.5, (S, O, 1, 2))
    sign = np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]], float)                        # counter-clockwise boxes
    verts = ctr + half * sign
    verts[:, 0, 3] = verts[:, 0, 0] + [0.0, -0.01]                                      # a thin sliver: near-degenerate edge
    n_obst = rng.integers(0, O + 1, S).astype(np.int32)
    obst_set = rng.integers(0, S, Q).astype(np.int32)
    theta = rng.uniform(0, 2 * np.pi, Q)
    gv, ga = gpu.sim.crowd_orca(P, _t(gpu, ag), _t(gpu, pv), _t(gpu, verts), _t(gpu, n_agents), _t(gpu, n_obst),
                                _t(gpu, obst_set), _t(gpu, theta))
    rv, ra = ref.crowd_orca(P, ag, pv, verts, n_agents, n_obst, obst_set, theta)
    _eq(gv.cpu().numpy(), rv, "ORCA velocity")
    _eq(ga.cpu().numpy(), ra, "ORCA action")
    assert (np.linalg.norm(rv, axis=1) <= ag[:, 0, 5] * (1 + 5e-3)).all()               # the speed disc (float32 LP: a few 1e-4 over)
    assert (np.abs(rv - pv) > 1e-3).any(axis=1).mean() > 0.3                             # the constraints do bind
    d = np.load(os.path.join(golden_dir, "golden_crowd_agent.npz"))
    gp, gvel = gpu.sim.crowd_agent_step(_t(gpu, d["pose"]), _t(gpu, d["action"]), float(d["time_step"]))
    rp, rvel = ref.crowd_agent_step(d["pose"], d["action"], float(d["time_step"]))
    _eq(gp.cpu().numpy(), rp, "Agent.step pose"); _eq(gvel.cpu().numpy(), rvel, "Agent.step velocity")
    np.testing.assert_allclose(rp, d["pose_out"], rtol=0, atol=1e-12)


def test_crowd_sim_step_pipeline_vs_oracle(gpu):
    """CrowdSim.step (crowd_sim.py:724-997) for 64 envs x 5 ORCA pedestrians in closed loop, composed on the device by
    nav_gym_amd.crowd.CrowdSimStepper, against the same composition of the oracle's functions on the host: rewards,
    info codes, agent states and the robot's angular map bit for bit over 25 steps (pedestrians visibly avoid each
    other and the robot: their ORCA actions differ from their goal-seeking velocity)."""
    from nav_gym_amd.crowd import CrowdSimStepper
    torch = gpu.torch
    rng = np.random.default_rng(23)
    E, H, O, G = 64, 5, 4, 140
    params = dict(time_step=0.25, discomfort_dist=0.2, map_size_m=14.0, map_resolution=0.1, success_reward=1.0,
                  collision_penalty=-0.25, discomfort_penalty_factor=0.5, rotation_penalty_factor=-0.01,
                  timeout_penalty=-0.125, time_limit=25.0)
    mp = dict(angular_min=-np.pi, angular_max=np.pi, angular_max_range=6.0, angular_dim=72, normalize=1,
              map_size_m=14.0, map_resolution=0.1, submap_size_m=6.0)
    ctr = rng.uniform(-5, 5, (E, O, 1, 2)); half = rng.uniform(0.2, 0.8, (E, O, 1, 2))
    verts = ctr + half * np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]], float)
    n_obst = rng.integers(0, O + 1, E).astype(np.int32)
    free = np.ones((E, G, G), np.uint8)
    for e in range(E):
        for o in range(n_obst[e]):
            lo = np.floor((verts[e, o, 2] + 7.0) / 0.1).astype(int); hi = np.ceil((verts[e, o, 0] + 7.0) / 0.1).astype(int)
            free[e, max(lo[0], 0):hi[0], max(lo[1], 0):hi[1]] = 0
    ang = rng.uniform(0, 2 * np.pi, (E, H)); rad = rng.uniform(2.5, 4.0, (E, H))
    hp = np.stack([rad * np.cos(ang), rad * np.sin(ang)], -1)
    humans = np.concatenate([hp, np.zeros((E, H, 2)), np.full((E, H, 1), 0.3), rng.uniform(0.6, 1.2, (E, H, 1)), -hp,
                             rng.uniform(0, 2 * np.pi, (E, H, 1))], axis=-1)
    robot = np.concatenate([rng.uniform(-1, 1, (E, 2)), np.zeros((E, 2)), np.full((E, 1), 0.3), np.ones((E, 1)),
                            rng.uniform(-4, 4, (E, 2)), rng.uniform(0, 2 * np.pi, (E, 1))], axis=1)
    stp = CrowdSimStepper(_t(gpu, humans), _t(gpu, robot), _t(gpu, verts), _t(gpu, n_obst), _t(gpu, free), params,
                          map_params=mp)
    h, r, gt = humans.copy(), robot.copy(), np.zeros(E)
    orca_p = dict(time_step=0.25, neighbor_dist=10, time_horizon=5, time_horizon_obst=5, max_neighbors=10)
    deviated = 0
    for t in range(25):
        act = np.stack([rng.uniform(0, 1.0, E), rng.uniform(-0.5, 0.5, E)], axis=1)
        out = stp.step(_t(gpu, act))
        # ---- the same composition with the oracle
        ag = np.zeros((E, H, H + 1, 6)); pv = np.zeros((E, H, 2))
        for k in range(H):
            order = [k] + [j for j in range(H) if j != k]
            ag[:, k, :H, :4] = h[:, order, :4]; ag[:, k, :H, 4] = h[:, order, 4] + 0.01; ag[:, k, :H, 5] = h[:, k:k + 1, 5]
            ag[:, k, H, :4] = r[:, :4]; ag[:, k, H, 4] = r[:, 4] + 0.01; ag[:, k, H, 5] = h[:, k, 5]
            vel = h[:, k, 6:8] - h[:, k, 0:2]; sp = np.sqrt(vel[:, :1] * vel[:, :1] + vel[:, 1:] * vel[:, 1:])
            pv[:, k] = np.where(sp > 1, vel / sp, vel)
        _, hact = ref.crowd_orca(orca_p, ag.reshape(E * H, H + 1, 6), pv.reshape(E * H, 2), verts, None, n_obst,
                                 np.repeat(np.arange(E, dtype=np.int32), H), h[..., 8].reshape(-1))
        npose, nvel = ref.crowd_agent_step(np.stack([r[:, 0], r[:, 1], r[:, 8]], 1), act, 0.25)
        robot10 = np.stack([r[:, 0], r[:, 1], npose[:, 0], npose[:, 1], nvel[:, 0], nvel[:, 1], r[:, 6], r[:, 7], r[:, 4], act[:, 1]], 1)
        rew, done, info, md = ref.crowd_check(params, free, robot10, h[..., :5], gt)
        r[:, 0:2] = npose[:, 0:2]; r[:, 2:4] = nvel; r[:, 8] = npose[:, 2]
        hpn, hvn = ref.crowd_agent_step(np.stack([h[..., 0], h[..., 1], h[..., 8]], -1).reshape(E * H, 3), hact, 0.25)
        h[..., 0:2] = hpn[:, :2].reshape(E, H, 2); h[..., 2:4] = hvn.reshape(E, H, 2); h[..., 8] = hpn[:, 2].reshape(E, H)
        gt = gt + 0.25
        amap = ref.crowd_angular_map(mp, np.stack([r[:, 0], r[:, 1], r[:, 8], r[:, 4]], 1), verts, n_obst)
        _eq(out["reward"].cpu().numpy(), rew, "reward at step %d" % t)
        _eq(out["info"].cpu().numpy(), info, "info at step %d" % t)
        _eq(out["done"].cpu().numpy(), done, "done at step %d" % t)
        _eq(stp.h.cpu().numpy(), h, "pedestrian states at step %d" % t)
        _eq(stp.r.cpu().numpy(), r, "robot states at step %d" % t)
        _eq(out["local_map"].cpu().numpy(), amap, "angular map at step %d" % t)
        deviated += int((np.abs(hvn.reshape(E, H, 2) - pv * 1.0).max(axis=-1) > 0.05).sum())
    assert deviated > 100


def test_crowd_sim_env_reset_and_steps(gpu, golden_dir):
    """'CrowdSim-v0' as an env (crowd_sim/__init__.py:3-6, crowd_sim.py:626-722): reset(phase='test', test_case=0) of ten
    envs = the reference's own test cases 0..9 (tests/golden/golden_crowd_reset.npz) -- the pedestrians' observable states,
    their ragged counts, the static obstacles as pedestrians, and the robot's angular local map equal to what the
    reference's reset() RETURNED (1e-12).  Then 20 steps in closed loop against the oracle's functions composed per env with
    ITS humans only (ragged lists, the robot appended for the humans that see it): reward, info, done, every agent state
    and the local map bit for bit."""
    import crowd_sim
    from nav_gym_amd import crowd
    d = np.load(os.path.join(golden_dir, "golden_crowd_reset.npz"))
    E = 10
    env = crowd_sim.make("CrowdSim-v0", num_envs=E, device=gpu.dev)
    ob, local_map = env.reset(phase="test", test_case=0)
    nh = ob["n_humans"].cpu().numpy()
    lm = local_map.cpu().numpy()
    for e in range(E):
        assert str(d["phase"][e]) == "test" and int(d["case"][e]) == e
        hg = d["humans_%d" % e]
        assert nh[e] == len(hg)
        _eq(ob["humans"][e, :nh[e]].cpu().numpy(), np.stack([hg[:, 0], hg[:, 1], hg[:, 4], hg[:, 5], hg[:, 7]], 1), "humans of case %d" % e)
        ks = int(ob["n_static"][e])
        _eq(ob["static"][e, :ks].cpu().numpy(), d["static_%d" % e], "static obstacles of case %d" % e)
        np.testing.assert_allclose(lm[e], d["local_map_%d" % e], rtol=0, atol=1e-12)
    assert len(set(nh.tolist())) >= 3 and env.case_counter["test"] == E
    # ---- closed loop against the oracle, env by env
    st = env.stepper
    params, mp = env._params(), env._map_params()
    orca_p = dict(time_step=params["time_step"], neighbor_dist=10, time_horizon=5, time_horizon_obst=5, max_neighbors=10)
    dt = float(params["time_step"])
    h = [st.h[e, :nh[e]].cpu().numpy().copy() for e in range(E)]
    r = st.r.cpu().numpy().copy()
    sees = [st.sees_robot[e, :nh[e]].cpu().numpy() for e in range(E)]
    verts = [st.verts[e, : int(st.n_obst[e])].cpu().numpy() for e in range(E)]
    free = st.free_map.cpu().numpy()
    gt = np.zeros(E)
    rng = np.random.default_rng(3)
    moved = 0
    for t in range(20):
        act = np.stack([rng.uniform(0.2, 1.0, E), rng.uniform(-0.3, 0.3, E)], axis=1)
        ob, local_map, reward, done, info = env.step(_t(gpu, act))
        for e in range(E):
            n = nh[e]
            hact = np.zeros((n, 2))
            for k in range(n):
                order = [k] + [j for j in range(n) if j != k]
                A = n + int(sees[e][k])
                ag = np.zeros((1, A, 6))
                ag[0, :n, :4] = h[e][order, :4]; ag[0, :n, 4] = h[e][order, 4] + 0.01; ag[0, :n, 5] = h[e][k, 5]
                if sees[e][k]:
                    ag[0, n, :4] = r[e, :4]; ag[0, n, 4] = r[e, 4] + 0.01; ag[0, n, 5] = h[e][k, 5]
                vel = h[e][k, 6:8] - h[e][k, 0:2]; sp = np.sqrt(vel[0] * vel[0] + vel[1] * vel[1])
                pv = (vel / sp if sp > 1 else vel)[None]
                _, a1 = ref.crowd_orca(orca_p, ag, pv, verts[e][None] if len(verts[e]) else None, None,
                                       np.array([len(verts[e])], np.int32) if len(verts[e]) else None,
                                       np.zeros(1, np.int32) if len(verts[e]) else None, h[e][k:k + 1, 8])
                hact[k] = a1[0]
            npose, nvel = ref.crowd_agent_step(np.array([[r[e, 0], r[e, 1], r[e, 8]]]), act[e:e + 1], dt)
            robot10 = np.array([[r[e, 0], r[e, 1], npose[0, 0], npose[0, 1], nvel[0, 0], nvel[0, 1], r[e, 6], r[e, 7], r[e, 4], act[e, 1]]])
            rew, dn, inf, md = ref.crowd_check(params, free[e:e + 1], robot10, h[e][None, :, :5], gt[e:e + 1])
            r[e, 0:2] = npose[0, 0:2]; r[e, 2:4] = nvel[0]; r[e, 8] = npose[0, 2]
            hpn, hvn = ref.crowd_agent_step(np.stack([h[e][:, 0], h[e][:, 1], h[e][:, 8]], -1), hact, dt)
            moved += int((np.abs(hvn).max(axis=1) > 0.05).sum())
            h[e][:, 0:2] = hpn[:, :2]; h[e][:, 2:4] = hvn; h[e][:, 8] = hpn[:, 2]
            gt[e] += dt
            amap = ref.crowd_angular_map(mp, np.array([[r[e, 0], r[e, 1], r[e, 8], r[e, 4]]]),
                                         verts[e][None] if len(verts[e]) else np.zeros((1, 0, 4, 2)),
                                         np.array([len(verts[e])], np.int32))
            assert float(reward[e]) == float(rew[0]) and int(info[e]) == int(inf[0]) and bool(done[e]) == bool(dn[0]), (t, e)
            _eq(st.h[e, :n].cpu().numpy(), h[e], "pedestrians of env %d at step %d" % (e, t))
            _eq(st.r[e].cpu().numpy(), r[e], "robot of env %d at step %d" % (e, t))
            _eq(local_map[e].cpu().numpy(), amap[0], "angular map of env %d at step %d" % (e, t))
            if n < st.h.shape[1]:                     # padding rows never move
                assert float(st.h[e, n:, 0].min()) >= 1e6
    assert moved > 200
    ob2, _ = env.reset(phase="test")                  # the next ten cases
    assert env.case_counter["test"] == 2 * E and not gpu.torch.equal(ob2["humans"], ob["humans"])


@pytest.mark.parametrize("shape", [(63, 64), (64, 65), (97, 131), (128, 129), (150, 90), (200, 200)])
def test_plan_random_costmaps_of_many_shapes(gpu, shape):
    """navsim_plan's breadth-first search on bitmaps (round 4) against the oracle on random costmaps whose widths sit around the
    64-bit word boundaries and whose sizes lie on both sides of the word-per-thread limit (128 x 128), square and not: sparse,
    dense and wall-with-a-gap obstacle patterns, unreachable goals, start == goal.  Waypoints, counts, path cells and path
    length identical.  (profiles/_diag/plan_fuzz.py is the wide form: 34 560 queries, profiles/r04_soak/plan_fuzz.txt.)"""
    torch = gpu.torch
    Hc, Wc = shape
    rng = np.random.default_rng(Hc * 1000 + Wc)
    n_maps, Q, res = 3, 30, 0.25
    cost = np.zeros((n_maps, Hc, Wc), np.uint8)
    cost[0] = rng.random((Hc, Wc)) < 0.08
    cost[1] = rng.random((Hc, Wc)) < 0.33
    for _ in range(max(Hc, Wc) // 6):
        if rng.random() < 0.5:
            r0 = rng.integers(1, Hc - 1); cost[2, r0, :] = 1; cost[2, r0, rng.integers(0, Wc)] = 0
        else:
            c0 = rng.integers(1, Wc - 1); cost[2, :, c0] = 1; cost[2, rng.integers(0, Hc), c0] = 0
    mi = np.repeat(np.arange(n_maps, dtype=np.int32), Q)
    start = np.stack([rng.uniform(0, Wc * res, n_maps * Q), rng.uniform(0, Hc * res, n_maps * Q)], 1)
    goal = np.stack([rng.uniform(0, Wc * res, n_maps * Q), rng.uniform(0, Hc * res, n_maps * Q)], 1)
    goal[::13] = start[::13]
    joined = 0
    for interval, P in ((2.0, 64), (0.6, 16)):
        exp = ref.plan(cost, start, goal, interval, max_wp=P, res_c=res, map_index=mi)
        got = gpu.sim.plan(_t(gpu, cost), _t(gpu, start), _t(gpu, goal), interval, max_wp=P, res_c=res, map_index=_t(gpu, mi))
        live = np.arange(P)[None, :] < exp[1][:, None]
        _eq(got[1].cpu().numpy(), exp[1], "waypoint counts")
        _eq(got[2].cpu().numpy(), exp[2], "path cells")
        _eq(got[3].cpu().numpy(), exp[3], "path length")
        _eq(got[0].cpu().numpy()[live], exp[0][live], "waypoints")
        joined += int((exp[1] > 0).sum())
    assert joined > 20


def test_config1_single_env_64_beams(gpu):
    """BASELINE config 1: 1 env, 64-beam lidar, 100x100 static map, no pedestrians."""
    cfg = gpu.lib.default_config(n_envs=1, map_h=100, map_w=100, n_spawn=4, auto_reset=0, seed=7)
    gpu.world.lidar_full_circle(cfg, 64)
    occ = gpu.world.make_maps(1, 100, 7)
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=0, steps=80, seed=2):
        _eq(go, ro, "obs at step %d" % t)
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))


def test_scan_noise_statistics(gpu):
    """Noise cannot be bit-compared with numpy's global stream: check that it is
    zero-mean with the requested std, applied only where range != range_max (env.py:437-440)."""
    E, size = 64, 200
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, n_spawn=4, add_scan_noise=1, seed=99)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 99)
    from nav_gym_amd import robots
    arrays = gpu.world.make_world(cfg, occ, device=gpu.dev, noise_std_range=(0.03, 0.03))
    arrays["scan_threshold"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "threshold_footprint")))
    arrays["scan_discomfort"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "discomfort_threshold_footprint")))
    noisy = gpu.sim.NavSim(cfg, arrays).reset_obs().cpu().numpy()[:, :1081]
    cfg.add_scan_noise = 0
    clean = gpu.sim.NavSim(cfg, arrays).reset_obs().cpu().numpy()[:, :1081]
    d = (noisy - clean)[clean != 25.0]
    assert abs(d.mean()) < 1e-3 and abs(d.std() - 0.03) < 1e-3
    assert np.array_equal(noisy[clean == 25.0], clean[clean == 25.0])
    k = np.mean(d ** 4) / d.var() ** 2
    assert 2.8 < k < 3.2                              # Gaussian kurtosis


def test_env_wrapper_gym_api(gpu):
    """gym.make('NavGym-v0')-style usage: reference shapes / dtypes for one arena, batched tensors
    for many, HER batch API consistent with step()."""
    import nav_gym_env
    env = nav_gym_env.make("NavGym-v0", map_size=400, num_humans=5, seed=3)
    obs = env.reset()
    assert obs["observation"].shape == (519,) and obs["observation"].dtype == np.float64
    assert obs["achieved_goal"].shape == (2,) and obs["desired_goal"].shape == (2,)
    assert len(env.humans) == 5 and env.map_info["data"].shape == (400, 400)
    o2, r, d, info = env.step(env.action_space.sample())
    assert isinstance(r, float) and isinstance(d, bool)
    assert set(info) == {"is_success", "is_crash", "distance"} and info["is_success"].dtype == np.float32
    if not info["is_crash"]:      # after a crash the returned obs is the reverted pose's (env.py:707-723)
        assert abs(env.compute_reward(np.zeros(2), o2) - r) < 1e-4      # float32 obs row vs float64 state
    assert env.compute_done(o2) == d
    assert abs(env.compute_info(o2)["distance"] - info["distance"]) < 1e-4
    assert abs(env.robot.px - o2["achieved_goal"][0]) < 1e-5
    img = env.render(mode="rgb_array")                      # env.py:833-1050 for the arena (render.py)
    assert img.shape == (800, 800, 3) and img.dtype == np.float32 and (img == 0).any() and (img == 1).any()
    # the debug text (env.py:182-217, 1035-1046): twelve lines, the six reward terms sum to the step's reward
    lines = env.render_obs_txt.split("\n") + env.render_reward_txt.split("\n")
    assert len(lines) == 12 and lines[0] == "t: %d" % (0 if d else 1) and lines[6].startswith("reward_success: ")
    assert lines[2] == "pose: ({:.2f} {:.2f})".format(o2["observation"][-5], o2["observation"][-4])
    plain = env.render(mode="rgb_array", text=False)
    red = (img == np.array([0, 0, 1], np.float32)).all(axis=2) & ~(plain == np.array([0, 0, 1], np.float32)).all(axis=2)
    assert red[30:52, 50:120].any() and red[580:602, 50:300].any() and not red[:, 700:].any()
    if not info["is_crash"] and not d:
        assert abs(sum(env._make_render_txt(0).values()) - r) < 1e-4
    from nav_gym_amd import export                          # ros_env.py:65-185 field lists
    assert export.reset_map_fields(env)["data"].shape == (400, 400)
    assert len(export.strict_update_fields(env)["humans"]) == 5
    # batched
    benv = nav_gym_env.make("NavGym-v0", num_envs=32, map_size=200, n_beams=1081, num_scan_stack=2, seed=4)
    bo = benv.reset()
    assert tuple(bo["observation"].shape) == (32, 2 * 1081 + 7) and bo["observation"].is_cuda
    acts = np.tile([[0.4, 0.1]], (32, 1))
    for _ in range(5):
        bo, br, bd, binfo = benv.step(acts)
    assert tuple(br.shape) == (32,) and bd.dtype == gpu.torch.bool
    assert benv.render(arena=17).shape == (800, 800, 3)
    rr = benv.compute_rewards(acts, {k: v for k, v in bo.items()})
    assert rr.shape == (32,)
    done_again = benv.compute_terminals(bo)
    assert done_again.shape == (32,)
    # pedestrians follow planned paths (plan_paths defaults to True): resident costmap, multi-waypoint routes
    assert "costmap" in env.sim.t and tuple(env.sim.t["costmap"].shape) == (1, 80, 80)
    assert int(env.sim.t["ped_n_waypoints"][0, :5].max()) > 1
    renv = nav_gym_env.make("NavGym-v0", num_envs=16, map_size=300, num_humans=4, seed=5, randomize_maps=True,
                            min_goal_dist=3, max_goal_dist=8)
    renv.reset()
    assert renv.cfg.regen_plan == 1
    straight = gpu.torch.tensor([[0.5, 0.0]] * 16, dtype=gpu.torch.float64, device=gpu.dev)
    finished = 0
    for _ in range(60):
        _, _, bd, _ = renv.step(straight)
        finished += int(bd.sum())
    assert finished > 0 and int(renv.sim.t["episode"].sum()) == finished
    # one arena in the layout RosEnv puts on the wire (ros_env.py:65-185)
    from nav_gym_amd import export
    m = export.reset_map_fields(benv, arena=3)
    assert m["data"].shape == (200, 200) and set(np.unique(m["data"])) <= {0, 100} and m["data"][0, 0] == 100
    su = export.strict_update_fields(env, arena=0)
    assert len(su["humans"]) == 5 and su["scan"]["ranges"].shape == (512,) and su["footprint"].shape == (4, 2)
    assert abs(su["pose"]["position"][0] - env.robot.px) < 1e-12
    assert np.allclose(su["scan"]["ranges"], env.prev_obs["observation"][:512])
    # pedestrians driven by the HumanPolicy actor on the device
    penv = nav_gym_env.make("NavGym-v0", num_envs=8, map_size=400, num_humans=6, seed=6, pedestrian_model="policy",
                            policy_weights=_policy_weights_random(3), indoor_ratio=0.0)
    penv.reset()
    p0 = penv.sim.t["ped_pose"].clone()
    for _ in range(5):
        penv.step(np.tile([[0.2, 0.0]], (8, 1)))
    moved = (penv.sim.t["ped_pose"][..., :2] - p0[..., :2]).norm(dim=2)
    assert float(moved.max()) > 0.05 and bool(gpu.torch.isfinite(penv.sim.t["ped_pose"]).all())


def test_env_graph_replay_equals_eager_steps(gpu):
    """NavGymEnv(use_graphs=True) -- the default when randomize_maps makes a step several launches -- replays navsim_step +
    navsim_regen + navsim_replan as ONE captured hipGraph per observation-buffer parity (NavSim.enable_graphs, behind
    navsim_prepare): observations, rewards, flags, info and the state equal the eager calls step for step, and the
    counters of the reset path (env.counters()) agree."""
    import nav_gym_env
    torch = gpu.torch
    envs = [nav_gym_env.make("NavGym-v0", num_envs=24, map_size=300, num_humans=4, seed=9, randomize_maps=True, indoor_ratio=0.0,
                             min_goal_dist=3, max_goal_dist=8, use_graphs=ug) for ug in (True, False)]
    assert envs[0]._graphed is False and envs[1]._graphed is False
    o = [e.reset() for e in envs]
    assert envs[0]._graphed and not envs[1]._graphed
    _eq(o[0]["observation"].cpu().numpy(), o[1]["observation"].cpu().numpy(), "reset observation")
    g = torch.Generator(device=gpu.dev); g.manual_seed(2)
    finished = 0
    for t in range(70):
        act = torch.rand((24, 2), generator=g, device=gpu.dev, dtype=torch.float64)
        act[:, 0] = 0.5 if t % 5 else act[:, 0] * 0.5
        act[:, 1] = act[:, 1] * 1.28 - 0.64
        res = [e.step(act) for e in envs]
        for k in ("observation", "achieved_goal", "desired_goal"):
            _eq(res[0][0][k].cpu().numpy(), res[1][0][k].cpu().numpy(), "%s at step %d" % (k, t))
        _eq(res[0][1].cpu().numpy(), res[1][1].cpu().numpy(), "reward at step %d" % t)
        _eq(res[0][2].cpu().numpy(), res[1][2].cpu().numpy(), "done at step %d" % t)
        finished += int(res[1][2].sum())
    for k in ("robot_pose", "ped_pose", "ped_waypoints", "episode", "n_peds"):
        _eq(envs[0].sim.t[k].cpu().numpy(), envs[1].sim.t[k].cpu().numpy(), "state " + k)
    c0, c1 = envs[0].counters(), envs[1].counters()
    assert finished > 3 and c0 == c1 and c1["regen_served"] >= finished and c1["regen_unserved"] == 0, (finished, c0, c1)
    assert envs[1].counters() == {k: 0 for k in abi.COUNTERS}            # counters() reads and clears


def test_env_reset_runs_on_the_device_and_matches_the_oracle(gpu, monkeypatch):
    """NavGymEnv.reset() of 4096 arenas x 500x500 maps: no map is generated on the host (world.make_maps must not
    be called); maps, fields, start / goal pairs joined by planned paths, pedestrians with planned waypoints, the
    per-episode env_param draws (num_humans, obstacle_number, corridor_width, iterations, scan_noise_std:
    env.py:281-292) and the first observations of sampled arenas equal navsim_regen_cpu run on ONE arena with
    the same global index, bit for bit.  A second reset() draws new maps."""
    import nav_gym_amd
    from nav_gym_amd import world
    def boom(*a, **k):
        raise AssertionError("reset() generated maps on the host")
    monkeypatch.setattr(world, "make_maps", boom)
    monkeypatch.setattr(world, "make_world", boom)
    kw = dict(nav_gym_amd.DEFAULT_KWARGS)
    kw["env_param_range"] = dict(kw["env_param_range"], obstacle_number=([6, 12], 'int'))
    E, size = 4096, 500
    env = nav_gym_amd.NavGymEnv(num_envs=E, n_beams=1081, map_size=size, seed=11, n_spawn=8, **kw)
    obs = env.reset()
    o = obs["observation"].cpu().numpy()
    assert o.shape == (E, 1088) and np.isfinite(o).all()
    st = env.sim.numpy_state("robot_pose", "robot_goal", "spawn_pose", "spawn_goal", "n_peds", "ped_pose", "ped_v_pref",
                             "ped_has_legs", "ped_waypoints", "ped_n_waypoints", "scan_noise_std", "episode")
    assert st["n_peds"].min() >= 5 and st["n_peds"].max() <= 15 and len(np.unique(st["n_peds"])) > 5
    assert 0.0 <= st["scan_noise_std"].min() and st["scan_noise_std"].max() <= 0.05 and st["scan_noise_std"].std() > 0.005
    occupied = np.array([env.sim.occupancy(e).mean() for e in range(24)])
    assert (occupied > 0.3).any() and (occupied < 0.2).any()          # indoor_ratio 0.5: both kinds of map
    d = np.linalg.norm(st["robot_goal"] - st["robot_pose"][:, :2], axis=1)
    assert np.mean((d > 10.0) & (d < 20.0)) > 0.9                      # env.py:748-783 (fallbacks are rare)
    assert "rect_table" in env.sim.t and (st["ped_n_waypoints"] > 1).mean() > 0.5
    # the scans of `o` carry each arena's freshly drawn scan noise (env.py:437-440; the oracle has no noise
    # generator): compare the noise-free first observation of the same state
    env.sim.cfg.add_scan_noise = 0
    o_clean = env.sim.reset_obs().cpu().numpy()
    env.sim.cfg.add_scan_noise = 1
    assert np.array_equal(o[:, 1081:], o_clean[:, 1081:]) and 0 < np.abs(o[:, :1081] - o_clean[:, :1081]).max() < 0.5
    cfg = env.sim.cfg
    for e in (0, 1, 777, 2048, 4095):
        c1 = cfg.copy(); c1.n_envs = 1; c1.env_index_base = int(e); c1.regen_cap = 1
        host = {k: v.cpu().numpy() for k, v in gpu.world.empty_world(c1, device="cpu", plan_paths=True).items()}
        host["field"] = np.zeros((1, size, size), np.float32)
        host["scan_threshold"] = env.scan_threshold.cpu().numpy(); host["scan_discomfort"] = env.scan_discomfort_threshold.cpu().numpy()
        r = ref.RefSim(c1, host)
        r.out["done"][:] = 1
        ro = r.regen()
        _eq(o_clean[e:e + 1], ro, "first observation of arena %d" % e)
        for k, v in st.items():
            _eq(v[e:e + 1], r.a[k], "arena %d state %s" % (e, k))
        _eq(env.sim.occupancy(e), (r.a["field"][0] == 0).astype(np.uint8), "arena %d map" % e)
    first_maps = [env.sim.occupancy(e) for e in range(4)]
    env.reset()
    assert all(not np.array_equal(env.sim.occupancy(e), first_maps[e]) for e in range(4))
    assert (env.sim.t["episode"] == 1).all()
    out = env.step(np.tile([[0.3, 0.1]], (E, 1)))
    assert out[0]["observation"].shape == (E, 1088)


@pytest.fixture(scope="module", params=["c2", "c3", "c4", "c5"])
def full_c2(gpu, request):
    """BASELINE configs at full per-GPU size, exactly as bench.py builds them (built once each):
    c2 = 4096 arenas x 1081 beams x 500x500 maps; c3 = c2 + 20 social-force pedestrians;
    c4 = 2048 arenas x 1000x1000 maps (one GPU's share of the 16384); c5 = 512 arenas (one GPU's share of
    the 4096), Husky kinematics, 20 pedestrians, a NEW random map for every finished arena (navsim_regen
    after every step)."""
    import bench
    wl = dict(bench.WORKLOADS[request.param]); wl["field"] = "u16t"
    cfg, sim, arrays, occ = bench.build_sim(wl, 0, wl["envs"])
    sim.regen_every_step = bool(wl.get("regen"))
    yield cfg, sim, arrays, occ
    del sim, arrays
    gpu.torch.cuda.empty_cache()


def test_full_size_c2_properties_and_sampled_oracle(gpu, full_c2):
    """Full-size run (the oracle cannot step 4096 arenas in seconds): size-independent properties on
    every arena + bit-exact oracle comparison on a sample of arenas (arenas are independent, so a
    sampled arena's rows must equal an oracle run of just that arena)."""
    torch = gpu.torch
    cfg, sim, arrays, occ = full_c2
    E, B = cfg.n_envs, cfg.n_beams
    sample = np.array([0, 1, 7, 63, 64, 511, 1000, 2047, 2048, 3000, 4094, 4095])
    sample = np.unique(np.minimum(sample, E - 1))
    regen = sim.regen_every_step
    if regen:
        # c5: put a few sampled robots ON their goals so that they finish at step 0 and are regenerated -- new map,
        # distance field, spawn table, pedestrians, first observation -- then keep stepping on the new map
        sample = np.unique(np.concatenate([sample, [2, 3, 100, 300]]))
        on_goal = torch.as_tensor([2, 3, 100, 300, 511], device=gpu.dev)
        sim.t["robot_goal"][on_goal] = sim.t["robot_pose"][on_goal, :2]
    sub_cfg = cfg.copy(); sub_cfg.n_envs = 1
    refs = []
    for e in sample:                                  # one single-arena oracle per sampled arena
        c1 = sub_cfg.copy(); c1.env_index_base = int(e)
        host = {}
        for k, t in sim.t.items():
            if k in ("field", "field_overflow", "rect_table", "beam_table", "arena_cost", "launch_order", "regen_ws", "counters"):
                continue                              # (counters, job board: per simulator, not per arena)
            a = t.detach().cpu().numpy()
            host[k] = a if k in ("scan_threshold", "scan_discomfort") else a[e:e + 1]
        host["field"] = ref.build_dt(occ[e:e + 1])
        r = ref.RefSim(c1, host)
        r.obs[r.cur][...] = sim.obs[e:e + 1].cpu().numpy()
        refs.append(r)
    rng = np.random.default_rng(11)
    g = torch.Generator(device=gpu.dev); g.manual_seed(3)
    total_done = regenerated_samples = 0
    for t in range(12):
        act = torch.rand((E, 2), generator=g, device=gpu.dev, dtype=torch.float64)
        act[:, 0] *= 0.5; act[:, 1] = act[:, 1] * 1.28 - 0.64
        if t % 4 == 1 and not regen:
            act[:, 0] = 0.5; act[:, 1] = 0.0
        obs, out = sim.step(act)
        o = obs.cpu().numpy(); scan = o[:, :B]
        assert np.isfinite(o).all() and (scan >= 0).all() and (scan <= 25.0).all()
        done = out["done"].cpu().numpy().astype(bool)
        succ = out["is_success"].cpu().numpy() > 0; crash = out["is_crash"].cpu().numpy() > 0
        assert np.array_equal(done, succ | crash)
        assert np.array_equal(succ, out["distance"].cpu().numpy() < cfg.distance_threshold)
        total_done += int(done.sum())
        a_host = act.cpu().numpy()
        for e, r in zip(sample, refs):
            ro, rout = r.step(a_host[e:e + 1])
            _eq(o[e:e + 1], ro, "arena %d obs at step %d" % (e, t))
            for k in rout:
                _eq(out[k][e:e + 1].cpu().numpy(), rout[k], "arena %d %s at step %d" % (e, k, t))
        if regen:
            # navsim_regen serves the finished arenas lowest index first, at most regen_cap of them: below the cap
            # an arena's fate does not depend on the others and the single-arena oracles stay comparable
            assert int(done.sum()) <= cfg.regen_cap, "more arenas finished than regen_cap: raise the cap of this test"
            o2 = sim.regen().cpu().numpy()
            for e, r in zip(sample, refs):
                _eq(o2[e:e + 1], r.regen(), "arena %d obs after regen at step %d" % (e, t))
                regenerated_samples += int(done[e])
            if t in (0, 11):
                gs = sim.numpy_state("robot_pose", "robot_goal", "spawn_pose", "spawn_goal", "ped_pose", "ped_v_pref",
                                     "ped_has_legs", "ped_waypoints", "ped_n_waypoints", "episode", "steps")
                for e, r in zip(sample, refs):
                    for k, v in gs.items():
                        _eq(v[e:e + 1], r.a[k], "arena %d state %s after regen at step %d" % (e, k, t))
    assert total_done > 0
    if regen:
        assert regenerated_samples >= 4, "no sampled arena went through navsim_regen"


def test_full_size_c2_determinism_and_shard_invariance(gpu, full_c2):
    """Two half-shards (env_index_base 0 and 2048) reproduce the full batch bit for bit, and a
    repeated run reproduces itself: arenas never interact and all randomness is keyed by the
    global arena index."""
    torch = gpu.torch
    cfg, sim, arrays, occ = full_c2
    E = cfg.n_envs
    state0 = {k: v.clone() for k, v in sim.t.items() if k != "regen_ws"}
    obs0 = sim.obs.clone()
    g = torch.Generator(device=gpu.dev); g.manual_seed(5)
    acts = torch.rand((6, E, 2), generator=g, device=gpu.dev, dtype=torch.float64)
    acts[..., 0] *= 0.5; acts[..., 1] = acts[..., 1] * 1.28 - 0.64

    def run(lo, hi, base):
        c = cfg.copy(); c.n_envs = hi - lo; c.env_index_base = base
        arr = {}
        for k, v in state0.items():
            if k in ("arena_cost", "launch_order", "counters"):      # scheduling hints / totals: each NavSim owns its own
                continue
            if k == "field":
                per = v.numel() // E
                arr[k] = v.reshape(E, per)[lo:hi].reshape(-1).clone()
            elif k in ("scan_threshold", "scan_discomfort", "beam_table"):
                arr[k] = v.clone()
            else:
                arr[k] = v[lo:hi].clone()
        s = gpu.sim.NavSim(c, arr)
        s.obs_buf[s.cur].copy_(obs0[lo:hi])
        outs = []
        for t in range(6):
            o, out = s.step(acts[t, lo:hi])
            outs.append((o.clone(), {k: v.clone() for k, v in out.items()}))
        return outs
    full = run(0, E, 0)
    again = run(0, E, 0)
    a, b = run(0, E // 2, 0), run(E // 2, E, E // 2)
    for t in range(6):
        assert torch.equal(full[t][0], again[t][0])
        assert torch.equal(full[t][0], torch.cat([a[t][0], b[t][0]]))
        for k in full[t][1]:
            assert torch.equal(full[t][1][k], torch.cat([a[t][1][k], b[t][1][k]])), k


@pytest.mark.parametrize("size,live,step_block", [(200, 0, 0), (253, 0, 256), (300, 200, 1024)])
def test_regen_writes_the_rect_records_of_outdoor_maps_from_the_generator(gpu, size, live, step_block):
    """A world of outdoor maps that keeps rect records: navsim_regen writes the records of a new map from the generator's
    rectangles (regen_rect_records: no search, no verification pass).  Every valid record reproduces the exact field on
    every cell of its tile, almost every tile has one, and the rollout (whose first observations after a regen march
    through the records staged in LDS) stays identical to the oracle's.  253: ragged edge tiles; 300 / 200: a live map
    inside a larger arena (map_size="reference")."""
    E, N = 24, 4
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_SFM, n_spawn=6,
                                 auto_reset=1, seed=23, field_format=abi.FIELD_U16T, regen_cap=8, min_goal_dist=3.0,
                                 max_goal_dist=8.0, spawn_clearance=0.9, ped_min_robot_dist=2.0, ped_min_goal_dist=4.0,
                                 regen_indoor_ratio=0.0, step_block=step_block)
    if live:
        cfg.outdoor_map_size = live
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 23)
    regenerated = 0
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=3, steps=40, seed=9):
        _eq(go, ro, "obs at step %d" % t)
        assert "rect_table" in g.t
        n_done = int(rout["done"].sum())
        _eq(g.regen().cpu().numpy(), r.regen(), "obs after regen at step %d" % t)
        regenerated += min(n_done, 8)
        if n_done:
            gs = g.numpy_state()
            for k, v in r.a.items():
                if k in gs and k not in ("field", "field_overflow", "rect_table", "rect_index"):
                    _eq(gs[k], v, "state %s after regen at step %d" % (k, t))
            d2, valid = _decode_rect_table(g.t["rect_table"].cpu().numpy(), size, size)
            exact = np.rint(r.a["field"].astype(np.float64) ** 2).astype(np.int64)
            assert np.array_equal(d2[valid], exact[valid]), "a record differs from the exact field at step %d" % t
            assert valid.mean() > 0.93          # 200 x 200 cells with 10 boxes: 0.97; the bench's 500 x 500: 0.995
    assert regenerated >= 6


def _decode_rect_table(table, H, W):
    """Host evaluation of navsim_build_rects records: -> (d2 int64 [E,H,W], valid bool [E,H,W])."""
    E = table.shape[0]
    tpr = (W + 7) // 8
    rec = table.view(np.uint32).reshape(E, -1, 4)
    yy, xx = np.mgrid[0:H, 0:W]
    tile = (yy // 8) * tpr + (xx // 8)
    r = rec[:, tile]                                            # [E,H,W,4]
    def sx(w): return (w & 0xFFFF).astype(np.int16).astype(np.int64)
    def sy(w): return (w >> 16).astype(np.int16).astype(np.int64)
    def dist2(lo, hi):
        ddx = np.maximum(0, np.maximum(sx(lo) - xx, xx - sx(hi)))
        ddy = np.maximum(0, np.maximum(sy(lo) - yy, yy - sy(hi)))
        return ddx * ddx + ddy * ddy
    d2 = np.minimum(dist2(r[..., 0], r[..., 1]), dist2(r[..., 2], r[..., 3]))
    valid = (r[..., 0] & 0xFFFF) != 0x7FFF
    return d2, valid


@pytest.mark.parametrize("size,indoor,fmt", [(100, 0.0, abi.FIELD_U16T), (253, 0.5, abi.FIELD_U16T), (500, 0.0, abi.FIELD_U16T),
                                             (500, 1.0, abi.FIELD_F32), (1000, 0.0, abi.FIELD_U16T)])
def test_rect_table_reproduces_field(gpu, size, indoor, fmt):
    """navsim_build_rects: every VALID record gives the exact squared distance on all in-map cells of its 8x8
    tile (min over its two rectangles); both rectangles consist of occupied cells only; tiles without such a
    pair are marked invalid (the march then reads the field).  On the reference's kind of maps (unions of
    boxes / corridors) nearly every tile is valid.  1000x1000 outdoor maps have saturated cells (d2 >= 65535):
    the float32 plane supplies their exact distance."""
    n = 3 if size <= 500 else 1
    occ = gpu.world.make_maps(n, size, 31, indoor_ratio=indoor)
    field, f32, nsat = gpu.sim.build_field(_t(gpu, occ), fmt)
    table = gpu.sim.build_rects(_t(gpu, occ), field, fmt, f32 if fmt == abi.FIELD_U16T else None).cpu().numpy()
    d2, valid = _decode_rect_table(table, size, size)
    exact = np.rint(ref.build_dt(occ).astype(np.float64) ** 2).astype(np.int64)
    assert np.array_equal(d2[valid], exact[valid]), "a valid record disagrees with the exact field"
    frac = valid.mean()                     # small maps: a larger share of the tiles sits between several obstacles
    assert frac > (0.8 if size <= 100 else 0.93), "only %.3f of the cells lie in tiles with a valid record" % frac
    # rectangles are made of occupied cells (the upper-bound argument of kernels_rect.hpp relies on it)
    rec = table.view(np.uint32).reshape(n, -1, 4)
    rng = np.random.default_rng(0)
    for m in range(n):
        ok = np.where((rec[m, :, 0] & 0xFFFF) != 0x7FFF)[0]
        for t in rng.choice(ok, 200):
            for lo, hi in ((rec[m, t, 0], rec[m, t, 1]), (rec[m, t, 2], rec[m, t, 3])):
                x0, y0, x1, y1 = int(lo & 0xFFFF), int(lo >> 16), int(hi & 0xFFFF), int(hi >> 16)
                assert 0 <= x0 <= x1 < size and 0 <= y0 <= y1 < size
                assert occ[m, y0:y1 + 1, x0:x1 + 1].all()
    if size == 1000 and fmt == abi.FIELD_U16T:
        assert nsat > 0
        # without the float plane the tiles holding a saturated cell must come out invalid, never wrong
        t2 = gpu.sim.build_rects(_t(gpu, occ), field, fmt, None).cpu().numpy()
        d2b, validb = _decode_rect_table(t2, size, size)
        assert np.array_equal(d2b[validb], exact[validb]) and validb.sum() < valid.sum()


def test_rect_table_on_arbitrary_maps(gpu):
    """Maps that are NOT rectangle unions (random speckle, a disc, diagonal walls): fewer valid records, never a
    wrong one, and the fused step through such a table still equals the oracle bit for bit."""
    size, E = 160, 6
    rng = np.random.default_rng(5)
    occ = np.zeros((E, size, size), np.uint8)
    occ[:, :3] = 1; occ[:, -3:] = 1; occ[:, :, :3] = 1; occ[:, :, -3:] = 1
    yy, xx = np.mgrid[0:size, 0:size]
    occ[0][rng.random((size, size)) < 0.01] = 1
    occ[1][(yy - 80) ** 2 + (xx - 70) ** 2 < 30 ** 2] = 1
    occ[2][np.abs(yy - xx) < 2] = 1
    occ[3][(np.abs(yy + xx - size) < 2) & (xx > 40)] = 1
    occ[4][rng.random((size, size)) < 0.002] = 1
    occ[5][60:100, 50:120] = 1
    for e in range(E):                                   # keep a free patch for the robots
        occ[e, 20:45, 20:45] = 0
    field, f32, _ = gpu.sim.build_field(_t(gpu, occ), abi.FIELD_U16T)
    table = gpu.sim.build_rects(_t(gpu, occ), field, abi.FIELD_U16T, f32).cpu().numpy()
    d2, valid = _decode_rect_table(table, size, size)
    exact = np.rint(ref.build_dt(occ).astype(np.float64) ** 2).astype(np.int64)
    assert np.array_equal(d2[valid], exact[valid])
    assert valid[5].mean() > 0.95 and valid[0].mean() < 0.9          # a box: valid; speckle: mostly not
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, n_spawn=4, auto_reset=1, seed=8,
                                 field_format=abi.FIELD_U16T)
    gpu.world.lidar_1081(cfg)
    for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=0, steps=25, seed=2, min_goal_dist=1.0,
                                                     max_goal_dist=3.0, robot_clearance=0.4):
        assert "rect_table" in g.t
        _eq(go, ro, "obs at step %d" % t)
        for k in rout:
            _eq(gout[k], rout[k], "%s at step %d" % (k, t))


def _decode_rect_index(rows, H, W):
    """Host evaluation of the index form (navsim_build_rect_index): rows uint8 [E, R] -> the 16-byte records it stands for
    (uint32 [E, T, 4]) and which tiles carry an index."""
    E = rows.shape[0]
    T_ = ((H + 7) // 8) * ((W + 7) // 8)
    lst = rows[:, :2048].copy().view(np.uint32).reshape(E, 256, 2)
    pair = rows[:, 2048:2048 + 2 * T_].copy().view(np.uint16).reshape(E, T_)
    has = pair != 0xFFFF
    ia, ib = (pair & 0xFF).astype(np.int64), (pair >> 8).astype(np.int64)
    rec = np.zeros((E, T_, 4), np.uint32)
    ar = np.arange(E)[:, None]
    rec[..., 0:2] = lst[ar, ia]; rec[..., 2:4] = lst[ar, ib]
    return rec, has


@pytest.mark.parametrize("size,indoor", [(500, 0.0), (500, 1.0), (253, 0.5), (1000, 1.0)])
def test_rect_index_is_the_record_table(gpu, size, indoor):
    """Round 4, "map tiles staged through LDS": the index form of an arena's record table -- the distinct rectangles of its
    records and two list indices per tile, 10 KB for 500 x 500 -- names, for every tile that has an index, exactly the two
    rectangles of the tile's 16-byte record; every valid record of the reference's kind of maps gets one (the lists hold
    14 rectangles for an outdoor map, a few dozen to ~200 for a corridor map); closed maps are recognised."""
    n = 3 if size <= 500 else 1
    occ = gpu.world.make_maps(n, size, 9 + size, indoor_ratio=indoor)
    field, f32, _ = gpu.sim.build_field(_t(gpu, occ), abi.FIELD_U16T)
    table = gpu.sim.build_rects(_t(gpu, occ), field, abi.FIELD_U16T, f32)
    rows, n_rects = gpu.sim.build_rect_index(table, size, size)
    closed = gpu.sim.maps_closed(_t(gpu, occ))
    assert rows.shape[1] == abi.rect_index_row_bytes(size, size) == gpu.lib.load().navsim_rect_index_bytes(1, size, size)
    rec, has = _decode_rect_index(rows.cpu().numpy(), size, size)
    tab = table.cpu().numpy().view(np.uint32).reshape(n, -1, 4)
    valid = (tab[..., 0] & 0xFFFF) != 0x7FFF
    assert not (has & ~valid).any()                                  # an index only where there is a valid record
    _eq(rec[has], tab[has], "records named by the index rows")
    nr = n_rects.cpu().numpy()
    assert (nr >= 5).all()
    if (nr <= 255).all():
        assert np.array_equal(has, valid), "every valid record has an index"
    assert (closed.cpu().numpy() == 1).all(), "the reference's maps are closed"
    # a gap in the border wall: not closed
    occ2 = occ.copy(); occ2[0, :8, 40:60] = 0
    occ2[-1, 100, size - 3] = 0                                      # one cell of the ring's inner layer
    c2 = gpu.sim.maps_closed(_t(gpu, occ2)).cpu().numpy()
    assert c2[0] == 0 and c2[-1] == 0 and (c2[1:-1] == 1).all()


def test_open_maps_keep_the_bounds_test(gpu):
    """A world with a gap in a border wall is not closed (cfg.closed_maps = 0 after make_world): the LDS form of the march,
    which carries no bounds test, is not used, rays leave the map through the gap (range_max), and the rollout equals
    the oracle bit for bit.  The same world with the gap filled is closed and takes the LDS form: also bit for bit."""
    E, size = 12, 200
    for gap in (True, False):
        occ = gpu.world.make_maps(E, size, 61)
        if gap:
            occ[:, 60:140, :8] = 0                                   # the west wall is open for four metres
        cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, n_spawn=6, auto_reset=1, seed=61, field_format=abi.FIELD_U16T)
        gpu.world.lidar_1081(cfg)
        left_the_map = 0
        for t, go, gout, ro, rout, g, r in _rollout_pair(gpu, cfg, occ, n_peds=0, steps=30, seed=6, min_goal_dist=2.0,
                                                         max_goal_dist=6.0, robot_clearance=0.6):
            _eq(go, ro, "obs at step %d (gap %s)" % (t, gap))
            for k in rout:
                _eq(gout[k], rout[k], "%s at step %d" % (k, t))
            left_the_map += int((go[:, :1081] == 25.0).sum())
        assert "rect_index" in g.t and g.cfg.closed_maps == (0 if gap else 1)
        assert (left_the_map > 0) == gap
    # a WRONG assertion is refused where the world is assembled (round-4 advisor: nothing ever checked cfg.closed_maps):
    # NavSim verifies it once against the fields it was given (navsim_world_closed)
    occ = gpu.world.make_maps(E, size, 61)
    occ[3, 60:140, :8] = 0
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, n_spawn=6, auto_reset=1, seed=61, field_format=abi.FIELD_U16T)
    gpu.world.lidar_1081(cfg)
    arrays = gpu.world.make_world(cfg, occ, n_peds=0, device=gpu.dev, min_goal_dist=2.0, max_goal_dist=6.0, robot_clearance=0.6)
    from nav_gym_amd import robots
    for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
        arrays[key] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", name)))
    assert cfg.closed_maps == 0
    cfg.closed_maps = 1                                              # "every map of this world is closed": not true of arena 3
    with pytest.raises(ValueError, match="1 arenas have a map without a closed ring"):
        gpu.sim.NavSim(cfg, arrays)


@pytest.mark.parametrize("size,indoor", [(400, 0.0), (500, 1.0), (1000, 1.0)])
def test_costmap_and_planner_vs_oracle(gpu, size, indoor):
    """Reset path: costmap (env.py:312-332) and shortest-path waypoints (env.py:343-354, 1261-1277) on
    outdoor and corridor maps; several queries per map; bit-exact vs the oracle."""
    torch = gpu.torch
    n_maps, per = 3, 12
    occ = gpu.world.make_maps(n_maps, size, 3 + size, indoor_ratio=indoor)
    cost = gpu.sim.costmap(_t(gpu, occ))
    rc = ref.costmap(occ)
    _eq(cost.cpu().numpy(), rc, "costmap")
    rng = np.random.default_rng(size)
    start, goal, mi = [], [], []
    for m in range(n_maps):
        free = np.argwhere(rc[m] == 0)
        a = free[rng.integers(0, len(free), per)]; b = free[rng.integers(0, len(free), per)]
        start.append(np.stack([(a[:, 1] + 0.5) * 0.25, (a[:, 0] + 0.5) * 0.25], 1))
        goal.append(np.stack([(b[:, 1] + 0.5) * 0.25, (b[:, 0] + 0.5) * 0.25], 1))
        mi.append(np.full(per, m))
    start, goal, mi = np.concatenate(start), np.concatenate(goal), np.concatenate(mi).astype(np.int32)
    start[0] = [0.1, 0.1]                                # inside the border wall: no path
    for interval in (2.0, 5.0):
        gw, gn, gc, gl = gpu.sim.plan(cost, _t(gpu, start), _t(gpu, goal), interval, max_wp=8, map_index=_t(gpu, mi))
        rw, rn, rcells, rl = ref.plan(rc, start, goal, interval, max_wp=8, map_index=mi)
        _eq(gn.cpu().numpy(), rn, "n_wp"); _eq(gc.cpu().numpy(), rcells, "path cells")
        _eq(gl.cpu().numpy(), rl, "path length")
        for q in range(len(rn)):
            _eq(gw.cpu().numpy()[q, : rn[q]], rw[q, : rn[q]], "waypoints of query %d" % q)
        assert rn[0] == 0
        if True:
            assert (rn[1:] > 0).sum() > len(rn) // 3, rn


def test_sharded_envs_reproduce_the_single_shard(gpu):
    """sharding.make_sharded_env: two shards of 24 arenas (ranks 0 and 1 of a world of 2, built one after the other
    on this box's GPU) reset and step exactly like the first and second half of ONE 48-arena env: maps, spawns,
    pedestrians, per-episode draws and noise are keyed by the global arena index."""
    from nav_gym_amd import sharding
    import nav_gym_amd
    torch = gpu.torch
    kw = dict(map_size=200, n_beams=512, seed=21, num_humans=4, min_goal_dist=3, max_goal_dist=7, device=gpu.dev)
    full = nav_gym_amd.NavGymEnv(num_envs=48, **dict(nav_gym_amd.DEFAULT_KWARGS, **kw))
    shards = [sharding.make_sharded_env(48, rank=r, world_size=2, **kw) for r in (0, 1)]
    assert [s.cfg.env_index_base for s in shards] == [0, 24] and [s.num_envs for s in shards] == [24, 24]
    of = full.reset()["observation"]
    os_ = torch.cat([s.reset()["observation"] for s in shards])
    assert torch.equal(of, os_)
    g = torch.Generator(device=gpu.dev); g.manual_seed(1)
    for t in range(8):
        act = torch.rand((48, 2), generator=g, device=gpu.dev, dtype=torch.float64)
        act[:, 0] *= 0.5; act[:, 1] = act[:, 1] * 1.28 - 0.64
        o, r, d, info = full.step(act)
        parts = [s.step(act[24 * i:24 * i + 24]) for i, s in enumerate(shards)]
        assert torch.equal(o["observation"], torch.cat([p[0]["observation"] for p in parts])), t
        assert torch.equal(r, torch.cat([p[1] for p in parts])) and torch.equal(d, torch.cat([p[2] for p in parts]))
    assert torch.equal(full.sim.t["ped_pose"], torch.cat([s.sim.t["ped_pose"] for s in shards]))


@pytest.mark.parametrize("peds", [False, True])
def test_soak_reduced(gpu, peds):
    """The soak of profiles/_diag/soak.py (2 x 384 k env-steps there) at suite length: 2 passes (scan stack 1 and 2;
    with pedestrians: update inside the step and ahead of it) of 96 arenas x 150 steps (60 with pedestrians) through
    crash reverts and respawns, every output of every step identical to the oracle."""
    from soak import run_soak
    passes = run_soak(60 if peds else 150, 96, peds)
    assert len(passes) == 2 and all(d > 0 for _, _, d in passes), passes      # episodes did end and restart


def test_bench_collectives_through_rccl_on_one_gpu(gpu):
    """The branches of bench.py that only exist under the nccl (= RCCL) backend -- the process group bound to the device, the
    device-side barrier and MAX reduction of the timed region, sharding.RowGather's all_gather_into_tensor after every step
    (value_with_obs_gather) -- executed on this box's one GPU: NAVSIM_BENCH_FORCE_DIST=1 makes a process group of ONE rank.
    Nothing crosses a link, but it is RCCL's init, communicator and kernels that run (the gloo stand-ins of the other tests
    never reach these lines).  stdout must hold the ONE json line and nothing else: RCCL prints a version banner through C
    stdio, which bench.py keeps off the real stdout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NAVSIM_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "NAVSIM_BENCH_BACKEND", "NAVSIM_BENCH_ONE_GPU", "MASTER_PORT"):
        env.pop(k, None)                                   # (MASTER_PORT: bench.py takes a free one)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--envs", "512", "--steps", "6", "--warmup", "2", "--repeats", "1",
                        "--no-cpu-baseline", "--no-extras"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=900)
    ready = "navsim-bench: process group ready" in r.stderr
    if r.returncode != 0 and not ready and any(k in r.stderr for k in ("ncclSystemError", "ncclUnhandledCudaError", "ncclInternalError",
                                                                        "hipIpcGetMemHandle")):
        # a system-level RCCL failure AT COMMUNICATOR INIT -- no usable IPC, no device for the communicator -- before any
        # kernel of this library has run (bench.py prints the marker behind its first collective).  Behind the marker the same
        # error names would be a fault of ours surfacing at the next collective: that fails the test (round-5 advisor).
        pytest.skip("RCCL could not initialise on this box: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{"metric"'), r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["envs_per_gpu"] == 512
    g = out["obs_gather"]
    assert out["value_with_obs_gather"] > 0 and "RCCL" in g["collective"] and g["equal_shards"]
    assert g["bytes_total"] == g["bytes_per_rank"] == 512 * 1088 * 4


def test_bench_two_ranks_on_one_gpu(gpu):
    """bench.py --gpus 2 end to end through the HIP library: the script spawns both ranks itself; they share
    this box's only GPU (NAVSIM_BENCH_ONE_GPU) and rendezvous over gloo (the driver's 8-GPU runs use RCCL, one
    rank per GPU).  The line must say n_gpus = 2 and count both ranks' arenas."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NAVSIM_BENCH_ONE_GPU="1", NAVSIM_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--envs", "256", "--steps", "6", "--scaling", "weak",
                        "--warmup", "2", "--repeats", "1", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["envs_per_gpu"] == 256
    assert abs(out["value"] - 2 * 256 * 6 / (out["ms_per_step"] * 6e-3)) < 1e-6 * out["value"]
    assert out["roofline"]["kernel_ms"] > 0 and out["noise_off"]["value"] > 0
    assert out["scaling"] == "weak" and out["config"]["envs_total"] == 512
    assert out["value_with_obs_gather"] is None             # the gather is an RCCL call: not under this gloo stand-in
    # strong scaling, ragged: 301 arenas split 151 + 150 over the two ranks, every arena counted once
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scaling", "strong", "--total-envs",
                        "301", "--steps", "6", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["config"]["envs_total"] == 301 and out["config"]["envs_per_gpu"] == 151
    assert abs(out["value"] - 301 * 6 / (out["ms_per_step"] * 6e-3)) < 1e-6 * out["value"]
    # the default with N > 1: BOTH curves from one invocation -- value (= value_strong) on the fixed total, value_weak on
    # the per-GPU count, each with its own envs_total (round-3 verdict)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--total-envs", "600", "--envs", "256",
                        "--steps", "6", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["scaling"] == "strong" and out["config"]["envs_total"] == 600 and out["value_strong"] == out["value"]
    assert out["value_weak"] > 0 and out["weak"]["envs_total"] == 512 and out["weak"]["envs_per_gpu"] == 256
    assert abs(out["value_weak"] - 512 * 6 / (out["weak"]["ms_per_step"] * 6e-3)) < 1e-6 * out["value_weak"]
    # Round-4 verdict: no 8-GPU node has run this yet -- the first one must not produce a line with a hole in it.  Every
    # key the N > 1 line is read by, with the defaults the driver uses (no --envs / --total-envs: c2's 4096 strong, 4096 per
    # GPU weak); sizes kept small through --steps only.
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--repeats", "1",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "value_strong", "value_weak", "weak", "value_with_obs_gather"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["scaling"] == "strong"
    assert out["config"]["envs_total"] == 4096 and out["config"]["envs_per_gpu"] == 2048
    assert out["weak"]["envs_total"] == 8192 and out["weak"]["envs_per_gpu"] == 4096 and out["weak"]["scaling"] == "weak"
    assert out["value"] == out["value_strong"] > 0 and out["value_weak"] > 0
    assert abs(out["value"] - 4096 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]
    assert abs(out["value_weak"] - 8192 * 4 / (out["weak"]["ms_per_step"] * 4e-3)) < 1e-6 * out["value_weak"]
    assert out["roofline"]["kernel_ms"] > 0 and 0 < out["roofline"]["frac"] <= 1 and "valu" in out["roofline"]
    assert out["config"]["workload"].startswith("c2: 4096 arenas (strong scaling: 2048 on rank 0)")
    assert "other_workloads" not in out and "cpu_baseline" not in out       # N = 1 extras only


def test_bench_line_contract(gpu):
    """The one-GPU bench line in the driver's shape (--steps 20 --warmup 5): every key of the contract, the roofline
    object with its live kernel time and both readings of the SURVEY 8d figure, and the CPU baseline object."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--envs", "1024", "--cpu-seconds", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    out = json.loads(lines[-1])                                  # ONE JSON line, the last thing printed
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["steps"] == 20 and out["warmup"] == 5 and out["n_gpus"] == 1 and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["data"] == "synthetic" and "workload" in out["config"]
    roof = out["roofline"]
    # the contract's fraction is a fraction: SURVEY 8d bytes with the s_map of what the kernel reads (rect records,
    # 16 B per 8x8 cells) over the kernel's own time -- never above the peak, and not above it on the step's wall time
    assert roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and roof["bound"].startswith("valu-issue")
    assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    # what the kernel streams of a map: the index form of its record table, copied to LDS once per arena-step
    row = 256 * 8 + ((63 * 63 * 2 + 15) // 16) * 16
    assert abs(roof["s_map"] - row / 250000.0) < 1e-12 and abs(roof["algorithmic_bytes_per_env_step"] - (row + 4 * 1081 + 4 * 1092 + 96)) < 1e-6
    reps = roof["frac_by_representation"]
    assert abs(reps["index_rows_in_lds (round 4)"] - roof["frac"]) < 1e-9 and reps["occupancy_int8 (SURVEY 8d, s_map = 1)"] > reps["rect_records_16B_per_tile (rounds 2-3)"] > roof["frac"]
    assert abs(roof["achieved"] * 1e9 * roof["kernel_ms"] * 1e-3 - roof["algorithmic_bytes_per_launch"]) < 1e-3 * roof["algorithmic_bytes_per_launch"]
    assert 0.0 < roof["frac_of_ms_per_step"] <= roof["frac"] * 1.02
    assert 0 < roof["kernel_ms"] <= out["ms_per_step"] * 1.02 and roof["kernel_ms_from"]
    assert roof["frac_s_map_1"] > roof["frac"] and roof["note_s_map_1"]
    # counter figures are quoted only from a profile of the same sources and launch shape (1024 arenas here: none)
    assert roof["traffic"] is None and roof["traffic_unavailable"] and roof["hbm_frac_measured"] is None
    assert out["config"]["kernel_src_sha"] and 0.5 < out["config"]["rect_valid_tile_frac"] <= 1.0
    assert out["value_no_spinup"] > 0 and out["value_weak"] == out["value"]
    work = out["work"]
    assert abs(work["rays_per_s"] - out["value"] * 1081) < 1e-6 * work["rays_per_s"]
    assert work["worst_case_probes_per_env_step"] == 540500
    assert 2.0 < work["probes_per_ray_mean"] < 40.0 and work["probes_per_ray_p99"] >= work["probes_per_ray_mean"]
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["sample"]


def test_kernel_arguments_read_in_place(gpu):
    """The step kernels bind references into the kernarg segment instead of copying their struct arguments (round 6,
    kernels_step.hpp NAVSIM_KERNARGS): the layout that rests on -- arguments laid out like a struct's members -- is checked by a
    probe launch on this toolchain and device."""
    gpu.sim.debug_kernarg_layout()


def test_bench_c1_window(gpu):
    """BASELINE.json configs[0] (1 arena, 64 beams, 100 x 100, no pedestrians) in the bench's default line: `other_workloads.c1`
    with the GPU's microseconds per step (wall clock, device, kernel alone) and the oracle's on one CPU thread (round-5 review,
    item 5).  The default line's extras run at the configured size only; `--only-windows c1` prints that one window."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--only-windows", "c1"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("c1 ")]
    assert len(rows) == 1, r.stdout[-2000:]
    c1 = json.loads(rows[0][3:])
    assert "64 beams" in c1["workload"] and "1 arena" in c1["workload"]
    assert 0.0 < c1["kernel_us"] <= c1["us_per_step_device"] * 1.05 and c1["us_per_step_device"] <= c1["us_per_step"] * 1.05
    assert c1["cpu_us_per_step_1_thread"] > 0.0 and "restatement" in c1["cpu_kind"]
    assert abs(c1["value"] - 1e6 / c1["us_per_step"]) < 1e-6 * c1["value"]


def test_edge_shapes(gpu):
    """Ragged / extreme shapes: odd map size (edge tiles), beam count not a multiple of 64, deep scan
    stack, the compiled maximum of 64 pedestrians with ragged n_peds (0, 1, 64), wide action range,
    turning-radius clamp; plus empty batches through every entry point."""
    torch = gpu.torch
    E, size, N = 5, 253, 64
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, n_scan_stack=4,
                                 ped_model=abi.PED_EXTERNAL, auto_reset=1, n_spawn=3, seed=91,
                                 field_format=abi.FIELD_U16T, min_turning_radius=0.33712)
    gpu.world.lidar_full_circle(cfg, 77)
    occ = gpu.world.make_maps(E, size, 91)
    n_peds = torch.tensor([0, 1, 64, 17, 64], dtype=torch.int32)
    arrays = gpu.world.make_world(cfg, occ, n_peds=n_peds, device=gpu.dev, min_goal_dist=2, max_goal_dist=6,
                                  robot_clearance=0.8)
    from nav_gym_amd import robots
    arrays["scan_threshold"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "threshold_footprint")))
    arrays["scan_discomfort"] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", "discomfort_threshold_footprint")))
    host = {k: v.cpu().numpy() for k, v in arrays.items() if k not in ("field", "field_overflow", "rect_table", "rect_index")}
    host["field"] = ref.build_dt(occ)
    g = gpu.sim.NavSim(cfg, arrays); r = ref.RefSim(cfg, host)
    _eq(g.reset_obs().cpu().numpy(), r.reset_obs(), "reset obs")
    rng = np.random.default_rng(4)
    for t in range(30):
        act = np.stack([rng.uniform(-0.3, 0.9, E), rng.uniform(-1.5, 1.5, E)], axis=1)    # out of range: not clipped
        cmd = np.stack([rng.uniform(0, 0.6, (E, N)), rng.uniform(-0.6, 0.6, (E, N))], axis=2)
        g.set_ped_cmd(cmd); r.set_ped_cmd(cmd)
        go, gout = g.step(torch.from_numpy(act).to(gpu.dev)); ro, rout = r.step(act)
        _eq(go.cpu().numpy(), ro, "obs at step %d" % t)
        for k in rout:
            _eq(gout[k].cpu().numpy(), rout[k], "%s at step %d" % (k, t))
    _eq(g.ped_scans().cpu().numpy()[2], r.ped_scans()[2], "64-pedestrian scans")
    # empty batches
    L = gpu.lib.load()
    cfg0 = cfg.copy(); cfg0.n_envs = 0
    st0 = abi.NavsimState(); io0 = abi.NavsimStepIO()
    for name in ("field", "scan_threshold", "scan_discomfort", "robot_pose", "robot_goal", "prev_action", "prev_pose",
                 "n_hist", "episode", "steps", "n_peds", "ped_pose", "ped_vel", "ped_prev_yaw", "ped_dist", "ped_has_legs",
                 "ped_waypoints", "ped_n_waypoints", "ped_wp_head", "ped_cmd", "spawn_pose", "spawn_goal"):
        setattr(st0, name, g.t.get(name, g.t["robot_pose"]).data_ptr())
    for name in ("action", "obs", "obs_prev", "reward", "done", "is_success", "is_crash", "distance"):
        setattr(io0, name, g.obs.data_ptr())
    assert L.navsim_step(C.byref(cfg0), C.byref(st0), C.byref(io0), None) == 0
    assert L.navsim_reset_obs(C.byref(cfg0), C.byref(st0), C.byref(io0), None, None) == 0
    z = torch.zeros(4, device=gpu.dev, dtype=torch.float64)
    assert L.navsim_integrate(z.data_ptr(), z.data_ptr(), None, 0, 0.2, 0.0, None) == 0
    assert L.navsim_cast_static(g.t["field"].data_ptr(), 0, size, size, None, 7, 1.0, 0, None, None) == 0


def test_pipelined_ped_policy_equals_the_two_calls(gpu):
    """Round 5: NavSim.ped_policy(pipeline=n) takes the pedestrians' scans and runs the network in slices of arenas on two
    streams (navsim_ped_scans_part beside navsim_ped_policy_part; an option, not the default: it measured slower).  Same kernels on the same rows: commands, clipped means, the
    waypoint heads it advances and the scan rows equal those of navsim_ped_scans followed by navsim_ped_policy."""
    torch = gpu.torch
    E, size, N = 700, 200, 12                      # 8400 pedestrian slots: above the pipeline's threshold, ragged slices
    cfg = gpu.lib.default_config(n_envs=E, map_h=size, map_w=size, max_peds=N, ped_model=abi.PED_EXTERNAL, n_spawn=4,
                                 auto_reset=1, seed=31, field_format=abi.FIELD_U16T)
    gpu.world.lidar_1081(cfg)
    occ = gpu.world.make_maps(E, size, 31)
    arrays = gpu.world.make_world(cfg, occ, n_peds=9, device=gpu.dev, min_goal_dist=2.0, max_goal_dist=6.0, robot_clearance=0.6)
    from nav_gym_amd import robots
    for key, name in (("scan_threshold", "threshold_footprint"), ("scan_discomfort", "discomfort_threshold_footprint")):
        arrays[key] = gpu.sim.scan_threshold(cfg, _t(gpu, robots.footprint_array("keti", name)))
    arrays["n_peds"][::5] = 12
    arrays["n_peds"][3] = 0
    g = gpu.sim.NavSim(cfg, arrays)
    g.set_policy(_policy_weights_random(7))
    g.t["policy_prev_actions"].copy_(torch.rand((E, N, 2), device=gpu.dev) * 0.5)
    keep = {k: g.t[k].clone() for k in ("policy_prev_actions", "ped_wp_head")}
    scans = g.ped_scans()
    cmd_a, mean_a = [x.clone() for x in g.ped_policy(scans, pipeline=False)]
    head_a = g.t["ped_wp_head"].clone()
    for n_slices in (3, 5):
        for k, v in keep.items():
            g.t[k].copy_(v)
        g.t["ped_cmd"].zero_()
        cmd_b, mean_b = g.ped_policy(pipeline=n_slices)
        torch.cuda.synchronize()
        assert torch.equal(cmd_a, cmd_b) and torch.equal(mean_a, mean_b) and torch.equal(head_a, g.t["ped_wp_head"])
        live = (torch.arange(N, device=gpu.dev)[None, :] < g.t["n_peds"][:, None].clamp(max=N))
        assert torch.equal(g.t["ped_scan_rows"][live], scans[live])
