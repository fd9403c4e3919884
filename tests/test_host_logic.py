"""CPU: host-side logic -- C ABI surface, struct layouts, registry / env wrapper, world generation,
sharding (world_size 2 over gloo).  No GPU compute."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import ref
from nav_gym_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    """include/navsim.h <-> libnavsim_hip.so <-> abi.EXPORTS agree (the .so loads without a GPU)."""
    from nav_gym_amd import lib
    L = lib.load()
    header = open(os.path.join(ROOT, "include", "navsim.h")).read()
    declared = set(re.findall(r"\b(navsim_[a-z0-9_]+)\s*\(", header))
    declared -= {"navsim_config", "navsim_state", "navsim_step_io"}
    assert declared == set(abi.EXPORTS), (declared ^ set(abi.EXPORTS))
    for name in abi.EXPORTS:
        assert hasattr(L, name), name
    assert L.navsim_abi_version() == abi.ABI_VERSION
    assert L.navsim_sizeof_config() == C.sizeof(abi.NavsimConfig)
    assert L.navsim_sizeof_state() == C.sizeof(abi.NavsimState)
    assert L.navsim_sizeof_step_io() == C.sizeof(abi.NavsimStepIO)
    assert L.navsim_error_string(-1) == b"invalid argument"
    assert L.navsim_step_kernel_name() == b"navsim_step_kernel"


def test_default_config_matches_oracle_and_reference():
    from nav_gym_amd import lib
    a, b = lib.default_config(), ref.default_config()
    assert bytes(a) == bytes(b)                       # HIP library and oracle agree field by field
    assert (a.n_beams, a.time_step, a.range_max, a.axle_offset) == (512, 0.2, 25.0, 0.14474)


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch (error codes, never exceptions across the ABI)."""
    from nav_gym_amd import lib
    L = lib.load()
    cfg = lib.default_config()
    st, io = abi.NavsimState(), abi.NavsimStepIO()
    assert L.navsim_step(C.byref(cfg), C.byref(st), C.byref(io), None) == abi.E_ARG
    assert L.navsim_step(None, None, None, None) == abi.E_ARG
    cfg.max_peds = 1000
    assert L.navsim_reset_obs(C.byref(cfg), C.byref(st), C.byref(io), None, None) == abi.E_UNSUPPORTED
    assert L.navsim_field_bytes(2, 500, 500, abi.FIELD_U16T) == 2 * 63 * 63 * 64 * 2
    assert L.navsim_field_bytes(2, 500, 500, abi.FIELD_F32) == 2 * 500 * 500 * 4
    assert L.navsim_build_dt_workspace_bytes(3, 10, 20) == 3 * 10 * 20 * 2


def test_registry_and_env_surface(golden_dir):
    import nav_gym_env
    from nav_gym_amd import registry
    u = np.load(os.path.join(golden_dir, "golden_units.npz"))
    spec = registry.spec("NavGym-v0")
    kw = spec["kwargs"]
    # the registered kwargs equal the reference's (captured by make_golden.py)
    ref_kw = dict(zip(u["kwargs_keys"].tolist(), u["kwargs_vals"].tolist()))
    assert set(ref_kw) | {"env_param_range"} == set(kw)
    for k, v in ref_kw.items():
        assert str(kw[k]) == v, k
    ref_ep = dict(zip(u["env_param_keys"].tolist(), u["env_param_vals"].tolist()))
    assert {k: str(v) for k, v in kw["env_param_range"].items()} == ref_ep
    env = nav_gym_env.make("NavGym-v0", num_envs=3)
    assert env.observation_space.spaces["observation"].shape == (512 + 7,)
    assert env.action_space.low.tolist() == [0.0, pytest.approx(-0.64)] and env.action_space.high.tolist() == [0.5, pytest.approx(0.64)]
    assert env.cfg.n_beams == 512 and env.cfg.angle_min == -3.141592
    for m in ("reset", "step", "compute_reward", "compute_rewards", "compute_terminals", "compute_done",
              "compute_info", "_override_reward_factor", "render"):
        assert callable(getattr(env, m))
    env._override_reward_factor(reward_scale=3.0)
    assert env.cfg.reward_scale == 3.0
    with pytest.raises(NotImplementedError):
        nav_gym_env.make("NavGym-v0", robot_type="turtlebot")
    import torch
    if not torch.cuda.is_available():                 # the product path has no CPU fallback
        from nav_gym_amd.lib import NavsimError
        with pytest.raises(NavsimError):
            env.reset()


def test_world_generation_is_shard_invariant():
    import torch
    from nav_gym_amd import lib, world
    cfg = lib.default_config(n_envs=6, map_h=120, map_w=120, max_peds=4, n_spawn=4, ped_model=abi.PED_SFM, seed=9)
    occ = world.make_maps(6, 120, 9)
    assert occ.shape == (6, 120, 120) and set(np.unique(occ)) == {0, 1}
    assert (occ[:, :5] == 1).all() and (occ[:, :, -5:] == 1).all()          # 5-cell border wall
    field = torch.from_numpy(ref.build_dt(occ))
    a = world.make_world(cfg, occ, n_peds=3, device="cpu", field=field, min_goal_dist=2, max_goal_dist=5, robot_clearance=0.6)
    cells = (a["spawn_pose"][:, :, :2] / 0.05).long()
    for e in range(6):                                 # spawns sit on free cells with clearance
        assert (field[e][cells[e, :, 1], cells[e, :, 0]] >= 0.6 / 0.05).all()
    cfg2 = cfg.copy(); cfg2.n_envs = 2; cfg2.env_index_base = 4
    occ2 = world.make_maps(2, 120, 9, env_index_base=4)
    assert np.array_equal(occ2, occ[4:])
    b = world.make_world(cfg2, occ2, n_peds=3, device="cpu", field=field[4:], min_goal_dist=2, max_goal_dist=5, robot_clearance=0.6)
    for k in b:
        if k != "n_peds":
            assert torch.equal(b[k], a[k][4:]), k
    ind = world.indoor_map(np.random.default_rng(1), 200, 3, 100)
    assert ind.shape == (200, 200) and 0.05 < (ind == 0).mean() < 0.9


def test_shard_range():
    from nav_gym_amd.sharding import shard_range
    for n, w in ((4096, 8), (16384, 8), (10, 3), (5, 8)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert sum(c for _, c in spans) == n
        assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(w - 1))


_WORKER = r"""
import os, sys
sys.path[:0] = [%(root)r + "/nav-gym_amd", %(root)r + "/oracle", %(root)r + "/tests"]
import numpy as np, torch, torch.distributed as dist
import ref
from nav_gym_amd import abi, lib, robots, world
from nav_gym_amd.sharding import shard_range, gather_rows
rank, ws = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=ws)
E_total, size = 5, 100            # ragged: 3 + 2 arenas
def run(start, count):
    cfg = lib.default_config(n_envs=count, map_h=size, map_w=size, max_peds=3, n_spawn=4, ped_model=abi.PED_SFM,
                             auto_reset=1, seed=21, env_index_base=start)
    world.lidar_full_circle(cfg, 64)
    occ = world.make_maps(count, size, 21, env_index_base=start)
    f = ref.build_dt(occ)
    a = world.make_world(cfg, occ, n_peds=2, device="cpu", field=torch.from_numpy(f), min_goal_dist=1.5, max_goal_dist=4, robot_clearance=0.6)
    host = {k: v.numpy() for k, v in a.items()}
    host["scan_threshold"] = ref.scan_threshold(cfg, robots.footprint_array("keti", "threshold_footprint"))
    host["scan_discomfort"] = ref.scan_threshold(cfg, robots.footprint_array("keti", "discomfort_threshold_footprint"))
    sim = ref.RefSim(cfg, host); sim.reset_obs()
    rng = np.random.default_rng(3)
    acts = np.stack([rng.uniform(0, 0.5, (12, E_total)), rng.uniform(-0.6, 0.6, (12, E_total))], axis=2)
    for t in range(12):
        obs, out = sim.step(acts[t, start:start + count])
    return obs.copy(), out["reward"].copy(), out["done"].copy()
start, count = shard_range(E_total, rank, ws)
obs, rew, done = run(start, count)
g_obs = gather_rows(torch.from_numpy(obs)); g_rew = gather_rows(torch.from_numpy(rew)); g_done = gather_rows(torch.from_numpy(done))
if rank == 0:
    f_obs, f_rew, f_done = run(0, E_total)
    assert np.array_equal(g_obs.numpy(), f_obs), "sharded obs differ from the single-shard run"
    assert np.array_equal(g_rew.numpy(), f_rew) and np.array_equal(g_done.numpy(), f_done)
    print("SHARD_OK", g_obs.shape)
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_sharding_over_gloo(tmp_path):
    """N > 1 path on CPU: two ranks step their arena blocks (oracle as the stepper), gather rows over
    gloo; the result must equal the single-shard run bit for bit (global-index keyed worlds)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % {"root": ROOT})
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "SHARD_OK" in outs[0]


def _bench(args, **env):
    e = dict(os.environ, OMP_NUM_THREADS="1", **env)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=240)


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` with no launcher around it spawns two ranks itself (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*), they rendezvous (gloo here, RCCL on the GPU box), the max over ranks is taken and
    rank 0's line says n_gpus = 2.  --dry-run: control flow only, no GPU work (there is no GPU here)."""
    import json
    r = _bench(["--gpus", "2", "--dry-run", "--steps", "7"], NAVSIM_BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["steps"] == 7 and out["dry_run"] is True
    assert out["max_rank_time"] == 2.0                    # rank 1 reports 2.0, rank 0 1.0: MAX over ranks


def test_bench_launcher_propagates_a_failed_rank():
    r = _bench(["--gpus", "2", "--dry-run"], NAVSIM_BENCH_BACKEND="gloo", NAVSIM_BENCH_FAIL_RANK="1")
    assert r.returncode != 0 and "rank 1 exit 3" in r.stderr


def test_bench_rejects_mismatched_world_size():
    e = dict(os.environ, WORLD_SIZE="2", RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=e,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_without_gpu_fails_loudly():
    """The product path has no CPU fallback: bench.py refuses to run without a MI355X."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = _bench(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_export_transform_matches_reference_utils():
    """export._transform restates utils.transform_xys (ros_env.py:100-135 uses it for the three footprint
    polygons); golden: the reference's own output for a translation (3.25, -1.5) and yaw 0.7."""
    from nav_gym_amd import export
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_units.npz"))
    got = export._transform(d["tf_xys_in"], 3.25, -1.5, 0.7)
    assert np.abs(got - d["tf_xys_out"][:, :2]).max() < 1e-12
    q = export._quaternion_from_yaw(0.7)
    assert abs(q[2] - np.sin(0.35)) < 1e-15 and abs(q[3] - np.cos(0.35)) < 1e-15 and q[0] == 0.0 and q[1] == 0.0
