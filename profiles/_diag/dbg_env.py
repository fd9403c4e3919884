import sys; sys.path[:0]=['/root/repo/nav-gym_amd']
import numpy as np, nav_gym_env
env = nav_gym_env.make("NavGym-v0", map_size=200, num_humans=5, seed=3)
obs = env.reset()
a=env.action_space.sample(); print('a',a)
o2, r, d, info = env.step(a)
print(r, d, info, env.compute_reward(np.zeros(2), o2), env.compute_info(o2), env.compute_done(o2))
print(o2['observation'][-7:], o2['desired_goal'], obs['observation'][-7:])
