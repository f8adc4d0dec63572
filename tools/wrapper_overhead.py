"""Host-side cost of the dm_env-style wrapper per step, next to the bare engine (8192 envs)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tests import test_device_sampler as T
from spriteworld_amd import environment, action_spaces, gym_wrapper
sampler, task, rend = T._cobra_like()
env = environment.BatchedEnvironment(task=task, action_space=action_spaces.SelectMove(scale=0.25), renderers=rend,
                                     init_sprites=sampler, num_envs=8192, episodes_per_env=8, max_episode_length=50)
acts = [env.sample_actions() for _ in range(8)]
acts = [a if isinstance(a, torch.Tensor) else torch.as_tensor(a) for a in acts]
acts = [a.cuda() for a in acts]
env.reset()
def timeit(fn, n=300):
    for i in range(20): fn(i)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
g = gym_wrapper.BatchedGymWrapper(env)
for rep in range(3):
  print('engine.step        %.4f ms' % timeit(lambda i: env.engine.step(acts[i % 8])))
  print('env.step           %.4f ms' % timeit(lambda i: env.step(acts[i % 8])))
  print('gym.step           %.4f ms' % timeit(lambda i: g.step(acts[i % 8])))
env._check_errors = 0
print('env.step, no error read-back %.4f ms' % timeit(lambda i: env.step(acts[i % 8])))
t = time.perf_counter()
for i in range(300): env.step(acts[i % 8])
print('host time per env.step (no sync) %.4f ms' % ((time.perf_counter() - t) / 300 * 1e3)); torch.cuda.synchronize()
print('sample_actions     %.4f ms' % timeit(lambda i: env.sample_actions()))
