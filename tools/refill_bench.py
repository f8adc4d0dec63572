import time, sys
sys.path.insert(0, '.')
import numpy as np, torch
from tests import test_device_sampler as T
from spriteworld_amd import environment, action_spaces, sprite_generators
sampler, task, rend = T._cobra_like()
env = environment.BatchedEnvironment(task=task, action_space=action_spaces.SelectMove(scale=0.25), renderers=rend,
                                     init_sprites=sampler, num_envs=8192, episodes_per_env=8)
torch.cuda.synchronize()
t = time.time()
for _ in range(10):
    env.refill_pool()
torch.cuda.synchronize()
dev = (time.time() - t) / 10
t = time.time()
eps = [sampler() for _ in range(2048)]
host = (time.time() - t) / 2048 * 65536
print('refill 65536 episodes: device %.3f ms, host sampling alone (extrapolated) %.1f s' % (dev * 1e3, host))
ts = env.reset()
for _ in range(5):
    ts = env.step(env.sample_actions())
env.check()
print('ok', float(ts.reward.mean()))
