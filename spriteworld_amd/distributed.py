"""Multi-GPU use of the engine: shard environments, optionally all-gather observations.

`Environment.step()` of one environment never reads another environment, so N
environments split into contiguous ranges, one per GPU / process, with no
exchange.  The only collective a consumer may want is the
stacked observation tensor: one RCCL all-gather of the u8 [N/G, H, W, 3] shard
per step (`torch.distributed`, backend "nccl" == RCCL over xGMI).
"""
import time

import numpy as np
import torch
import torch.distributed as dist


def shard_range(num_envs, rank, world):
  """Contiguous [begin, end) of environments owned by `rank`."""
  return rank * num_envs // world, (rank + 1) * num_envs // world


def sharded_environment(total_envs, rank=None, world=None, **config):
  """This rank's BatchedEnvironment of a `total_envs` job (one process per GPU, LOCAL_RANK device).

  `config` is what BatchedEnvironment takes.  The shard knows its global offset, so device-side
  reset sampling (device_sampler.DeviceSampler) draws the episodes a single process would have
  drawn for the same environments."""
  import os
  from spriteworld_amd import environment
  if rank is None:
    rank = dist.get_rank() if dist.is_initialized() else int(os.environ.get('RANK', 0))
  if world is None:
    world = dist.get_world_size() if dist.is_initialized() else int(os.environ.get('WORLD_SIZE', 1))
  begin, end = shard_range(total_envs, rank, world)
  config.setdefault('device', int(os.environ.get('LOCAL_RANK', 0)))
  return environment.BatchedEnvironment(num_envs=end - begin, global_env_offset=begin, **config)


def all_gather_observations(obs_shard, out=None):
  """Stacks every rank's u8 [n, H, W, 3] shard into [world*n, H, W, 3] on each rank."""
  world = dist.get_world_size()
  if out is None:
    out = torch.empty((world * obs_shard.shape[0],) + tuple(obs_shard.shape[1:]),
                      dtype=obs_shard.dtype, device=obs_shard.device)
  dist.all_gather_into_tensor(out, obs_shard.contiguous())
  return out


def bench_allgather(workload, envs_per_gpu, steps, warmup, aa, device, rank):
  """Steps + all-gather of the observation shard each step; returns timing dict (rank 0)."""
  from spriteworld_amd import engine, workloads
  cfg, pool, sample = workloads.build(workload, envs_per_gpu, episodes_per_env=4, seed=rank,
                                      anti_aliasing=aa)
  eng = engine.Engine(cfg, pool, device=device)
  rng = np.random.default_rng(3000 + rank)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(8)]
  world = dist.get_world_size()
  full = torch.empty((world * envs_per_gpu,) + eng.obs_shape, dtype=torch.uint8, device=eng.device)
  for i in range(warmup):
    eng.step(acts[i % 8])
    all_gather_observations(eng.obs, full)
  torch.cuda.synchronize()
  dist.barrier()
  t0 = time.perf_counter()
  for i in range(steps):
    eng.step(acts[i % 8])
    all_gather_observations(eng.obs, full)
  torch.cuda.synchronize()
  dist.barrier()
  dt = time.perf_counter() - t0
  t = torch.tensor([dt], dtype=torch.float64, device=eng.device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  eng.close()
  return {'env_steps_per_s': world * envs_per_gpu * steps / float(t.item()),
          'gathered_bytes_per_step': int(full.numel())}
