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


def shard_pool_entries(total_envs, rank, world, episodes_per_env):
  """[first, last) global pool-entry indices of `rank`'s shard: environment n owns entries
  [n * episodes_per_env, (n + 1) * episodes_per_env), so a shard's Philox streams (device-side reset
  sampling, `first_entry` of swb_sample_pool) are the ones a single process would use for the same
  environments.  This is the arithmetic BatchedEnvironment(global_env_offset=begin) applies."""
  begin, end = shard_range(total_envs, rank, world)
  return begin * episodes_per_env, end * episodes_per_env


def all_gather_observations(obs_shard, out=None, method='auto'):
  """Stacks every rank's u8 [n, H, W, 3] shard into [world*n, H, W, 3] on each rank.

  method 'ring':   one `all_gather_into_tensor` (RCCL ring: every byte crosses world-1 links in turn, so
                   the step is bound by ONE xGMI link, ~153 GB/s -> ~4.6 ms for 65 536 64x64 frames);
         'direct': every rank sends its shard straight to each peer and receives theirs, all
                   point-to-point copies batched in one group (`batch_isend_irecv`): on the fully
                   connected xGMI mesh all seven links of a GPU carry traffic at once (~0.66 ms for the
                   same gather);
         'auto':   'direct' on RCCL with more than two ranks, else 'ring'.
  Both give the same tensor; `out` may be passed to reuse a buffer."""
  world = dist.get_world_size()
  n = obs_shard.shape[0]
  if out is None:
    out = torch.empty((world * n,) + tuple(obs_shard.shape[1:]), dtype=obs_shard.dtype, device=obs_shard.device)
  shard = obs_shard.contiguous()
  if method == 'auto':
    method = 'direct' if (world > 2 and dist.get_backend() == 'nccl') else 'ring'
  if method == 'ring' or world == 1:
    dist.all_gather_into_tensor(out, shard)
    return out
  if method != 'direct':
    raise ValueError('method must be auto, ring or direct')
  rank = dist.get_rank()
  ops = []
  for step in range(1, world):                  # peer order staggered per rank: no two ranks start on the same target
    dst, src = (rank + step) % world, (rank - step) % world
    ops.append(dist.P2POp(dist.isend, shard, dst))
    ops.append(dist.P2POp(dist.irecv, out[src * n:(src + 1) * n], src))
  out[rank * n:(rank + 1) * n].copy_(shard)
  for req in dist.batch_isend_irecv(ops):
    req.wait()
  return out


def bench_allgather(workload, envs_per_gpu, steps, warmup, aa, device, rank, method='auto'):
  """Times the step and the all-gather of the observation shard separately (HIP events on the
  stream, per step), plus the two together; returns a dict (every rank, same values after the
  MAX reduction)."""
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
    all_gather_observations(eng.obs, full, method)
  torch.cuda.synchronize()
  dist.barrier()
  ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
  t0 = time.perf_counter()
  for i in range(steps):
    ev[i][0].record()
    eng.step(acts[i % 8])
    ev[i][1].record()
    all_gather_observations(eng.obs, full, method)
    ev[i][2].record()
  torch.cuda.synchronize()
  dist.barrier()
  dt = time.perf_counter() - t0
  step_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
  gather_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
  t = torch.tensor([dt, step_ms, gather_ms], dtype=torch.float64, device=eng.device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  eng.close()
  return {'env_steps_per_s': world * envs_per_gpu * steps / float(t[0].item()),
          'step_ms': float(t[1].item()), 'gather_ms': float(t[2].item()),
          'gathered_bytes_per_step': int(full.numel()), 'method': method}
