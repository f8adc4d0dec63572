"""N > 1 path on CPU: env sharding and the observation all-gather over gloo (world_size 2)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spriteworld_amd import distributed as swd


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, n_envs, q, method='auto'):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    b, e = swd.shard_range(n_envs, rank, world)
    # each rank "renders" its shard: frame of env i is filled with i % 251
    shard = torch.stack([torch.full((8, 8, 3), i % 251, dtype=torch.uint8) for i in range(b, e)])
    full = swd.all_gather_observations(shard, method=method)
    q.put((rank, b, e, full.numpy()))
  finally:
    dist.destroy_process_group()


def test_shard_ranges_partition_the_batch():
  for n in (8192, 65536, 10, 7):
    for world in (1, 2, 4, 8):
      ranges = [swd.shard_range(n, r, world) for r in range(world)]
      assert ranges[0][0] == 0 and ranges[-1][1] == n
      assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))


def test_shard_pool_entries_partition_the_job():
  """sharded_environment's global_env_offset / first_entry arithmetic: the shards' Philox entry ranges tile
  the single-process range, in rank order, for every world size the driver uses."""
  for total, k in ((8192, 8), (65536, 4), (1000, 3)):
    for world in (1, 2, 4, 8):
      ranges = [swd.shard_pool_entries(total, r, world, k) for r in range(world)]
      assert ranges[0][0] == 0 and ranges[-1][1] == total * k
      for r in range(world):
        b, e = swd.shard_range(total, r, world)
        assert ranges[r] == (b * k, e * k)                      # global_env_offset * episodes_per_env
        assert r == 0 or ranges[r][0] == ranges[r - 1][1]


@pytest.mark.parametrize('world,method', [(2, 'ring'), (2, 'direct'), (4, 'direct'), (4, 'auto')])
def test_all_gather_equals_concatenation(world, method):
  n_envs = 6 * world
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, n_envs, q, method)) for r in range(world)]
  for p in procs:
    p.start()
  results = [q.get(timeout=120) for _ in range(world)]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  want = np.stack([np.full((8, 8, 3), i % 251, dtype=np.uint8) for i in range(n_envs)])
  for rank, b, e, full in results:
    assert (b, e) == swd.shard_range(n_envs, rank, world)
    assert np.array_equal(full, want)
