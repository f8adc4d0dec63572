#!/usr/bin/env python
"""bench.py -- env-steps/s of the batched Spriteworld step+render hot path on MI355X.

Contract (one JSON line on rank 0):
  python bench.py --gpus N --steps K --warmup W
  N > 1: launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (one process per GPU, RCCL barrier; environments are sharded across ranks with NO data-path
  collective -- weak scaling, 8192 environments per GPU).

A "step" is one Environment.step() of every environment of the batch, including the 64x64 RGB
render, with actions and state resident in HBM (reference: spriteworld/environment.py:88-108).
Workload (BASELINE.json `metric`, configs[2]): 8192 envs x 5 sprites, SelectMove(0.25),
Clustering reward, 64x64 PILRenderer with anti_aliasing=5 (the COBRA renderer), synthetic pools.

roofline:     algorithmic bytes per launch (12 461 B/env-step) / the fused
              kernel's mean duration measured with HIP events on the launch stream, vs 8 TB/s HBM.
              `kernel`, `metric` and `config` are derived from what ran (swb_variant / the lowered
              config), `traffic` and `instructions` come from the committed PMC passes of exactly
              this build (profiles/r02_counters.json, keyed by the library's build id) or are null.
cpu_baseline: the CPU oracle (a C port of the reference algorithm, oracle/sw_oracle.c) stepping a
              bounded sample of the same workload on all host cores of rank 0, with the survey's
              rate of the unmodified Python reference beside it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ENVS_PER_GPU = 8192
WORKLOAD = 'cluster_s5'
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
N_ACTION_SETS = 16


def algorithmic_bytes(cfg):
  """A = 3*H*W + 28*S + 17 + action_bytes per env-step."""
  action_bytes = 8 if cfg.action_space == 2 else 16
  return 3 * cfg.image_h * cfg.image_w + 28 * cfg.max_sprites + 17 + action_bytes


# Survey-time rate of the unmodified reference (Python + PIL + sklearn) on one host core, BASELINE.md section 2.
# /root/reference does not exist on the GPU box, so the same-run CPU baseline is the C port (oracle/sw_oracle.c),
# which is 3-9x faster per core than the reference; both figures are put in the line.
REFERENCE_RATE_PER_CORE = {'cluster_s5': 228.0, 'goal_s5': 637.0, 'embodied_s12': 156.0}
VALU_CYCLES_PER_INST = 4.24      # profiles/r02_ubench_valu.md + SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of the step kernel


def profiled_counters(workload, envs, aa, build_id):
  """PMC figures of the committed rocprofv3 passes (profiles/r02_counters.json) for this exact build and workload.

  bench.py cannot collect PMC counters itself.  The file records the build id (content hash of the kernel sources)
  the passes ran on; for any other build, workload or batch the figures are stale and None is returned."""
  path = os.path.join(ROOT, 'profiles', 'r02_counters.json')
  if not os.path.exists(path):
    return None
  with open(path) as f:
    data = json.load(f)
  for rec in data.get('records', []):
    if (rec['build_id'] == build_id and rec['workload'] == workload and rec['envs'] == envs and
        rec['anti_aliasing'] == aa):
      return rec
  return None


def gpu_run(name, n_envs, steps, warmup, aa, device, barrier=None, seed=0):
  import torch
  from spriteworld_amd import engine, workloads
  cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=4, seed=seed, anti_aliasing=aa)
  eng = engine.Engine(cfg, pool, device=device)
  rng = np.random.default_rng(2000 + seed)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(N_ACTION_SETS)]
  for i in range(warmup):
    eng.step(acts[i % N_ACTION_SETS])
  torch.cuda.synchronize(eng.device)
  eng.timing(True)
  if barrier:
    barrier()
  torch.cuda.synchronize(eng.device)
  t0 = time.perf_counter()
  for i in range(steps):
    eng.step(acts[i % N_ACTION_SETS])
  torch.cuda.synchronize(eng.device)
  if barrier:
    barrier()
    torch.cuda.synchronize(eng.device)
  elapsed = time.perf_counter() - t0
  kernel_ms, launches = eng.step_time_ms()
  eng.timing(False)
  errors = int(eng.error.max().item())
  a_bytes = algorithmic_bytes(cfg)
  variant = eng.variant()
  facts = dict(sprites=cfg.max_sprites, image=[cfg.image_w, cfg.image_h], anti_aliasing=cfg.anti_aliasing,
               action_space={0: 'SelectMove', 1: 'DragAndDrop', 2: 'Embodied'}[cfg.action_space],
               task={0: 'NoReward', 1: 'FindGoalPosition', 2: 'Clustering'}[cfg.tasks[0].kind] if not cfg.is_meta
               else 'MetaAggregated', max_episode_length=cfg.max_episode_length)
  eng.close()
  return dict(elapsed=elapsed, kernel_ms=kernel_ms, launches=launches, a_bytes=a_bytes, errors=errors,
              variant=variant, facts=facts)


def gpu_run_groups(name, n_envs, groups, steps, warmup, aa, device):
  """The same batch as `groups` independent engines stepped on `groups` HIP streams (no cross-group
  ordering between steps): the double-buffered stepping pattern of RL samplers.  Returns env-steps/s."""
  import torch
  from spriteworld_amd import engine, workloads
  n = n_envs // groups
  engs, acts, streams = [], [], []
  for g in range(groups):
    cfg, pool, sample = workloads.build(name, n, episodes_per_env=4, seed=g, anti_aliasing=aa)
    engs.append(engine.Engine(cfg, pool, device=device))
    rng = np.random.default_rng(2000 + g)
    acts.append([torch.as_tensor(sample(rng), device=engs[g].device) for _ in range(N_ACTION_SETS)])
    streams.append(torch.cuda.Stream(device=engs[g].device))

  def run(k):
    for i in range(k):
      for g in range(groups):
        with torch.cuda.stream(streams[g]):
          engs[g].step(acts[g][i % N_ACTION_SETS])
  run(warmup)
  torch.cuda.synchronize(engs[0].device)
  t0 = time.perf_counter()
  run(steps)
  torch.cuda.synchronize(engs[0].device)
  elapsed = time.perf_counter() - t0
  errors = max(int(e.error.max().item()) for e in engs)
  for e in engs:
    e.close()
  return n * groups * steps / elapsed, errors


def usable_cores():
  """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
  try:
    cores = len(os.sched_getaffinity(0))
  except AttributeError:
    cores = os.cpu_count() or 1
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()
    if quota != 'max':
      cores = max(1, min(cores, int(int(quota) / int(period))))
  except (OSError, ValueError):
    pass
  return cores


def cpu_baseline(name, aa, budget_s=12.0):
  """Oracle env-steps/s on all host cores (threads; the C calls release the GIL)."""
  from concurrent.futures import ThreadPoolExecutor
  from oracle import oracle
  from spriteworld_amd import workloads
  cores = usable_cores()
  n_envs = 64 * cores
  cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=4, seed=1, anti_aliasing=aa)
  eng = oracle.Engine(cfg, pool)
  rng = np.random.default_rng(5)
  import ctypes as C
  lib = oracle.lib()
  obs = np.zeros((n_envs,) + eng.obs_shape, np.uint8)
  rew = np.zeros(n_envs)
  bounds = [(i * n_envs // cores, (i + 1) * n_envs // cores) for i in range(cores)]

  def one_step(actions):
    a = np.ascontiguousarray(actions)

    def work(b):
      lib.swo_step_range(eng._h, b[0], b[1], a.ctypes.data_as(C.c_void_p), obs.ctypes.data_as(C.c_void_p),
                         rew.ctypes.data_as(C.c_void_p), None, None, None, None)
    list(pool_exec.map(work, bounds))

  with ThreadPoolExecutor(cores) as pool_exec:
    one_step(sample(rng))                       # reset step (untimed)
    one_step(sample(rng))
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < budget_s:
      one_step(sample(rng))
      steps += 1
    dt = time.perf_counter() - t0
  out = dict(value=n_envs * steps / dt, unit='env-steps/s', cores=cores, kind='port',
             sample='%d envs x %d steps of %s (AA=%d) with oracle/sw_oracle.c on %d threads, %.1f s' %
             (n_envs, steps, name, aa, cores, dt))
  if name in REFERENCE_RATE_PER_CORE and aa == 5:
    out['reference_env_steps_per_s_per_core'] = REFERENCE_RATE_PER_CORE[name]
    out['reference_note'] = ('the unmodified Python reference measured %.0f env-steps/s on one core for this config '
                             '(BASELINE.md section 2, survey container); it cannot run on the GPU box, so the timed '
                             'baseline here is its C port' % REFERENCE_RATE_PER_CORE[name])
  return out


def assemble_line(args, res, elapsed):
  """The JSON line (without the extras / cpu_baseline blocks) from one timed run: `res` is gpu_run()'s result,
  `elapsed` the max-over-ranks wall time of the timed region.  Pure host code (tests call it without a GPU)."""
  total_envs = args.envs_per_gpu * args.gpus
  value = total_envs * args.steps / elapsed
  kernel_s = res['kernel_ms'] / 1e3 / max(res['launches'], 1)
  achieved = res['a_bytes'] * args.envs_per_gpu / kernel_s / 1e9
  facts, variant = res['facts'], res['variant']
  image = '%dx%d' % (facts['image'][1], facts['image'][0])
  counters = profiled_counters(args.workload, args.envs_per_gpu, args.aa, variant['build_id'])
  roofline = {
      'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
      'frac': achieved / HBM_PEAK_GBS, 'frac_of_measured_copy_peak': achieved / 6290.0,
      'traffic': counters['hbm_traffic_bytes_per_launch'] if counters else None,
      'kernel': variant['kernel'], 'kernel_ms': kernel_s * 1e3,
      'algorithmic_bytes_per_env_step': res['a_bytes'],
      'lds_bytes_per_wave': variant['lds_bytes_per_wave'], 'waves_per_simd': variant['waves_per_simd'],
      'build_id': variant['build_id'],
  }
  if counters:
    # instruction side (SURVEY 8d asks for both): the kernel is bound by instruction issue, not by HBM
    waves_per_simd_resident = counters.get('resident_waves_per_simd', variant['waves_per_simd'])
    roofline['instructions'] = {
        'insts_valu_per_wave': counters['insts_valu_per_wave'], 'insts_salu_per_wave': counters['insts_salu_per_wave'],
        'insts_lds_per_wave': counters['insts_lds_per_wave'],
        'valu_cycles_per_inst': VALU_CYCLES_PER_INST,
        'valu_issue_frac': counters['active_inst_valu_per_wave'] * waves_per_simd_resident / counters['wave_cycles_per_wave'],
        'source': counters['source'],
    }
  else:
    roofline['instructions'] = None       # no committed PMC pass for this build / workload (see profiles/)
  out = {
      'metric': 'env-steps/sec (incl. %s RGB render) at %d envs' % (image, args.envs_per_gpu),
      'value': value,
      'unit': 'env-steps/s',
      'n_gpus': args.gpus,
      'steps': args.steps,
      'warmup': args.warmup,
      'ms_per_step': elapsed / args.steps * 1e3,
      'higher_is_better': True,
      'scaling': 'weak',
      'vs_baseline': None,
      'dtype': 'i32 fixed-point raster/resample + f64 state',
      'data': 'synthetic',
      'config': {
          'workload': '%s: %d envs/GPU x %d sprites, %s, %s reward, %s PILRenderer anti_aliasing=%d, auto-reset from '
                      'an HBM pool%s' % (args.workload, args.envs_per_gpu, facts['sprites'], facts['action_space'],
                                         facts['task'], image, facts['anti_aliasing'],
                                         ' (BASELINE configs[2])' if (args.workload, args.envs_per_gpu, args.aa) ==
                                         (WORKLOAD, ENVS_PER_GPU, 5) else ''),
          'envs_per_gpu': args.envs_per_gpu, 'sprites': facts['sprites'], 'image': facts['image'],
          'anti_aliasing': facts['anti_aliasing'],
          'parallelism': 'env-sharded x%d, no data-path collective' % args.gpus,
      },
      'roofline': roofline,
      'env_errors': res['errors'],
  }
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=20)
  ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
  ap.add_argument('--workload', default=WORKLOAD)
  ap.add_argument('--aa', type=int, default=5)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extra', action='store_true')
  ap.add_argument('--gather-obs', action='store_true',
                  help='also all-gather the observation shards over RCCL every step (BASELINE configs[3])')
  args = ap.parse_args()

  import torch
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus and world > 1:
    raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
  if args.gpus > 1 and world == 1:
    raise SystemExit('launch with torch.distributed.run for --gpus > 1')
  barrier = None
  dist = None
  if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:   # launched by torch.distributed.run
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # SWB_BENCH_ONE_DEVICE=1: functional check of the N > 1 path on a 1-GPU box (all ranks on cuda:0, gloo
    # instead of RCCL, which refuses two ranks on one device); never set by the driver
    one_device = os.environ.get('SWB_BENCH_ONE_DEVICE') == '1'
    if one_device:
      local_rank = 0
    torch.cuda.set_device(local_rank)
    if one_device:
      dist.init_process_group('gloo')
    else:
      dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    barrier = dist.barrier
  device = local_rank if dist is not None else 0

  res = gpu_run(args.workload, args.envs_per_gpu, args.steps, args.warmup, args.aa, device,
                barrier=barrier, seed=rank)
  elapsed = res['elapsed']
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if dist.get_backend() == 'gloo' else 'cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  gather = None
  if args.gather_obs and dist is not None:
    from spriteworld_amd import distributed as swd
    gather = swd.bench_allgather(args.workload, args.envs_per_gpu, args.steps, args.warmup, args.aa, device, rank)

  if rank != 0:
    if dist is not None:
      dist.barrier()
      dist.destroy_process_group()
    return

  out = assemble_line(args, res, elapsed)
  if gather is not None:
    out['obs_allgather'] = gather
  if args.gpus == 1 and not args.no_extra:
    extra = {}
    short = max(args.steps // 4, 10)
    for label, (nm, n, aa) in {
        'cluster_s5_aa1': ('cluster_s5', args.envs_per_gpu, 1),
        'goal_s5_1024_aa5': ('goal_s5', 1024, 5),
        'embodied_s12_128_aa5': ('embodied_s12', args.envs_per_gpu, 5),
        # BASELINE configs[3]'s per-GPU share is 8192; this is the same scene with 8x the batch in ONE launch
        # (the launch's fixed fill/drain cost amortised)
        'cluster_s5_65536_aa5': ('cluster_s5', 65536, 5),
    }.items():
      r = gpu_run(nm, n, short, 5, aa, device)
      ks = r['kernel_ms'] / 1e3 / max(r['launches'], 1)
      extra[label] = {'env_steps_per_s': n * short / r['elapsed'], 'kernel': r['variant']['kernel'], 'kernel_ms': ks * 1e3,
                      'hbm_GBs': r['a_bytes'] * n / ks / 1e9, 'hbm_frac': r['a_bytes'] * n / ks / 1e9 / HBM_PEAK_GBS,
                      'env_errors': r['errors']}
    # the headline batch as two groups of 4096 on two HIP streams: consecutive steps of different groups overlap,
    # which hides the fill/drain of each launch (an application-level choice; `value` above is one launch per step)
    rate, errs = gpu_run_groups(args.workload, args.envs_per_gpu, 2, short, 5, args.aa, device)
    extra['%s_2_groups_2_streams' % args.workload] = {'env_steps_per_s': rate, 'env_errors': errs}
    out['extra'] = extra
  if args.gpus == 1 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline(args.workload, args.aa)
  print(json.dumps(out))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
