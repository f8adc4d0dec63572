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

roofline:     a step is two kernels on one stream (cover: state + geometry + coverage -> run lists; resample: run
              lists -> frames).  `achieved` = algorithmic bytes per step (12 461 B/env-step)
              / the mean duration of BOTH kernels together, measured with HIP events on the launch stream, vs
              8 TB/s HBM -- the conservative reading; `kernels` gives each kernel's own duration, and for the
              dominant one (resample, which writes the frames) the rate of its own bytes.  `kernel`, `metric`
              and `config` are derived from what ran (swb_variant / the lowered config); `traffic` and
              `instructions` come from the committed PMC passes of exactly this build
              (profiles/rNN_counters.json, keyed by the library's build id) or are null.
cpu_baseline: kind "reference" -- the UNMODIFIED reference (Python + PIL + matplotlib + sklearn; on the GPU node the
              sourceless bytecode of /root/reference under oracle/_ref, oracle/stage_ref.py) stepping a bounded sample
              of the headline scene on ALL host cores of rank 0 in this same run (tools/reference_cpu_baseline.py:
              one process per core, core count and CPU model in the block), with the C port of the same algorithm
              (oracle/sw_oracle.c, threads) beside it as `port`.  Falls back to kind "port" -- with the reason --
              only when the reference or one of its libraries cannot be imported.
verified_envs: after the timed region the last step's positions, rewards, step types and frames of 64 sampled
              environments are compared with the oracle stepped through the same actions (outside the timed region).
"""
import argparse
import copy
import json
import os
import subprocess
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


def handoff_bytes_per_env(variant, n_envs):
  b = variant.get('run_list_bytes')
  return round(b / max(n_envs, 1), 1) if b else 0


def algorithmic_bytes(cfg):
  """A = 3*H*W + 28*S + 17 + action_bytes per env-step."""
  action_bytes = 8 if cfg.action_space == 2 else 16
  return 3 * cfg.image_h * cfg.image_w + 28 * cfg.max_sprites + 17 + action_bytes


VALU_CYCLES_PER_INST = 4.24      # profiles/r02_ubench_valu.md + SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of the step kernels


def profiled_counters(workload, envs, aa, build_id):
  """PMC figures of the committed rocprofv3 passes (profiles/rNN_counters.json, newest round first) for this exact build and
  workload -> (record or None, state).

  bench.py cannot collect PMC counters itself.  A file records the build id (content hash of the kernel sources) its passes
  ran on.  state: 'fresh' -- a record of exactly this build; 'stale' -- the newest record of this workload is of ANOTHER
  build (a kernel edit without a re-run of tools/final_evidence.sh): the line then says so instead of quoting it or
  silently dropping the block; 'none' -- no committed passes for this workload at all."""
  import glob
  state = 'none'
  for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_counters.json')), reverse=True):
    with open(path) as f:
      data = json.load(f)
    for rec in data.get('records', []):
      if rec['workload'] == workload and rec['envs'] == envs and rec['anti_aliasing'] == aa:
        if rec['build_id'] == build_id:
          return rec, 'fresh'
        state = 'stale'
  return None, state


class TimedRun(object):
  """One timed run in three parts, so that several of them can be set up first and then run back to back on the device:
  build() puts the engine and its action sets on the device; go() does the warm-up steps and times exactly `steps` steps;
  finish() reads the error flags and (`verify` > 0) the final state / outputs of that many sampled environments for
  verify_against_oracle(), times the two kernels of a step apart, and closes the engine.

  Timers.  The timed region is bracketed by ONE pair of HIP events on the launch stream (and by the host clock behind a
  device synchronisation on either side): `kernel_ms` = the events' elapsed time = the kernels of `steps` steps and the
  gaps between them.  The engine's own timing mode -- three events per step: before the cover kernel, between the two
  kernels, after the second -- is NOT on in the timed region: measured, its events cost 6 % of a run of back-to-back steps
  (0.1867 against 0.1758 ms per step at 8192 environments, tools/exp_timing_overhead.py; rounds 1-4 timed with them).  The
  durations of the two kernels apart come from a short pass in that mode AFTER the timed region (`split`): each carries the
  cost of its own completion event, as it does under rocprofv3's kernel trace, so their sum exceeds `kernel_ms`."""

  def __init__(self, name, n_envs, steps, warmup, aa, device, seed=0, verify=0, protocol_8d=None):
    """protocol_8d: None, or (index of this run's first environment in the job, environments of the whole job)."""
    self.name, self.n_envs, self.steps, self.warmup, self.aa, self.device = name, n_envs, steps, warmup, aa, device
    self.seed, self.verify, self.protocol_8d = seed, verify, protocol_8d
    self.run_error, self.elapsed = None, None

  def build(self, inputs=None):
    """`inputs`: (cfg, pool, action sets) of another run of the same workload to share (the clock-ramp twin)."""
    import torch
    from spriteworld_amd import engine, workloads
    if inputs is not None:
      self.cfg, self.pool, self.acts_host, self.data = inputs
    elif self.protocol_8d:
      # SURVEY 8d, literally: every environment's pool from the reference's own generators under np.random.seed(1000 + env),
      # the actions of step t from RandomState(2000 + t) -- N_ACTION_SETS of them, cycled
      self.cfg, self.pool, actions_of_step = workloads.build_protocol_8d(self.name, self.n_envs, 4, self.aa,
                                                                       env_offset=self.protocol_8d[0], total_envs=self.protocol_8d[1])
      self.acts_host = [actions_of_step(t) for t in range(N_ACTION_SETS)]
      self.data = ('synthetic, SURVEY 8d protocol: reset pools drawn by the reference\'s generators (this package\'s mirrors: the same '
                   'draws) under np.random.seed(1000 + env), actions RandomState(2000 + (step mod %d)).uniform' % N_ACTION_SETS)
    else:
      self.cfg, self.pool, sample = workloads.build(self.name, self.n_envs, episodes_per_env=4, seed=self.seed, anti_aliasing=self.aa)
      rng = np.random.default_rng(2000 + self.seed)
      self.acts_host = [sample(rng) for _ in range(N_ACTION_SETS)]
      self.data = 'synthetic'
    self.eng = engine.Engine(self.cfg, self.pool, device=self.device)
    self.acts = [torch.as_tensor(a, device=self.eng.device) for a in self.acts_host]
    return self

  def inputs(self):
    return self.cfg, self.pool, self.acts_host, self.data

  def go(self, barrier=None):
    """The --warmup steps, then exactly `steps` timed ones between two barriers (N > 1) and two device synchronisations.
    N > 1: the ranks have been through the gate by now (gpu_run), so every rank runs the same sequence of collectives whatever
    happens on it: an exception in the warm-up or in the timed region is carried in the result (`error`) instead of being
    raised past a barrier the other ranks are waiting in."""
    import torch
    eng, acts = self.eng, self.acts
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    self.event_ms = 0.0
    try:
      for i in range(self.warmup):
        eng.step(acts[i % N_ACTION_SETS])
      torch.cuda.synchronize(eng.device)
    except Exception as e:  # pylint: disable=broad-except
      self.run_error = repr(e)
    if barrier:
      barrier()
    torch.cuda.synchronize(eng.device)
    t0 = time.perf_counter()
    try:
      if self.run_error is None:
        ev0.record()                                 # (torch's current stream: the one the engine launches on)
        for i in range(self.steps):
          eng.step(acts[i % N_ACTION_SETS])
        ev1.record()
        torch.cuda.synchronize(eng.device)
        self.event_ms = float(ev0.elapsed_time(ev1))
    except Exception as e:  # pylint: disable=broad-except
      self.run_error = repr(e)
    if barrier:
      barrier()
      torch.cuda.synchronize(eng.device)
    self.elapsed = time.perf_counter() - t0
    return True

  def ramp(self, ms, until=None):
    """Keeps the device busy with this engine's steps for `ms` milliseconds (untimed) and, after that, for as long as
    `until()` is false (N > 1: until every rank has got this far -- an idle device drops its clocks within milliseconds, and
    a rank that waited idle in a barrier would start its timed steps on a cold device); returns the steps taken."""
    import torch
    t0, k = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms or (until is not None and not until() and time.perf_counter() - t0 < 60.0):
      for _ in range(8):
        self.eng.step(self.acts[k % N_ACTION_SETS])
        k += 1
      torch.cuda.synchronize(self.eng.device)
    return k

  def close(self):
    try:
      import torch
      torch.cuda.synchronize(self.eng.device)
      self.eng.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def finish(self):
    eng, cfg, elapsed = self.eng, self.cfg, self.elapsed
    if self.run_error is not None:
      return dict(error=self.run_error, elapsed=elapsed, kernel_ms=0.0, cover_ms=0.0, resample_ms=0.0, launches=0, errors=-1)
    try:
      import torch
      kernel_ms, launches = self.event_ms, self.steps
      errors = int(eng.error.max().item())
      a_bytes = algorithmic_bytes(cfg)
      variant = eng.variant()
      sample_out = None
      if self.verify:
        idx = np.sort(np.random.default_rng(99 + self.seed).choice(self.n_envs, size=min(self.verify, self.n_envs), replace=False))
        got, st = eng.outputs_host(), eng.state()
        sample_out = dict(idx=idx, cfg=cfg, pool=self.pool, warmup=self.warmup, steps=self.steps,
                          actions=[np.ascontiguousarray(a[idx]) for a in self.acts_host],
                          got={k: got[k][idx].copy() for k in ('obs', 'reward', 'step_type', 'success', 'discount')},
                          state={k: st[k][idx].copy() for k in ('x', 'y', 'step_count', 'episode', 'n_sprites')})
      # the two kernels apart: a pass in the engine's timing mode behind the timed region (its state is no longer needed)
      split_steps = max(10, min(self.steps, 40))
      eng.timing(True)
      for i in range(split_steps):
        eng.step(self.acts[i % N_ACTION_SETS])
      torch.cuda.synchronize(eng.device)
      split_total, k = eng.step_time_ms()
      split_cover, split_second, _ = eng.kernel_times_ms()
      eng.timing(False)
      cover_ms, resample_ms = split_cover / max(k, 1) * launches, split_second / max(k, 1) * launches     # (per step x launches)
      split = dict(steps=int(k), step_ms=split_total / max(k, 1), cover_ms=split_cover / max(k, 1), second_ms=split_second / max(k, 1))
      errors = max(errors, int(eng.error.max().item()))
      facts = dict(sprites=cfg.max_sprites, image=[cfg.image_w, cfg.image_h], anti_aliasing=cfg.anti_aliasing,
                   action_space={0: 'SelectMove', 1: 'DragAndDrop', 2: 'Embodied'}[cfg.action_space],
                   task={0: 'NoReward', 1: 'FindGoalPosition', 2: 'Clustering'}[cfg.tasks[0].kind] if not cfg.is_meta
                   else 'MetaAggregated', max_episode_length=cfg.max_episode_length)
      eng.close()
    except Exception as e:  # pylint: disable=broad-except
      return dict(error=repr(e), elapsed=elapsed, kernel_ms=0.0, cover_ms=0.0, resample_ms=0.0, launches=0, errors=-1)
    return dict(elapsed=elapsed, kernel_ms=kernel_ms, cover_ms=cover_ms, resample_ms=resample_ms, launches=launches,
                a_bytes=a_bytes, errors=errors, variant=variant, facts=facts, sample=sample_out, error=None, split=split,
                data=getattr(self, 'data', 'synthetic'))


def gpu_run(name, n_envs, steps, warmup, aa, device, barrier=None, seed=0, gate=None, verify=0, before=None, after_gate=None,
            protocol_8d=None):
  """One timed run, start to end.  `before(run)`: called once the engine and its action sets are on the device (builds the
  clock-ramp twin: see main()); `gate` (N > 1): every rank reports whether it could set up -- returns False when some rank could
  not, then nothing is timed and None is returned (every rank leaves together instead of hanging in a barrier); it also brings
  the ranks to the same point in time, so `after_gate()` (the clock ramp; must not raise) and the warm-up steps behind it end
  on every rank within microseconds of each other and no device idles in the barrier in front of the timed steps."""
  run = TimedRun(name, n_envs, steps, warmup, aa, device, seed=seed, verify=verify, protocol_8d=protocol_8d).build()
  if before is not None:
    before(run)
  if gate is not None and not gate(None):
    run.close()
    return None
  if after_gate is not None:
    after_gate()
  run.go(barrier=barrier)
  return run.finish()


def verify_against_oracle(sample):
  """The in-run check of the timed run: the oracle (checker; outside the timed region) replays the sampled environments
  through exactly the actions the engine saw -- warm-up and timed steps -- and the LAST step's positions, step counts,
  rewards, step types, success flags and frames must agree bit for bit (frames: differing bytes are counted)."""
  import ctypes as C
  from oracle import oracle
  idx = sample['idx']
  n = len(idx)
  cfg = type(sample['cfg']).from_buffer_copy(bytes(sample['cfg']))
  cfg.n_envs = n
  pool = copy.copy(sample['pool'])
  pool.pool_base = np.ascontiguousarray(sample['pool'].pool_base[idx])
  pool.pool_len = np.ascontiguousarray(sample['pool'].pool_len[idx])
  ora = oracle.Engine(cfg, pool)
  t0 = time.perf_counter()
  out = None
  for k in (sample['warmup'], sample['steps']):          # the engine cycles the action sets from 0 in both regions
    for i in range(k):
      out = ora.step(sample['actions'][i % N_ACTION_SETS])
  st = ora.state()
  got, gst = sample['got'], sample['state']
  bad = np.zeros(n, bool)
  for key in ('x', 'y'):
    bad |= (gst[key].view(np.uint64) != st[key].view(np.uint64)).any(axis=1)
  for key in ('step_count', 'episode', 'n_sprites'):
    bad |= gst[key] != st[key]
  bad |= got['step_type'] != out['step_type']
  bad |= got['success'] != out['success']
  rg, ro = got['reward'], out['reward']
  bad |= ~((np.isnan(rg) & np.isnan(ro)) | (rg.view(np.uint64) == ro.view(np.uint64)))
  frame_bytes = int((got['obs'] != out['obs']).sum())
  max_lsb = int(np.abs(got['obs'].astype(np.int16) - out['obs'].astype(np.int16)).max()) if n else 0
  bad |= (got['obs'] != out['obs']).reshape(n, -1).any(axis=1)
  import zlib
  return {'verified_envs': int(n), 'mismatches': int(bad.sum()), 'frame_bytes_differing': frame_bytes,
          'frame_max_abs_diff': max_lsb, 'steps_replayed': int(sample['warmup'] + sample['steps']),
          'frames_crc32': '%08x' % zlib.crc32(np.ascontiguousarray(got['obs']).tobytes()),
          'checker': 'oracle/sw_oracle.c through the same actions, after the timed region (%.1f s)' % (time.perf_counter() - t0)}


class GroupsRun(object):
  """The same batch as `groups` independent engines stepped on `groups` HIP streams (no cross-group ordering between
  steps): the double-buffered stepping pattern of RL samplers.  build(), then go() -> (env-steps/s, error flags)."""

  def __init__(self, name, n_envs, groups, steps, warmup, aa, device):
    self.name, self.n, self.groups, self.steps, self.warmup, self.aa, self.device = name, n_envs // groups, groups, steps, warmup, aa, device

  def build(self):
    import torch
    from spriteworld_amd import engine, workloads
    self.engs, self.acts, self.streams = [], [], []
    for g in range(self.groups):
      cfg, pool, sample = workloads.build(self.name, self.n, episodes_per_env=4, seed=g, anti_aliasing=self.aa)
      self.engs.append(engine.Engine(cfg, pool, device=self.device))
      rng = np.random.default_rng(2000 + g)
      self.acts.append([torch.as_tensor(sample(rng), device=self.engs[g].device) for _ in range(N_ACTION_SETS)])
      self.streams.append(torch.cuda.Stream(device=self.engs[g].device))
    return self

  def go(self):
    import torch
    engs, acts, streams = self.engs, self.acts, self.streams

    def run(k):
      for i in range(k):
        for g in range(self.groups):
          with torch.cuda.stream(streams[g]):
            engs[g].step(acts[g][i % N_ACTION_SETS])
    run(self.warmup)
    torch.cuda.synchronize(engs[0].device)
    t0 = time.perf_counter()
    run(self.steps)
    torch.cuda.synchronize(engs[0].device)
    self.elapsed = time.perf_counter() - t0

  def close(self):
    for e in getattr(self, 'engs', []):
      try:
        e.close()
      except Exception:  # pylint: disable=broad-except
        pass

  def finish(self):
    errors = max(int(e.error.max().item()) for e in self.engs)
    self.close()
    return self.n * self.groups * self.steps / self.elapsed, errors


def gpu_run_groups(name, n_envs, groups, steps, warmup, aa, device):
  run = GroupsRun(name, n_envs, groups, steps, warmup, aa, device).build()
  run.go()
  return run.finish()


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


def port_cpu_baseline(name, aa, budget_s=6.0):
  """The C port of the reference algorithm (oracle/sw_oracle.c) on all host cores (threads; the C calls release the GIL)."""
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
  return dict(value=n_envs * steps / dt, unit='env-steps/s', cores=cores, kind='port',
              sample='%d envs x %d steps of %s (AA=%d) with oracle/sw_oracle.c on %d threads, %.1f s' %
              (n_envs, steps, name, aa, cores, dt))


def reference_cpu_baseline(envs_per_core=4, steps=2000, warmup=100, timeout_s=300):
  """The UNMODIFIED reference on every usable core of THIS host, in this run: tools/reference_cpu_baseline.py as a child
  process (a fresh interpreter -- nothing forks beside the HIP context), one worker process per core, each stepping
  `envs_per_core` reference Environments of the headline scene (BASELINE configs[2]) for `warmup` + `steps` steps -- SURVEY 8d's
  protocol: the reference's own generators under np.random.seed(1000 + env), actions RandomState(2000 + step), 2000 timed steps
  after 100 warm-up steps (about 15 s on 16 cores).
  Returns (block, None) or (None, reason)."""
  from oracle import ref_harness
  if not ref_harness.reference_available():
    return None, 'reference not present: neither /root/reference nor oracle/_ref (python oracle/stage_ref.py)'
  cores = usable_cores()
  env = dict(os.environ, SWB_REF_WHERE='the bench host (GPU node when a GPU is visible): %s' % os.uname().nodename,
             SPRITEWORLD_REFERENCE=ref_harness.REFERENCE_ROOT)
  cmd = [sys.executable, os.path.join(ROOT, 'tools', 'reference_cpu_baseline.py'), str(envs_per_core * cores), str(steps),
         str(warmup), str(cores)]
  try:
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
  except subprocess.TimeoutExpired:
    return None, 'reference baseline timed out after %d s' % timeout_s
  if proc.returncode != 0:
    return None, 'reference baseline failed: ' + (proc.stderr.strip().splitlines() or ['?'])[-1][:300]
  rec = json.loads(proc.stdout)
  block = dict(value=rec['env_steps_per_s_all_cores'], unit='env-steps/s', cores=rec['processes'], kind='reference',
               per_core=rec['env_steps_per_s_per_core'], cpu=rec['cpu'], host_cpus=rec['host_cpus'],
               sample='%d reference Environments (%d per process, %d processes = usable cores) x %d timed steps of the headline '
                      'scene (5 sprites, 2 hue clusters, SelectMove, Clustering, 64x64 AA=5) after %d warm-up steps; slowest '
                      'worker %.1f s' % (rec['envs'], envs_per_core, rec['processes'], rec['timed_steps_per_env'],
                                         rec['warmup_steps_per_env'], rec['timed_seconds_slowest_worker']),
               reference=dict(root=rec['reference_root'], kind=rec['reference_kind'], third_party=rec['third_party'],
                              frame_checksum=rec['frame_checksum']),
               where=rec['where'])
  return block, None


def cpu_baseline(name, aa):
  """The reference itself on the host's cores (kind "reference") with the C port beside it; kind "port" when the reference
  cannot run here (reason given).  The reference leg times the headline scene (configs[2]: cluster_s5, anti_aliasing = 5): it
  is the line's baseline only when that is the workload measured -- for any other workload the top-level value is the C port
  of THAT workload, with the reference's headline figure nested and labelled as such."""
  port = port_cpu_baseline(name, aa)
  ref, why = reference_cpu_baseline()
  if ref is None:
    port['reference_error'] = why
    return port
  if name == 'cluster_s5' and aa == 5:
    ref['port'] = port
    return ref
  ref['note'] = 'the reference on the HEADLINE scene (cluster_s5, anti_aliasing = 5), not on the workload of this line'
  port['reference_headline_scene'] = ref
  return port


def assemble_line(args, res, elapsed):
  """The JSON line (without the extras / cpu_baseline blocks) from one timed run: `res` is gpu_run()'s result,
  `elapsed` the max-over-ranks wall time of the timed region.  Pure host code (tests call it without a GPU)."""
  total_envs = args.envs_per_gpu * args.gpus
  value = total_envs * args.steps / elapsed
  n_launch = max(res['launches'], 1)
  kernel_s = res['kernel_ms'] / 1e3 / n_launch                       # both kernels of a step, HIP events on the launch stream
  cover_s, resample_s = res.get('cover_ms', 0.0) / 1e3 / n_launch, res.get('resample_ms', 0.0) / 1e3 / n_launch
  achieved = res['a_bytes'] * args.envs_per_gpu / kernel_s / 1e9
  facts, variant = res['facts'], res['variant']
  image = '%dx%d' % (facts['image'][1], facts['image'][0])
  obs_bytes = 3 * facts['image'][0] * facts['image'][1]
  counters, counters_state = profiled_counters(args.workload, args.envs_per_gpu, args.aa, variant['build_id'])
  second = variant['kernel']
  first = variant.get('cover_kernel', 'swb_cover_kernel')
  roofline = {
      'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
      'frac': achieved / HBM_PEAK_GBS, 'frac_of_measured_copy_peak': achieved / 6290.0,
      'traffic': counters['hbm_traffic_bytes_per_launch'] if counters else None,
      # 'fresh': traffic / instructions below are the committed PMC passes of exactly this build; 'stale': the committed
      # passes of this workload are of another build (not quoted); 'none': no passes committed for this workload
      'counters': counters_state,
      'kernel': ('%s + %s' % (first, second)) if not second.startswith('none') else first + ' (paints the frame: anti_aliasing = 1)',
      'kernel_ms': kernel_s * 1e3,
      'kernel_ms_source': 'one pair of HIP events on the launch stream around the timed steps / steps',
      'algorithmic_bytes_per_env_step': res['a_bytes'],
      # each kernel on its own (the second one writes the frames, 98.6 % of the algorithmic bytes): per-step events of a
      # separate pass behind the timed region -- every kernel then carries its own completion event, as under rocprofv3's
      # kernel trace, and their sum exceeds kernel_ms (TimedRun)
      'kernels_source': ('engine timing mode, %d steps after the timed region (step %.4f ms with its three events)' %
                         (res['split']['steps'], res['split']['step_ms'])) if res.get('split') else 'engine timing mode',
      'kernels': [
          {'name': first, 'ms': cover_s * 1e3, 'waves_per_simd': variant['waves_per_simd'],
           'lds_bytes_per_wave': variant['lds_bytes_per_wave']},
          {'name': second, 'ms': resample_s * 1e3, 'waves_per_simd': variant.get('resample_waves_per_simd'),
           'bands': variant.get('n_bands'), 'column_groups': variant.get('n_column_groups'),
           'achieved_GBs_frames_only': (obs_bytes * args.envs_per_gpu / resample_s / 1e9) if resample_s > 0 else None,
           'frac_frames_only': (obs_bytes * args.envs_per_gpu / resample_s / 1e9 / HBM_PEAK_GBS) if resample_s > 0 else None},
      ],
      'lds_bytes_per_wave': variant['lds_bytes_per_wave'], 'waves_per_simd': variant['waves_per_simd'],
      # device memory of the hand-off lists between the two kernels (fixed parts + arena, after the engine's trim) per environment
      'handoff_list_bytes_per_env': handoff_bytes_per_env(variant, args.envs_per_gpu),
      'build_id': variant['build_id'],
  }
  if second.startswith('none'):
    roofline['kernels'] = roofline['kernels'][:1]
    roofline['kernels'][0]['ms'] = kernel_s * 1e3
  if counters:
    # instruction side (SURVEY 8d asks for both): the step is bound by instruction issue, not by HBM.  Per environment:
    # what the two kernels retire, the time that alone takes on a SIMD's vector ALU, and the cost model's minimum for
    # the resample kernel from exact event counts (tools/assemble_evidence.py, tools/emu_stats.py)
    valu = counters['insts_valu_per_env']
    envs_per_simd = args.envs_per_gpu / 1024.0
    issue_frac = valu * envs_per_simd * VALU_CYCLES_PER_INST / 2.4e9 / kernel_s
    roofline['instructions'] = {
        'insts_valu_per_env': valu, 'insts_salu_per_env': counters['insts_salu_per_env'],
        'insts_valu_per_env_by_kernel': counters['insts_valu_per_env_by_kernel'],
        'valu_cycles_per_inst': VALU_CYCLES_PER_INST,
        'valu_issue_ms_per_step': valu * envs_per_simd * VALU_CYCLES_PER_INST / 2.4e9 * 1e3,
        'valu_issue_frac_of_step': issue_frac,
        'valu_busy_by_kernel': counters.get('valu_busy_by_kernel'),
        'resample_valu_model_min_per_env': counters.get('resample_valu_model_min_per_env'),
        'resample_valu_measured_over_model': (counters['insts_valu_per_env_by_kernel'].get('resample', 0) /
                                               counters['resample_valu_model_min_per_env'])
        if counters.get('resample_valu_model_min_per_env') else None,
        'source': counters['source'],
    }
    # `bound` above is the roofline SURVEY 8d defines for this path (HBM: raster / indexing, no MFMA); what the counters say
    # actually limits the step is named here, so that a reader of this block alone is not misled
    roofline['bound_measured'] = ('valu-issue' if issue_frac >= 0.5 else 'latency (vector ALU %.0f %% busy)' % (100 * issue_frac))
    roofline['bound_measured_source'] = ('vector instructions per environment x %.2f cycles on 1024 SIMDs at 2.4 GHz = %.0f %% of the '
                                         'step (PMC passes of this build)' % (VALU_CYCLES_PER_INST, 100 * issue_frac))
  else:
    roofline['instructions'] = None       # no committed PMC pass for this build / workload (see `counters`)
    roofline['bound_measured'] = None
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
      'data': res.get('data', 'synthetic'),
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
  ap.add_argument('--protocol', choices=['8d', 'synthetic'], default='8d',
                  help="inputs of the timed run: '8d' = SURVEY 8d literally (reference generators seeded 1000 + env, actions "
                       "RandomState(2000 + step); rank 0 of cluster_s5 / goal_s5), 'synthetic' = the same distributions drawn with numpy")
  ap.add_argument('--ramp-ms', type=float, default=300.0,
                  help='milliseconds of device load (a twin engine of the same workload) in front of the --warmup steps, on every '
                       'rank at every N; 0: none')
  ap.add_argument('--no-verify', action='store_true', help='skip the oracle check of 64 sampled environments after the timed region')
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

  # N > 1: every rank reports whether it got as far as the timed region before any rank enters it, so that rank 0 prints
  # its line (with the failing ranks' errors) whatever happens on the others
  setup_errors = []

  def gate(error):
    if dist is None:
      return error is None
    status = [None] * world
    dist.all_gather_object(status, error)
    setup_errors[:] = [(r, e) for r, e in enumerate(status) if e is not None]
    return not setup_errors

  # Clock ramp -- the SAME on every path (N = 1 and every rank of N > 1).  A device that comes out of idle raises its clocks
  # over tens of milliseconds of load (profiles/r05_launch_convergence.json: a compute-bound control kernel drifts 7 % over its
  # first 64 launches) and drops them again within milliseconds of idling; the driver's 5 + 20 steps are 5 ms of load.  So
  # before the timed engine's --warmup steps a TWIN engine of the same workload (a) times the same W + K steps cold -- the
  # line's `cold` block: what a fresh process gets -- and (b) keeps the device busy for --ramp-ms milliseconds.  The timed
  # engine itself sees exactly --warmup untimed steps, then exactly --steps timed ones; `warmup_effective` says what ran in
  # front of them.  (Round 5 ran the `extra` workloads in front of the headline instead, on the N = 1 path only: the advisor's
  # point that an N > 1 line would then have been compared with a differently warmed N = 1 line.)  The `extra` workloads now
  # run AFTER the headline, every engine built first, the first of them behind a ramp of its own.
  ramp_info = {}

  def clock_ramp_build(run):
    ramp_info['data'] = run.data
    if args.ramp_ms <= 0:
      return
    # (closed BEHIND the timed region: freeing its buffers in front of it idles the device for milliseconds, and an idle device
    # drops its clocks again -- tools/exp_warm_engine.py: a fresh engine's launches 5 .. 24 take +4.4 % out of idle, +1.6 %
    # directly behind another engine's steps; the driver's 5 + 20 steps +0.8 % with the twin kept, gpurun_out/r06l/ab.txt)
    ramp_info['twin'] = TimedRun(args.workload, args.envs_per_gpu, args.steps, args.warmup, args.aa, device, seed=rank + 1000).build(run.inputs())

  def clock_ramp_go():
    """Behind the gate (N > 1: every rank at the same point in time).  Never raises: a failing twin only costs the ramp."""
    twin, steps_taken = ramp_info.get('twin'), 0
    try:
      if twin is not None:
        twin.go()
        if twin.run_error is not None:
          raise RuntimeError(twin.run_error)
        ramp_info['cold_ms_per_step'] = twin.event_ms / max(args.steps, 1)
        ramp_info['cold_wall_ms_per_step'] = twin.elapsed / max(args.steps, 1) * 1e3
        steps_taken = twin.ramp(args.ramp_ms)
    except Exception as e:  # pylint: disable=broad-except
      ramp_info['ramp_error'], twin = repr(e), None
    if dist is not None and args.ramp_ms > 0:
      # N > 1: the ramp lasts until EVERY rank has done its --ramp-ms -- one all-reduce (issued by every rank, whatever happened
      # to its twin) polled while the twin keeps stepping, so that no device waits idle for a slower rank
      work = dist.all_reduce(torch.zeros(1, device='cpu' if dist.get_backend() == 'gloo' else 'cuda'), async_op=True)
      def arrived():
        try:
          return bool(work.is_completed())
        except Exception:  # pylint: disable=broad-except     (a backend without a completion query: stop polling, wait below)
          return True
      try:
        if twin is not None:
          steps_taken += twin.ramp(0.0, until=arrived)
      except Exception as e:  # pylint: disable=broad-except
        ramp_info['ramp_error'] = repr(e)
      work.wait()
    ramp_info['clock_ramp_steps'] = steps_taken

  extra = {}
  extra_runs = []

  def build_extras():
    short = max(args.steps // 4, 10)
    for label, (nm, n, aa) in {
        # anti_aliasing = 5 (reference-faithful) AND 1 (pure raster) for each configuration
        'cluster_s5_aa1': ('cluster_s5', args.envs_per_gpu, 1),
        'goal_s5_1024_aa5': ('goal_s5', 1024, 5),
        'goal_s5_1024_aa1': ('goal_s5', 1024, 1),
        'embodied_s12_128_aa5': ('embodied_s12', args.envs_per_gpu, 5),
        'embodied_s12_128_aa1': ('embodied_s12', args.envs_per_gpu, 1),
        # BASELINE configs[3]'s per-GPU share is 8192; this is the same scene with 8x the batch in ONE launch
        # (the launch's fixed fill/drain cost amortised)
        'cluster_s5_65536_aa5': ('cluster_s5', 65536, 5),
        'cluster_s5_65536_aa1': ('cluster_s5', 65536, 1),
    }.items():
      extra_runs.append((label, TimedRun(nm, n, short, 5, aa, device).build()))
    # the headline batch as two groups of 4096 on two HIP streams: consecutive steps of different groups overlap,
    # which hides part of the fill/drain of each launch: +5 % for the same total work (profiles/r06_queue_concurrency.md; an
    # application-level choice -- environment.EnvironmentGroups -- `value` is one launch per step).  At least 100 steps: the
    # two queues need a few launches to interleave, and this figure is wall time over host-side launches
    extra_runs.append(('%s_2_groups_2_streams' % args.workload,
                       GroupsRun(args.workload, args.envs_per_gpu, 2, max(args.steps, 100), 20, args.aa, device).build()))

  extra_error = []

  def run_extras():
    # (a failure here must not cost the line its headline -- which has been timed by now: it is recorded, every engine built so
    # far is closed and the extras are dropped)
    try:
      build_extras()
      if args.ramp_ms > 0:
        extra_runs[0][1].ramp(args.ramp_ms)
      for _, run in extra_runs:
        run.go()
    except Exception as e:  # pylint: disable=broad-except
      extra_error.append(repr(e))
      try:
        import torch as _t
        _t.cuda.synchronize()
      except Exception:  # pylint: disable=broad-except
        pass
      for _, run in extra_runs:
        run.close()
      del extra_runs[:]

  def finish_extras():
    for label, run in extra_runs:
      if isinstance(run, GroupsRun):
        rate, errs = run.finish()
        extra[label] = {'env_steps_per_s': rate, 'env_errors': errs}
        continue
      r = run.finish()
      if r.get('error'):
        extra[label] = {'error': r['error']}
        continue
      n, short = run.n_envs, run.steps
      ks = r['kernel_ms'] / 1e3 / max(r['launches'], 1)
      extra[label] = {'env_steps_per_s': n * short / r['elapsed'], 'kernel': r['variant']['kernel'], 'kernel_ms': ks * 1e3,
                      'cover_ms': r['cover_ms'] / max(r['launches'], 1), 'resample_ms': r['resample_ms'] / max(r['launches'], 1),
                      'hbm_GBs': r['a_bytes'] * n / ks / 1e9, 'hbm_frac': r['a_bytes'] * n / ks / 1e9 / HBM_PEAK_GBS,
                      'env_errors': r['errors'], 'handoff_list_bytes_per_env': handoff_bytes_per_env(r['variant'], n)}

  res = None
  gated = [False]                            # this rank has been through the gate (set by the wrapper below)

  def gate_once(error):
    gated[0] = True
    return gate(error)

  try:
    res = gpu_run(args.workload, args.envs_per_gpu, args.steps, args.warmup, args.aa, device,
                  barrier=barrier, seed=rank, gate=gate_once if dist is not None else None,
                  verify=64 if (rank == 0 and not args.no_verify) else 0,
                  before=clock_ramp_build, after_gate=clock_ramp_go,
                  protocol_8d=((rank * args.envs_per_gpu, args.gpus * args.envs_per_gpu)
                               if (args.protocol == '8d' and args.workload in ('cluster_s5', 'goal_s5')) else None))
  except Exception as e:  # pylint: disable=broad-except
    if dist is None:
      raise
    if not gated[0]:                        # failed before the gate: tell the others (they are waiting in it)
      gate_once('rank %d: %r' % (rank, e))
    # (after the gate gpu_run raises nothing: failures of the timed region come back in res['error'], so that every
    # rank runs the same collectives below)
  if ramp_info.get('twin') is not None:
    ramp_info.pop('twin').close()
  if dist is None and res is not None and res.get('error'):
    raise SystemExit('timed region failed: ' + res['error'])
  per_rank = None
  elapsed = res['elapsed'] if res else None
  if dist is not None and res is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if dist.get_backend() == 'gloo' else 'cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    per_rank = [None] * world
    n_l = max(res['launches'], 1)
    dist.all_gather_object(per_rank, {'rank': rank, 'device': torch.cuda.get_device_name(device), 'local_device': device,
                                      'elapsed_s': res['elapsed'], 'kernel_ms': res['kernel_ms'] / n_l,
                                      'cover_ms': res['cover_ms'] / n_l, 'resample_ms': res['resample_ms'] / n_l,
                                      'env_errors': res['errors'], 'error': res.get('error')})
    failed = [(r['rank'], r['error']) for r in per_rank if r.get('error')]
    if failed:                               # some rank failed inside the timed region: no value, one line with the errors
      setup_errors[:] = failed
      res = None

  gather = None
  if args.gather_obs and dist is not None and res is not None:
    from spriteworld_amd import distributed as swd
    # both schedules of the gather (distributed.py): 'ring' = all_gather_into_tensor, 'direct' = point-to-point copies to
    # every peer at once (all seven xGMI links of a GPU busy); step_ms and gather_ms apart, HIP events per step
    gather = {m: swd.bench_allgather(args.workload, args.envs_per_gpu, args.steps, args.warmup, args.aa, device, rank, method=m)
              for m in ('ring', 'direct')}

  if rank != 0:
    if dist is not None:
      dist.barrier()
      dist.destroy_process_group()
    return

  if res is None:                            # some rank could not set up: still ONE line from rank 0
    print(json.dumps({'metric': 'env-steps/sec (incl. RGB render) at %d envs' % args.envs_per_gpu, 'value': None,
                      'unit': 'env-steps/s', 'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup,
                      'world_size': world, 'backend': dist.get_backend() if dist is not None else None,
                      'rank_errors': [{'rank': r, 'error': e} for r, e in setup_errors]}))
    if dist is not None:
      dist.barrier()
      dist.destroy_process_group()
    return

  out = assemble_line(args, res, elapsed)
  if res.get('sample') is not None:
    try:
      out.update(verify_against_oracle(res['sample']))
    except Exception as e:  # pylint: disable=broad-except
      out.update({'verified_envs': 0, 'mismatches': None, 'verify_error': repr(e)})
  if dist is not None:
    # what torch.distributed itself saw (the driver can check that RCCL really ran N ranks) and every rank's own kernel time
    out['world_size'] = dist.get_world_size()
    out['backend'] = dist.get_backend()
    out['per_rank'] = per_rank
  if gather is not None:
    out['obs_allgather'] = gather
  out['warmup_effective'] = {
      'timed_engine_warmup_steps': args.warmup,
      'clock_ramp_ms': args.ramp_ms if 'cold_ms_per_step' in ramp_info else 0.0,
      'clock_ramp_steps': ramp_info.get('clock_ramp_steps', 0),
      'how': ('a twin engine of the same workload timed W + K steps cold, then kept the device busy for clock_ramp_ms, before the '
              "timed engine's own W warm-up steps; identical on every rank at every N") if 'cold_ms_per_step' in ramp_info else 'none (--ramp-ms 0)',
  }
  if 'ramp_error' in ramp_info:
    out['warmup_effective']['ramp_error'] = ramp_info['ramp_error']
  if 'cold_ms_per_step' in ramp_info:
    cold_s = ramp_info['cold_ms_per_step'] / 1e3
    out['cold'] = {'value': args.envs_per_gpu * args.gpus / cold_s if cold_s > 0 else None, 'unit': 'env-steps/s',
                   'ms_per_step': ramp_info['cold_ms_per_step'], 'wall_ms_per_step': ramp_info['cold_wall_ms_per_step'],
                   'what': 'the same W warm-up + K timed steps on the twin engine, first thing in the process (device out of '
                           'idle); rank 0, HIP events'}
  if args.gpus == 1 and dist is None and not args.no_extra:
    run_extras()
  if extra_error:
    out['extra_error'] = extra_error[0]
  if extra_runs:
    try:
      finish_extras()
    except Exception as e:  # pylint: disable=broad-except
      out['extra_error'] = repr(e)
    out['extra'] = extra
    out['order'] = 'the extra workloads ran after the timed steps of this line, behind a clock ramp of their own'
  if args.gpus == 1 and not args.no_cpu_baseline:
    out['cpu_baseline'] = cpu_baseline(args.workload, args.aa)
  print(json.dumps(out))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
