#!/usr/bin/env python
"""Turns the outputs of tools/r02_last.sh (gpurun_out/<TAG>/) into the committed artefacts under profiles/:
the rocprofv3 summaries of the three judged workloads (kernel trace + the PMC passes), r02_counters.json keyed by the
library's build id (what bench.py reads), and the default bench line.
usage: python tools/r02_assemble.py gpurun_out/r02last"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, 'profiles')
WORKLOADS = (('default', 'cluster_s5', 5), ('aa1', 'cluster_s5', 1), ('embodied_s12_128', 'embodied_s12', 5))
N = 8192


def main():
  src = sys.argv[1]
  pmc = {}
  for name in ('insts', 'active', 'write', 'fetch'):
    with open(os.path.join(src, 'pmc_%s.json' % name)) as f:
      for key, vals in json.load(f).items():
        pmc.setdefault(key, {}).update(vals)
  records = []
  for tag, wl, aa in WORKLOADS:
    c = pmc['%s:%d:0' % (wl, aa)]
    with open(os.path.join(src, tag, 'bench_unprofiled.json')) as f:
      b = json.loads(f.readlines()[-1])
    fetch, write = c['FETCH_SIZE'], c['WRITE_SIZE']
    rec = {
        'build_id': b['roofline']['build_id'], 'workload': wl, 'envs': N, 'anti_aliasing': aa, 'kernel': b['roofline']['kernel'],
        'insts_valu_per_wave': c['SQ_INSTS_VALU'] / N, 'insts_salu_per_wave': c['SQ_INSTS_SALU'] / N,
        'insts_lds_per_wave': c['SQ_INSTS_LDS'] / N, 'insts_smem_per_wave': c['SQ_INSTS_SMEM'] / N,
        'insts_vmem_wr_per_wave': c['SQ_INSTS_VMEM_WR'] / N, 'insts_vmem_rd_per_wave': c['SQ_INSTS_VMEM_RD'] / N,
        'wave_cycles_per_wave': c['SQ_WAVE_CYCLES'] / N,
        'active_inst_valu_per_wave': c['SQ_ACTIVE_INST_VALU'] / N, 'active_inst_sca_per_wave': c['SQ_ACTIVE_INST_SCA'] / N,
        'wait_any_per_wave': c['SQ_WAIT_ANY'] / N, 'wait_inst_any_per_wave': c['SQ_WAIT_INST_ANY'] / N,
        'lds_bank_conflict_per_wave': c['SQ_LDS_BANK_CONFLICT'] / N,
        'resident_waves_per_simd': b['roofline']['waves_per_simd'],
        'fetch_size_kb': fetch, 'write_size_kb': write, 'fetch_correction': 2.0,
        'hbm_traffic_bytes_per_launch': int((2.0 * fetch + write) * 1024),
        'algorithmic_bytes_per_launch': b['roofline']['algorithmic_bytes_per_env_step'] * N,
        'kernel_ms_unprofiled': b['roofline']['kernel_ms'],
        'source': 'profiles/r02_rocprofv3_summary_%s.md (rocprofv3 --pmc, four separate passes of tools/phase_profile.py '
                  'pmc-run over the three workloads, 8 measured launches each; SQ counters in quad-cycles, per wave = per '
                  'environment; FETCH_SIZE doubled per MI355X_MICROARCH.md)' % tag,
    }
    records.append(rec)
    # summary = the kernel trace of `python bench.py ...` + the PMC figures of the same kernel
    with open(os.path.join(src, tag, 'summary.md')) as f:
      text = f.read()
    head, _, tail = text.partition('\n```\n')
    lines = [head.rstrip(), '', '## PMC passes (`rocprofv3 --pmc`, separate processes; averages over 8 launches of this kernel at '
             '%d environments)' % N, '', '| counter | per dispatch | per wave (= per environment) |', '|---|---|---|']
    for k in sorted(c):
      per_wave = '' if k in ('FETCH_SIZE', 'WRITE_SIZE') else '%.1f' % (c[k] / N)
      lines.append('| %s | %.6g | %s |' % (k, c[k], per_wave))
    traffic = rec['hbm_traffic_bytes_per_launch']
    lines += ['', 'HBM traffic per launch = WRITE_SIZE + 2 x FETCH_SIZE (gfx950 correction) = %.1f MB; algorithmic bytes = %.1f MB '
              '(x%.2f).  VALU issue: SQ_ACTIVE_INST_VALU x %d resident waves / SQ_WAVE_CYCLES = %.0f %%.' %
              (traffic / 1e6, rec['algorithmic_bytes_per_launch'] / 1e6, traffic / rec['algorithmic_bytes_per_launch'],
               rec['resident_waves_per_simd'],
               100.0 * rec['active_inst_valu_per_wave'] * rec['resident_waves_per_simd'] / rec['wave_cycles_per_wave']),
              '', 'Unprofiled bench line of the same command:', '', '```', tail.replace('```', '').strip(), '```', '']
    with open(os.path.join(PROFILES, 'r02_rocprofv3_summary_%s.md' % tag), 'w') as f:
      f.write('\n'.join(lines))
  with open(os.path.join(PROFILES, 'r02_counters.json'), 'w') as f:
    json.dump({'records': records}, f, indent=1)
  shutil.copy(os.path.join(src, 'bench_default.json'), os.path.join(PROFILES, 'r02_bench_default.json'))
  for rec in records:
    print(rec['workload'], rec['anti_aliasing'], rec['kernel'], 'VALU %.0f SALU %.0f' % (rec['insts_valu_per_wave'], rec['insts_salu_per_wave']),
          'traffic x%.2f' % (rec['hbm_traffic_bytes_per_launch'] / rec['algorithmic_bytes_per_launch']), 'kernel %.4f ms' % rec['kernel_ms_unprofiled'])


if __name__ == '__main__':
  main()
