#!/usr/bin/env python
"""ISA resource table of every kernel of the shipped sources (cover, resample, fill, factors, sampler): compiles both translation units with the
flags of spriteworld_amd/build.py plus -save-temps (in a temporary directory) and reads the kernel descriptors' metadata.
usage: python tools/isa_resources.py [CSRC_DIR] > table.md"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spriteworld_amd import build  # noqa: E402


def kernels_of(asm):
  """{demangled-ish name: {field: value}} from the .amdgpu_metadata block of an assembly file."""
  out = {}
  meta = asm[asm.index('.amdgpu_metadata'):]
  for block in meta.split('  - .agpr_count:')[1:]:
    block = '  - .agpr_count:' + block
    name = re.search(r'\.name:\s+(\S+)', block).group(1)
    if 'swb_' not in name:
      continue
    m = re.search(r'swb_cover_kernelILi(\d+)ELb([01])ELb([01])E', name)
    if m:
      key = 'swb_cover_kernel<%s%s%s>' % (m.group(1), ',OV' if m.group(2) == '1' else '', ',PAINT' if m.group(3) == '1' else '')
    else:
      m = re.search(r'swb_resample_kernelILi(\d+)E', name)
      key = 'swb_resample_kernel<%s>' % m.group(1) if m else re.sub(r'^_Z\d+', '', name).split('1')[0] if False else None
      if key is None:
        m = re.search(r'_Z\d+(swb_\w+?_kernel)', name)
        key = m.group(1) if m else name
    f = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, block).group(1))
    out[key] = dict(vgpr=f('vgpr_count'), agpr=f('agpr_count'), sgpr=f('sgpr_count'), vspill=f('vgpr_spill_count'),
                    sspill=f('sgpr_spill_count'), scratch=f('private_segment_fixed_size'))
  return out


def main():
  rows = []
  csrc = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else build.CSRC        # another copy of the sources (tools/try_patch.py)
  with tempfile.TemporaryDirectory() as tmp:
    for unit, extra in build.UNITS:
      cmd = ['hipcc'] + build.COMMON + extra + ['-DSWB_BUILD_ID="isa"', '-save-temps', '-c', '-o', os.path.join(tmp, unit + '.o'),
                                               os.path.join(csrc, unit)]
      subprocess.check_call(cmd, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
      asm = open(os.path.join(tmp, unit.replace('.hip', '') + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
      for key, r in kernels_of(asm).items():
        rows.append((unit, key, r))
  def order(row):
    nums = [int(v) for v in re.findall(r'\d+', row[1])]
    return (row[1].split('<')[0], nums, 'OV' in row[1])
  rows.sort(key=order)
  print('| kernel | translation unit | VGPRs | AGPRs | SGPRs | VGPR spills | SGPR spills | scratch B/lane | waves/SIMD by registers |')
  print('|---|---|---|---|---|---|---|---|---|')
  for unit, key, r in rows:
    alloc = (r['vgpr'] + r['agpr'] + 7) // 8 * 8
    print('| `%s` | %s | %d | %d | %d | %d | %d | %d | %d |' % (key, unit, r['vgpr'], r['agpr'], r['sgpr'], r['vspill'], r['sspill'],
                                                             r['scratch'], min(8, 512 // max(alloc, 1))))


if __name__ == '__main__':
  main()
