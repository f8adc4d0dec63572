"""Rewrites a plain `#define NAME value` of the kernel source (experiment builds): ARG = NAME=VALUE."""
import re


def apply(files, arg, replace_once):
  name, _, value = arg.partition('=')
  k = 'swb_kernels.hip.inc'
  pat = re.compile(r'^#define %s\s+\S+' % re.escape(name), re.M)
  assert len(pat.findall(files[k])) == 1, name
  files[k] = pat.sub('#define %s %s' % (name, value), files[k])
