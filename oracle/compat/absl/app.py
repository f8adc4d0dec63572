"""absl.app stand-in."""
import sys


def run(main, argv=None):
  sys.exit(main(argv if argv is not None else sys.argv))
