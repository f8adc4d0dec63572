"""absl.flags stand-in: DEFINE_integer / DEFINE_string and a FLAGS object with attribute access."""


class _Flags(object):

  def __init__(self):
    object.__setattr__(self, '_values', {})

  def __getattr__(self, name):
    try:
      return self._values[name]
    except KeyError:
      raise AttributeError(name)

  def __setattr__(self, name, value):
    self._values[name] = value


FLAGS = _Flags()


def _define(name, default, help_text=None):
  del help_text
  FLAGS._values.setdefault(name, default)


DEFINE_integer = DEFINE_string = DEFINE_float = DEFINE_bool = DEFINE_boolean = _define
