"""Minimal stand-in for `dm_env.specs` (test infrastructure only)."""
import numpy as np


class Array(object):

  def __init__(self, shape, dtype, name=None):
    self.shape = tuple(int(d) for d in shape)
    self.dtype = np.dtype(dtype)
    self.name = name

  def __repr__(self):
    return 'Array(shape={}, dtype={}, name={})'.format(self.shape, self.dtype,
                                                       self.name)


class BoundedArray(Array):

  def __init__(self, shape, dtype, minimum, maximum, name=None):
    super(BoundedArray, self).__init__(shape, dtype, name)
    self.minimum = np.asarray(minimum)
    self.maximum = np.asarray(maximum)


class DiscreteArray(BoundedArray):

  def __init__(self, num_values, dtype=np.int32, name=None):
    super(DiscreteArray, self).__init__((), dtype, 0, num_values - 1, name)
    self.num_values = num_values
