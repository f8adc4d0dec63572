"""absl.logging stand-in: messages go to the standard logging module (logger 'absl')."""
import logging as _logging

_logger = _logging.getLogger('absl')


def info(msg, *args):
  _logger.info(msg, *args)


def warning(msg, *args):
  _logger.warning(msg, *args)
