"""Loggers with the interface of nwaylib/logger.py:28-53 (``log``, ``warn``, ``progress``)."""
from __future__ import division, print_function

import sys
import warnings


class _PassThrough(object):
	"""progress "bar" that does nothing; iterating over it yields the wrapped iterable"""

	def __init__(self, *args, **kwargs):
		pass

	def __call__(self, iterable):
		return iterable

	def start(self):
		return self

	def increment(self):
		pass

	def finish(self):
		pass


FakeProgressBar = _PassThrough


class NullOutputLogger(object):
	"""silent logger"""

	def log(self, *msg):
		pass

	def warn(self, msg):
		warnings.warn(msg, stacklevel=3)

	def progress(self, *args, **kwargs):
		return _PassThrough()


class NormalLogger(object):
	"""messages go to stderr, one per line"""

	def log(self, msg):
		sys.stderr.write('%s\n' % msg)

	def warn(self, msg):
		warnings.warn(msg, stacklevel=3)

	def progress(self, ndigits=6, *args, **kwargs):
		from . import progress
		return progress.bar(ndigits=ndigits, *args, **kwargs)
