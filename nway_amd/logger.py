"""Message sinks accepted by every ``logger=`` argument of the package.

Same surface as the reference's logger module (``NormalLogger``, ``NullOutputLogger``,
``FakeProgressBar``; methods ``log``, ``warn``, ``progress``), so objects written for nwaylib
can be passed in unchanged.  Warnings always go through the ``warnings`` module; only the
destination of ordinary messages differs between the two loggers.
"""
from __future__ import division, print_function

import sys
import warnings


class FakeProgressBar(object):
	"""Stands in for a progress bar: wraps an iterable without reporting anything."""

	def __init__(self, *unused_args, **unused_kwargs):
		self.steps = 0

	def __call__(self, iterable):
		return iterable

	def start(self):
		return self

	def increment(self):
		self.steps += 1

	def finish(self):
		return None


class _Logger(object):
	stream = None  # file-like object for ordinary messages, None = discard

	def log(self, *messages):
		if self.stream is not None:
			self.stream.write(' '.join(str(m) for m in messages) + '\n')

	def warn(self, message):
		# stacklevel 3: attribute the warning to the caller of the library function
		warnings.warn(message, stacklevel=3)

	def progress(self, *args, **kwargs):
		return FakeProgressBar()


class NullOutputLogger(_Logger):
	"""keeps quiet (warnings are still raised through ``warnings``)"""


class NormalLogger(_Logger):
	"""one line per message on stderr; tqdm progress bars when tqdm is installed"""

	@property
	def stream(self):
		return sys.stderr

	def progress(self, ndigits=6, *args, **kwargs):
		from . import progress
		return progress.bar(ndigits=ndigits, *args, **kwargs)
