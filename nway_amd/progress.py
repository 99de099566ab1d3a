"""Progress-bar factory and FITS overwrite keyword (interface of nwaylib/progress.py:7-17)."""


def bar(**kwargs):
	try:
		import tqdm
		return tqdm.tqdm
	except ImportError:
		from .logger import FakeProgressBar
		return FakeProgressBar()


# our own FITS writer (nway_amd/_fits.py) takes ``overwrite``
arg_overwrite = 'overwrite'
kwargs_overwrite_true = {arg_overwrite: True}
kwargs_overwrite_false = {arg_overwrite: False}
