"""``import nwaylib`` resolves to the MI355X implementation (nway_amd) so that existing
callers of the reference package keep working unchanged."""
import sys

import nway_amd
from nway_amd import *  # noqa: F401,F403
from nway_amd import (nway_match, UndersampledException, EmptyResultException, NormalLogger, NullOutputLogger,
	default_logger, __version__, _create_match_table, _compute_source_densities, _truncate_table)
from nway_amd import bayesdistance, fastskymatch, magnitudeweights, logger, progress
from nway_amd import bayesdistance as bayesdist, fastskymatch as match

for _name in ('bayesdistance', 'fastskymatch', 'magnitudeweights', 'logger', 'progress'):
	sys.modules['nwaylib.' + _name] = getattr(nway_amd, _name)
