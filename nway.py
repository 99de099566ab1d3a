#!/usr/bin/env python
"""nway.py -- same command line as the reference's script, running on the GPU.

    nway.py --radius 10 cat_primary.fits :pos_err cat_secondary.fits 0.1 --out=out.fits

All of the work is in nway_amd/cli.py."""
import sys

from nway_amd.cli import main

if __name__ == '__main__':
	sys.exit(main())
