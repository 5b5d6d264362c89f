"""``python -m tbmodels_amd <command>`` -- see :mod:`tbmodels_amd._cli`."""

import sys

from ._cli import main

sys.exit(main())
