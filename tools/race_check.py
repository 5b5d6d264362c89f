#!/usr/bin/env python3
"""Repeat-run bit identity of the reduction paths (race detector): python tools/race_check.py <n> <nk> <reps>
Environment (read once per process): TBK_BAND_FUSE, TBK_BAND."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbmodels_amd import _lib  # noqa: E402

n, nk, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(n)
m = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
h = np.ascontiguousarray((m + m.conj().transpose(0, 2, 1)) / 2)
lib = _lib.lib()
first = None
bad_total = 0
for rep in range(reps):
    d, e = np.empty((nk, n)), np.empty((nk, n))
    _lib.check(lib.tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(h), 0, _lib.ptr(d), _lib.ptr(e), None))
    if first is None:
        first = (d.copy(), e.copy())
    else:
        bad = np.flatnonzero(np.any(d != first[0], axis=1) | np.any(e != first[1], axis=1))
        bad_total += len(bad)
        if len(bad):
            i = bad[0]
            print("rep %d: %d matrices differ; first %d, max |dd| %.2e |de| %.2e" % (
                rep, len(bad), i, np.abs(d[i] - first[0][i]).max(), np.abs(e[i] - first[1][i]).max()))
print("n=%d nk=%d reps=%d env FUSE=%s BAND=%s: %d differing matrices" % (
    n, nk, reps, os.environ.get("TBK_BAND_FUSE"), os.environ.get("TBK_BAND"), bad_total))
