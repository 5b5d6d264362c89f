#!/usr/bin/env python3
"""Repeat-run bit identity (race detector, GPU box).

    python tools/race_check.py <n> <nk> <reps>                 the reduction alone (tbk_tridiagonal_reduce) on random matrices
    python tools/race_check.py --model <n> <n_r> <nk> <reps>   the whole path (H(k) + reduction + bisection) through Model.eigenval_array

Environment (read once per process): TBK_BAND_FUSE, TBK_BAND.  The second form found the one-in-250 000 race of round 3 (a wave
overwriting partial sums another wave was still adding up, csrc/tbk_eig_band.hip)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tbmodels_amd import _lib  # noqa: E402


def reduction_alone(n, nk, reps):
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


def whole_path(n, n_r, nk, reps):
    import tbmodels_amd
    from tbmodels_amd import synthetic as syn

    r_vec, hop, pos = syn.dense_model_arrays(n, n_r, syn.MODEL_SEED + n)
    k = syn.random_kpoints(nk, seed=n)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    first = model.eigenval_array(k).copy()
    ham0 = model.hamilton(k[:256]).copy()
    bad_total = 0
    for rep in range(reps):
        again = model.eigenval_array(k)
        bad = np.flatnonzero(np.any(again != first, axis=1))
        bad_total += len(bad)
        if len(bad):
            i = bad[0]
            print("rep %d: %d rows differ; first %d: max |dE| %.2e, columns %s" % (
                rep, len(bad), i, np.abs(again[i] - first[i]).max(), np.flatnonzero(again[i] != first[i])[:8]))
        if rep % 8 == 0 and not np.array_equal(model.hamilton(k[:256]), ham0):
            print("rep %d: hamilton differs" % rep)
    print("n=%d n_r=%d nk=%d reps=%d FUSE=%s: %d differing rows" % (n, n_r, nk, reps, os.environ.get("TBK_BAND_FUSE"), bad_total))


if __name__ == "__main__":
    if sys.argv[1] == "--model":
        whole_path(*(int(x) for x in sys.argv[2:6]))
    else:
        reduction_alone(*(int(x) for x in sys.argv[1:4]))
