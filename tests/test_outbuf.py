"""Recycled result buffers (tbmodels_amd/_outbuf.py): a mapping is reused only when nothing looks into it."""

import gc

import numpy as np

from tbmodels_amd import _outbuf


def _addr(arr):
    return arr.__array_interface__["data"][0]


def test_small_arrays_are_plain_numpy():
    arr = _outbuf.empty((10, 8, 8), np.complex128)
    assert arr.flags.owndata and arr.shape == (10, 8, 8) and arr.dtype == np.complex128


def test_reuse_after_the_result_and_all_its_views_are_gone():
    _outbuf.clear()
    shape = (300, 64, 64)  # 19.7 MB
    first = _outbuf.empty(shape, np.complex128)
    assert first.shape == shape and first.flags.writeable and first.flags.c_contiguous and not first.flags.owndata
    first[...] = 1.5
    addr = _addr(first)
    rows = list(first)  # what Model.eigenval hands out
    view = first[7].T[3:5]
    del first
    gc.collect()
    second = _outbuf.empty(shape, np.complex128)
    assert _addr(second) != addr  # rows / view still look into the first mapping
    assert rows[7][0, 0] == 1.5
    del rows
    third = _outbuf.empty(shape, np.complex128)
    assert _addr(third) not in (addr, _addr(second))
    assert view[0, 0] == 1.5
    del view
    gc.collect()
    fourth = _outbuf.empty((290, 64, 64), np.complex128)  # a little smaller: still fits the idle mapping
    assert _addr(fourth) == addr
    assert _outbuf.stats()[0] == 3
    del second, third, fourth
    assert _outbuf.stats()[1] == 3
    _outbuf.clear()
    assert _outbuf.stats() == (0, 0, 0)


def test_other_dtypes_and_much_smaller_requests_get_their_own_mapping():
    _outbuf.clear()
    big = _outbuf.empty((4_000_000,), np.float64)  # 32 MB
    addr = _addr(big)
    del big
    small = _outbuf.empty((1_100_000,), np.float64)  # 8.8 MB: not worth pinning 32 MB for it
    assert _addr(small) != addr
    again = _outbuf.empty((2_000_000, 2), np.float64)
    assert _addr(again) == addr and again.shape == (2_000_000, 2)
    _outbuf.clear()


def test_bounded_number_of_mappings():
    _outbuf.clear()
    keep = [_outbuf.empty((1_100_000 + i,), np.float64) for i in range(_outbuf.MAX_ENTRIES + 4)]
    assert _outbuf.stats()[0] <= _outbuf.MAX_ENTRIES
    for i, arr in enumerate(keep):  # forgotten mappings stay valid for their arrays
        arr[-1] = i
    assert [int(arr[-1]) for arr in keep] == list(range(len(keep)))
    del keep, arr
    gc.collect()
    _outbuf.clear()
