"""
Result arrays for large batches, recycled once the caller has dropped them.

``hamilton()`` returns 64 KiB per k-point at 64 orbitals.  A fresh ``np.empty`` of that size has no pages behind it,
and the device-to-host copy into it then runs at the kernel's page-fault rate (~21 GB/s on the GPU hosts, whoever takes
the faults) instead of the PCIe rate (~52 GB/s): glibc hands every array above 32 MiB back to the system when it is
freed, so a loop over batches pays that on every call.  :func:`empty` therefore backs large results with anonymous
mappings that it keeps and reuses as soon as no array looks into them any more.

Liveness: the result is ``flat.reshape(shape)`` where ``flat = np.frombuffer(mapping)``; NumPy makes ``flat`` the
``base`` of the result and of every view derived from it (slices, rows of ``list(out)``, ``.T``, ``.view()``, a
``memoryview``), so the mapping is idle exactly when the weak reference to ``flat`` is dead.  Arrays that stay alive
keep their mapping for themselves; it is unmapped with them.

The reference returns arrays that own their data (``out.flags.owndata``); these do not -- nothing else differs.
"""

import math
import mmap
import threading
import weakref

import numpy as np

MIN_BYTES = 8 << 20  # below that malloc recycles freed blocks by itself
MAX_ENTRIES = 8
MAX_IDLE_BYTES = 4 << 30
_GRANULE = 2 << 20

_lock = threading.Lock()
_entries = []  # [mapping, size in bytes, weak reference to the flat array over it or None]


def _idle(entry):
    return entry[2] is None or entry[2]() is None


def _trim():
    idle_bytes = sum(e[1] for e in _entries if _idle(e))
    while _entries and (len(_entries) > MAX_ENTRIES or idle_bytes > MAX_IDLE_BYTES):
        # idle mappings go first (largest first); a mapping that is still looked into is merely forgotten here
        idle = [e for e in _entries if _idle(e)]
        victim = max(idle, key=lambda e: e[1]) if idle else _entries[0]
        if _idle(victim):
            idle_bytes -= victim[1]
        _entries.remove(victim)


def empty(shape, dtype):
    """Like ``np.empty(shape, dtype)`` (C order); large arrays come from the recycled mappings."""
    dtype = np.dtype(dtype)
    count = math.prod(shape)  # (np.prod of a three-element tuple: 2.9 us of a 25 us one-k call)
    nbytes = count * dtype.itemsize
    if nbytes < MIN_BYTES:
        return np.empty(shape, dtype)
    with _lock:
        chosen = None
        for entry in _entries:
            if _idle(entry) and nbytes <= entry[1] <= nbytes + nbytes // 2 and (chosen is None or entry[1] < chosen[1]):
                chosen = entry
        if chosen is None:
            size = (nbytes + _GRANULE - 1) // _GRANULE * _GRANULE
            try:
                mapping = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
            except (OSError, ValueError, AttributeError):
                return np.empty(shape, dtype)
            try:
                mapping.madvise(mmap.MADV_HUGEPAGE)
            except (OSError, ValueError, AttributeError):
                pass
            chosen = [mapping, size, None]
            _entries.append(chosen)
        flat = np.frombuffer(chosen[0], dtype=dtype, count=count)
        chosen[2] = weakref.ref(flat)
        _trim()
    return flat.reshape(shape)


def stats():
    """``(mappings, idle mappings, idle bytes)`` -- for tests."""
    with _lock:
        idle = [e for e in _entries if _idle(e)]
        return len(_entries), len(idle), sum(e[1] for e in idle)


def clear():
    """Forget every mapping (idle ones are unmapped at once, the others with their arrays)."""
    with _lock:
        del _entries[:]
