"""
Sharded evaluation of ``Model.eigenval`` over the GPUs of one node: one process per GPU.

The reference has no distributed code; its only concession is that models pickle, so users can farm
k-points out with ``multiprocessing`` (``/root/reference/doc/source/tutorial.rst:85``).  k-points are
independent (``_tb_model.py:1111-1123`` has no cross-k term), so the path shards trivially:

* the staged hoppings are replicated on every rank's GPU;
* the k list is cut into ``world`` contiguous slabs (order preserving, so the gathered result is
  already in caller order);
* the one exchange is an all-gather of eigenvalue slabs: RCCL over xGMI on device buffers
  (``tbk_comm_allgather_f64``), or -- when the process group has no GPUs (the CPU test-suite, gloo) --
  a host all-gather through ``torch.distributed``.

The process group (``tbmodels_amd.rendezvous``: torch-free ``FileGroup`` or a ``TorchGroup`` adapter) is
used for rendezvous only (ranks, barrier, handing the RCCL unique id around); the data path is
``libtbk.so``.
"""

import ctypes

import numpy as np

from . import _lib

__all__ = ("slab_bounds", "ShardedEigenval")


def slab_bounds(n_k, world, rank):
    """Rows ``[start, stop)`` of rank ``rank``: contiguous slabs of ``ceil(n_k / world)`` (the last ones may be short or empty)."""
    per = -(-n_k // world) if world > 0 else n_k
    start = min(n_k, rank * per)
    return start, min(n_k, start + per)


class ShardedEigenval:
    """
    ``ShardedEigenval(model, dist)(k)`` returns, on every rank, the eigenvalues of all k-points as a
    ``(NK, N)`` array in the order of ``k``.

    Parameters
    ----------
    model :
        a :class:`tbmodels_amd.Model` (every rank holds the same one).
    group :
        a ``tbmodels_amd.rendezvous`` process group (``FileGroup`` / ``TorchGroup``), or an initialised
        ``torch.distributed`` module (wrapped in a ``TorchGroup``).
    device :
        this rank's GPU index; ``None`` = no GPU collective: slabs are evaluated by ``evaluate`` and
        gathered on the host.
    evaluate :
        ``evaluate(k_slab) -> (n, N) array``; defaults to ``model.eigenval`` on ``device``.  The CPU
        test-suite injects the oracle here to exercise the slab / gather logic under gloo.
    """

    def __init__(self, model, group, device=None, evaluate=None):
        from .rendezvous import TorchGroup  # pylint: disable=import-outside-toplevel

        if hasattr(group, "get_world_size"):  # a torch.distributed module
            group = TorchGroup(group)
        self.model = model
        self.group = group
        self.world = group.world
        self.rank = group.rank
        self.device = device
        self.evaluate = evaluate
        self._comm = None
        self._buffers = {}  # name -> (device pointer, bytes): reused from call to call
        if device is not None:
            model.device = int(device)

    # ------------------------------------------------------------------ RCCL communicator
    def _communicator(self):
        if self._comm is None:
            lib = _lib.lib()
            uid = np.zeros(128, dtype=np.uint8)
            if self.rank == 0:
                _lib.check(lib.tbk_comm_unique_id(_lib.ptr(uid)))
            uid = np.frombuffer(self.group.broadcast_bytes(uid.tobytes(), src=0), dtype=np.uint8).copy()
            comm = ctypes.c_void_p()
            _lib.check(lib.tbk_comm_create(self.device, self.world, self.rank, _lib.ptr(uid), ctypes.byref(comm)))
            self._comm = comm
        return self._comm

    def close(self):
        for pointer, _ in self._buffers.values():
            _lib.lib().tbk_device_free(self.device, pointer)
        self._buffers = {}
        if self._comm is not None:
            _lib.lib().tbk_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass

    # ------------------------------------------------------------------ evaluation
    def __call__(self, k):
        k = np.ascontiguousarray(np.array(k, ndmin=2), dtype=np.float64)
        n_k = k.shape[0]
        n_orb = self.model.size
        start, stop = slab_bounds(n_k, self.world, self.rank)
        per = -(-n_k // self.world)
        if self.device is None or self.evaluate is not None:
            return self._host_gather(k, start, stop, per, n_k, n_orb)
        return self._device_gather(k, start, stop, per, n_k, n_orb)

    def _host_gather(self, k, start, stop, per, n_k, n_orb):
        evaluate = self.evaluate or (lambda ks: np.array(self.model.eigenval(ks)).reshape(len(ks), n_orb))
        slab = np.zeros((per, n_orb), dtype=np.float64)
        failure = None
        if stop > start:
            try:
                slab[: stop - start] = np.asarray(evaluate(k[start:stop])).reshape(stop - start, n_orb)
            except Exception as exc:  # pylint: disable=broad-except  # e.g. NaN in this rank's slab only, out of memory, ...
                failure = exc
        # all ranks take the same branch: the failure of one is raised on every rank, after the same collectives
        failed = self.group.allreduce_max(1.0 if failure is not None else 0.0)
        pieces = self.group.all_gather_array(slab)
        if failure is not None:
            raise failure
        if failed:
            raise ValueError("a peer rank failed while evaluating its k slab")
        return np.concatenate(pieces, axis=0)[:n_k].copy()

    def _device_gather(self, k, start, stop, per, n_k, n_orb):
        with self.model._call_lock:  # pylint: disable=protected-access  # staged handle stays valid for the whole exchange
            return self._device_gather_locked(k, start, stop, per, n_k, n_orb)

    def _buffer(self, name, nbytes):
        """Persistent device buffer `name` of at least `nbytes` (grow-only: no malloc / free -- each one a device
        synchronisation -- on the calls after the largest one)."""
        have = self._buffers.get(name)
        if have is not None and have[1] >= nbytes:
            return have[0]
        lib = _lib.lib()
        if have is not None:
            lib.tbk_device_free(self.device, have[0])
            del self._buffers[name]
        p = ctypes.c_void_p()
        size = max(int(nbytes), 8)
        _lib.check(lib.tbk_device_malloc(self.device, size, ctypes.byref(p)))
        self._buffers[name] = (p, size)
        return p

    def _device_gather_locked(self, k, start, stop, per, n_k, n_orb):
        lib = _lib.lib()
        dev = self.device
        # Step 1, agreement: everything that can fail on ONE rank in front of the collectives -- the communicator's
        # id exchange aside, which every rank walks alike: staging the model, growing the device buffers -- happens
        # inside the try, and its outcome is all-gathered as one status word per rank (8 bytes through buffers the
        # communicator owns since its creation) before anybody enters the data collectives.  A rank that raised on
        # its own in front of the gather used to leave its peers hanging in it.
        comm = self._communicator()  # first call: a host broadcast of the RCCL id, taken by every rank alike
        failure, handle, d_k, d_all, d_status, k_slab = None, None, None, None, None, None
        try:
            handle = self.model._staged()  # pylint: disable=protected-access
            d_all = self._buffer("all", self.world * per * n_orb * 8)
            d_status = self._buffer("status", self.world * 8)
            _lib.check(lib.tbk_comm_prepare_gather(comm, n_orb, per))  # the gather's landing area: its allocation can fail too
            if stop > start:
                k_slab = np.ascontiguousarray(k[start:stop])
                d_k = self._buffer("k", k_slab.nbytes)
                _lib.check(lib.tbk_memcpy_h2d(dev, d_k, _lib.ptr(k_slab), k_slab.nbytes))
        except Exception as exc:  # pylint: disable=broad-except
            failure = exc
        verdict = np.zeros(self.world)
        _lib.check(lib.tbk_comm_agree(comm, _status_of(failure), _lib.ptr(verdict)))
        if failure is not None:
            raise failure
        if verdict.max() != 0:
            bad = int(np.argmax(verdict))
            _raise_status(int(verdict[bad]), "rank %d failed (status %d) while preparing its k slab" % (bad, int(verdict[bad])))
        # Step 2, the call: this rank's slab is evaluated into its rows of the result, finished blocks of rows leave on
        # the communicator's stream while later k chunks compute (tbk_eigenval_device_gather), the solvers' flags travel
        # as the LAST collective -- a NaN in one rank's slab raises on every rank, after the same collectives.  The host
        # slab goes along as the structure hint: mesh slabs are folded (include/tbk.h).
        local = lib.tbk_eigenval_device_gather(comm, handle, d_k, _lib.ptr(k_slab) if k_slab is not None else None,
                                               stop - start, per, 0, d_all, d_status)
        local_message = _lib.last_error() if local != 0 else ""
        _lib.check(lib.tbk_comm_synchronize(comm))  # ONE wait: eigenvalues of every rank and the verdict are in place
        _lib.check(lib.tbk_memcpy_d2h(dev, _lib.ptr(verdict), d_status, verdict.nbytes))
        worst = int(verdict.max())
        if worst != 0:
            bad = int(np.argmax(verdict))
            if bad == self.rank and local_message:
                _raise_status(worst, local_message)
            if bad == self.rank and worst == _lib.TBK_ERR_NOT_FINITE:
                _raise_status(worst, "array must not contain infs or NaNs")  # scipy's message (tbk_eigenval_check)
            _raise_status(worst, "rank %d failed (status %d) while evaluating its k slab" % (bad, worst))
        out = np.empty((self.world * per, n_orb), dtype=np.float64)
        _lib.check(lib.tbk_memcpy_d2h(dev, _lib.ptr(out), d_all, out.nbytes))
        return out[:n_k].copy() if n_k != len(out) else out


def _status_of(exc):
    """The ``tbk_status`` that `_lib.check` maps to the type of `exc` (the inverse of :func:`_raise_status`)."""
    if exc is None:
        return 0
    if isinstance(exc, np.linalg.LinAlgError):
        return _lib.TBK_ERR_NO_CONVERGENCE
    if isinstance(exc, ValueError):
        return _lib.TBK_ERR_NOT_FINITE
    if isinstance(exc, MemoryError):
        return _lib.TBK_ERR_MEMORY
    return _lib.TBK_ERR_DEVICE


def _raise_status(status, message):
    """The exception `_lib.check` would raise for `status`, with an explicit message."""
    if status in (_lib.TBK_ERR_ARGUMENT, _lib.TBK_ERR_NOT_FINITE):
        raise ValueError(message)
    if status == _lib.TBK_ERR_MEMORY:
        raise MemoryError(message)
    if status == _lib.TBK_ERR_NO_CONVERGENCE:
        raise np.linalg.LinAlgError(message)
    raise RuntimeError(message)
