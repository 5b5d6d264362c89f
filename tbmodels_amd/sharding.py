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
        if stop > start:
            slab[: stop - start] = np.asarray(evaluate(k[start:stop])).reshape(stop - start, n_orb)
        pieces = self.group.all_gather_array(slab)
        return np.concatenate(pieces, axis=0)[:n_k].copy()

    def _device_gather(self, k, start, stop, per, n_k, n_orb):
        with self.model._call_lock:  # pylint: disable=protected-access  # staged handle stays valid for the whole exchange
            return self._device_gather_locked(k, start, stop, per, n_k, n_orb)

    def _device_gather_locked(self, k, start, stop, per, n_k, n_orb):
        lib = _lib.lib()
        handle = self.model._staged()  # pylint: disable=protected-access
        comm = self._communicator()
        dev = self.device

        def dmalloc(nbytes):
            p = ctypes.c_void_p()
            _lib.check(lib.tbk_device_malloc(dev, max(nbytes, 8), ctypes.byref(p)))
            return p

        k_slab = np.ascontiguousarray(k[start:stop])
        d_k = dmalloc(k_slab.nbytes)
        d_send = dmalloc(per * n_orb * 8)
        d_recv = dmalloc(self.world * per * n_orb * 8)
        try:
            if stop > start:
                _lib.check(lib.tbk_memcpy_h2d(dev, d_k, _lib.ptr(k_slab), k_slab.nbytes))
                _lib.check(lib.tbk_eigenval_device(handle, d_k, stop - start, d_send))
            _lib.check(lib.tbk_comm_allgather_f64(comm, handle, d_send, d_recv, per * n_orb))
            _lib.check(lib.tbk_eigenval_check(handle))  # synchronises
            out = np.empty((self.world * per, n_orb), dtype=np.float64)
            _lib.check(lib.tbk_memcpy_d2h(dev, _lib.ptr(out), d_recv, out.nbytes))
        finally:
            for p in (d_k, d_send, d_recv):
                lib.tbk_device_free(dev, p)
        return out[:n_k].copy()
