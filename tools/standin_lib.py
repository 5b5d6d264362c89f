"""A stand-in for libtbk on a machine WITHOUT a GPU -- test infrastructure, never a product path (the product has no CPU
implementation: `tbmodels_amd` raises without libtbk.so and a device).

"Device" buffers live in NumPy arrays, slabs are evaluated by the oracle (oracle/tbk_oracle.py), the collectives go through the
process group of tbmodels_amd.rendezvous.  Two users:

* tests/test_bench_helpers.py -- bench.py's strong-scaling leg at 2 and 8 ranks (slabs, result layout, status words);
* ``bench.py --dry-ranks`` -- the exact command the scaling driver runs (``python -m torch.distributed.run ... bench.py --gpus N``)
  on the CPU with a tiny model: every rank walks main()'s control flow (rendezvous, communicator, step loop with the overlapped
  gather, per-rank split, strong-scaling leg, the one JSON line) -- what no one-GPU box can exercise at N > 1.
"""

import numpy as np


class StandInLib:
    def __init__(self, group, world, rank, arrays):
        import ctypes

        self.ctypes = ctypes
        self.group, self.world, self.rank, self.arrays = group, world, rank, arrays
        self.buffers = {}
        self.next_id = 1

    def _buf(self, pointer):
        value = pointer.value if hasattr(pointer, "value") else pointer
        return self.buffers[int(value)]

    def tbk_device_malloc(self, device, nbytes, out):
        handle = self.next_id
        self.next_id += 1
        self.buffers[handle] = np.zeros(int(nbytes) // 8 + 1)
        out._obj.value = handle
        return 0

    def tbk_device_free(self, device, pointer):
        self.buffers.pop(int(pointer.value), None)
        return 0

    def _host(self, pointer, count):
        return np.ctypeslib.as_array(self.ctypes.cast(pointer, self.ctypes.POINTER(self.ctypes.c_double)), shape=(count,))

    def tbk_memcpy_h2d(self, device, dst, src, nbytes):
        self._buf(dst)[: nbytes // 8] = self._host(src, nbytes // 8)
        return 0

    def tbk_memcpy_d2h(self, device, dst, src, nbytes):
        self._host(dst, nbytes // 8)[:] = self._buf(src)[: nbytes // 8]
        return 0

    def tbk_model_set_option(self, *args):
        return 0

    def tbk_synchronize(self, *args):
        return 0

    def tbk_comm_synchronize(self, *args):
        return 0

    def tbk_comm_ranks(self, comm, count, rank):
        count._obj.value = self.world
        if rank is not None:
            rank._obj.value = self.rank
        return 0

    def _eig(self, d_k, nk):
        from oracle import tbk_oracle as oracle

        k = self._buf(d_k)[: nk * 3].reshape(nk, 3)
        return np.array(oracle.eigenval(self.arrays["R"], self.arrays["hop"], k)).reshape(nk, -1)

    def tbk_eigenval_device_hint(self, model, d_k, h_k, nk, d_out):
        eig = self._eig(d_k, nk)
        self._buf(d_out)[: eig.size] = eig.reshape(-1)
        return 0

    fail_rank = -1  # this rank's solver "fails" inside the gather: like the library, it still walks every collective and
    # its verdict reaches every rank through the status words (the LAST collective)

    def tbk_eigenval_device_gather(self, comm, model, d_k, h_k, nk, per, host_status, d_all, d_status):
        n = self.arrays["n_orb"]
        slab = np.zeros((per, n))
        if nk:
            slab[:nk] = self._eig(d_k, nk)
        pieces = self.group.all_gather_array(slab)
        self._buf(d_all)[: self.world * per * n] = np.concatenate(pieces).reshape(-1)
        mine = np.array([3.0 if self.rank == self.fail_rank else 0.0])
        self._buf(d_status)[: self.world] = np.concatenate(self.group.all_gather_array(mine))
        return 0

    # ---- what bench.py's main() needs beyond the strong-scaling leg (``--dry-ranks``) ------------------------------------
    def _set(self, ref, value):
        ref._obj.value = value  # pylint: disable=protected-access
        return 0

    def tbk_model_create_dense(self, device, dim, n_orb, n_r, r_ptr, hop_ptr, out):
        return self._set(out, 1)

    def tbk_model_destroy(self, model):
        return 0

    def tbk_comm_unique_id(self, uid):
        return 0

    def tbk_comm_create(self, device, world, rank, uid, out):
        return self._set(out, 2)

    def tbk_comm_destroy(self, comm):
        return 0

    def tbk_comm_wait_slot(self, comm, model, slot):
        return 0

    def tbk_eigenval_check(self, model):
        return 0

    def tbk_get_timing(self, model, ms, launches, reset):
        if ms is not None:
            for i in range(len(ms)):
                ms[i] = 0.0
        if launches is not None:
            for i in range(len(launches)):
                launches[i] = 0
        return 0

    def tbk_mfma_f64_peak(self, device, out):
        return self._set(out, 0.0)

    def _allgather(self, src, dst, count):
        pieces = self.group.all_gather_array(self._buf(src)[:count].copy()) if self.group is not None else [self._buf(src)[:count]]
        self._buf(dst)[: self.world * count] = np.concatenate(pieces)
        return 0

    def tbk_comm_allgather_f64(self, comm, model, src, dst, count):
        return self._allgather(src, dst, count)

    def tbk_comm_allgather_f64_overlapped(self, comm, model, src, dst, count, slot):
        return self._allgather(src, dst, count)
