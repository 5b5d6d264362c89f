#!/usr/bin/env python3
"""One-k (and short-line) hamilton calls on the sparse BASELINE model (cfg3: CSR, 256 orbitals, 512 lattice vectors), host buffers."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import tbmodels_amd
from tbmodels_amd import synthetic as syn, _lib
import bench
arrays = bench.build_model_arrays("cfg3")
lib = _lib.lib()
h = bench.stage(lib, 0, arrays)
k = syn.random_kpoints(64)
H = np.empty((8, 256, 256), dtype=np.complex128)
E = np.empty((8, 256))
for nk in (1, 3, 8):
    for name, call in (("hamilton", lambda q: lib.tbk_hamilton(h, _lib.ptr(np.ascontiguousarray(k[q:q+nk])), nk, 2, None, _lib.ptr(H))),):
        for q in range(5): _lib.check(call(q))
        t0 = time.perf_counter()
        for q in range(40): _lib.check(call(q % 50))
        print("cfg3 model nk=%d %s %.1f us" % (nk, name, (time.perf_counter() - t0) / 40 * 1e6), flush=True)
from oracle import tbk_oracle as oracle
hop = syn.csr_to_dense(256, arrays["r_ptr"], arrays["row"], arrays["col"], arrays["val"])
_lib.check(lib.tbk_hamilton(h, _lib.ptr(np.ascontiguousarray(k[:3])), 3, 2, None, _lib.ptr(H)))
print("max |dH| vs oracle", np.abs(H[:3] - oracle.hamilton(arrays["R"], hop, k[:3], 2)).max())
