"""
BASELINE.json's full sizes on the GPU, checked through size-independent properties (the oracle needs
~17 ms per k-point at the headline shape, so only a small subset is compared with it directly):

* sum_i E_i(k) = tr H(k) = sum_R 2 Re(exp(2 pi i k.R) tr hop[R])   (computed independently on the host)
* E(k + G) = E(k) for integer G                                      (periodicity of the Fourier sum)
* eigenvalues ascending per k; result independent of the order / chunking of the k list
* a random subset against the oracle at 1e-10
"""

import numpy as np
import pytest

import tbmodels_amd
from tbmodels_amd import _lib
from tbmodels_amd import synthetic as syn
from oracle import tbk_oracle as oracle

pytestmark = pytest.mark.gpu


def _trace_from_hoppings(r_vec, traces, k):
    """tr H(k) from the per-R traces: O(NK * N_R) host work, independent of the GPU path."""
    out = np.empty(len(k))
    r_f = r_vec.astype(float)
    for lo in range(0, len(k), 8192):
        phase = np.exp(2j * np.pi * (k[lo:lo + 8192] @ r_f.T))
        out[lo:lo + 8192] = 2.0 * (phase @ traces).real
    return out


def test_config2_headline_shape_100k():
    """Config 2: dense N_orb=64, N_R=4096, 100 000 random k-points."""
    r_vec, hop, pos = syn.dense_model_arrays(64, 4096, syn.MODEL_SEED + 2)
    k = syn.random_kpoints(100_000)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    model.pin_staging()
    eig = np.array(model.eigenval(k))
    assert eig.shape == (100_000, 64) and np.isfinite(eig).all()
    assert np.all(np.diff(eig, axis=1) >= 0)
    traces = np.einsum("rii->r", hop)
    assert np.abs(eig.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
    import bench

    assert bench.second_moment_error(_lib.lib(), model._staged(), 64, k, eig, 4096) < 1e-10  # sum E^2 = ||H(k)||_F^2 on 4096 rows
    # periodicity, and independence of position in the batch (different chunk / tile for every k)
    perm = np.random.default_rng(3).permutation(len(k))
    shifted = k[perm] + np.array([2.0, -1.0, 5.0])
    eig2 = np.array(model.eigenval(shifted))
    assert np.abs(eig2 - eig[perm]).max() < 1e-10
    # direct comparison on a subset
    idx = np.random.default_rng(4).choice(len(k), 192, replace=False)  # (~3 s of oracle)
    ref = np.array(oracle.eigenval(r_vec, hop, k[idx]))
    assert np.abs(eig[idx] - ref).max() < 1e-10
    # H(k) itself on a few points, both conventions
    ham = model.hamilton(k[idx[:8]])
    assert np.abs(ham - oracle.hamilton(r_vec, hop, k[idx[:8]])).max() < 1e-10
    ham1 = model.hamilton(k[idx[:8]], convention=1)
    assert np.abs(ham1 - oracle.hamilton(r_vec, hop, k[idx[:8]], 1, pos=pos)).max() < 1e-10


def test_headline_shape_intermediate_batches():
    """Batches between the matrix-vector path and the long launches at the headline shape, where the H(k) launch is
    K-split, tail-split, or both (tbk_hk_dense.hip: launch()): every k-point must come out as it does in a
    20-point call (matrix-vector path, checked against the oracle in the test above), and tr H(k) must match."""
    r_vec, hop, pos = syn.dense_model_arrays(64, 4096, syn.MODEL_SEED + 2)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    model.pin_staging()
    traces = np.einsum("rii->r", hop)
    rng = np.random.default_rng(11)
    for n_k in (40, 129, 600, 1000, 1500, 2100, 4200, 6000):
        k = syn.random_kpoints(n_k, seed=n_k)
        ham = model.hamilton(k, convention=1)
        eig = model.eigenval_array(k)
        assert np.abs(eig.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
        assert np.abs(np.einsum("kii->k", ham).real - eig.sum(axis=1)).max() < 1e-10
        for start in {0, n_k - 20, int(rng.integers(0, n_k - 20)), int(rng.integers(0, n_k - 20))}:
            window = slice(start, start + 20)
            assert np.abs(ham[window] - model.hamilton(k[window], convention=1)).max() < 1e-12
            assert np.abs(eig[window] - model.eigenval_array(k[window])).max() < 1e-12
        assert np.array_equal(ham, model.hamilton(k, convention=1))  # fixed summation order


def _csr_model(n_orb, n_r, seed):
    import scipy.sparse as sp

    r_vec, r_ptr, row, col, val, pos = syn.csr_model_arrays(n_orb, n_r, seed)
    hop_dict = {}
    for idx, r in enumerate(r_vec):
        sl = slice(r_ptr[idx], r_ptr[idx + 1])
        hop_dict[tuple(int(x) for x in r)] = sp.csr_matrix((val[sl], (row[sl], col[sl])), shape=(n_orb, n_orb))
    model = tbmodels_amd.Model(hop=hop_dict, pos=pos, size=n_orb, contains_cc=False, sparse=True)
    diag = row == col
    traces = np.zeros(len(r_vec), dtype=complex)
    np.add.at(traces, np.searchsorted(r_ptr, np.flatnonzero(diag), side="right") - 1, val[diag])
    return model, (r_vec, r_ptr, row, col, val, pos), traces


def test_config3_sparse_shape_full_50k():
    """Config 3 at full size: CSR N_orb=256, N_R=512, 2 % fill, all 50 000 random k-points (per-k independence:
    ``_tb_model.py:1111-1123``; eigenvalues: ``:1147-1150``).  The call runs through many k chunks of the
    reduction || bisection pipeline; every row is checked by the trace identity, 48 rows against the oracle, and the
    rows shared with a 4096-point call / with other chunkings must come out bit-identical."""
    model, (r_vec, r_ptr, row, col, val, pos), traces = _csr_model(256, 512, syn.MODEL_SEED + 3)
    k = syn.random_kpoints(50_000)
    eig = model.eigenval_array(k)
    assert eig.shape == (50_000, 256) and np.isfinite(eig).all()
    assert np.all(np.diff(eig, axis=1) >= 0)
    assert np.abs(eig.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
    # the second moment: sum_i E_i^2 = ||H(k)||_F^2 with H(k) from the sparse H(k) kernel (pinned to the oracle separately) on
    # 4096 rows from the whole list -- holds the eigensolver to H itself where the trace only sees the first moment
    import bench

    assert bench.second_moment_error(_lib.lib(), model._staged(), 256, k, eig, 4096) < 1e-10
    dense_hop = syn.csr_to_dense(256, r_ptr, row, col, val)
    idx = np.sort(np.random.default_rng(5).choice(len(k), 48, replace=False))
    ref = np.array(oracle.eigenval(r_vec, dense_hop, k[idx]))
    assert np.abs(eig[idx] - ref).max() < 1e-10
    # the 4096-point call of the shortened test: same rows, same bits (one chunk there, many here)
    head = model.eigenval_array(k[:4096])
    assert np.array_equal(head, eig[:4096])
    # the chunking is not allowed to change anything either
    for chunk in (1024, 3000):
        model.set_option(_lib.TBK_OPT_K_CHUNK, chunk)
        assert np.array_equal(model.eigenval_array(k[:9000]), eig[:9000])
    model.set_option(_lib.TBK_OPT_K_CHUNK, 0)
    # the sparse kernel and the dense MFMA kernel agree on H(k) (tests/test_sparse_dense.py of the reference)
    dense = tbmodels_amd.Model.from_packed(r_vec, dense_hop, pos=pos)
    assert np.abs(model.hamilton(k[:16]) - dense.hamilton(k[:16])).max() < 1e-12
    assert np.abs(dense.eigenval_array(k[:512]) - eig[:512]).max() < 1e-11


def test_config5_large_orbital_shape_reduced_R():
    """Config 5 orbital count (N_orb=512) at reduced N_R: the quick form of the full-size test below."""
    r_vec, hop, pos = syn.dense_model_arrays(512, 64, syn.MODEL_SEED + 5)
    k = syn.random_kpoints(64)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    eig = np.array(model.eigenval(k))
    traces = np.einsum("rii->r", hop)
    assert np.abs(eig.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
    ref = np.array(oracle.eigenval(r_vec, hop, k[:4]))
    assert np.abs(eig[:4] - ref).max() < 1e-10


def test_config5_large_orbital_shape_full():
    """Config 5 at full size: dense N_orb=512, N_R=2048 (8.6 GB of hoppings), 10 000 random k-points, staged straight
    through the C ABI (the host model class would copy the 8.6 GB twice).  Trace identity on every row, the second moment on 512 rows, 32 rows
    against the oracle, and a second run must give the same bits (race detector for the N=512 reduction pipeline)."""
    import ctypes

    n_orb, n_r, n_k = 512, 2048, 10_000
    r_vec, hop, _ = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 5)
    k = syn.random_kpoints(n_k)
    lib = _lib.lib()
    handle = ctypes.c_void_p()
    _lib.check(lib.tbk_model_create_dense(0, 3, n_orb, n_r, _lib.ptr(r_vec), _lib.ptr(hop), ctypes.byref(handle)))
    try:
        eig = np.empty((n_k, n_orb))
        _lib.check(lib.tbk_eigenval(handle, _lib.ptr(k), n_k, _lib.ptr(eig)))
        assert np.isfinite(eig).all() and np.all(np.diff(eig, axis=1) >= 0)
        traces = np.einsum("rii->r", hop)
        assert np.abs(eig.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
        import bench

        assert bench.second_moment_error(lib, handle, n_orb, k, eig, 512) < 1e-10  # sum E^2 = ||H(k)||_F^2 on 512 rows
        idx = np.unique(np.concatenate([[0, 1111, 2222, 3333, 5000, 6667, 8888, n_k - 1],
                                        np.random.default_rng(8).choice(n_k, 24, replace=False)]))  # 32 rows (~20 s of oracle)
        ref = np.array(oracle.eigenval(r_vec, hop, k[idx]))
        assert np.abs(eig[idx] - ref).max() < 1e-10
        again = np.empty_like(eig)
        _lib.check(lib.tbk_eigenval(handle, _lib.ptr(k), n_k, _lib.ptr(again)))
        assert np.array_equal(again, eig)
        # a short call (one chunk; its H(k) launch is tail-split along K, so sums are ordered differently)
        head = np.empty((700, n_orb))
        _lib.check(lib.tbk_eigenval(handle, _lib.ptr(k), 700, _lib.ptr(head)))
        assert np.abs(head - eig[:700]).max() < 1e-12
    finally:
        lib.tbk_model_destroy(handle)


def test_config4_headline_model_on_the_100_cubed_mesh():
    """Config 4 at full size on one GPU: the cfg2 model (N_orb=64, N_R=4096) on the 100 x 100 x 100 mesh, i.e. the
    two-level folded evaluation (csrc/tbk_fold.hip) over 10^6 k-points, and the 8 contiguous slabs the 8-GPU run
    hands to its ranks (``sharding.slab_bounds``), evaluated one after the other."""
    from tbmodels_amd.sharding import slab_bounds

    r_vec, hop, pos = syn.dense_model_arrays(64, 4096, syn.MODEL_SEED + 2)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    model.pin_staging()
    n = 100
    k = syn.uniform_grid(n)
    whole = model.eigenval_array(k)
    assert whole.shape == (n ** 3, 64) and np.isfinite(whole).all()
    assert np.all(np.diff(whole, axis=1) >= 0)
    # trace identity on 4096 rows drawn from the whole mesh (every chunk of the pipeline)
    rows = np.sort(np.random.default_rng(6).choice(n ** 3, 4096, replace=False))
    traces = np.einsum("rii->r", hop)
    assert np.abs(whole[rows].sum(axis=1) - _trace_from_hoppings(r_vec, traces, k[rows])).max() < 1e-10
    import bench

    # (the second moment with H(k) from the DIRECT evaluation -- random rows are no mesh -- against the folded eigenvalues)
    assert bench.second_moment_error(_lib.lib(), model._staged(), 64, k, whole, 4096) < 1e-10
    # 192 rows against the oracle
    idx = np.sort(np.random.default_rng(7).choice(n ** 3, 192, replace=False))
    ref = np.array(oracle.eigenval(r_vec, hop, k[idx]))
    assert np.abs(whole[idx] - ref).max() < 1e-10
    # folded against direct evaluation on three whole planes (first, an interior one, last)
    model.set_option(_lib.TBK_OPT_FOLD, 0)
    try:
        for plane in (0, 37, n - 1):
            sl = slice(plane * n * n, (plane + 1) * n * n)
            direct = model.eigenval_array(k[sl])
            assert np.abs(direct - whole[sl]).max() < 1e-11
    finally:
        model.set_option(_lib.TBK_OPT_FOLD, 1)
    # the 8 slabs of the sharded run, concatenated, equal the whole (slab edges cut planes in half: ragged runs)
    parts = []
    for rank in range(8):
        lo, hi = slab_bounds(n ** 3, 8, rank)
        parts.append(model.eigenval_array(syn.grid_slab(n, lo, hi)))
    assert np.abs(np.concatenate(parts) - whole).max() < 1e-11
    # and the whole call is reproducible bit for bit
    assert np.array_equal(model.eigenval_array(k), whole)


def test_config4_grid_slabs_match_whole():
    """Config 4 sharding logic on one GPU: contiguous slabs of the uniform grid, evaluated one by one, equal the whole."""
    from tbmodels_amd.sharding import slab_bounds

    r_vec, hop, pos = syn.dense_model_arrays(16, 64, syn.MODEL_SEED + 4)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    n = 24
    whole = np.array(model.eigenval(syn.uniform_grid(n)))
    world = 8
    parts = []
    for rank in range(world):
        lo, hi = slab_bounds(n ** 3, world, rank)
        parts.append(np.array(model.eigenval(syn.grid_slab(n, lo, hi))).reshape(hi - lo, 16))
    # slabs of <= 4096 k-points take the bisection kernel, the whole grid the QL kernel: same eigenvalues to
    # rounding (the reference's own batch-vs-single comparisons are `isclose`, tests/test_hamilton.py:21-32)
    assert np.abs(np.concatenate(parts) - whole).max() < 1e-13
    again = [np.array(model.eigenval(syn.grid_slab(n, *slab_bounds(n ** 3, world, rank)))) for rank in range(world)]
    assert np.abs(np.concatenate(again).reshape(-1, 16) - np.concatenate(parts)).max() == 0.0  # and deterministic


@pytest.mark.parametrize("n_orb,n_r,n_k,reps", [(64, 256, 100_000, 12), (40, 64, 70_000, 8), (96, 32, 24_576, 8), (128, 32, 12_288, 6), (256, 16, 4_096, 4)])
def test_repeated_runs_are_bit_identical(n_orb, n_r, n_k, reps):
    """Fixed summation orders everywhere (no floating-point atomics): the same call gives the same bits every time.
    A race between the waves of a reduction workgroup or between the streams of the chunk pipeline shows up here
    (the missing-waitcnt bug of the n <= 64 reduction changed ~50 rows per 100 000 only under LDS load)."""
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + n_orb)
    k = syn.random_kpoints(n_k, seed=n_orb)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    first = np.array(model.eigenval(k))
    traces = np.einsum("rii->r", hop)
    assert np.abs(first.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
    for _ in range(reps):
        again = np.array(model.eigenval(k))
        assert np.array_equal(again, first)


@pytest.mark.parametrize("n_orb,n_k,reps", [(256, 4096, 60), (150, 4096, 40), (130, 2048, 30), (400, 1024, 24), (512, 1024, 16),
                                            (640, 160, 8), (1024, 24, 5), (300, 64, 30), (512, 7, 20)])
def test_two_stage_reduction_is_race_free(n_orb, n_k, reps):
    """The two-stage reduction (csrc/tbk_eig_band.hip) has a dozen phases per panel that meet at workgroup barriers and
    share LDS; a missing meeting shows up as a WRONG matrix once in ~10^5 (round 2: the last QR step's partial sums
    were overwritten by a wave that had run ahead -- 2 rows in 500 000, off by 1e-2).  Many repetitions of one call:
    every row must come out bit-identical every time, and right.  (130 and 150 orbitals take the one-stage cascade since
    round 3 -- streaming kernel, eight-wave register kernel, 64-row and packed kernels, each handing its trailing block
    to the next through the head of the matrix' storage: the same kind of meeting points.  Round 4: 640 and 1024 orbitals take
    the eight-wave / two-rows-per-thread kernel and the second stage whose diagonals live in global memory -- its ticks meet
    through s_waitcnt vmcnt(0) + barrier instead of LDS ordering --, calls of <= 128 matrices the eight-wave / one-row
    kernels.)"""
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, 16 if n_orb <= 512 else 4, syn.MODEL_SEED + n_orb)
    k = syn.random_kpoints(n_k, seed=n_orb)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    first = model.eigenval_array(k).copy()
    traces = np.einsum("rii->r", hop)
    assert np.abs(first.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
    ref = np.array(oracle.eigenval(r_vec, hop, k[:3]))
    assert np.abs(first[:3] - ref).max() < 1e-10
    for rep in range(reps):
        again = model.eigenval_array(k)
        bad = np.flatnonzero(np.any(again != first, axis=1))
        assert len(bad) == 0, "repetition %d: rows %s differ by up to %.2e" % (
            rep, bad[:5], np.abs(again[bad] - first[bad]).max())
