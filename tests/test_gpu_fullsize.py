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
    # periodicity, and independence of position in the batch (different chunk / tile for every k)
    perm = np.random.default_rng(3).permutation(len(k))
    shifted = k[perm] + np.array([2.0, -1.0, 5.0])
    eig2 = np.array(model.eigenval(shifted))
    assert np.abs(eig2 - eig[perm]).max() < 1e-10
    # direct comparison on a subset
    idx = np.random.default_rng(4).choice(len(k), 48, replace=False)
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


def test_config3_sparse_shape():
    """Config 3: CSR N_orb=256, N_R=512, 2 % fill (k list shortened: the N=256 eigensolve is ~0.1 ms per k)."""
    r_vec, r_ptr, row, col, val, pos = syn.csr_model_arrays(256, 512, syn.MODEL_SEED + 3)
    hop_dict = {}
    import scipy.sparse as sp

    for idx, r in enumerate(r_vec):
        sl = slice(r_ptr[idx], r_ptr[idx + 1])
        hop_dict[tuple(int(x) for x in r)] = sp.csr_matrix((val[sl], (row[sl], col[sl])), shape=(256, 256))
    model = tbmodels_amd.Model(hop=hop_dict, pos=pos, size=256, contains_cc=False, sparse=True)
    k = syn.random_kpoints(4096)
    eig = np.array(model.eigenval(k))
    assert np.all(np.diff(eig, axis=1) >= 0)
    diag = row == col
    traces = np.zeros(len(r_vec), dtype=complex)
    np.add.at(traces, np.searchsorted(r_ptr, np.flatnonzero(diag), side="right") - 1, val[diag])
    assert np.abs(eig.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
    dense_hop = syn.csr_to_dense(256, r_ptr, row, col, val)
    ref = np.array(oracle.eigenval(r_vec, dense_hop, k[:6]))
    assert np.abs(eig[:6] - ref).max() < 1e-10
    # the sparse kernel and the dense MFMA kernel agree on H(k) (tests/test_sparse_dense.py of the reference)
    dense = tbmodels_amd.Model.from_packed(r_vec, dense_hop, pos=pos)
    assert np.abs(model.hamilton(k[:16]) - dense.hamilton(k[:16])).max() < 1e-12


def test_config5_large_orbital_shape_reduced_R():
    """Config 5 orbital count (N_orb=512, rocSOLVER path) at reduced N_R so the host model stays small."""
    r_vec, hop, pos = syn.dense_model_arrays(512, 64, syn.MODEL_SEED + 5)
    k = syn.random_kpoints(64)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    eig = np.array(model.eigenval(k))
    traces = np.einsum("rii->r", hop)
    assert np.abs(eig.sum(axis=1) - _trace_from_hoppings(r_vec, traces, k)).max() < 1e-10
    ref = np.array(oracle.eigenval(r_vec, hop, k[:4]))
    assert np.abs(eig[:4] - ref).max() < 1e-10


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


@pytest.mark.parametrize("n_orb,n_r,n_k,reps", [(64, 256, 100_000, 12), (40, 64, 70_000, 8), (128, 32, 12_288, 6), (256, 16, 4_096, 4)])
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
