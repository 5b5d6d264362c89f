"""
GPU parity tests (``-m gpu``): the HIP path, called through tbmodels_amd.Model -> ctypes ->
libtbk.so, against (a) the golden fixtures generated from the imported reference (which embed the
reference's own stored goldens) and (b) the oracle on seeded inputs.  Tolerance: 1e-10 absolute on
eigenvalues and on H(k) (BASELINE.json north_star; the reference's own known-answer test uses
atol 1e-10, tests/test_cli_eigenvals.py:48-50).
"""

import pickle

import numpy as np
import pytest

import tbmodels_amd
from tbmodels_amd import synthetic as syn
from oracle import tbk_oracle as oracle

from conftest import KPT, T_VALUES
from test_host_model import toy_model
from test_oracle_golden import SYN_TAGS, synthetic_case

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _close(a, b, tol=TOL):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol, err


def test_silicon_known_answer(silicon):
    """The reference's known-answer test (tests/test_cli_eigenvals.py): 11 k-points, atol 1e-10."""
    model = tbmodels_amd.Model.from_packed(silicon["R"], silicon["hop"], pos=silicon["pos"], uc=silicon["uc"])
    eig = model.eigenval(silicon["known_kpoints"])
    assert isinstance(eig, list) and len(eig) == 11 and eig[0].shape == (8,)
    _close(np.array(eig), silicon["known_eigenvals"])


def test_silicon_config1_grid(silicon):
    """BASELINE config 1: silicon, 10x10x10 grid, against the reference's output."""
    model = tbmodels_amd.Model.from_packed(silicon["R"], silicon["hop"], pos=silicon["pos"])
    _close(np.array(model.eigenval(silicon["grid"])), silicon["grid_eig"])
    _close(model.hamilton(silicon["grid"][:16]), silicon["grid_h2_first16"])
    _close(model.hamilton(silicon["grid"][:16], convention=1), silicon["grid_h1_first16"])
    _close(model.hamilton(silicon["kpt"]), silicon["kpt_h2"])
    _close(model.hamilton(silicon["kpt"], convention=1), silicon["kpt_h1"])
    _close(np.array(model.eigenval(silicon["kpt"])), silicon["kpt_eig"])
    ham = model.hamilton(silicon["grid"])
    _close(ham, ham.conj().transpose(0, 2, 1), 0.0)  # exactly Hermitian, like H += H^H
    assert np.all(ham.imag[:, np.arange(8), np.arange(8)] == 0.0)


def test_silicon_wannier_stored_golden(silicon):
    model = tbmodels_amd.Model.from_packed(silicon["wannier_R"], silicon["wannier_hop"], pos=silicon["wannier_pos"])
    _close(np.array([model.hamilton(k) for k in KPT]), silicon["wannier_kpt_h2_stored"])


@pytest.mark.parametrize("t_idx", range(6))
@pytest.mark.parametrize("sparse", [False, True])
def test_toy_stored_goldens(toy, t_idx, sparse):
    """tests/test_hamilton.py / tests/test_eigenval.py of the reference, through the mirrored API."""
    model = toy_model(*T_VALUES[t_idx], sparse=sparse)
    tag = "t%d_%s" % (t_idx, "sparse" if sparse else "dense")
    for conv in (1, 2):
        per_k = np.array([model.hamilton(k, convention=conv) for k in KPT])
        _close(per_k, toy[tag + "_h%d_stored" % conv])
        # batch == per-k (tests/test_hamilton.py:21-32)
        _close(model.hamilton(KPT, convention=conv), per_k, 1e-14)
    eig = np.array([model.eigenval(k) for k in KPT])
    _close(eig, toy[tag + "_eig_stored"])
    _close(np.array(model.eigenval(KPT)), eig, 1e-14)


@pytest.mark.parametrize("dim", [2, 4])
def test_toy_other_dims(toy, dim):
    tag = "dim%d" % dim
    model = tbmodels_amd.Model.from_packed(toy[tag + "_R"], toy[tag + "_hop"], pos=toy[tag + "_pos"])
    k = toy[tag + "_k"]
    _close(model.hamilton(k), toy[tag + "_h2"])
    _close(model.hamilton(k, convention=1), toy[tag + "_h1"])
    _close(np.array(model.eigenval(k)), toy[tag + "_eig"])


@pytest.mark.parametrize("tag", SYN_TAGS)
@pytest.mark.parametrize("sparse", [False, True])
def test_synthetic_cases(synthetic, tag, sparse):
    """Dense kernel and CSR kernel on the same models: both must match the reference (sparse == dense)."""
    r_vec, hop, pos, k = synthetic_case(synthetic, tag)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos, sparse=sparse)
    n_h = len(synthetic[tag + "_h2"])
    _close(model.hamilton(k[:n_h]), synthetic[tag + "_h2"])
    _close(model.hamilton(k[:n_h], convention=1), synthetic[tag + "_h1"])
    _close(np.array(model.eigenval(k)), synthetic[tag + "_eig"])


@pytest.mark.parametrize("n_orb,n_r", [(8, 30), (64, 200), (100, 12)])
def test_one_k_calls_take_k_in_the_kernel_arguments_and_keep_the_positions_on_the_device(n_orb, n_r):
    """ONE k-point per call is the Z2Pack call shape (_tb_model.py:1103-1108): the k-point goes into the kernel arguments, the
    convention-1 phases of that k-point are formed inside the H(k) kernel from the raw positions, and those stay on the device
    between calls -- until their bytes change.  Every call must equal the oracle and the batched call on the same k-points."""
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 500 + n_orb)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    k = syn.random_kpoints(6, seed=n_orb) * 3.0 - 1.0  # (outside [0, 1) too)
    batch1, batch2 = model.hamilton(k, convention=1), model.hamilton(k, convention=2)
    for q in range(len(k)):
        _close(model.hamilton(k[q], convention=1), oracle.hamilton(r_vec, hop, k[q], 1, pos=pos))
        _close(model.hamilton(k[q], convention=1), batch1[q], 1e-13)
        _close(model.hamilton(k[q]), batch2[q], 1e-13)
        _close(model.eigenval(k[q]), oracle.eigenval(r_vec, hop, k[q]))
    # new positions: the cached copy on the device must go
    moved = np.ascontiguousarray(pos[::-1] * 0.5 + 0.1)
    model.pos = moved
    _close(model.hamilton(k[2], convention=1), oracle.hamilton(r_vec, hop, k[2], 1, pos=moved))
    model.pos = pos
    _close(model.hamilton(k[2], convention=1), batch1[2], 1e-13)
    h = model.hamilton(k[3], convention=1)
    assert np.array_equal(h, h.conj().T) and not np.diagonal(h).imag.any()  # exactly Hermitian, real diagonal


def test_scalar_k_single_point_and_empty(synthetic):
    model = tbmodels_amd.Model.from_packed(synthetic["dim1_R"], synthetic["dim1_hop"], pos=synthetic["dim1_pos"])
    k = float(synthetic["dim1_scalar_k"])
    ham = model.hamilton(k)
    assert ham.shape == (6, 6)
    _close(ham, synthetic["dim1_scalar_h2"])
    _close(model.hamilton(k, convention=1), synthetic["dim1_scalar_h1"])
    eig = model.eigenval(k)
    assert isinstance(eig, np.ndarray) and eig.shape == (6,)
    _close(eig, synthetic["dim1_scalar_eig"])
    model = tbmodels_amd.Model.from_packed(synthetic["single_R"], synthetic["single_hop"])
    _close(model.hamilton(synthetic["single_k"][0]), synthetic["single_k0_h2"])
    _close(model.eigenval(synthetic["single_k"][0]), synthetic["single_k0_eig"])
    empty = tbmodels_amd.Model(size=3, dim=3)
    k = [[0.1, 0.2, 0.3], [0.5, 0.5, 0.5]]
    _close(empty.hamilton(k), synthetic["empty_h2"])
    _close(np.array(empty.eigenval(k)), synthetic["empty_eig"])
    assert len(empty.eigenval(np.zeros((0, 3)))) == 0  # an empty batch is an empty list
    assert empty.hamilton(np.zeros((0, 3))).shape == (0, 3, 3)


def test_restaging_after_mutation(toy):
    """add_hop / in-place edits / set_sparse between calls must be seen by the next call."""
    model = toy_model(0.2, -0.2)
    first = model.hamilton(KPT)
    model.add_hop(0.3 + 0.1j, 0, 1, (2, -1, 0))
    r_vec, hop = model.packed_hop()
    _close(model.hamilton(KPT), oracle.hamilton(r_vec, hop, KPT))
    assert np.abs(model.hamilton(KPT) - first).max() > 1e-3
    model.hop[(0, 0, 0)] += 0.05 * np.eye(2)
    r_vec, hop = model.packed_hop()
    _close(np.array(model.eigenval(KPT)), np.array(oracle.eigenval(r_vec, hop, KPT)))
    model.set_sparse(True)
    _close(np.array(model.eigenval(KPT)), np.array(oracle.eigenval(r_vec, hop, KPT)))
    clone = pickle.loads(pickle.dumps(model))
    _close(clone.hamilton(KPT, convention=1), model.hamilton(KPT, convention=1), 0.0)


def test_restaging_after_edit_through_a_handed_out_reference():
    """No content hash while the model is untouched; an edit through a reference taken earlier is still seen."""
    model = toy_model(0.2, -0.2)
    assert model._staging_key()[0] == "version"
    first = np.array(model.eigenval(KPT))
    assert model._staging_key()[0] == "version"  # evaluating does not expose anything
    on_site = model.hop[(0, 0, 0)]
    model.eigenval(KPT)  # staged again under the content fingerprint
    on_site += 0.25 * np.eye(2)  # no dict access: only the bytes changed
    r_vec, hop = model.packed_hop()
    got = np.array(model.eigenval(KPT))
    _close(got, np.array(oracle.eigenval(r_vec, hop, KPT)))
    _close(got, first + 0.5, 1e-12)  # hop[0] holds half of the on-site block


@pytest.mark.parametrize("n_orb,n_r,n_k", [(64, 96, 300), (24, 17, 1000), (96, 40, 130), (3, 5, 4097)])
def test_seeded_dense_vs_oracle(n_orb, n_r, n_k):
    """Sizes that cross tile boundaries (k tiles of 128, element tiles of 64, K stages of 8 R) and the XCD walk."""
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + n_orb)
    k = syn.random_kpoints(n_k, seed=n_k) * 2 - 1
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    _close(np.array(model.eigenval(k)), np.array(oracle.eigenval(r_vec, hop, k)))
    sub = k[: min(n_k, 64)]
    _close(model.hamilton(sub), oracle.hamilton(r_vec, hop, sub))
    _close(model.hamilton(sub, convention=1), oracle.hamilton(r_vec, hop, sub, 1, pos=pos))


@pytest.mark.parametrize("n_k", [1, 2, 3, 5, 8, 9, 16, 17, 32, 33, 130, 700, 3000])
def test_small_batches_vs_oracle(n_k):
    """Up to 32 k-points take the matrix-vector kernel (one instantiation per 1 / 2 / 4 / 8 / 16 / 32 accumulator
    pairs), larger small batches the split-K launch of the MFMA kernel; both park partial sums that the finish
    kernel adds in fixed order.  3000 k-points are past both.  All modes of the epilogue."""
    r_vec, hop, pos = syn.dense_model_arrays(16, 300, syn.MODEL_SEED + 31)
    k = syn.random_kpoints(n_k, seed=n_k) * 3 - 1.5
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    sub = slice(0, min(n_k, 40))
    _close(model.hamilton(k)[sub], oracle.hamilton(r_vec, hop, k[sub]))
    _close(model.hamilton(k, convention=1)[sub], oracle.hamilton(r_vec, hop, k[sub], 1, pos=pos))
    _close(np.array(model.eigenval(k))[sub], np.array(oracle.eigenval(r_vec, hop, k[sub])))
    again = model.hamilton(k)
    assert np.array_equal(again, model.hamilton(k))  # fixed summation order: bit-identical run to run
    assert np.array_equal(again, np.conj(np.swapaxes(again, 1, 2)))  # exactly Hermitian


@pytest.mark.parametrize("n_orb", [1, 2, 3, 7, 11, 12, 23, 33, 64, 65, 100, 181, 300])
def test_matrix_vector_path_over_shapes(n_orb):
    """The matrix-vector kernel of round 6 cuts its work into (blocks of 64 packed elements) x (slices of whole lattice vectors)
    per WAVE, the slices ragged (K rows / slices is no whole number), one round of two workgroups per CU or many rounds, the
    phase rows in a wave-private LDS strip or -- strip too short -- handed in (`csrc/tbk_hk_dense.hip`: gemv_plan,
    tbk_hk_inline_phases).  Every combination of 13 orbital counts (1 .. 300: one block of 64 elements .. 706 blocks, with and
    without padding), 8 lattice-vector counts (1 .. 1500: one slice of 16 rows .. hundreds) and 7 batch sizes (every
    instantiation, full and ragged) against the oracle: H(k) in both conventions, and the eigenvalues
    (`_tb_model.py:1109-1128`, `:1147-1150`)."""
    rng = np.random.default_rng(600 + n_orb)
    for n_r in (1, 2, 7, 8, 9, 100, 257, 1500):
        if n_orb * n_orb * n_r > 3e7:  # (the oracle's time, not the kernel's limit)
            continue
        r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 7 * n_orb + n_r)
        model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
        for n_k in (1, 2, 3, 5, 9, 17, 32):
            k = rng.random((n_k, 3)) * 4.0 - 2.0
            arg = k[0] if n_k == 1 else k
            want = oracle.hamilton(r_vec, hop, k)
            _close(np.asarray(model.hamilton(arg)).reshape(want.shape), want)
            want1 = oracle.hamilton(r_vec, hop, k, 1, pos=pos)
            _close(np.asarray(model.hamilton(arg, convention=1)).reshape(want1.shape), want1)
            if n_k in (1, 5, 32):
                _close(np.asarray(model.eigenval(arg)).reshape(n_k, n_orb), np.array(oracle.eigenval(r_vec, hop, k)))


def _grid(shape, offset=(0.0, 0.0, 0.0), order="ij"):
    axes = [np.linspace(0, 1, n, endpoint=False) + o for n, o in zip(shape, offset)]
    mesh = np.meshgrid(*axes, indexing=order)
    return np.stack([m.reshape(-1) for m in mesh], axis=1)


@pytest.mark.parametrize(
    "dim,shape,offset,order",
    [
        (3, (3, 40, 40), (0.0, 0.0, 0.0), "ij"),       # planes of constant k_1
        (3, (40, 3, 40), (0.13, -0.4, 2.5), "xy"),     # shifted mesh; component 1 is the slow one
        (2, (5, 1100), (0.0, 0.25), "ij"),             # 2-D model: lines of constant k_1 fold to a 1-D model
        (3, (2, 30, 50), (0.0, 0.0, 0.0), "ij"),
        (3, (2, 12, 200), (0.0, 0.1, 0.0), "ij"),      # lines longer than a k tile: planes only
        (4, (2, 40, 6, 6), (0.0, 0.0, 0.5, 0.0), "ij"),  # 4-D: lines are 2-D models
    ],
)
def test_folded_grid_matches_direct_evaluation(dim, shape, offset, order):
    """k lists with long runs of one shared component (uniform meshes) are evaluated on the model folded along that
    component (tbk_fold.hip): same eigenvalues as the direct path (TBK_OPT_FOLD = 0) and as the oracle."""
    from tbmodels_amd import _lib

    n_orb, n_r = 12, {2: 160, 3: 300, 4: 600}[dim]
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 50 + dim, dim=dim)
    k = _grid(shape[:dim], offset[:dim], order)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    folded = np.array(model.eigenval(k))
    model.set_option(_lib.TBK_OPT_FOLD, 0)
    direct = np.array(model.eigenval(k))
    model.set_option(_lib.TBK_OPT_FOLD, 1)
    assert np.abs(folded - direct).max() < 1e-12
    assert np.abs(folded - direct).max() > 0.0  # the folded path really ran (different summation order)
    idx = np.random.default_rng(1).choice(len(k), 40, replace=False)
    _close(folded[idx], np.array(oracle.eigenval(r_vec, hop, k[idx])))
    # a contiguous slab of the mesh (what one rank of a sharded run gets): ragged first and last runs
    lo, hi = len(k) // 7, len(k) - len(k) // 5
    model.set_option(_lib.TBK_OPT_FOLD, 1)
    assert np.abs(np.array(model.eigenval(k[lo:hi])) - direct[lo:hi]).max() < 1e-12
    # a list without long runs takes the direct path: bit-identical with folding switched off
    shuffled = k[np.random.default_rng(2).permutation(len(k))]
    a = np.array(model.eigenval(shuffled))
    model.set_option(_lib.TBK_OPT_FOLD, 0)
    assert np.array_equal(a, np.array(model.eigenval(shuffled)))


def test_folded_mesh_over_many_chunks_is_reproducible():
    """The folded H(k) of chunk c + 1 is built beside the reduction of chunk c (second H buffer, `h_overlap` in
    csrc/tbk_api.hip): many small chunks of a mesh must give the one-chunk result, the same bits on every repeat (a
    missing wait between the streams shows up as a few wrong rows now and then), and the oracle's eigenvalues."""
    from tbmodels_amd import _lib

    n_orb, n_r = 12, 300
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 53)
    k = _grid((4, 40, 40))
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    whole = model.eigenval_array(k)
    for chunk in (512, 1024):
        model.set_option(_lib.TBK_OPT_K_CHUNK, chunk)
        first = model.eigenval_array(k)
        assert np.abs(first - whole).max() < 1e-12
        for _ in range(6):
            assert np.array_equal(model.eigenval_array(k), first)
    model.set_option(_lib.TBK_OPT_FOLD, 0)
    assert 0.0 < np.abs(model.eigenval_array(k) - first).max() < 1e-12  # the folded path really ran
    model.set_option(_lib.TBK_OPT_FOLD, 1)
    model.set_option(_lib.TBK_OPT_K_CHUNK, 0)
    idx = np.random.default_rng(3).choice(len(k), 32, replace=False)
    _close(whole[idx], np.array(oracle.eigenval(r_vec, hop, k[idx])))


def test_chunks_of_whole_mesh_planes_take_one_contraction():
    """Chunks that consist of whole mesh planes (here two planes of 40 lines per chunk: TBK_OPT_K_CHUNK = 3200) fold the
    lines of all their planes into consecutive operand slots and contract them in ONE launch (`batched` in
    csrc/tbk_api.hip): same eigenvalues as the direct path, as plane-by-plane chunks, and the same bits on every repeat."""
    from tbmodels_amd import _lib

    n_orb, n_r = 12, 300
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 54)
    k = _grid((6, 40, 40), (0.1, 0.0, 0.3))
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    model.set_option(_lib.TBK_OPT_K_CHUNK, 1600)  # one plane per chunk: the per-plane path
    per_plane = model.eigenval_array(k)
    model.set_option(_lib.TBK_OPT_K_CHUNK, 3200)  # two whole planes per chunk: batched
    first = model.eigenval_array(k)
    assert np.abs(first - per_plane).max() < 1e-12
    for _ in range(5):
        assert np.array_equal(model.eigenval_array(k), first)
    # a slab that starts and ends inside planes: ragged chunks fall back to the per-piece path
    lo, hi = 700, len(k) - 900
    assert np.abs(model.eigenval_array(k[lo:hi]) - first[lo:hi]).max() < 1e-12
    model.set_option(_lib.TBK_OPT_FOLD, 0)
    assert 0.0 < np.abs(model.eigenval_array(k) - first).max() < 1e-12
    model.set_option(_lib.TBK_OPT_FOLD, 1)
    model.set_option(_lib.TBK_OPT_K_CHUNK, 0)
    idx = np.random.default_rng(4).choice(len(k), 32, replace=False)
    _close(first[idx], np.array(oracle.eigenval(r_vec, hop, k[idx])))


def test_device_entry_point_never_reads_k_back_and_folds_through_the_hint():
    """include/tbk.h: the device entry points enqueue and return.  ``tbk_eigenval_device`` used to copy a
    device-resident k list back (plus a stream synchronisation) to look for mesh structure, and remembered a miss by
    (pointer, length) -- a buffer refilled with a mesh was never folded.  Now the structure comes from the caller's
    host copy (``tbk_eigenval_device_hint``) or not at all: ONE device buffer holds a random list, then a mesh."""
    import ctypes

    from tbmodels_amd import _lib

    lib = _lib.lib()
    n_orb, n_r = 12, 300
    r_vec, hop, _ = syn.dense_model_arrays(n_orb, n_r, syn.MODEL_SEED + 53)
    handle = ctypes.c_void_p()
    _lib.check(lib.tbk_model_create_dense(0, 3, n_orb, n_r, _lib.ptr(r_vec), _lib.ptr(hop), ctypes.byref(handle)))
    mesh = _grid((3, 40, 40), (0.0, 0.0, 0.0), "ij")
    rand = syn.random_kpoints(len(mesh), seed=9)
    d_k, d_e = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.tbk_device_malloc(0, mesh.nbytes, ctypes.byref(d_k)))
    _lib.check(lib.tbk_device_malloc(0, len(mesh) * n_orb * 8, ctypes.byref(d_e)))

    def counter(which):
        value = ctypes.c_int64(-1)
        _lib.check(lib.tbk_model_counter(handle, which, ctypes.byref(value)))
        return value.value

    def run(k_host, hint):
        out = np.empty((len(k_host), n_orb))
        _lib.check(lib.tbk_memcpy_h2d(0, d_k, _lib.ptr(k_host), k_host.nbytes))
        if hint:
            _lib.check(lib.tbk_eigenval_device_hint(handle, d_k, _lib.ptr(k_host), len(k_host), d_e))
        else:
            _lib.check(lib.tbk_eigenval_device(handle, d_k, len(k_host), d_e))
        _lib.check(lib.tbk_eigenval_check(handle))
        _lib.check(lib.tbk_memcpy_d2h(0, _lib.ptr(out), d_e, out.nbytes))
        return out

    try:
        e_rand = run(rand, hint=True)  # random list: nothing to fold
        assert counter(_lib.TBK_CNT_FOLDED_CALLS) == 0
        e_mesh = run(mesh, hint=True)  # the SAME device buffer, refilled with a mesh: folded
        assert counter(_lib.TBK_CNT_FOLDED_CALLS) == 1 and counter(_lib.TBK_CNT_FOLDED_KPOINTS) == len(mesh)
        e_plain = run(mesh, hint=False)  # no host copy, no read-back: direct evaluation
        assert counter(_lib.TBK_CNT_FOLDED_CALLS) == 1 and counter(_lib.TBK_CNT_EIGENVAL_CALLS) == 3
        assert 0.0 < np.abs(e_mesh - e_plain).max() < 1e-12
        idx = np.random.default_rng(3).choice(len(mesh), 24, replace=False)
        _close(e_mesh[idx], np.array(oracle.eigenval(r_vec, hop, mesh[idx])))
        _close(e_rand[idx], np.array(oracle.eigenval(r_vec, hop, rand[idx])))
        host = np.empty_like(e_mesh)  # the host entry point always has the list
        _lib.check(lib.tbk_eigenval(handle, _lib.ptr(mesh), len(mesh), _lib.ptr(host)))
        assert counter(_lib.TBK_CNT_FOLDED_CALLS) == 2 and np.array_equal(host, e_mesh)
    finally:
        lib.tbk_device_free(0, d_k)
        lib.tbk_device_free(0, d_e)
        lib.tbk_model_destroy(handle)


def test_seeded_csr_vs_oracle():
    """BASELINE config 3 shape at reduced size: N=128, N_R=64, 2 % fill."""
    r_vec, r_ptr, row, col, val, pos = syn.csr_model_arrays(128, 64, syn.MODEL_SEED + 3)
    hop = syn.csr_to_dense(128, r_ptr, row, col, val)
    k = syn.random_kpoints(200)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos, sparse=True)
    _close(np.array(model.eigenval(k)), np.array(oracle.eigenval(r_vec, hop, k)))
    _close(model.hamilton(k[:32], convention=1), oracle.hamilton(r_vec, hop, k[:32], 1, pos=pos))
    dense = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos, sparse=False)
    _close(dense.hamilton(k[:32]), model.hamilton(k[:32]), 1e-13)  # tests/test_sparse_dense.py


@pytest.mark.parametrize(
    "n_orb,n_r,fill,nk",
    [
        (24, 40, 0.10, 37),      # phase tile of 8 k-points; 300 packed elements: five wave rounds, the last one ragged
        (23, 600, 0.05, 21),     # tile of 4 k-points (rows of 5 slots)
        (16, 1500, 0.05, 9),     # tile of 2 k-points (rows of 3 slots)
        (12, 2400, 0.08, 5),     # tile of 1 k-point (rows of 2 slots: only 8 distinct classes)
        (70, 64, 0.003, 33),     # most packed elements without a single record
        (9, 30, 1.0, 16),        # fully dense blocks stored as CSR: every lane's list has every lattice vector
    ],
)
def test_sparse_walk_order_for_every_tile_shape(n_orb, n_r, fill, nk):
    """The LDS kernel of the sparse path walks its records in the conflict-free order built at staging
    (csrc/tbk_hk_csr.hip: tbk_csr_schedule, an edge colouring per 16-lane LDS group) -- for every phase-tile shape, ragged
    last wave rounds, empty record lists and full ones: equal to the dense storage of the same model (the reference's
    tests/test_sparse_dense.py) and to the oracle, both conventions."""
    r_vec, r_ptr, row, col, val, pos = syn.csr_model_arrays(n_orb, n_r, syn.MODEL_SEED + 500 + n_r, fill=fill)
    hop = syn.csr_to_dense(n_orb, r_ptr, row, col, val)
    k = syn.random_kpoints(nk, seed=n_r)
    sparse = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos, sparse=True)
    dense = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos, sparse=False)
    for convention in (1, 2):
        h_sparse = sparse.hamilton(k, convention=convention)
        _close(h_sparse, dense.hamilton(k, convention=convention), 1e-13)
        _close(h_sparse[:4], oracle.hamilton(r_vec, hop, k[:4], convention, pos=pos))
        _close(h_sparse, h_sparse.conj().transpose(0, 2, 1), 0.0)  # exactly Hermitian, like H += H^H
    _close(np.array(sparse.eigenval(k)), np.array(dense.eigenval(k)), 1e-12)
    _close(np.array(sparse.eigenval(k[:4])), np.array(oracle.eigenval(r_vec, hop, k[:4])))


def test_chunked_pipeline_matches_single_chunk():
    """k chunks (TBK_OPT_K_CHUNK) must not change results; order of k is preserved."""
    from tbmodels_amd import _lib

    r_vec, hop, pos = syn.dense_model_arrays(32, 24, syn.MODEL_SEED + 77)
    k = syn.random_kpoints(1000)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    whole = np.array(model.eigenval(k))
    model.set_option(_lib.TBK_OPT_K_CHUNK, 128)
    _close(np.array(model.eigenval(k)), whole, 0.0)
    _close(np.array(model.eigenval(k[::-1]))[::-1], whole, 1e-13)


def test_kdotp_models(kdotp_golden):
    g = kdotp_golden
    for order in (0, 1, 2, 3):
        powers, coeffs = g["order%d_powers" % order], g["order%d_coeffs" % order]
        kp = tbmodels_amd.KdotpModel({tuple(p): c for p, c in zip(powers.tolist(), coeffs)})
        dk = g["order%d_dk" % order]
        _close(kp.hamilton(dk), g["order%d_h" % order])
        _close(np.array(kp.eigenval(dk)), g["order%d_eig" % order])
        _close(kp.hamilton(dk[0]), g["order%d_h_single" % order])
        _close(kp.eigenval(dk[0]), g["order%d_eig_single" % order])


def test_kdotp_model_follows_edits_of_its_coefficients(kdotp_golden):
    """a ``KdotpModel`` staged once kept answering for its old coefficients after
    ``taylor_coefficients`` (a public dict the reference reads on every call, kdotp.py:51-82) was edited."""
    g = kdotp_golden
    powers, coeffs, dk = g["order2_powers"], g["order2_coeffs"], g["order2_dk"]
    kp = tbmodels_amd.KdotpModel({tuple(p): c for p, c in zip(powers.tolist(), coeffs)})
    _close(kp.hamilton(dk), g["order2_h"])
    first = tuple(powers[0].tolist())
    kp.taylor_coefficients[first] = kp.taylor_coefficients[first] + np.eye(coeffs.shape[1])  # replaced
    expect = oracle.kdotp_hamilton(powers, np.array([kp.taylor_coefficients[tuple(p)] for p in powers.tolist()]), dk)
    _close(kp.hamilton(dk), expect)
    kp.taylor_coefficients[first][0, 0] -= 0.25  # edited in place
    expect = oracle.kdotp_hamilton(powers, np.array([kp.taylor_coefficients[tuple(p)] for p in powers.tolist()]), dk)
    _close(kp.hamilton(dk), expect)
    _close(np.array(kp.eigenval(dk)), np.linalg.eigvalsh(expect))
    del kp.taylor_coefficients[tuple(powers[-1].tolist())]  # key set changed
    kept = powers[:-1]
    expect = oracle.kdotp_hamilton(kept, np.array([kp.taylor_coefficients[tuple(p)] for p in kept.tolist()]), dk)
    _close(kp.hamilton(dk), expect)


def test_eigenval_array_is_the_list_as_one_array(silicon, kdotp_golden):
    model = tbmodels_amd.Model.from_packed(silicon["R"], silicon["hop"])
    k = silicon["known_kpoints"]
    as_list, as_array = model.eigenval(k), model.eigenval_array(k)
    assert isinstance(as_list, list) and isinstance(as_array, np.ndarray) and as_array.shape == (len(k), 8)
    assert np.array_equal(np.array(as_list), as_array)
    assert np.array_equal(model.eigenval(k[3]), model.eigenval_array(k[3])) and model.eigenval_array(k[3]).shape == (8,)
    g = kdotp_golden
    kp = tbmodels_amd.KdotpModel({tuple(p): c for p, c in zip(g["order2_powers"].tolist(), g["order2_coeffs"])})
    assert np.array_equal(np.array(kp.eigenval(g["order2_dk"])), kp.eigenval_array(g["order2_dk"]))


def test_non_finite_k_is_value_error(silicon, kdotp_golden):
    """One NaN / Inf k component anywhere in a batch: the device-side check of the eigenvalues raises scipy's
    ValueError for the whole call (every solver path), the next call on the same handle is clean again, and
    hamilton() returns the NaNs like the reference."""
    from tbmodels_amd import _lib
    from tbmodels_amd.kdotp import KdotpModel

    model = tbmodels_amd.Model.from_packed(silicon["R"], silicon["hop"])
    with pytest.raises(ValueError, match="infs or NaNs"):
        model.eigenval([[0.1, np.nan, 0.2]])
    assert np.isnan(model.hamilton([0.1, np.nan, 0.2])).any()
    for n_orb, n_k, solver in [(8, 20000, "auto"), (40, 700, "auto"), (40, 9000, "rocsolver"), (64, 45000, "auto"), (100, 300, "auto")]:
        r_vec, hop, pos = syn.dense_model_arrays(n_orb, 5, syn.MODEL_SEED + n_orb)
        model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
        model.set_option(_lib.TBK_OPT_EIGENSOLVER, {"auto": _lib.TBK_EIG_AUTO, "rocsolver": _lib.TBK_EIG_ROCSOLVER}[solver])
        k = syn.random_kpoints(n_k, seed=n_orb)
        clean = model.eigenval_array(k)
        for where, bad in [(0, np.nan), (n_k // 2, np.inf), (n_k - 1, -np.inf), (n_k - 1, np.nan)]:
            broken = k.copy()
            broken[where, 1] = bad
            with pytest.raises(ValueError, match="infs or NaNs"):
                model.eigenval_array(broken)
        assert np.array_equal(model.eigenval_array(k), clean)
    g = kdotp_golden
    kp = KdotpModel({tuple(p): c for p, c in zip(g["order2_powers"].tolist(), g["order2_coeffs"])})
    k = np.array(g["order2_dk"], dtype=float)
    k[3, 0] = np.inf
    with pytest.raises(ValueError, match="infs or NaNs"):
        kp.eigenval(k)
    constant = KdotpModel({(0, 0, 0): g["order0_coeffs"][0]})  # k does not enter: like the reference, no error
    assert np.isfinite(constant.eigenval(k)).all()


@pytest.mark.parametrize("n_orb,solver", [(40, "rocsolver"), (64, "rocsolver"), (1030, "auto")])
def test_one_k_calls_on_the_library_solver_branch_follow_k(n_orb, solver):
    """ONE k-point per call (the Z2Pack call shape, `_tb_model.py:1103-1108`) on the branch that hands the matrices to
    rocSOLVER on request: the k-point of the CURRENT call must reach H(k) -- the chunk pipeline takes it from the kernel
    arguments and skips the upload, this branch reads the uploaded copy (it was evaluated at the previous call's
    k).  (1030, "auto") takes the OWN launch chain since round 5 (own range: up to 4096 orbitals) -- the same call sequence
    on the other side of the switch; AUTO falling through to rocSOLVER is the child-process test below.)"""
    from tbmodels_amd import _lib

    r_vec, hop, pos = syn.dense_model_arrays(n_orb, 3, syn.MODEL_SEED + 700 + n_orb)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    model.set_option(_lib.TBK_OPT_EIGENSOLVER, {"auto": _lib.TBK_EIG_AUTO, "rocsolver": _lib.TBK_EIG_ROCSOLVER}[solver])
    ks = syn.random_kpoints(3, seed=900 + n_orb)
    batch = model.eigenval_array(ks)
    ref = np.array(oracle.eigenval(r_vec, hop, ks))
    _close(batch, ref)
    for i in (2, 0, 1, 1):  # every call after one at a different k (and once at the same)
        one = model.eigenval(ks[i])
        assert one.shape == (n_orb,)
        _close(one, ref[i])


def test_auto_falls_through_to_the_library_solver_above_the_own_range():
    """AUTO above the own kernels' range is rocSOLVER (`tbk_api.hip: eigenval_device_solve`; the reference has no size limit,
    `_tb_model.py:1149-1150`).  The own range ends at 4096 orbitals, too big for a test -- ``TBK_BAND_XL=0`` (read once per
    process, hence the child) ends it at 1024, so that a 1030-orbital model with the DEFAULT solver option really reaches the
    library: a 3-point batch, one-k calls each after a call at another k (the stale-k fix of ADVICE r4 on THIS branch), and a
    NaN hopping -> ValueError through `flag_nonfinite_kernel` behind the library call."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import numpy as np
import tbmodels_amd
from tbmodels_amd import _lib, synthetic as syn
from oracle import tbk_oracle as oracle
n_orb = 1030
r_vec, hop, pos = syn.dense_model_arrays(n_orb, 3, syn.MODEL_SEED + 700 + n_orb)
model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
ks = syn.random_kpoints(3, seed=900 + n_orb)
import ctypes
def library_calls(mdl):
    value = ctypes.c_int64(0)
    _lib.check(_lib.lib().tbk_model_counter(mdl._staged(), _lib.TBK_CNT_LIBRARY_CALLS, ctypes.byref(value)))
    return value.value
batch = model.eigenval_array(ks)
ref = np.array(oracle.eigenval(r_vec, hop, ks))
assert np.abs(batch - ref).max() < 1e-10, np.abs(batch - ref).max()
for i in (2, 0, 1, 1):
    one = model.eigenval(ks[i])
    assert one.shape == (n_orb,) and np.abs(one - ref[i]).max() < 1e-10, (i, np.abs(one - ref[i]).max())
assert library_calls(model) == 5, library_calls(model)  # every one of these calls went to rocSOLVER
broken = hop.copy()
broken[1, 0, n_orb - 1] = np.nan
bad = tbmodels_amd.Model.from_packed(r_vec, broken, pos=pos)
for arg in (ks, ks[1]):
    try:
        bad.eigenval(arg)
    except ValueError as exc:
        assert "infs or NaNs" in str(exc), exc
    else:
        raise AssertionError("NaN hopping went through")
assert library_calls(bad) == 2
assert np.abs(model.eigenval(ks[2]) - ref[2]).max() < 1e-10
print("CHILD OK")
"""
    env = dict(os.environ, TBK_BAND_XL="0", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    run = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert run.returncode == 0 and "CHILD OK" in run.stdout, (run.stdout[-1000:], run.stderr[-2000:])


@pytest.mark.parametrize("n_orb,n_k", [(8, 5), (8, 6000), (40, 3), (64, 50000), (100, 4), (300, 2)])
def test_non_finite_hopping_is_value_error(n_orb, n_k):
    """NaN / Inf in the model: every eigensolver path (1-wave and 4-wave reduction, QL and bisection, streaming
    reduction) terminates and `eigenval` raises like scipy's check_finite (`_tb_model.py:1147-1150`)."""
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, 4, syn.MODEL_SEED + n_orb)
    for bad in (np.nan, np.inf):
        broken = hop.copy()
        broken[1, 0, n_orb - 1] = bad
        model = tbmodels_amd.Model.from_packed(r_vec, broken, pos=pos)
        k = syn.random_kpoints(n_k)
        with pytest.raises(ValueError):
            model.eigenval(k)
        assert not np.isfinite(model.hamilton(k[:2])).all()
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)  # the library is still usable afterwards
    assert np.isfinite(np.array(model.eigenval(k[:3]))).all()


def test_periodicity_and_linearity_properties():
    """Size-independent properties at the headline orbital count: H(k + G) = H(k); H is additive in hop."""
    r_vec, hop, pos = syn.dense_model_arrays(64, 256, syn.MODEL_SEED + 1)
    k = syn.random_kpoints(256)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    h0 = model.hamilton(k)
    shift = np.array([1.0, -2.0, 3.0])
    _close(model.hamilton(k + shift), h0, 1e-11)
    half_a = tbmodels_amd.Model.from_packed(r_vec[:100], hop[:100], pos=pos)
    half_b = tbmodels_amd.Model.from_packed(r_vec[100:], hop[100:], pos=pos)
    _close(half_a.hamilton(k) + half_b.hamilton(k), h0, 1e-12)
    eig = np.array(model.eigenval(k))
    assert np.all(np.diff(eig, axis=1) >= 0)  # ascending, like eigvalsh
    _close(eig.sum(axis=1), np.trace(h0, axis1=1, axis2=2).real, 1e-10)


def _onsite_model(mat):
    """A model whose H(k) is the constant Hermitian matrix `mat` (R = 0 block stored halved)."""
    mat = np.asarray(mat, dtype=complex)
    return tbmodels_amd.Model(hop={(0, 0, 0): mat / 2}, size=len(mat), dim=3, contains_cc=False)


@pytest.mark.parametrize("solver", ["auto", "rocsolver"])
@pytest.mark.parametrize(
    "n", [1, 2, 3, 8, 9, 12, 13, 16, 17, 32, 33, 40, 48, 49, 63, 64, 65, 100, 127, 128, 129, 191, 192, 193, 200, 256, 257, 300, 384, 385, 512,
          513, 520, 768, 1000, 1024, 1030, 1536, 2048, 2050]
)
def test_eigensolver_structured_matrices(solver, n):
    """
    Every eigensolver path (register-resident reduction + QL up to 64 orbitals, blocked streaming reduction +
    bisection above, rocSOLVER) on matrices that stress deflation, zero reflectors and clustered spectra:
    diagonal, multiples of the identity, block-diagonal, graded, rank one, purely imaginary couplings, random.
    """
    from tbmodels_amd import _lib

    if solver == "rocsolver" and n > 64 and n not in (65, 128, 256, 520):
        pytest.skip("rocSOLVER path sampled at a few sizes only (slow)")
    if solver == "rocsolver" and n > 1024:
        pytest.skip("rocSOLVER path sampled at a few sizes only (slow); above 4096 orbitals 'auto' is that path")

    rng = np.random.default_rng(100 + n)
    rand = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    rand = (rand + rand.conj().T) / 2
    cases = {
        "diagonal": np.diag(rng.standard_normal(n)).astype(complex),
        "identity": np.eye(n, dtype=complex) * 0.75,
        "random": rand,
        "graded": rand * np.outer(10.0 ** -np.arange(n) / max(1, n // 8), np.ones(n)),
        "imag_offdiag": np.diag(np.arange(n, dtype=float)) + 1j * (np.eye(n, k=1) - np.eye(n, k=-1)),
        "tiny": rand * 1e-30,  # no fixed thresholds anywhere: eigenvalues scale with the matrix
        "huge": rand * 1e30,
    }
    cases["graded"] = (cases["graded"] + cases["graded"].conj().T) / 2
    if n >= 4:
        blk = np.zeros((n, n), dtype=complex)
        h = n // 2
        blk[:h, :h] = rand[:h, :h]
        blk[h:, h:] = rand[:n - h, :n - h]
        cases["two_equal_blocks"] = blk
        proj = np.outer(rand[:, 0], rand[:, 0].conj())
        cases["rank_one"] = proj
        # exactly dependent columns inside panels (round 5: the panel QR takes all reflectors of a round from ONE Gram matrix
        # and must notice that a column's remaining norm has cancelled): every fourth row / column a copy of its neighbour
        dup = rand.copy()
        for q in range(1, n - 1, 4):
            dup[q + 1, :] = dup[q, :]
            dup[:, q + 1] = dup[:, q]
        cases["duplicate_columns"] = (dup + dup.conj().T) / 2
    # (rocSOLVER's zheevd deflates against absolute thresholds -- on "tiny" its eigenvalues were 17 % off at n = 17 --
    # so tbk_eig_batched scales every matrix to unit size first, like LAPACK's zheev* do)
    code = {"auto": _lib.TBK_EIG_AUTO, "rocsolver": _lib.TBK_EIG_ROCSOLVER}[solver]
    for name, mat in cases.items():
        model = _onsite_model(mat)
        if not model.hop:  # all-zero matrices are dropped, like the reference
            continue
        model.set_option(_lib.TBK_OPT_EIGENSOLVER, code)
        eig = np.array(model.eigenval([[0.1, 0.2, 0.3], [0.0, 0.0, 0.0]]))
        ref = np.linalg.eigvalsh(mat)
        err = np.abs(eig - ref[None]).max()
        scale = np.abs(ref).max() if name in ("tiny", "huge") else max(1.0, np.abs(ref).max())
        assert err <= 1e-12 * scale * n, (name, err)
        if solver == "auto" and n <= 64:
            # two k-points take the bisection kernel; a batch past max(4096, 768 n) takes the QL pipeline (several
            # chunks, last one bisection): both must agree with LAPACK on every row
            many = np.array(model.eigenval(np.zeros((768 * n + 4100, 3))))
            err = np.abs(many - ref[None]).max()
            assert err <= 1e-12 * scale * n, (name, "large batch", err)


@pytest.mark.parametrize("n_orb", [20, 33, 40, 48, 64])
def test_ql_pipeline_agrees_with_bisection_on_every_row(n_orb):
    """One call past max(4096, 768 n) k-points takes the lane-per-matrix QL (64 DIFFERENT matrices per wave, every lane
    with its own deflation state); the same k-points in calls of 4096 take the bisection kernel.  Every row must agree
    (two independent tridiagonal eigensolvers behind the same reduction), a sample must match the oracle."""
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, 8, syn.MODEL_SEED + 300 + n_orb)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    nk = 768 * n_orb + 5003
    k = syn.random_kpoints(nk, seed=77 + n_orb)
    big = model.eigenval_array(k)
    small = np.concatenate([model.eigenval_array(k[i:i + 4096]) for i in range(0, nk, 4096)])
    assert big.shape == small.shape == (nk, n_orb)
    assert np.abs(big - small).max() < 1e-12
    assert np.all(np.diff(big, axis=1) >= 0)
    idx = np.random.default_rng(5).choice(nk, 48, replace=False)
    _close(big[idx], np.array(oracle.eigenval(r_vec, hop, k[idx])))


@pytest.mark.parametrize("n, nk", [(513, 3), (600, 3), (777, 3), (1030, 3), (1300, 3), (300, 260), (385, 260), (449, 257), (512, 257), (600, 260), (768, 257)])
def test_second_stage_in_the_lds_window_equals_the_one_in_global_memory(n, nk):
    """Above 512 orbitals the 16 working diagonals of the second stage do not fit the LDS: the ~490 columns the 32 sweeps in
    flight touch live in a cyclic LDS window in front of the global buffer (csrc/tbk_eig_band_chase.hip, band_chase4w_kernel;
    tools/two_stage_model.py: stage2_window).  ``TBK_CHASE_WINDOW=0`` (read once per process, hence the child) works in global
    memory throughout (band_chase4g_kernel, the round-4 form).  257 - 768 orbitals: calls of more than 256 matrices take a window
    of 16 sweep slots and 272 columns in 78 KiB, so that two workgroups share a CU; with the switch off the plain LDS form up to
    512 orbitals (band_chase4_kernel).  The same sweeps in another schedule: (d, e) agree bit for
    bit, and the spectra are the matrices' (scipy's eigvalsh at _tb_model.py:1149)."""
    import os
    import subprocess
    import sys
    import tempfile

    import scipy.linalg as la

    from tbmodels_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys
import numpy as np
from tbmodels_amd import _lib
n, nk, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(4000 + n)
h = np.empty((nk, n, n), dtype=complex)
for i in range(nk):
    m = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    h[i] = (m + m.conj().T) / 2
h[1] *= 1e-20
h[2, : n // 2, n // 2 :] = 0.0
h[2, n // 2 :, : n // 2] = 0.0
d, e = np.empty((nk, n)), np.empty((nk, n))
_lib.check(_lib.lib().tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(h), _lib.TBK_REDUCE_TWO_STAGE, _lib.ptr(d), _lib.ptr(e), None))
np.savez(out, d=d, e=e, h=h[:3])
"""
    with tempfile.TemporaryDirectory() as tmp:
        got = {}
        for label, window in (("window", "1"), ("global", "0")):
            out = os.path.join(tmp, label + ".npz")
            env = dict(os.environ, TBK_CHASE_WINDOW=window, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
            run = subprocess.run([sys.executable, "-c", code, str(n), str(nk), out], env=env, capture_output=True, text=True, timeout=600, cwd=root)
            assert run.returncode == 0, run.stderr[-2000:]
            got[label] = np.load(out)
    assert np.array_equal(got["window"]["d"], got["global"]["d"]) and np.array_equal(got["window"]["e"], got["global"]["e"])
    d, e, h = got["window"]["d"], got["window"]["e"], got["window"]["h"]
    assert np.isfinite(d).all() and np.isfinite(e).all()
    for i in range(3):
        ref = np.linalg.eigvalsh(h[i])
        assert np.abs(la.eigvalsh_tridiagonal(d[i], e[i, :-1]) - ref).max() <= 1e-13 * n * np.abs(ref).max(), i
    assert _lib is not None


def test_the_read_once_sweep_above_1024_orbitals_stays_correct():
    """``TBK_BAND_XL_SWEEP4=1`` (a measurement switch of the EXPERIMENTS build of the library, `libtbk_experiments.so` -- the default
    library does not read it; read once per process, hence the child process): the panels' sweeps run four
    block rows per workgroup, every tile is read once and the transposed products are added up through partial sums
    (csrc/tbk_eig_band_xl.hip, band_xl_sweep4_kernel + band_xl_xsum_kernel; DESIGN_LOG.md R5.12: built, not faster, off by default).
    Same function as the one-row sweep: the spectra are those of the matrices (scipy's eigvalsh at _tb_model.py:1149), two runs
    give the same bits (the partial sums are added in a fixed order)."""
    import os
    import subprocess
    import sys

    from tbmodels_amd import _lib

    if not os.path.exists(_lib.EXPERIMENTS_LIB_PATH):
        pytest.skip("the read-once sweep is a dropped variant: only in the experiments build (make -C tbmodels_amd/csrc EXPERIMENTS=1)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import numpy as np, scipy.linalg as la
from tbmodels_amd import _lib
lib = _lib.lib()
assert b"+experiments" in lib.tbk_version()
n, nk = 1040, 9
rng = np.random.default_rng(78)
m = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
h = np.ascontiguousarray((m + m.conj().transpose(0, 2, 1)) / 2)
h[5] *= 1e-25
h[6] = np.diag(np.diagonal(h[6]).real)
out = []
for rep in range(2):
    d, e = np.empty((nk, n)), np.empty((nk, n))
    _lib.check(lib.tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(h), _lib.TBK_REDUCE_AUTO, _lib.ptr(d), _lib.ptr(e), None))
    out.append((d, e))
assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
d, e = out[0]
assert np.isfinite(d).all() and np.isfinite(e).all()
for i in (0, 5, 6, 8):
    ref = np.linalg.eigvalsh(h[i])
    got = la.eigvalsh_tridiagonal(d[i], e[i, :-1])
    assert np.abs(got - ref).max() <= 1e-13 * n * np.abs(ref).max(), i
print("ok")
"""
    env = dict(os.environ, TBK_BAND_XL_SWEEP4="1", TBK_LIBTBK=_lib.EXPERIMENTS_LIB_PATH,
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    run = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stderr[-2000:]


def test_a_batch_above_1024_orbitals_in_groups_equals_its_matrices_one_at_a_time():
    """Above 1024 orbitals a batch goes in groups of matrices on streams of their own (csrc/tbk_eig_band_xl.hip, tbk_band_launch_xl:
    one group's serial phases and second stage under the other groups' sweeps).  Per matrix nothing may change: (d, e) and the
    band left in the work copy of a batch of 11 (two groups of 6 and 5) equal those of the same matrices reduced one per call,
    bit for bit (the independence of k-points at _tb_model.py:1111-1123)."""
    from tbmodels_amd import _lib

    lib = _lib.lib()
    n, nk = 1040, 11
    rng = np.random.default_rng(77)
    m = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = np.ascontiguousarray((m + m.conj().transpose(0, 2, 1)) / 2)
    h[4] *= 1e-20
    d, e, red = np.empty((nk, n)), np.empty((nk, n)), np.empty_like(h)
    _lib.check(lib.tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(h), _lib.TBK_REDUCE_AUTO, _lib.ptr(d), _lib.ptr(e), _lib.ptr(red)))
    for i in range(nk):
        one = np.ascontiguousarray(h[i : i + 1])
        d1, e1, r1 = np.empty((1, n)), np.empty((1, n)), np.empty_like(one)
        _lib.check(lib.tbk_tridiagonal_reduce(0, n, 1, _lib.ptr(one), _lib.TBK_REDUCE_AUTO, _lib.ptr(d1), _lib.ptr(e1), _lib.ptr(r1)))
        assert np.array_equal(d[i], d1[0]) and np.array_equal(e[i], e1[0]), i
        band = lambda a: np.triu(a) - np.triu(a, 9)  # noqa: E731  (half-width 8)
        assert np.array_equal(band(red[i]), band(r1[0])), i
    ref = np.linalg.eigvalsh(h[10])
    import scipy.linalg as la

    assert np.abs(la.eigvalsh_tridiagonal(d[10], e[10, :-1]) - ref).max() <= 1e-13 * n * np.abs(ref).max()


@pytest.mark.parametrize("n", [1, 7, 33, 64, 65, 72, 96, 130, 257, 400, 512, 513, 700, 1024, 1030, 1300])
def test_tridiagonal_reduce_on_caller_supplied_matrices(n):
    """``tbk_tridiagonal_reduce``: the reduction stage of the eigensolver alone (scipy's eigvalsh at _tb_model.py:1149 is
    this plus the tridiagonal stage) on random Hermitian batches whose lower triangle is poisoned with NaN -- only the
    upper triangle may be read.  Above 64 orbitals (two-stage reduction, csrc/tbk_eig_band.hip) the work copy must be
    a band matrix of half-width 8 with the same spectrum, and equal the NumPy model of the algorithm
    (tools/two_stage_model.py) entry by entry."""
    import os
    import sys

    import scipy.linalg as la

    from tbmodels_amd import _lib

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import two_stage_model as model

    lib = _lib.lib()
    rng = np.random.default_rng(1000 + n)
    nk = 9 if n <= 512 else 5  # (above 1024 orbitals: the launch chain of band_xl_*, round 5)
    m = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))
    h = (m + m.conj().transpose(0, 2, 1)) / 2
    h[1] *= 1e-30  # no absolute thresholds
    h[2] *= 1e30
    h[3] = np.diag(np.diagonal(h[3]))  # already reduced: zero reflectors throughout
    poisoned = np.ascontiguousarray(h.copy())
    il = np.tril_indices(n, -1)
    poisoned[:, il[0], il[1]] = np.nan
    d, e, red = np.empty((nk, n)), np.empty((nk, n)), np.empty_like(poisoned)
    # the two-stage kernels at every size they handle (eigenval itself takes them from 185 orbitals on); the
    # production choice and the forced one-stage path are compared below
    method = _lib.TBK_REDUCE_TWO_STAGE if n > 64 else _lib.TBK_REDUCE_AUTO
    _lib.check(lib.tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(poisoned), method, _lib.ptr(d), _lib.ptr(e), _lib.ptr(red)))
    assert np.isfinite(d).all() and np.isfinite(e).all() and np.all(e[:, n - 1] == 0.0)
    for i in range(nk):
        ref = np.linalg.eigvalsh(h[i])
        scale = np.abs(ref).max()
        got = la.eigvalsh_tridiagonal(d[i], e[i, :-1]) if n > 1 else d[i]
        assert np.abs(got - ref).max() <= 1e-13 * n * scale, i
        if n > 64:
            up = np.triu(red[i])
            band = up - np.triu(up, model.B + 1)
            hb = band + np.triu(band, 1).conj().T
            assert np.abs(np.linalg.eigvalsh(hb) - ref).max() <= 1e-13 * n * scale, i
    if 64 < n <= 130:
        mband, _ = model.stage1_band(h[0])
        gband = np.array([[red[0][r, r + dd] if r + dd < n else 0.0 for dd in range(model.B + 1)] for r in range(n)])
        assert np.abs(gband - mband).max() < 1e-12 * n
    # repeated calls give the same bits; bad sizes are refused before any device work
    d2, e2 = np.empty_like(d), np.empty_like(e)
    _lib.check(lib.tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(poisoned), method, _lib.ptr(d2), _lib.ptr(e2), None))
    assert np.array_equal(d, d2) and np.array_equal(e, e2)
    for other in (_lib.TBK_REDUCE_AUTO, _lib.TBK_REDUCE_ONE_STAGE):  # every path: the same spectrum
        if other == _lib.TBK_REDUCE_ONE_STAGE and n > 512:  # the one-stage kernel stops at 512 orbitals
            assert lib.tbk_tridiagonal_reduce(0, n, 1, _lib.ptr(poisoned), other, _lib.ptr(d2), _lib.ptr(e2), None) == _lib.TBK_ERR_ARGUMENT
            continue
        _lib.check(lib.tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(poisoned), other, _lib.ptr(d2), _lib.ptr(e2), None))
        for i in range(nk):
            ref = np.linalg.eigvalsh(h[i])
            got = la.eigvalsh_tridiagonal(d2[i], e2[i, :-1]) if n > 1 else d2[i]
            assert np.abs(got - ref).max() <= 1e-13 * n * np.abs(ref).max(), (other, i)
    bad = lib.tbk_tridiagonal_reduce(0, 4097, 1, _lib.ptr(poisoned), 0, _lib.ptr(d), _lib.ptr(e), None)
    assert bad == _lib.TBK_ERR_ARGUMENT
    if n <= 64:
        assert lib.tbk_tridiagonal_reduce(0, n, 1, _lib.ptr(poisoned), 2, _lib.ptr(d), _lib.ptr(e), None) == _lib.TBK_ERR_ARGUMENT


@pytest.mark.parametrize("n", [3000, 4096])
def test_largest_sizes_of_the_own_path(n):
    """The launch chain above 1024 orbitals at the top of its validated range (the structured-matrix test stops at 2050: LAPACK
    on the host needs ~10 - 30 s per matrix here): a random Hermitian matrix through ``eigenval`` against numpy.linalg.eigvalsh,
    and two STRUCTURED matrices whose spectra are known without LAPACK -- a graded and a clustered diagonal D (16 decades; 64
    values repeated 64 times) turned dense by two Householder similarities, M = H2 H1 D H1 H2 (O(n^2) to build, spectrum = D up
    to the rounding of that) -- each at two k-points (the bisection's eigenvalues span three / four workgroups per matrix).
    Tolerance n eps-like, as for the (d, e) tests: 1e-13 n max|lambda| (it was 1e-12 n, ~4e-7 absolute at 4096)."""
    rng = np.random.default_rng(7000 + n)
    rand = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    rand = (rand + rand.conj().T) / 2

    def reflected(diag):
        mat = np.diag(diag).astype(complex)
        for _ in range(2):
            u = rng.standard_normal(n) + 1j * rng.standard_normal(n)
            u /= np.linalg.norm(u)
            mu = mat @ u
            mat = mat - 2.0 * np.outer(u, u.conj() @ mat) - 2.0 * np.outer(mu, u.conj()) + 4.0 * (u.conj() @ mu) * np.outer(u, u.conj())
        return (mat + mat.conj().T) / 2

    graded = np.sort(10.0 ** (-16.0 * np.arange(n) / n) * np.where(np.arange(n) % 2, -1.0, 1.0))
    clustered = np.sort(np.repeat(np.linspace(-3.0, 5.0, 64), (n + 63) // 64)[:n])
    cases = [("random", rand, None), ("graded", reflected(graded), graded), ("clustered", reflected(clustered), clustered)]
    for name, mat, known in cases:
        model = _onsite_model(mat)
        eig = np.array(model.eigenval([[0.1, 0.2, 0.3], [0.0, 0.0, 0.0]]))
        ref = np.linalg.eigvalsh(mat) if known is None else known
        assert np.abs(eig - ref[None]).max() <= 1e-13 * max(1.0, np.abs(ref).max()) * n, (name, np.abs(eig - ref[None]).max())
        assert np.array_equal(eig[0], eig[1])


@pytest.mark.parametrize("n", [66, 97, 130, 200, 300])
def test_launch_chain_above_1024_orbitals_equals_the_model_at_small_sizes(n):
    """Above 1024 orbitals the first stage is a chain of launches with nothing per row in registers or LDS
    (csrc/tbk_eig_band_xl.hip, band_xl_*: serial phases / update sweep / product sweep per panel).  TBK_BAND_XL_FROM=64 (read
    once per process) sends EVERY two-stage size down that chain: the band it leaves must equal the NumPy model of the
    algorithm (tools/two_stage_model.py with the Gram-matrix panel QR) entry by entry, the tridiagonal behind the second
    stage must have the matrix' spectrum, a batch with scaled / diagonal members included, and the chain is deterministic."""
    import os
    import subprocess
    import sys
    import tempfile

    import scipy.linalg as la

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import two_stage_model as model

    script = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from tbmodels_amd import _lib\n"
        "n = %d; nk = 5; rng = np.random.default_rng(4000 + n)\n"
        "m = rng.standard_normal((nk, n, n)) + 1j * rng.standard_normal((nk, n, n))\n"
        "h = (m + m.conj().transpose(0, 2, 1)) / 2\n"
        "h[1] *= 1e-30; h[2] *= 1e30; h[3] = np.diag(np.diagonal(h[3]))\n"
        "p = np.ascontiguousarray(h.copy()); il = np.tril_indices(n, -1); p[:, il[0], il[1]] = np.nan\n"
        "d, e, red = np.empty((nk, n)), np.empty((nk, n)), np.empty_like(p)\n"
        "lib = _lib.lib()\n"
        "_lib.check(lib.tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(p), _lib.TBK_REDUCE_TWO_STAGE, _lib.ptr(d), _lib.ptr(e), _lib.ptr(red)))\n"
        "d2, e2 = np.empty_like(d), np.empty_like(e)\n"
        "_lib.check(lib.tbk_tridiagonal_reduce(0, n, nk, _lib.ptr(p), _lib.TBK_REDUCE_TWO_STAGE, _lib.ptr(d2), _lib.ptr(e2), None))\n"
        "assert np.array_equal(d, d2) and np.array_equal(e, e2)\n"
        "np.savez(sys.argv[1], h=h, d=d, e=e, red=red)\n" % (root, n)
    )
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "xl.npz")
        subprocess.run([sys.executable, "-c", script, out], check=True, env=dict(os.environ, TBK_BAND_XL_FROM="64"), timeout=600)
        got = np.load(out)
    h, d, e, red = got["h"], got["d"], got["e"], got["red"]
    assert np.isfinite(d).all() and np.isfinite(e).all()
    for i in range(len(h)):
        ref = np.linalg.eigvalsh(h[i])
        scale = np.abs(ref).max()
        assert np.abs(la.eigvalsh_tridiagonal(d[i], e[i, :-1]) - ref).max() <= 1e-13 * n * scale, i
        up = np.triu(red[i])
        band = up - np.triu(up, model.B + 1)
        hb = band + np.triu(band, 1).conj().T
        assert np.abs(np.linalg.eigvalsh(hb) - ref).max() <= 1e-13 * n * scale, i
    saved = model.panel_qr
    model.panel_qr = model.panel_qr_gram
    try:
        mband, _ = model.stage1_band(h[0])
    finally:
        model.panel_qr = saved
    gband = np.array([[red[0][r, r + dd] if r + dd < n else 0.0 for dd in range(model.B + 1)] for r in range(n)])
    assert np.abs(gband - mband).max() < 1e-12 * n


@pytest.mark.parametrize("n_orb", [200, 300, 520])
def test_launch_chain_of_small_calls_agrees_with_the_one_launch_kernel(n_orb):
    """Calls of a few matrices take the first stage of the two-stage reduction as a chain of launches (every tile pass on
    several CUs, csrc/tbk_eig_band.hip PHASE 1 / 2); TBK_BAND_SPLIT=0 (read once per process) keeps the one-launch kernels:
    the same eigenvalues to rounding, both within 1e-10 of the oracle; the chain is deterministic."""
    import os
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r_vec, hop, pos = syn.dense_model_arrays(n_orb, 5, syn.MODEL_SEED + 400 + n_orb)
    k = syn.random_kpoints(7, seed=n_orb)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    here = model.eigenval_array(k)
    assert np.array_equal(here, model.eigenval_array(k))
    one = model.eigenval(k[0])
    assert np.abs(one - here[0]).max() < 1e-12
    _close(here, np.array(oracle.eigenval(r_vec, hop, k)))
    script = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import tbmodels_amd; from tbmodels_amd import synthetic as syn\n"
        "r, h, p = syn.dense_model_arrays(%d, 5, syn.MODEL_SEED + 400 + %d)\n"
        "m = tbmodels_amd.Model.from_packed(r, h, pos=p)\n"
        "np.save(sys.argv[1], m.eigenval_array(syn.random_kpoints(7, seed=%d)))\n" % (root, n_orb, n_orb, n_orb)
    )
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "e.npy")
        subprocess.run([sys.executable, "-c", script, out], check=True, env=dict(os.environ, TBK_BAND_SPLIT="0"), timeout=300)
        other = np.load(out)
    assert np.abs(other - here).max() < 1e-11


def test_launch_chain_above_1024_orbitals_in_the_chunk_pipeline():
    """1040 orbitals in THREE k chunks: the launch chain of band_xl_* on the reduction stream, the global-memory chase and the
    bisection of the previous chunk on the tridiagonal stream beside it, two band buffers in turn -- the same bits on a second
    run, what other call sizes give to rounding, the oracle's values on a sample, and (TBK_EIG_ROCSOLVER) the library's."""
    from tbmodels_amd import _lib

    r_vec, hop, pos = syn.dense_model_arrays(1040, 3, syn.MODEL_SEED + 1040)
    k = syn.random_kpoints(250, seed=1040)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    model.set_option(_lib.TBK_OPT_K_CHUNK, 100)  # 100 + 100 + 50: calls of more than 96 matrices, chunks that are not
    whole = model.eigenval_array(k)
    assert np.all(np.diff(whole, axis=1) >= 0)
    pieces = np.concatenate([model.eigenval_array(k[i:i + 125]) for i in (0, 125)])  # (two calls of two chunks each)
    assert np.abs(whole - pieces).max() < 1e-12 and np.array_equal(whole, model.eigenval_array(k))
    _close(whole[[0, 99, 100, 249]], np.array(oracle.eigenval(r_vec, hop, k[[0, 99, 100, 249]])))
    model.set_option(_lib.TBK_OPT_EIGENSOLVER, _lib.TBK_EIG_ROCSOLVER)
    lib_rows = model.eigenval_array(k[:6])
    assert np.abs(lib_rows - whole[:6]).max() < 1e-11


def test_two_stage_and_one_stage_reductions_agree():
    """TBK_BAND=0 (read once per process) selects the one-stage streaming reduction of tbk_eig_stream.hip: same
    eigenvalues to rounding as the default two-stage path, on a multi-chunk call."""
    import os
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r_vec, hop, pos = syn.dense_model_arrays(200, 6, syn.MODEL_SEED + 321)  # (above the crossover at 185 orbitals)
    k = syn.random_kpoints(2600, seed=5)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    from tbmodels_amd import _lib

    model.set_option(_lib.TBK_OPT_K_CHUNK, 1024)  # three chunks: the stages of neighbouring chunks overlap
    here = model.eigenval_array(k)
    _close(here[:40], np.array(oracle.eigenval(r_vec, hop, k[:40])))
    script = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import tbmodels_amd; from tbmodels_amd import synthetic as syn, _lib\n"
        "r, h, p = syn.dense_model_arrays(200, 6, syn.MODEL_SEED + 321)\n"
        "m = tbmodels_amd.Model.from_packed(r, h, pos=p); m.set_option(_lib.TBK_OPT_K_CHUNK, 1024)\n"
        "np.save(sys.argv[1], m.eigenval_array(syn.random_kpoints(2600, seed=5)))\n" % root
    )
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "e.npy")
        env = dict(os.environ, TBK_BAND="0")
        subprocess.run([sys.executable, "-c", script, out], check=True, env=env, timeout=300)
        other = np.load(out)
    assert 0.0 < np.abs(other - here).max() < 1e-11  # a different algorithm, the same spectrum


@pytest.mark.parametrize("n_orb,switch", [(72, "TBK_REG128"), (96, "TBK_REG128_NW2"), (128, "TBK_REG128"), (150, "TBK_REG128"), (230, "TBK_REG128")])
def test_register_cascade_agrees_with_the_streaming_kernel(n_orb, switch):
    """65-128 orbitals never leave the registers (herm_tridiag8_kernel: four waves per matrix up to 96 orbitals, eight
    above; 128 -> 64 rows), and above 128 the streaming kernel hands over its trailing 128 x 128 block.  TBK_REG128=0
    (read once per process) keeps the streaming kernel down to 64 rows, TBK_REG128_NW2=0 the eight-wave form at every
    size: same eigenvalues to rounding on a batch of several rounds of workgroups, and the oracle's on a sample.
    (230 orbitals: the one-stage path on request, TBK_BAND=0 in both processes.)"""
    import os
    import subprocess
    import sys
    import tempfile

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    nk = 1500 if n_orb <= 150 else 600
    script = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import tbmodels_amd; from tbmodels_amd import synthetic as syn\n"
        "r, h, p = syn.dense_model_arrays(%d, 5, syn.MODEL_SEED + 77)\n"
        "m = tbmodels_amd.Model.from_packed(r, h, pos=p)\n"
        "np.save(sys.argv[1], m.eigenval_array(syn.random_kpoints(%d, seed=9)))\n" % (root, n_orb, nk)
    )
    results = []
    with tempfile.TemporaryDirectory() as tmp:
        for value in (None, "0"):
            out = os.path.join(tmp, "e%s.npy" % value)
            env = dict(os.environ)
            env.pop(switch, None)
            if n_orb > 184:
                env["TBK_BAND"] = "0"
            if value is not None:
                env[switch] = value
            subprocess.run([sys.executable, "-c", script, out], check=True, env=env, timeout=300)
            results.append(np.load(out))
    assert results[0].shape == (nk, n_orb)
    assert 0.0 < np.abs(results[0] - results[1]).max() < 1e-11  # different kernels, the same spectrum
    r_vec, hop, _ = syn.dense_model_arrays(n_orb, 5, syn.MODEL_SEED + 77)
    k = syn.random_kpoints(nk, seed=9)
    idx = np.random.default_rng(3).choice(nk, 6, replace=False)
    _close(results[0][idx], np.array(oracle.eigenval(r_vec, hop, k[idx])))


def test_wave_solver_rejects_large_n():
    from tbmodels_amd import _lib

    r_vec, hop, pos = syn.dense_model_arrays(80, 3, syn.MODEL_SEED + 5)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    model.set_option(_lib.TBK_OPT_EIGENSOLVER, _lib.TBK_EIG_WAVE)
    with pytest.raises(ValueError):
        model.eigenval([0.1, 0.2, 0.3])
    model.set_option(_lib.TBK_OPT_EIGENSOLVER, _lib.TBK_EIG_AUTO)  # falls back to rocSOLVER above 64
    k = syn.random_kpoints(5)
    _close(np.array(model.eigenval(k)), np.array(oracle.eigenval(r_vec, hop, k)))


@pytest.mark.parametrize("sparse", [False, True])
def test_construct_kdotp(kdotp_golden, sparse):
    """Model.construct_kdotp (reference _tb_model.py:942-982; tests/test_kdotp.py) against the reference's output."""
    g = kdotp_golden
    model = tbmodels_amd.Model.from_packed(g["R"], g["hop"], pos=g["pos"], sparse=sparse)
    for order in (0, 1, 2, 3):
        kp = model.construct_kdotp(g["k0"], order)
        powers = [tuple(p) for p in g["order%d_powers" % order].tolist()]
        assert sorted(kp.taylor_coefficients) == powers
        got = np.array([kp.taylor_coefficients[p] for p in powers])
        _close(got, g["order%d_coeffs" % order], 1e-9)  # coefficients grow like (2 pi |R|)^order
        dk = g["order%d_dk" % order]
        _close(kp.hamilton(dk), g["order%d_h" % order])
        _close(np.array(kp.eigenval(dk)), g["order%d_eig" % order])
    # tests/test_kdotp.py:47-57 of the reference: at dk = 0 the k.p model reproduces the TB eigenvalues at k0
    kp = model.construct_kdotp(g["k0"], 2)
    _close(kp.eigenval([0.0, 0.0, 0.0]), model.eigenval(g["k0"]))
    with pytest.raises(ValueError):
        model.construct_kdotp(g["k0"], -1)


@pytest.mark.parametrize("n_r", [300, 700, 1500, 2400, 3000])
def test_csr_kernel_variants(n_r):
    """Every phase-tile width of the LDS sparse kernel (8/4/2/1 k-points) and the global-gather fallback."""
    r_vec, r_ptr, row, col, val, pos = syn.csr_model_arrays(12, n_r, syn.MODEL_SEED + n_r, fill=0.2)
    hop = syn.csr_to_dense(12, r_ptr, row, col, val)
    k = syn.random_kpoints(77, seed=n_r)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos, sparse=True)
    _close(model.hamilton(k), oracle.hamilton(r_vec, hop, k))
    _close(model.hamilton(k, convention=1), oracle.hamilton(r_vec, hop, k, 1, pos=pos))
    _close(np.array(model.eigenval(k)), np.array(oracle.eigenval(r_vec, hop, k)))


@pytest.mark.parametrize("n,n_r", [(10, 4), (70, 9)])
def test_csr_duplicates_unsorted_and_explicit_zeros(n, n_r):
    """Straight through the C ABI (the Python model canonicalises its matrices): duplicate (row, col) entries are
    summed like scipy's toarray(), entries come in any order inside an R block, stored zeros are harmless
    (include/tbk.h, tbk_model_create_csr)."""
    import ctypes

    from tbmodels_amd import _lib

    rng = np.random.default_rng(99 + n)
    r_vec = np.ascontiguousarray(syn.half_space_vectors(n_r), dtype=np.int32)  # R = 0 first
    assert not r_vec[0].any()
    dense = np.zeros((n_r, n, n), dtype=complex)
    r_ptr, rows, cols, vals = [0], [], [], []
    for idx in range(n_r):
        nnz = 4 * n  # with replacement: duplicates are certain
        row = rng.integers(0, n, nnz).astype(np.int32)  # not even grouped by row
        col = rng.integers(0, n, nnz).astype(np.int32)
        val = rng.normal(size=nnz) + 1j * rng.normal(size=nnz)
        val[::7] = 0.0
        if idx == 0:  # the stored R = 0 block is half of a Hermitian block (a8): add every entry's mirror image
            row, col, val = np.concatenate([row, col]), np.concatenate([col, row]), np.concatenate([val, val.conj()]) / 2
        np.add.at(dense[idx], (row, col), val)
        rows.append(row)
        cols.append(col)
        vals.append(val)
        r_ptr.append(r_ptr[-1] + len(row))
    r_ptr = np.array(r_ptr, dtype=np.int64)
    rows, cols = np.ascontiguousarray(np.concatenate(rows)), np.ascontiguousarray(np.concatenate(cols))
    vals = np.ascontiguousarray(np.concatenate(vals), dtype=np.complex128)
    lib = _lib.lib()
    handle = ctypes.c_void_p()
    _lib.check(lib.tbk_model_create_csr(0, 3, n, n_r, _lib.ptr(r_vec), _lib.ptr(r_ptr), _lib.ptr(rows), _lib.ptr(cols),
                                        _lib.ptr(vals), ctypes.byref(handle)))
    try:
        k = syn.random_kpoints(50, seed=5)
        ham = np.empty((len(k), n, n), dtype=np.complex128)
        eig = np.empty((len(k), n))
        _lib.check(lib.tbk_hamilton(handle, _lib.ptr(k), len(k), 2, None, _lib.ptr(ham)))
        _lib.check(lib.tbk_eigenval(handle, _lib.ptr(k), len(k), _lib.ptr(eig)))
    finally:
        lib.tbk_model_destroy(handle)
    _close(ham, oracle.hamilton(r_vec, dense, k))
    _close(eig, np.array(oracle.eigenval(r_vec, dense, k)))


def test_threads_sharing_one_model_and_threads_with_their_own():
    """ctypes drops the GIL during a call: concurrent host threads on ONE handle are serialised by the library (they
    share its workspaces), threads with their own models just run; both give the single-threaded results."""
    import threading

    r_vec, hop, pos = syn.dense_model_arrays(20, 30, syn.MODEL_SEED + 77)
    shared = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    batches = [syn.random_kpoints(n_k, seed=100 + i) for i, n_k in enumerate((1, 37, 5000, 300, 9000, 2, 700, 4097))]
    expected_e = [shared.eigenval_array(k) for k in batches]
    expected_h = [shared.hamilton(k[:50], convention=1) for k in batches]
    results = {}
    errors = []

    def work(tag, model, idx):
        try:
            for _ in range(3):
                results[(tag, idx, "e")] = model.eigenval_array(batches[idx])
                results[(tag, idx, "h")] = model.hamilton(batches[idx][:50], convention=1)
        except Exception as exc:  # pylint: disable=broad-except
            errors.append(exc)

    threads = [threading.Thread(target=work, args=("shared", shared, i)) for i in range(len(batches))]
    own = [tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos) for _ in range(4)]
    threads += [threading.Thread(target=work, args=("own", own[i], i)) for i in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for (tag, idx, what), got in results.items():
        ref = expected_e[idx] if what == "e" else expected_h[idx]
        _close(got, ref, 1e-13)


@pytest.mark.parametrize("n", [5, 13, 16, 17, 24, 32, 33, 64, 65, 100])
def test_block_diagonal_spectra_symmetric_about_zero(n):
    """Very sparse on-site blocks: the tridiagonal form is block diagonal (e_i = 0 exactly), most eigenvalues are
    exactly 0 and the rest come in +/- pairs, so the first bisection midpoint IS an eigenvalue.  The polynomial
    Sturm recurrence once zeroed out behind two consecutive zeros there (every positive eigenvalue came back as 0);
    found by tools/fuzz_parity.py."""
    rng = np.random.default_rng(1000 + n)
    for fill in (0.02, 0.05, 0.2):
        for _ in range(6):
            mat = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) * (rng.random((n, n)) < fill)
            mat = (mat + mat.conj().T) / 2
            model = tbmodels_amd.Model(hop={(0,): mat / 2}, size=n, dim=1, contains_cc=False)
            ref = np.linalg.eigvalsh(mat)
            for n_k in (1, 3, 4200):  # bisection for small calls, QL + bisection for the chunked pipeline
                got = model.eigenval_array(rng.random((n_k, 1)))
                _close(got, np.broadcast_to(ref, got.shape))


@pytest.mark.parametrize("n_k", [33, 64, 65, 1000, 4096, 4097])
def test_small_model_batches_take_the_grouped_matrix_vector_path(silicon, n_k):
    """Small models (<= 22 orbitals, < 128 lattice vectors) evaluate up to 4096 k-points in groups of 32 with the
    matrix-vector kernel (tbk_hk_gemv_path); 4097 k-points are back on the MFMA tiles.  Both conventions, both H modes."""
    model = tbmodels_amd.Model.from_packed(silicon["R"], silicon["hop"], pos=silicon["pos"])
    k = syn.random_kpoints(n_k, seed=n_k) * 2.0 - 0.7
    sub = np.unique(np.concatenate([np.arange(0, n_k, max(1, n_k // 37)), [n_k - 1, n_k - 2, 31, 32]]))
    ham = model.hamilton(k)
    _close(ham[sub], oracle.hamilton(silicon["R"], silicon["hop"], k[sub]))
    _close(model.hamilton(k, convention=1)[sub], oracle.hamilton(silicon["R"], silicon["hop"], k[sub], 1, pos=silicon["pos"]))
    eig = model.eigenval_array(k)
    _close(eig[sub], np.array(oracle.eigenval(silicon["R"], silicon["hop"], k[sub])))
    _close(eig.sum(axis=1), np.einsum("kii->k", ham).real, 1e-12)  # every row, not only the sampled ones
    assert np.array_equal(ham, np.conj(np.swapaxes(ham, 1, 2)))


def test_fixed_seed_fuzz_sample():
    """A fixed-seed sample of tools/fuzz_parity.py (random shapes around the kernel boundaries, structured hoppings,
    k.p models) so that the randomised comparison with the oracle is part of every GPU run, reproducibly."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    fuzz.configure(seconds=120.0, seed=20261002, cases=40)
    assert fuzz.main() == 0


def test_repeated_small_calls_through_the_pinned_staging_buffer():
    """One-k-point calls (the Z2Pack pattern) move k, the eigenvalues and the flag words through the handle's pinned staging
    buffer (csrc/tbk_api.hip: tbk_eigenval): right for every k of a long loop, for other small shapes, after a larger call
    in between (which grows the workspaces), the call counter keeps counting, and a NaN k-point still raises -- and
    leaves the flags clean for the calls behind it."""
    import ctypes

    from tbmodels_amd import _lib

    r_vec, hop, pos = syn.dense_model_arrays(20, 60, syn.MODEL_SEED + 321)
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    k = syn.random_kpoints(40, seed=77)
    ref = np.array(oracle.eigenval(r_vec, hop, k))

    def calls():
        value = ctypes.c_int64(-1)
        _lib.check(_lib.lib().tbk_model_counter(model._staged(), _lib.TBK_CNT_EIGENVAL_CALLS, ctypes.byref(value)))
        return value.value

    for q in range(12):
        _close(model.eigenval(k[q]), ref[q])
    assert calls() == 12
    for q in range(0, 12, 3):  # another shape: three k-points per call
        _close(np.array(model.eigenval(k[q:q + 3])), ref[q:q + 3])
    big = syn.random_kpoints(5000, seed=3)  # grows H / (d, e) / phase workspaces
    eig_big = model.eigenval_array(big)
    _close(eig_big[:3], np.array(oracle.eigenval(r_vec, hop, big[:3])))
    for q in range(12, 24):
        _close(model.eigenval(k[q]), ref[q])
    bad = k[5].copy()
    bad[1] = np.nan
    with pytest.raises(ValueError):
        model.eigenval(bad)
    for q in range(24, 30):
        _close(model.eigenval(k[q]), ref[q])
    before = calls()
    model.eigenval(k[0])
    assert calls() == before + 1
