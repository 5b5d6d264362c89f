"""
Host-side logic of tbmodels_amd.Model (no GPU): construction semantics against what the reference
itself built (fixtures), argument handling, error behaviour, pickling, the C-ABI surface.
"""

import ctypes
import os
import pickle
import re

import numpy as np
import pytest

import tbmodels_amd
from tbmodels_amd import _lib

from conftest import KPT, T_VALUES, ROOT


def toy_model(t1, t2, sparse=False, dim=3):
    """The toy model of the reference's tests/conftest.py:155-189, built through the mirrored API."""
    import itertools

    pos = [[0] * 2, [0.5] * 2]
    for position in pos:
        position.extend([0] * (dim - 2))
    model = tbmodels_amd.Model(pos=pos, occ=1, on_site=(1, -1), size=2, dim=None, sparse=sparse)
    for phase, r_part in zip([1, -1j, 1j, -1], itertools.product([0, -1], [0, -1])):
        r_vec = list(r_part) + [0] * (dim - 2)
        model.add_hop(t1 * phase, 0, 1, r_vec)
    for r_part in itertools.permutations([0, 1]):
        r_vec = list(r_part) + [0] * (dim - 2)
        model.add_hop(t2, 0, 0, r_vec)
        model.add_hop(-t2, 1, 1, r_vec)
    return model


@pytest.mark.parametrize("t_idx", range(6))
@pytest.mark.parametrize("sparse", [False, True])
def test_toy_hop_matches_reference_construction(toy, t_idx, sparse):
    """add_hop / on_site bookkeeping gives exactly the hop dict the reference builds."""
    model = toy_model(*T_VALUES[t_idx], sparse=sparse)
    tag = "t%d_%s" % (t_idx, "sparse" if sparse else "dense")
    r_ref, hop_ref = toy[tag + "_R"], toy[tag + "_hop"]
    got = {key: np.array(mat) for key, mat in model.hop.items()}
    assert set(got) == {tuple(r) for r in r_ref.tolist()}
    for r, h in zip(r_ref.tolist(), hop_ref):
        assert np.abs(got[tuple(r)] - h).max() < 1e-15
    assert np.allclose(model.pos, toy[tag + "_pos"])
    assert model._sparse is sparse


def test_packed_roundtrip_dense_and_sparse(synthetic):
    r_vec, hop, pos = synthetic["dense16_R"], synthetic["dense16_hop"], synthetic["dense16_pos"]
    model = tbmodels_amd.Model.from_packed(r_vec, hop, pos=pos)
    r2, h2 = model.packed_hop()
    assert np.array_equal(r2, r_vec) and np.array_equal(h2, hop)
    model.set_sparse(True)
    r3, (r_ptr, row, col, val) = model.packed_hop()
    assert np.array_equal(r3, r_vec)
    dense = tbmodels_amd.synthetic.csr_to_dense(16, r_ptr, row, col, val)
    assert np.array_equal(dense, hop)
    assert np.array_equal(np.array(model.hop[tuple(r_vec[3])]), hop[3])  # __array__ of the CSR wrapper
    model.set_sparse(False)
    assert np.array_equal(model.packed_hop()[1], hop)


def test_contains_cc_reduction():
    """Full (+R and -R) input is halved onto the half-space; non-Hermitian input is rejected (:247-279)."""
    a = np.array([[0.1, 0.2j], [0.3, -0.1]])
    on = np.array([[1.0, 0.5], [0.5, -1.0]])
    model = tbmodels_amd.Model(hop={(1, 0): a, (-1, 0): a.conj().T, (0, 0): on}, size=2)
    assert set(model.hop) == {(1, 0), (0, 0)}
    assert np.allclose(model.hop[(1, 0)], a)
    assert np.allclose(model.hop[(0, 0)], on / 2)
    with pytest.raises(ValueError):
        tbmodels_amd.Model(hop={(1, 0): a, (-1, 0): a}, size=2)
    # contains_cc=False: negative R folds to +R conjugated
    model = tbmodels_amd.Model(hop={(-1, 0): a}, size=2, contains_cc=False)
    assert np.allclose(model.hop[(1, 0)], a.conj().T)


def test_positions_mapped_into_home_cell():
    """pos outside [0,1) moves hoppings between lattice vectors (:221-245)."""
    a = np.array([[0, 0.3], [0, 0]], dtype=complex)
    model = tbmodels_amd.Model(hop={(1,): a}, pos=[[0.25], [1.5]], contains_cc=False)
    assert np.allclose(model.pos, [[0.25], [0.5]])
    assert set(model.hop) == {(2,)}
    assert model.hop[(2,)][0, 1] == 0.3


def test_size_dim_inference_errors():
    with pytest.raises(ValueError):
        tbmodels_amd.Model()
    with pytest.raises(ValueError):
        tbmodels_amd.Model(size=2)
    with pytest.raises(ValueError):
        tbmodels_amd.Model(size=2, dim=2, pos=[[0, 0]])
    with pytest.raises(ValueError):
        tbmodels_amd.Model(on_site=(1, 2), dim=2, hop={(1, 0): np.ones((3, 3))}, contains_cc=False)
    with pytest.raises(ValueError):
        tbmodels_amd.Model.from_hop_list(hop_list=[(1.0, 0, 1, (1, 0))])
    model = tbmodels_amd.Model.from_hop_list(
        hop_list=[(1.0, 0, 1, (1, 0)), (0.5, 0, 1, (1, 0))], size=2, contains_cc=False
    )
    assert model.hop[(1, 0)][0, 1] == 1.5
    with pytest.raises(ValueError):
        model.add_hop(1.0, 0, 1, (1, 0, 0))
    with pytest.raises(ValueError):
        model.add_on_site((1, 2, 3))


@pytest.mark.parametrize("convention", ["a", "1", None, 3])
def test_invalid_convention_raises_before_any_device_work(convention):
    """tests/test_hamilton.py:35-42 of the reference; must hold on a machine without a GPU."""
    model = toy_model(0, 0.1)
    with pytest.raises(ValueError):
        model.hamilton((0, 0, 0), convention=convention)


def test_k_shape_mismatch_is_value_error():
    model = toy_model(0.1, 0.2)
    with pytest.raises(ValueError):
        model.hamilton((0.0, 0.0))
    with pytest.raises(ValueError):
        model.eigenval([[0.0, 0.0, 0.0, 0.0]])


def test_pickle_drops_device_state():
    """tests/test_pickle.py of the reference: models survive pickling (multiprocessing use)."""
    model = toy_model(0.1, 0.2, sparse=True)
    clone = pickle.loads(pickle.dumps(model))
    assert clone._handle is None
    assert set(clone.hop) == set(model.hop)
    for key in model.hop:
        assert np.array_equal(np.array(clone.hop[key]), np.array(model.hop[key]))
    clone.add_hop(0.1, 0, 1, (5, 0, 0))  # defaultdict factory still works after unpickling
    assert (5, 0, 0) in clone.hop


def test_fingerprint_tracks_in_place_mutation():
    model = toy_model(0.1, 0.2)
    before = model._fingerprint()
    assert before == model._fingerprint()
    model.hop[(0, 0, 0)][0, 0] += 1e-9
    assert model._fingerprint() != before
    before = model._fingerprint()
    model.add_hop(0.01, 0, 1, (3, 0, 0))
    assert model._fingerprint() != before
    before = model._fingerprint()
    model.set_sparse(True)
    assert model._fingerprint() != before


def test_staging_key_uses_edit_counter_until_a_matrix_leaves_the_model():
    """`_HopDict`: the model's own mutators are counted; any outside access switches to content hashing for good."""
    model = toy_model(0.1, 0.2)
    key = model._staging_key()
    assert key[0] == "version" and key == model._staging_key()
    assert len(model.hop) > 0 and (0, 0, 0) in model.hop and list(model.hop)  # keys / len / in: not an exposure
    repr(model), model.packed_hop(), model.to_hdf5()
    assert model._staging_key() == key
    model.add_hop(0.01, 0, 1, (3, 0, 0))
    assert model._staging_key()[0] == "version" and model._staging_key() != key
    key = model._staging_key()
    model.add_on_site([0.1, 0.2])
    assert model._staging_key()[0] == "version" and model._staging_key() != key
    key = model._staging_key()
    model.set_sparse(True)
    assert model._staging_key()[0] == "version" and model._staging_key() != key
    model.set_sparse(False)

    handed_out = model.hop[(0, 0, 0)]  # from here on a reference is out there
    key = model._staging_key()
    assert key[0] != "version" and key == model._fingerprint()
    assert model._staging_key() == key
    handed_out[0, 0] += 1e-9  # written through the reference, no dict access at all
    assert model._staging_key() != key

    for expose in (
        lambda m: m.hop.values(),
        lambda m: m.hop.items(),
        lambda m: m.hop.get((0, 0, 0)),
        lambda m: m.hop.__setitem__((9, 0, 0), np.zeros((2, 2), dtype=complex)),
        lambda m: m.hop.update({}),
        lambda m: m.hop.setdefault((0, 0, 0)),
        lambda m: m.hop.pop((0, 0, 0)),
        lambda m: m.hop[(7, 7, 7)],  # the defaultdict insertion
    ):
        fresh = toy_model(0.1, 0.2)
        assert fresh._staging_key()[0] == "version"
        expose(fresh)
        assert fresh._staging_key()[0] != "version"

    replaced = toy_model(0.1, 0.2)
    replaced.hop = {(0, 0, 0): np.eye(2, dtype=complex)}  # a plain dict: always content-hashed
    assert replaced._staging_key() == replaced._fingerprint()
    replaced.add_hop(0.5, 0, 1, (1, 0, 0))
    assert (1, 0, 0) in replaced.hop and replaced.packed_hop()[0].shape == (2, 3)


def test_pickle_and_copy_keep_the_source_unexposed():
    import copy

    model = toy_model(0.1, 0.2)
    for clone in (pickle.loads(pickle.dumps(model)), copy.deepcopy(model)):
        assert model._staging_key()[0] == "version"  # serialising did not expose the source
        assert clone._staging_key()[0] == "version"  # nobody holds references into the clone
        assert clone._fingerprint() == model._fingerprint()
        clone.add_hop(0.1, 0, 1, (5, 0, 0))
        assert clone._staging_key()[0] == "version" and (5, 0, 0) in clone.hop and (5, 0, 0) not in model.hop
    model.hop[(0, 0, 0)]
    clone = pickle.loads(pickle.dumps(model))
    assert clone._staging_key()[0] == "version" and model._staging_key()[0] != "version"


def test_every_route_that_hands_out_a_matrix_marks_the_dict_exposed():
    """``dict(hop)``, ``{**hop}``, ``copy.copy(hop)`` went through C-level dict fast paths that bypassed
    the exposure tracking, ``hop |= ...`` changed contents behind the edit counter, and ``copy.copy(model)`` reset
    the flags of the dict it SHARES with the original: each of them left the staged copy silently stale."""
    import copy

    def fresh():
        model = toy_model(0.1, 0.2)
        assert model._staging_key()[0] == "version"
        return model

    routes = {
        "dict()": lambda m: dict(m.hop),
        "**": lambda m: {**m.hop},
        "copy.copy(hop)": lambda m: copy.copy(m.hop),
        "dict.update(other, hop)": lambda m: {}.update(m.hop),
        "hop | {}": lambda m: m.hop | {},
        "{} | hop": lambda m: {} | m.hop,
        "list(hop.values())": lambda m: list(m.hop.values()),
    }
    for name, route in routes.items():
        model = fresh()
        route(model)
        assert model._staging_key() == model._fingerprint(), name  # content-hashed from now on
    # a matrix obtained through one of them, edited in place, changes the key
    model = fresh()
    handed_out = dict(model.hop)
    before = model._staging_key()
    next(iter(handed_out.values()))[0, 0] += 1.0
    assert model._staging_key() != before
    # |= changes the contents: the key must change
    model = fresh()
    before = model._staging_key()
    model.hop |= {(7, 0, 0): np.eye(2, dtype=complex)}
    assert model._staging_key() != before and (7, 0, 0) in model.hop
    # keys-only access stays free
    model = fresh()
    assert sorted(model.hop) and list(model.hop.keys()) and len(model.hop) and (0, 0, 0) in model.hop
    assert model._staging_key()[0] == "version"
    # copy.copy(model) shares the dict: a reference handed out by the original stays tracked in both
    model = fresh()
    mat = model.hop[(0, 0, 0)]
    clone = copy.copy(model)
    assert clone.hop is model.hop
    before = clone._staging_key()
    assert before == clone._fingerprint()
    mat[0, 0] += 1.0
    assert clone._staging_key() != before and model._staging_key() == clone._staging_key()


def test_kdotp_model_revalidates_its_staged_copy():
    """``KdotpModel.taylor_coefficients`` is a public dict the reference reads on every call (kdotp.py:51-82): the
    staging key follows edits of the matrices, of the key set and of ``device``; HDF5 round trip."""
    from tbmodels_amd import io

    kp = tbmodels_amd.KdotpModel({(0, 0, 1): np.eye(2), (1, 0, 0): np.array([[0, 1j], [-1j, 0]])})
    key = kp._staging_key()
    assert kp._staging_key() == key
    kp.taylor_coefficients[(0, 0, 1)][0, 0] = 2.0
    key2 = kp._staging_key()
    assert key2 != key
    kp.taylor_coefficients[(2, 0, 0)] = np.eye(2, dtype=complex)
    key3 = kp._staging_key()
    assert key3 != key2
    kp.device = 1
    assert kp._staging_key() != key3
    clone = pickle.loads(pickle.dumps(kp))
    assert clone._handle is None and clone._staged_key is None
    assert clone._staging_key() == kp._staging_key()
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "kp.hdf5")
        io.save(kp, path)
        loaded = io.load(path)
    assert isinstance(loaded, tbmodels_amd.KdotpModel)
    assert list(loaded.taylor_coefficients) == list(kp.taylor_coefficients)
    for power, mat in kp.taylor_coefficients.items():
        assert np.array_equal(loaded.taylor_coefficients[power], mat)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads and exports exactly what include/tbk.h declares."""
    with open(os.path.join(ROOT, "include", "tbk.h")) as handle:
        header = handle.read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(tbk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    assert _lib.lib().tbk_version().startswith(b"tbk")


def test_no_cpu_fallback_without_device():
    """Without a GPU the compute entry points raise; nothing silently computes on the host."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    model = toy_model(0.1, 0.2)
    with pytest.raises(RuntimeError):
        model.hamilton((0, 0, 0))
    with pytest.raises(RuntimeError):
        model.eigenval(KPT)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(ROOT, "tbmodels_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(dirpath, name)) as handle:
                    text = handle.read()
                assert "import oracle" not in text and "from oracle" not in text, name


def test_c_abi_rejects_bad_arguments_before_touching_a_device():
    """Argument validation of the create calls comes first (no GPU needed): status TBK_ERR_ARGUMENT and a message in
    tbk_last_error(); the Python wrapper turns that status into ValueError."""
    lib = _lib.lib()
    handle = ctypes.c_void_p()
    r_vec = np.zeros((1, 3), dtype=np.int32)
    hop = np.zeros((1, 2, 2), dtype=np.complex128)

    def last():
        return lib.tbk_last_error().decode()

    create_dense = lib.tbk_model_create_dense
    assert create_dense(0, 0, 2, 1, _lib.ptr(r_vec), _lib.ptr(hop), ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    assert "dim" in last()
    assert create_dense(0, 3, 0, 1, _lib.ptr(r_vec), _lib.ptr(hop), ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    assert "n_orb" in last()
    assert create_dense(0, 3, 2, -1, _lib.ptr(r_vec), _lib.ptr(hop), ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    assert create_dense(0, 3, 2, 1, _lib.ptr(r_vec), None, ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    assert "hop" in last()
    assert create_dense(0, 3, 2, 1, None, _lib.ptr(hop), ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    assert create_dense(0, 3, 2, 1, _lib.ptr(r_vec), _lib.ptr(hop), None) == _lib.TBK_ERR_ARGUMENT

    create_csr = lib.tbk_model_create_csr
    r_ptr = np.array([0, 2], dtype=np.int64)
    row = np.array([0, 1], dtype=np.int32)
    col = np.array([1, 5], dtype=np.int32)  # 5 is outside a 2 x 2 block
    val = np.ones(2, dtype=np.complex128)
    args = (0, 3, 2, 1, _lib.ptr(r_vec))
    assert create_csr(*args, _lib.ptr(r_ptr), _lib.ptr(row), _lib.ptr(col), _lib.ptr(val),
                      ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    assert "out of range" in last()
    assert create_csr(*args, None, _lib.ptr(row), _lib.ptr(col), _lib.ptr(val),
                      ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    bad_ptr = np.array([2, 0], dtype=np.int64)
    assert create_csr(*args, _lib.ptr(bad_ptr), _lib.ptr(row), _lib.ptr(col), _lib.ptr(val),
                      ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    col[1] = 1
    assert create_csr(*args, _lib.ptr(r_ptr), _lib.ptr(row), _lib.ptr(col), None,
                      ctypes.byref(handle)) == _lib.TBK_ERR_ARGUMENT
    assert handle.value is None
    with pytest.raises(ValueError):
        _lib.check(create_dense(0, 0, 2, 1, _lib.ptr(r_vec), _lib.ptr(hop), ctypes.byref(handle)))
    # calls on a missing handle, and a communicator with an impossible rank
    k = np.zeros((1, 3))
    out = np.zeros((1, 2, 2), dtype=np.complex128)
    assert lib.tbk_hamilton(None, _lib.ptr(k), 1, 2, None, _lib.ptr(out)) == _lib.TBK_ERR_ARGUMENT
    assert lib.tbk_eigenval(None, _lib.ptr(k), 1, _lib.ptr(out)) == _lib.TBK_ERR_ARGUMENT
    assert lib.tbk_model_set_option(None, _lib.TBK_OPT_TIMING, 1) == _lib.TBK_ERR_ARGUMENT
    # the several-devices entry points: no handle array, an empty one, a NULL entry
    assert lib.tbk_eigenval_multi(None, 1, _lib.ptr(k), 1, _lib.ptr(out)) == _lib.TBK_ERR_ARGUMENT
    assert "handles" in last()
    holes = (ctypes.c_void_p * 2)(None, None)
    assert lib.tbk_eigenval_multi(holes, 0, _lib.ptr(k), 1, _lib.ptr(out)) == _lib.TBK_ERR_ARGUMENT
    assert lib.tbk_eigenval_multi(holes, 2, _lib.ptr(k), 1, _lib.ptr(out)) == _lib.TBK_ERR_ARGUMENT
    assert "NULL" in last()
    assert lib.tbk_hamilton_multi(holes, 2, _lib.ptr(k), 1, 2, None, _lib.ptr(out)) == _lib.TBK_ERR_ARGUMENT
    comm = ctypes.c_void_p()
    uid = np.zeros(128, dtype=np.uint8)
    assert lib.tbk_comm_create(0, 2, 2, _lib.ptr(uid), ctypes.byref(comm)) == _lib.TBK_ERR_ARGUMENT
    assert lib.tbk_comm_create(0, 2, 0, None, ctypes.byref(comm)) == _lib.TBK_ERR_ARGUMENT
    lib.tbk_model_destroy(None)  # a no-op, like free(NULL)
    if _lib.device_count() == 0:  # valid arguments, no device: the loud failure, not a host computation
        assert create_dense(0, 3, 2, 1, _lib.ptr(r_vec), _lib.ptr(hop), ctypes.byref(handle)) == _lib.TBK_ERR_DEVICE
        assert "no CPU path" in last()


def test_device_list_of_a_model(monkeypatch):
    """``Model.devices``: default from TBK_DEVICES / TBK_DEVICE, assignment re-stages, ``device`` is its first entry
    (several GPUs behind the unchanged methods: INTEGRATION.md section 4.1)."""
    import pickle

    from tbmodels_amd import _model

    monkeypatch.delenv("TBK_DEVICES", raising=False)
    monkeypatch.delenv("TBK_DEVICE", raising=False)
    assert _model._devices_from_env() == [0]
    monkeypatch.setenv("TBK_DEVICE", "3")
    assert _model._devices_from_env() == [3]
    monkeypatch.setenv("TBK_DEVICES", "0, 2,5")
    assert _model._devices_from_env() == [0, 2, 5]
    model = toy_model(0.2, -0.2)
    assert model.devices == [0, 2, 5] and model.device == 0
    key = model._staging_key()
    model.device = 4
    assert model.devices == [4] and model._staging_key() != key  # the staged copies belong to a device list
    model.devices = (1, 1)
    assert model.devices == [1, 1]
    with pytest.raises(ValueError):
        model.devices = []
    with pytest.raises(ValueError):
        model.devices = [0, -1]
    clone = pickle.loads(pickle.dumps(model))
    assert clone.devices == [1, 1] and clone._handles == []
    # states pickled before the device list existed
    state = model.__getstate__()
    state.pop("_devices")
    state["device"] = 2
    state["_handle"] = None
    old = _model.Model.__new__(_model.Model)
    old.__setstate__(state)
    assert old.devices == [2] and old._handles == []
