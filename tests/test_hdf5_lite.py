"""
HDF5 wire formats: the reader / writer of ``tbmodels_amd.hdf5_lite`` and the
``io`` / ``Model`` methods on top, checked against

* the reference's own sample files (``tests/samples/cli_eigenvals/*`` of the reference, committed as data
  fixtures under ``tests/golden/cli_eigenvals``),
* real ``h5py`` / the reference's own ``Model.from_hdf5_file`` / ``to_hdf5`` where an interpreter with h5py
  exists (``tools/hdf5_crosscheck.py`` as a child process; skipped otherwise -- the GPU box has none).
"""

import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

import tbmodels_amd
from tbmodels_amd import hdf5_lite, io
from conftest import GOLDEN, ROOT

SAMPLES = os.path.join(GOLDEN, "cli_eigenvals")
H5PY_PYTHON = "/opt/conda/bin/python3.9"
CROSSCHECK = os.path.join(ROOT, "tools", "hdf5_crosscheck.py")


def _have_h5py():
    if not os.path.exists(H5PY_PYTHON):
        return False
    return subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True, check=False).returncode == 0


needs_h5py = pytest.mark.skipif(not _have_h5py(), reason="no interpreter with h5py on this machine")
needs_reference = pytest.mark.skipif(
    not (_have_h5py() and os.path.isdir("/root/reference/src/tbmodels")), reason="reference sources not present"
)


def crosscheck(*args):
    run = subprocess.run([H5PY_PYTHON, CROSSCHECK, *args], capture_output=True, text=True, check=False)
    assert run.returncode == 0, run.stderr[-2000:]


def flatten(tree, prefix=""):
    out = {}
    for name, value in tree.items():
        if isinstance(value, dict):
            out.update(flatten(value, prefix + name + "|"))
        else:
            out[prefix + name] = np.asarray(value)
    return out


# ------------------------------------------------------------------------------------------------
# the reference's sample files
# ------------------------------------------------------------------------------------------------
def test_read_reference_model_file(silicon):
    tree = hdf5_lite.read(os.path.join(SAMPLES, "silicon_model.hdf5"))
    assert tree["type_tag"] == "tbmodels.model"
    assert tree["size"] == 8 and tree["dim"] == 3 and tree["sparse"] is np.False_ or tree["sparse"] == False  # noqa: E712
    assert len(tree["hop"]) == 95
    hop = {tuple(entry["R"].tolist()): entry["mat"] for entry in tree["hop"].values()}
    for r_vec, mat in zip(silicon["R"], silicon["hop"]):
        np.testing.assert_array_equal(hop[tuple(r_vec.tolist())], mat)
    np.testing.assert_array_equal(tree["pos"], silicon["pos"])
    np.testing.assert_array_equal(tree["uc"], silicon["uc"])


def test_read_reference_kpoints_and_eigenvals(silicon):
    kpts = io.load(os.path.join(SAMPLES, "kpoints.hdf5"))
    assert isinstance(kpts, io.KpointsExplicit)
    np.testing.assert_array_equal(kpts.kpoints, silicon["known_kpoints"])
    eig = io.load(os.path.join(SAMPLES, "silicon_eigenvals.hdf5"))
    assert isinstance(eig, io.EigenvalsData)
    np.testing.assert_array_equal(eig.eigenvals, silicon["known_eigenvals"])
    np.testing.assert_array_equal(eig.kpoints.kpoints, silicon["known_kpoints"])


def test_model_from_reference_file(silicon):
    model = tbmodels_amd.Model.from_hdf5_file(os.path.join(SAMPLES, "silicon_model.hdf5"))
    assert model.size == 8 and model.dim == 3 and not model._sparse and model.occ is None
    assert len(model.hop) == 95
    for r_vec, mat in zip(silicon["R"], silicon["hop"]):
        np.testing.assert_array_equal(np.array(model.hop[tuple(r_vec.tolist())]), mat)
    loaded = io.load(os.path.join(SAMPLES, "silicon_model.hdf5"))
    assert isinstance(loaded, tbmodels_amd.Model) and len(loaded.hop) == 95


# ------------------------------------------------------------------------------------------------
# writer -> reader
# ------------------------------------------------------------------------------------------------
def sample_tree(n_hop=20):
    rng = np.random.default_rng(3)
    return {
        "type_tag": "tbmodels.model",
        "size": np.int64(4),
        "dim": np.int64(3),
        "sparse": False,
        "yes": True,
        "flags": np.array([True, False, True]),
        "uc": 2.5 * np.eye(3),
        "pos": rng.random((4, 3)),
        "f32": rng.random(5).astype(np.float32),
        "i32": np.arange(6, dtype=np.int32).reshape(2, 3),
        "c64": (rng.random(3) + 1j * rng.random(3)).astype(np.complex64),
        "empty": np.zeros((0, 3)),
        "unicode": "grüße",
        "hop": {
            str(i): {"R": np.array([i, -i, 2 * i]), "mat": rng.random((4, 4)) + 1j * rng.random((4, 4))}
            for i in range(n_hop)
        },
        "nothing": {},
    }


def assert_trees_equal(a, b):
    assert set(a) == set(b)
    for key, value in a.items():
        if isinstance(value, dict):
            assert_trees_equal(value, b[key])
        elif isinstance(value, str):
            assert b[key] == value
        else:
            got = np.asarray(b[key])
            want = np.asarray(value)
            assert got.shape == want.shape and got.dtype == want.dtype, key
            np.testing.assert_array_equal(got, want)


def test_write_read_roundtrip(tmp_path):
    tree = sample_tree()
    path = tmp_path / "t.hdf5"
    hdf5_lite.write(path, tree)
    assert_trees_equal(tree, hdf5_lite.read(path))


@pytest.mark.parametrize("leaf_k,internal_k,n", [(2, 2, 3), (2, 2, 4), (2, 2, 5), (2, 2, 16), (2, 2, 17), (2, 2, 70), (4, 2, 200)])
def test_group_btree_levels(tmp_path, monkeypatch, leaf_k, internal_k, n):
    """Several symbol-table nodes and B-tree levels (small K forces them)."""
    monkeypatch.setattr(hdf5_lite, "_LEAF_K", leaf_k)
    monkeypatch.setattr(hdf5_lite, "_INTERNAL_K", internal_k)
    tree = {"g": {"name%d" % i: np.int64(i) for i in range(n)}, "type_tag": "x"}
    path = tmp_path / "t.hdf5"
    hdf5_lite.write(path, tree)
    back = hdf5_lite.read(path)
    assert_trees_equal(tree, back)
    assert list(back["g"]) == sorted(tree["g"])  # symbol tables are name-ordered


def test_large_group_default_k(tmp_path):
    tree = {"hop": {str(i): {"R": np.array([i, 0, 0])} for i in range(1500)}}
    path = tmp_path / "t.hdf5"
    hdf5_lite.write(path, tree)
    back = hdf5_lite.read(path)
    assert len(back["hop"]) == 1500 and back["hop"]["1499"]["R"][0] == 1499


def test_not_hdf5(tmp_path):
    path = tmp_path / "x.hdf5"
    path.write_bytes(b"this is not an HDF5 file" * 10)
    with pytest.raises(hdf5_lite.HDF5FormatError):
        hdf5_lite.read(path)
    with pytest.raises(TypeError):
        hdf5_lite.write(path, [1, 2, 3])
    with pytest.raises(TypeError):
        hdf5_lite.write(path, {"x": np.array(["a", "b"], dtype=object)})


# ------------------------------------------------------------------------------------------------
# io / Model on top
# ------------------------------------------------------------------------------------------------
def toy_model(sparse=False, **kwargs):
    hop = {
        (0, 0, 0): np.array([[0.5, 0.1 + 0.2j], [0.1 - 0.2j, -0.5]]),
        (1, 0, 0): np.array([[0.0, 0.3], [0.2j, 0.0]]),
        (0, 1, -1): np.array([[0.1, 0.0], [0.0, 0.4 - 0.1j]]),
    }
    return tbmodels_amd.Model(hop=hop, contains_cc=False, sparse=sparse, **kwargs)


def models_equal(m1, m2):
    assert m1.size == m2.size and m1.dim == m2.dim and m1.occ == m2.occ and m1._sparse == m2._sparse
    np.testing.assert_array_equal(m1.pos, m2.pos)
    if m1.uc is None:
        assert m2.uc is None
    else:
        np.testing.assert_array_equal(m1.uc, m2.uc)
    assert set(m1.hop) == set(m2.hop)
    for key in m1.hop:
        np.testing.assert_array_equal(np.array(m1.hop[key]), np.array(m2.hop[key]))


@pytest.mark.parametrize("sparse", [False, True])
@pytest.mark.parametrize(
    "kwargs", [dict(), dict(pos=None, dim=3), dict(uc=3 * np.eye(3)), dict(pos=np.zeros((2, 3)), uc=np.eye(3), occ=1)]
)
def test_model_file_consistency(tmp_path, kwargs, sparse):
    """tests/test_hdf5.py:22-45 of the reference: save -> load gives the same model (method and free function)."""
    model = toy_model(sparse=sparse, **kwargs)
    path = tmp_path / "m.hdf5"
    model.to_hdf5_file(path)
    models_equal(model, tbmodels_amd.Model.from_hdf5_file(path))
    io.save(model, path)
    models_equal(model, io.load(path))
    # explicit keywords take precedence over the file's
    assert tbmodels_amd.Model.from_hdf5_file(path, occ=7).occ == 7


def test_legacy_file_warns(tmp_path):
    tree = toy_model().to_hdf5()
    del tree["type_tag"]
    path = tmp_path / "legacy.hdf5"
    hdf5_lite.write(path, tree)
    with pytest.deprecated_call():
        model = io.load(path)
    models_equal(toy_model(), model)
    with pytest.deprecated_call():
        tbmodels_amd.Model.from_hdf5_file(path)
    hdf5_lite.write(path, {"type_tag": "who.knows", "x": np.int64(1)})
    with pytest.raises(ValueError):
        io.load(path)


def test_eigenvals_data_roundtrip(tmp_path):
    kpts = np.random.default_rng(0).random((7, 3))
    data = io.EigenvalsData.from_eigenval_function(
        kpoints=kpts, eigenval_function=lambda k: np.cumsum(np.asarray(k), axis=-1), listable=True
    )
    np.testing.assert_allclose(data.eigenvals, np.cumsum(kpts, axis=-1))
    single = io.EigenvalsData.from_eigenval_function(kpoints=kpts, eigenval_function=np.cumsum)
    np.testing.assert_allclose(single.eigenvals, data.eigenvals)
    path = tmp_path / "e.hdf5"
    io.save(data, path)
    back = io.load(path)
    np.testing.assert_array_equal(back.eigenvals, data.eigenvals)
    np.testing.assert_array_equal(back.kpoints.kpoints, kpts)
    with pytest.raises(ValueError):
        io.EigenvalsData(kpoints=kpts, eigenvals=np.zeros((3, 2)))


def test_cli_arguments(tmp_path, capsys):
    from tbmodels_amd._cli import main

    with pytest.raises(SystemExit):
        main(["eigenvals", "--help"])
    assert "--kpoints" in capsys.readouterr().out
    with pytest.raises(SystemExit):
        main(["no_such_command"])
    with pytest.raises(FileNotFoundError):
        main(["eigenvals", "-i", str(tmp_path / "missing.hdf5")])


# ------------------------------------------------------------------------------------------------
# against real h5py / the reference
# ------------------------------------------------------------------------------------------------
@needs_h5py
def test_h5py_reads_our_files(tmp_path):
    tree = sample_tree(n_hop=700)
    path = tmp_path / "ours.hdf5"
    hdf5_lite.write(path, tree)
    dump = tmp_path / "dump.npz"
    crosscheck("h5py-read", str(path), str(dump))
    want = flatten(tree)
    with np.load(dump) as got:
        assert set(got.files) == set(want)
        for key, value in want.items():
            if value.dtype.kind == "U":
                assert str(got[key]) == str(value)
            else:
                assert got[key].dtype == value.dtype and got[key].shape == value.shape, key
                np.testing.assert_array_equal(got[key], value)


@needs_h5py
def test_we_read_h5py_files(tmp_path):
    path = tmp_path / "theirs.hdf5"
    dump = tmp_path / "dump.npz"
    crosscheck("h5py-write", str(path))
    crosscheck("h5py-read", str(path), str(dump))
    got = flatten(hdf5_lite.read(path))
    with np.load(dump) as want:
        assert set(want.files) == set(got)
        for key in want.files:
            if want[key].dtype.kind in "US":
                assert str(got[key]) == (want[key].item().decode() if want[key].dtype.kind == "S" else str(want[key]))
            else:
                assert got[key].dtype == want[key].dtype and got[key].shape == want[key].shape, key
                np.testing.assert_array_equal(got[key], want[key])
    assert hdf5_lite.read(path)["empty_group"] == {}


@needs_h5py
def test_unsupported_features_are_named(tmp_path):
    path = tmp_path / "chunked.hdf5"
    crosscheck("h5py-chunked", str(path))
    with pytest.raises(hdf5_lite.HDF5FormatError, match="chunked|filtered"):
        hdf5_lite.read(path)


@needs_reference
@pytest.mark.parametrize("sparse", [False, True])
def test_reference_reads_our_model_files(tmp_path, silicon, sparse):
    """Our ``to_hdf5_file`` -> the reference's ``Model.from_hdf5_file`` (real h5py) gives the same model."""
    hop = {tuple(r.tolist()): mat for r, mat in zip(silicon["R"], silicon["hop"])}
    model = tbmodels_amd.Model(hop=hop, pos=silicon["pos"], uc=silicon["uc"], occ=4, contains_cc=False, sparse=sparse)
    path = tmp_path / "ours.hdf5"
    model.to_hdf5_file(path)
    dump = tmp_path / "ref.npz"
    crosscheck("ref-read", str(path), str(dump))
    with np.load(dump) as ref:
        assert bool(ref["sparse"]) == sparse and int(ref["occ"]) == 4 and int(ref["size"]) == 8
        got = {tuple(r.tolist()): mat for r, mat in zip(ref["R"], ref["hop"])}
        assert set(got) == set(model.hop)
        for key in got:
            np.testing.assert_array_equal(got[key], np.array(model.hop[key]))
        np.testing.assert_array_equal(ref["pos"], model.pos)
        np.testing.assert_array_equal(ref["uc"], model.uc)


@needs_reference
@pytest.mark.parametrize("sparse", [False, True])
def test_we_read_reference_written_model_files(tmp_path, silicon, sparse):
    """The reference's ``Model.to_hdf5`` (real h5py) -> our ``Model.from_hdf5_file``."""
    src = tmp_path / "in.npz"
    np.savez(src, R=silicon["R"], hop=silicon["hop"], pos=silicon["pos"], uc=silicon["uc"], occ=4, sparse=sparse)
    path = tmp_path / "theirs.hdf5"
    crosscheck("ref-write", str(src), str(path))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model = tbmodels_amd.Model.from_hdf5_file(path)
    assert model._sparse == sparse and model.occ == 4
    for r_vec, mat in zip(silicon["R"], silicon["hop"]):
        np.testing.assert_array_equal(np.array(model.hop[tuple(r_vec.tolist())]), mat)
