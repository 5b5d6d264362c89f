"""
Real-material sample data of the reference's own test-suite beyond silicon (fixtures and expected values by
``tools/gen_golden.py materials``, which imports the reference):

* bismuth Wannier90 output (hr + wsvec + xyz + win, both ``pos_kind``s), ``wannier90_hr.dat`` with and without
  stray empty lines, the two inconsistent ``_hr.dat`` files and the two broken ``_wsvec.dat`` files
  (``tests/test_wannier.py:18-345`` of the reference);
* InAs HDF5 models: the 14-orbital primitive cell and its (1, 2, 3) supercell -- 84 orbitals, CSR, a folded and
  therefore highly degenerate spectrum, through the streaming reduction + bisection (``tests/test_supercell.py``).

CPU part: parsers / HDF5 reader / oracle against the reference's results.  GPU part (``-m gpu``): the library.
"""

import gzip
import itertools
import os
import shutil

import numpy as np
import pytest

import tbmodels_amd
from tbmodels_amd import wannier
from oracle import tbk_oracle as oracle
from conftest import GOLDEN, KPT, load_golden

WANNIER = [
    "bi_hr.dat", "bi_wsvec.dat", "bi_centres.xyz", "bi.win", "bi_equivalent.win", "bi_wsvec_blocks_missing.dat",
    "bi_wsvec_blocks_incomplete.dat", "wannier90_hr.dat", "wannier90_hr_v2.dat", "wannier90_inconsistent.dat",
    "wannier90_inconsistent_v2.dat",
]
MODELS = ["InAs_nosym.hdf5", "InAs_supercell_reference.hdf5"]


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    out = tmp_path_factory.mktemp("materials")
    paths = {}
    for sub, names in (("wannier", WANNIER), ("models", MODELS)):
        for name in names:
            with gzip.open(os.path.join(GOLDEN, sub, name + ".gz"), "rb") as src, open(out / name, "wb") as dst:
                shutil.copyfileobj(src, dst)
            paths[name] = str(out / name)
    return paths


@pytest.fixture(scope="module")
def gold():
    return load_golden("materials")


def sorted_packed(model):
    r_vec, payload = model.packed_hop()
    if model._sparse:
        hop = tbmodels_amd.synthetic.csr_to_dense(model.size, *payload)
    else:
        hop = payload
    order = np.lexsort(r_vec.T[::-1])
    return r_vec[order], hop[order]


def check_model(model, gold, tag):
    r_vec, hop = sorted_packed(model)
    assert np.array_equal(r_vec, gold[tag + "_R"])
    assert np.abs(hop - gold[tag + "_hop"]).max() < 1e-14
    assert np.abs(model.pos - gold[tag + "_pos"]).max() < 1e-12
    if tag + "_uc" in gold:
        assert np.abs(model.uc - gold[tag + "_uc"]).max() < 1e-14


def bi_all(files, kind):
    return tbmodels_amd.Model.from_wannier_files(
        hr_file=files["bi_hr.dat"], wsvec_file=files["bi_wsvec.dat"], xyz_file=files["bi_centres.xyz"],
        win_file=files["bi.win"], pos_kind=kind, distance_ratio_threshold=1.0,
    )


# ------------------------------------------------------------------------------------------------
# host side
# ------------------------------------------------------------------------------------------------
def test_bismuth_hr_wsvec(files, gold):
    model = tbmodels_amd.Model.from_wannier_files(hr_file=files["bi_hr.dat"], wsvec_file=files["bi_wsvec.dat"])
    check_model(model, gold, "bi_hr_ws")
    r_vec, hop = model.packed_hop()
    assert np.abs(oracle.hamilton(r_vec, hop, KPT) - gold["bi_hr_ws_h2"]).max() < 1e-12


def test_bismuth_all_files(files, gold):
    model = bi_all(files, "wannier")
    check_model(model, gold, "bi_all_wannier")
    nearest = bi_all(files, "nearest_atom")
    assert np.abs(nearest.pos - gold["bi_all_nearest_atom_pos"]).max() < 1e-12
    assert np.abs(nearest.uc - gold["bi_all_nearest_atom_uc"]).max() < 1e-14
    # tests/test_wannier.py:160-178: two distinct atomic sites
    assert len({tuple(np.round(p, 5)) for p in nearest.pos}) == 2
    with pytest.raises(ValueError):  # no unit cell: reduced positions cannot be computed (:49-70)
        tbmodels_amd.Model.from_wannier_files(
            hr_file=files["bi_hr.dat"], wsvec_file=files["bi_wsvec.dat"], xyz_file=files["bi_centres.xyz"],
            distance_ratio_threshold=1.0,
        )


def test_length_unit_modes(files):
    """tests/test_wannier.py:355-373: the length unit given explicitly or inside the unit-cell block is equivalent."""
    kwargs = dict(hr_file=files["bi_hr.dat"], xyz_file=files["bi_centres.xyz"], wsvec_file=files["bi_wsvec.dat"])
    model1 = tbmodels_amd.Model.from_wannier_files(win_file=files["bi.win"], **kwargs)
    model2 = tbmodels_amd.Model.from_wannier_files(win_file=files["bi_equivalent.win"], **kwargs)
    assert np.allclose(model1.uc, model2.uc, rtol=0, atol=1e-10) and np.allclose(model1.pos, model2.pos, rtol=0, atol=1e-10)
    assert set(model1.hop) == set(model2.hop)
    for key in model1.hop:
        assert np.array_equal(np.array(model1.hop[key]), np.array(model2.hop[key]))


def test_wannier90_hr_with_and_without_empty_lines(files, gold):
    model1 = tbmodels_amd.Model.from_wannier_files(hr_file=files["wannier90_hr.dat"], occ=28)
    model2 = tbmodels_amd.Model.from_wannier_files(hr_file=files["wannier90_hr_v2.dat"], occ=28)
    check_model(model1, gold, "w90")
    assert model1.occ == 28 and model1.size == 7
    assert set(model1.hop) == set(model2.hop)
    for key in model1.hop:
        assert (np.array(model1.hop[key]) == np.array(model2.hop[key])).all()


@pytest.mark.parametrize("name", ["wannier90_inconsistent.dat", "wannier90_inconsistent_v2.dat"])
def test_inconsistent_hr_files(files, name):
    with pytest.raises(ValueError):
        tbmodels_amd.Model.from_wannier_files(hr_file=files[name])


def test_broken_wsvec_files(files, tmp_path):
    kwargs = dict(hr_file=files["bi_hr.dat"], xyz_file=files["bi_centres.xyz"], win_file=files["bi.win"])
    with pytest.raises(KeyError):
        tbmodels_amd.Model.from_wannier_files(wsvec_file=files["bi_wsvec_blocks_missing.dat"], **kwargs)
    with pytest.raises(wannier.WannierParseError, match="Incomplete wsvec iterator."):
        tbmodels_amd.Model.from_wannier_files(wsvec_file=files["bi_wsvec_blocks_incomplete.dat"], **kwargs)
    empty = tmp_path / "empty_wsvec.dat"
    empty.write_text("")
    with pytest.raises(wannier.WannierParseError, match="The 'wsvec' iterator is empty."):
        tbmodels_amd.Model.from_wannier_files(wsvec_file=str(empty), **kwargs)


def test_inas_models_from_hdf5(files, gold):
    prim = tbmodels_amd.Model.from_hdf5_file(files["InAs_nosym.hdf5"])
    assert (prim.size, prim.dim, prim.occ, prim._sparse) == (14, 3, 6, False) and len(prim.hop) == 140
    check_model(prim, gold, "inas")
    r_vec, hop = prim.packed_hop()
    assert np.abs(np.array(oracle.eigenval(r_vec, hop, gold["kpath"])) - gold["inas_eig"]).max() < 1e-12
    sup = tbmodels_amd.Model.from_hdf5_file(files["InAs_supercell_reference.hdf5"])
    assert (sup.size, sup.dim, sup.occ, sup._sparse) == (84, 3, 36, True) and len(sup.hop) == 56
    assert sum(m.count_nonzero() for m in sup.hop.values()) == int(gold["inas_sc_nnz"])
    assert np.abs(sup.pos - gold["inas_sc_pos"]).max() < 1e-12
    r_vec, payload = sup.packed_hop()
    dense = tbmodels_amd.synthetic.csr_to_dense(84, *payload)
    assert np.abs(np.array(oracle.eigenval(r_vec, dense, gold["kpath"][:3])) - gold["inas_sc_eig"][:3]).max() < 1e-12


# ------------------------------------------------------------------------------------------------
# the library
# ------------------------------------------------------------------------------------------------
def close(got, want, tol=1e-10):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= tol


@pytest.mark.gpu
def test_gpu_bismuth(files, gold):
    for tag, model in (
        ("bi_hr_ws", tbmodels_amd.Model.from_wannier_files(hr_file=files["bi_hr.dat"], wsvec_file=files["bi_wsvec.dat"])),
        ("bi_all_wannier", bi_all(files, "wannier")),
        ("bi_all_nearest_atom", bi_all(files, "nearest_atom")),
    ):
        close(model.hamilton(KPT), gold[tag + "_h2"])
        close(model.hamilton(KPT, convention=1), gold[tag + "_h1"])
        close(np.array(model.eigenval(KPT)), gold[tag + "_eig"])
    model = bi_all(files, "wannier")
    model.set_sparse(True)
    close(model.hamilton(KPT, convention=1), gold["bi_all_wannier_h1"])


@pytest.mark.gpu
def test_gpu_wannier90_hr(files, gold):
    model = tbmodels_amd.Model.from_wannier_files(hr_file=files["wannier90_hr.dat"], occ=28)
    close(model.hamilton(KPT), gold["w90_h2"])
    close(np.array(model.eigenval(KPT)), gold["w90_eig"])


@pytest.mark.gpu
def test_gpu_inas_primitive_and_supercell(files, gold):
    kpath = gold["kpath"]
    prim = tbmodels_amd.Model.from_hdf5_file(files["InAs_nosym.hdf5"])
    close(prim.hamilton(kpath), gold["inas_h2"])
    close(prim.hamilton(kpath, convention=1), gold["inas_h1"])
    close(np.array(prim.eigenval(kpath)), gold["inas_eig"])
    sup = tbmodels_amd.Model.from_hdf5_file(files["InAs_supercell_reference.hdf5"])
    close(sup.hamilton(kpath), gold["inas_sc_h2"])
    close(sup.hamilton(kpath, convention=1), gold["inas_sc_h1"])
    eig = np.array(sup.eigenval(kpath))  # CSR kernel + streaming reduction + bisection at n = 84
    close(eig, gold["inas_sc_eig"])
    # the same through the dense contraction and through rocSOLVER
    sup.set_sparse(False)
    close(np.array(sup.eigenval(kpath)), gold["inas_sc_eig"])
    from tbmodels_amd import _lib

    sup.set_option(_lib.TBK_OPT_EIGENSOLVER, _lib.TBK_EIG_ROCSOLVER)
    close(np.array(sup.eigenval(kpath)), gold["inas_sc_eig"])


@pytest.mark.gpu
def test_gpu_supercell_spectrum_is_the_folded_spectrum(files):
    """tests/test_supercell.py:24-46 as a property of OUR two evaluations: eigenvalues of the (1, 2, 3) supercell at k
    are the union of the primitive cell's at the 6 equivalent k-points (atol 1e-7 there; the models are stored
    independently, so the agreement is limited by the file's own hopping cut-offs)."""
    prim = tbmodels_amd.Model.from_hdf5_file(files["InAs_nosym.hdf5"])
    sup = tbmodels_amd.Model.from_hdf5_file(files["InAs_supercell_reference.hdf5"])
    size = (1, 2, 3)
    rng = np.random.default_rng(5)
    k_sc = np.vstack([rng.random((40, 3)), [[0.0, 0.0, 0.0], [0.5, 0.5, 0.5]]])
    eig_sc = np.array(sup.eigenval(k_sc))
    shifts = np.array(list(itertools.product(*[range(s) for s in size])), dtype=float)
    k_eq = ((k_sc[:, None, :] + shifts[None, :, :]) / np.array(size, dtype=float)).reshape(-1, 3)
    folded = np.sort(np.array(prim.eigenval(k_eq)).reshape(len(k_sc), -1), axis=1)
    assert folded.shape == eig_sc.shape
    assert np.abs(folded - eig_sc).max() < 1e-7
