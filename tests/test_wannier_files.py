"""
Model.from_wannier_files on the reference's own silicon sample files against what
the reference builds from them (tests/golden/wannier.npz, tools/gen_golden.py).  Host-side parsing: no GPU.
"""

import gzip
import os
import shutil

import numpy as np
import pytest

import tbmodels_amd
from tbmodels_amd import wannier

from conftest import GOLDEN, KPT, load_golden

FILES = ["silicon_hr.dat", "silicon_wsvec.dat", "silicon_centres.xyz", "silicon.win"]


@pytest.fixture(scope="module")
def wfiles(tmp_path_factory):
    out = tmp_path_factory.mktemp("wannier")
    for name in FILES:
        with gzip.open(os.path.join(GOLDEN, "wannier", name + ".gz"), "rb") as src, open(out / name, "wb") as dst:
            shutil.copyfileobj(src, dst)
    return {name: str(out / name) for name in FILES}


@pytest.fixture(scope="module")
def wgolden():
    return load_golden("wannier")


def _sorted_packed(model):
    r_vec, payload = model.packed_hop()
    if model._sparse:
        r_ptr, row, col, val = payload
        hop = tbmodels_amd.synthetic.csr_to_dense(model.size, r_ptr, row, col, val)
    else:
        hop = payload
    order = np.lexsort(r_vec.T[::-1])
    return r_vec[order], hop[order]


def _check(model, golden, tag):
    r_vec, hop = _sorted_packed(model)
    assert np.array_equal(r_vec, golden[tag + "_R"])
    assert np.abs(hop - golden[tag + "_hop"]).max() < 1e-14
    assert np.abs(model.pos - golden[tag + "_pos"]).max() < 1e-12
    if tag + "_uc" in golden:
        assert np.abs(model.uc - golden[tag + "_uc"]).max() < 1e-14
    else:
        assert model.uc is None


def test_hr_only(wfiles, wgolden):
    model = tbmodels_amd.Model.from_wannier_files(hr_file=wfiles["silicon_hr.dat"])
    assert model.size == 8 and model.dim == 3
    _check(model, wgolden, "hr")


def test_hr_and_wsvec(wfiles, wgolden):
    model = tbmodels_amd.Model.from_wannier_files(hr_file=wfiles["silicon_hr.dat"], wsvec_file=wfiles["silicon_wsvec.dat"])
    _check(model, wgolden, "hr_ws")


def test_all_files_and_position_kinds(wfiles, wgolden):
    kwargs = dict(hr_file=wfiles["silicon_hr.dat"], wsvec_file=wfiles["silicon_wsvec.dat"],
                  xyz_file=wfiles["silicon_centres.xyz"], win_file=wfiles["silicon.win"])
    _check(tbmodels_amd.Model.from_wannier_files(**kwargs), wgolden, "all")
    _check(tbmodels_amd.Model.from_wannier_files(pos_kind="nearest_atom", distance_ratio_threshold=1.0, **kwargs),
           wgolden, "nearest")
    with pytest.raises(ValueError):
        tbmodels_amd.Model.from_wannier_files(pos_kind="whatever", **kwargs)
    with pytest.raises(ValueError):
        tbmodels_amd.Model.from_wannier_files(uc=np.eye(3), **kwargs)
    with pytest.raises(ValueError):
        tbmodels_amd.Model.from_wannier_files(hr_file=wfiles["silicon_hr.dat"], xyz_file=wfiles["silicon_centres.xyz"])
    with pytest.raises(ValueError):  # the default threshold rejects silicon's bond-centred Wannier functions
        tbmodels_amd.Model.from_wannier_files(pos_kind="nearest_atom", **kwargs)


def test_cutoff_and_sparse(wfiles, wgolden):
    model = tbmodels_amd.Model.from_wannier_files(hr_file=wfiles["silicon_hr.dat"], h_cutoff=0.05, sparse=True)
    assert model._sparse
    _check(model, wgolden, "cutoff")


def test_malformed_files(wfiles, tmp_path):
    text = open(wfiles["silicon_wsvec.dat"]).read().splitlines()
    bad = tmp_path / "short_wsvec.dat"
    bad.write_text("\n".join(text[:-2]) + "\n")
    with pytest.raises(ValueError):
        wannier.read_wsvec(str(bad))
    empty = tmp_path / "empty_wsvec.dat"
    empty.write_text("")
    with pytest.raises(ValueError):
        wannier.read_wsvec(str(empty))
    hr = open(wfiles["silicon_hr.dat"]).read().splitlines()
    swapped = tmp_path / "swapped_hr.dat"
    parts = hr[10].split()
    parts[3], parts[4] = "2", "1"  # break the expected orbital order of one entry
    swapped.write_text("\n".join(hr[:10] + ["   ".join(parts)] + hr[11:]) + "\n")
    tbmodels_amd.wannier.read_hr(str(swapped), ignore_orbital_order=True)


@pytest.mark.gpu
def test_wannier_model_on_gpu_matches_reference(wfiles, wgolden, silicon):
    """The parsed model through the GPU path against the reference's H(k) (its stored test_wannier goldens)."""
    kwargs = dict(hr_file=wfiles["silicon_hr.dat"], wsvec_file=wfiles["silicon_wsvec.dat"],
                  xyz_file=wfiles["silicon_centres.xyz"], win_file=wfiles["silicon.win"])
    model = tbmodels_amd.Model.from_wannier_files(**kwargs)
    assert np.abs(model.hamilton(KPT) - wgolden["all_h2_stored"]).max() < 1e-10
    assert np.abs(model.hamilton(KPT, convention=1) - wgolden["all_h1"]).max() < 1e-10
    assert np.abs(np.array(model.eigenval(KPT)) - wgolden["all_eig"]).max() < 1e-10
    # the reference's known-answer eigenvalues (tests/samples/cli_eigenvals/silicon_eigenvals.hdf5, atol 1e-10)
    eig = np.array(model.eigenval(silicon["known_kpoints"]))
    assert np.abs(eig - silicon["known_eigenvals"]).max() < 1e-10
    hr_only = tbmodels_amd.Model.from_wannier_files(hr_file=wfiles["silicon_hr.dat"])
    assert np.abs(hr_only.hamilton(KPT) - wgolden["hr_h2_stored"]).max() < 1e-10
