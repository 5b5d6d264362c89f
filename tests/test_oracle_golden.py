"""
Pin the oracle (oracle/tbk_oracle.py) to the reference: every function against the golden
fixtures that tools/gen_golden.py produced by importing the unmodified reference, which embed the
reference's own stored goldens.  Bar: 1e-12.
"""

import numpy as np
import pytest

from oracle import tbk_oracle as oracle
from tbmodels_amd import synthetic as syn

from conftest import KPT

TOL = 1e-12


def _close(a, b, tol=TOL):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol, err


def test_silicon_known_answer(silicon):
    """tests/test_cli_eigenvals.py:21-50 of the reference: stored eigenvalues, atol 1e-10."""
    eig = oracle.eigenval(silicon["R"], silicon["hop"], silicon["known_kpoints"])
    assert isinstance(eig, list) and len(eig) == 11
    _close(np.array(eig), silicon["known_eigenvals"], 1e-10)


def test_silicon_reference_outputs(silicon):
    r_vec, hop, pos = silicon["R"], silicon["hop"], silicon["pos"]
    _close(oracle.hamilton(r_vec, hop, silicon["kpt"], 2), silicon["kpt_h2"])
    _close(oracle.hamilton(r_vec, hop, silicon["kpt"], 1, pos=pos), silicon["kpt_h1"])
    _close(np.array(oracle.eigenval(r_vec, hop, silicon["kpt"])), silicon["kpt_eig"])
    _close(np.array(oracle.eigenval(r_vec, hop, silicon["grid"])), silicon["grid_eig"])
    _close(oracle.hamilton(r_vec, hop, silicon["grid"][:16], 2), silicon["grid_h2_first16"])
    _close(oracle.hamilton(r_vec, hop, silicon["grid"][:16], 1, pos=pos), silicon["grid_h1_first16"])


def test_silicon_wannier_stored_golden(silicon):
    """tests/regression_data/test_wannier/...silicon_hr.dat-silicon_wsvec.dat (stored by the reference)."""
    ham = oracle.hamilton(silicon["wannier_R"], silicon["wannier_hop"], silicon["kpt"], 2)
    _close(ham, silicon["wannier_kpt_h2_stored"])


def test_grid_generator_matches_fixture(silicon):
    assert np.array_equal(syn.uniform_grid(10), silicon["grid"])
    assert np.array_equal(syn.grid_slab(10, 137, 412), silicon["grid"][137:412])


@pytest.mark.parametrize("t_idx", range(6))
@pytest.mark.parametrize("storage", ["dense", "sparse"])
def test_toy_stored_goldens(toy, t_idx, storage):
    """tests/test_hamilton.py:10-18 and tests/test_eigenval.py:10-14 regression goldens."""
    tag = "t%d_%s" % (t_idx, storage)
    r_vec, hop, pos = toy[tag + "_R"], toy[tag + "_hop"], toy[tag + "_pos"]
    for conv in (1, 2):
        per_k = np.array([oracle.hamilton(r_vec, hop, k, conv, pos=pos) for k in KPT])
        _close(per_k, toy[tag + "_h%d_stored" % conv])
        _close(per_k, toy[tag + "_h%d" % conv])
        _close(oracle.hamilton(r_vec, hop, KPT, conv, pos=pos), toy[tag + "_h%d_batch" % conv])
    eig = np.array([oracle.eigenval(r_vec, hop, k) for k in KPT])
    _close(eig, toy[tag + "_eig_stored"])


@pytest.mark.parametrize("dim", [2, 4])
def test_toy_other_dims(toy, dim):
    tag = "dim%d" % dim
    r_vec, hop, pos, k = toy[tag + "_R"], toy[tag + "_hop"], toy[tag + "_pos"], toy[tag + "_k"]
    _close(oracle.hamilton(r_vec, hop, k, 2), toy[tag + "_h2"])
    _close(oracle.hamilton(r_vec, hop, k, 1, pos=pos), toy[tag + "_h1"])
    _close(np.array(oracle.eigenval(r_vec, hop, k)), toy[tag + "_eig"])


def synthetic_case(synthetic, tag):
    """(R, hop, pos, k) of one synthetic fixture case, regenerating hop where only checksums are stored."""
    r_vec = synthetic[tag + "_R"]
    pos = synthetic[tag + "_pos"]
    k = synthetic[tag + "_k"]
    if tag + "_hop" in synthetic:
        hop = synthetic[tag + "_hop"]
    else:
        if tag == "dense64":
            _, hop, _ = syn.dense_model_arrays(64, 128, syn.MODEL_SEED + 101)
        elif tag == "csr64":
            _, r_ptr, row, col, val, _ = syn.csr_model_arrays(64, 40, syn.MODEL_SEED + 104)
            hop = syn.csr_to_dense(64, r_ptr, row, col, val)
        else:
            raise KeyError(tag)
        check = np.array([hop.sum(), np.abs(hop).sum(), (hop * np.arange(hop.size).reshape(hop.shape)).sum()])
        assert np.allclose(check, synthetic[tag + "_hop_sum"], rtol=1e-13, atol=0), "generator drifted from the fixture"
    return r_vec, hop, pos, k


SYN_TAGS = ["dense16", "dense64", "dense13", "dense1", "csr64", "dim1", "dim2", "single", "bigk"]


@pytest.mark.parametrize("tag", SYN_TAGS)
def test_synthetic_cases(synthetic, tag):
    r_vec, hop, pos, k = synthetic_case(synthetic, tag)
    n_h = len(synthetic[tag + "_h2"])
    _close(oracle.hamilton(r_vec, hop, k[:n_h], 2), synthetic[tag + "_h2"])
    _close(oracle.hamilton(r_vec, hop, k[:n_h], 1, pos=pos), synthetic[tag + "_h1"])
    _close(np.array(oracle.eigenval(r_vec, hop, k)), synthetic[tag + "_eig"])


def test_generator_reproduces_fixture_inputs(synthetic):
    """The numpy-2 generator on the GPU box must produce the arrays the reference was fed (numpy 1.26)."""
    r_vec, hop, pos = syn.dense_model_arrays(16, 33, syn.MODEL_SEED + 100)
    assert np.array_equal(r_vec, synthetic["dense16_R"])
    assert np.array_equal(hop, synthetic["dense16_hop"])
    assert np.array_equal(pos, synthetic["dense16_pos"])
    r_vec, r_ptr, row, col, val, _ = syn.csr_model_arrays(64, 40, syn.MODEL_SEED + 104)
    for name, arr in (("R", r_vec), ("r_ptr", r_ptr), ("row", row), ("col", col), ("val", val)):
        assert np.array_equal(arr, synthetic["csr64_in_" + name]), name
    assert np.array_equal(syn.random_kpoints(32), synthetic["dense64_k"])


def test_scalar_k_and_single_point(synthetic):
    """tests/test_convention.py:32 passes a bare scalar k for a 1-D model; 1-D k is ONE point."""
    r_vec, hop, pos = synthetic["dim1_R"], synthetic["dim1_hop"], synthetic["dim1_pos"]
    k = float(synthetic["dim1_scalar_k"])
    h2 = oracle.hamilton(r_vec, hop, k, 2)
    assert h2.shape == (6, 6)
    _close(h2, synthetic["dim1_scalar_h2"])
    _close(oracle.hamilton(r_vec, hop, k, 1, pos=pos), synthetic["dim1_scalar_h1"])
    eig = oracle.eigenval(r_vec, hop, k)
    assert isinstance(eig, np.ndarray) and eig.shape == (6,)
    _close(eig, synthetic["dim1_scalar_eig"])
    r_vec, hop = synthetic["single_R"], synthetic["single_hop"]
    _close(oracle.hamilton(r_vec, hop, synthetic["single_k"][0]), synthetic["single_k0_h2"])
    _close(oracle.eigenval(r_vec, hop, synthetic["single_k"][0]), synthetic["single_k0_eig"])


def test_empty_model(synthetic):
    k = [[0.1, 0.2, 0.3], [0.5, 0.5, 0.5]]
    ham = oracle.hamilton(np.zeros((0, 3), int), np.zeros((0, 3, 3), complex), k, n_orb=3)
    _close(ham, synthetic["empty_h2"])
    _close(np.array(oracle.eigenval(np.zeros((0, 3), int), np.zeros((0, 3, 3), complex), k, n_orb=3)), synthetic["empty_eig"])


@pytest.mark.parametrize("convention", ["a", "1", None, 3, 0])
def test_invalid_convention(silicon, convention):
    """tests/test_hamilton.py:35-42."""
    with pytest.raises(ValueError):
        oracle.hamilton(silicon["R"], silicon["hop"], (0, 0, 0), convention=convention)


@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_kdotp(kdotp_golden, order):
    g = kdotp_golden
    powers, coeffs = oracle.construct_kdotp(g["R"], g["hop"], g["k0"], order)
    assert np.array_equal(powers, g["order%d_powers" % order])
    _close(coeffs, g["order%d_coeffs" % order])
    dk = g["order%d_dk" % order]
    _close(oracle.kdotp_hamilton(powers, coeffs, dk), g["order%d_h" % order])
    _close(np.array(oracle.kdotp_eigenval(powers, coeffs, dk)), g["order%d_eig" % order])
    _close(oracle.kdotp_hamilton(powers, coeffs, dk[0]), g["order%d_h_single" % order])
    _close(oracle.kdotp_eigenval(powers, coeffs, dk[0]), g["order%d_eig_single" % order])
