"""
The NumPy model of the two-stage reduction (tools/two_stage_model.py) is the executable statement of what
csrc/tbk_eig_band.hip does; the GPU tests compare the kernels with it entry by entry.  Here (CPU): the model itself
against numpy.linalg.eigvalsh, and the pipelining of the second stage -- sweeps two chase steps apart touch disjoint
cells within a tick, one step apart they clash.
"""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import two_stage_model as model  # noqa: E402  pylint: disable=wrong-import-position


def _hermitian(rng, n):
    m = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    return (m + m.conj().T) / 2


@pytest.mark.parametrize("n", [10, 17, 65, 90])
def test_model_reduces_to_band_and_tridiagonal_with_the_same_spectrum(n):
    rng = np.random.default_rng(n)
    h = _hermitian(rng, n)
    ref = np.linalg.eigvalsh(h)
    band, work = model.stage1_band(h)
    outside = [abs(work[i, j]) for i in range(n) for j in range(i + model.B + 1, n) if not np.isnan(work[i, j])]
    assert not outside or max(outside) == 0.0
    hb = np.zeros((n, n), dtype=complex)
    for i in range(n):
        for dd in range(min(model.B, n - 1 - i) + 1):
            hb[i, i + dd] = band[i, dd]
            hb[i + dd, i] = np.conj(band[i, dd])
    assert np.abs(np.linalg.eigvalsh(hb) - ref).max() < 1e-12 * n
    d, e, _ = model.stage2_tridiag(band)
    assert np.abs(model.tridiag_eigvals(d, e) - ref).max() < 1e-12 * n


def test_structured_panels_take_the_zero_reflector_branches():
    """Diagonal and block-diagonal inputs: whole panels are already reduced (tau = 0 everywhere)."""
    rng = np.random.default_rng(5)
    n = 40
    for h in (np.diag(rng.standard_normal(n)).astype(complex), np.kron(np.eye(4), _hermitian(rng, 10))):
        band, _ = model.stage1_band(h)
        d, e, _ = model.stage2_tridiag(band)
        assert np.abs(model.tridiag_eigvals(d, e) - np.linalg.eigvalsh(h)).max() < 1e-12 * n


def test_chase_pipeline_two_steps_apart_is_conflict_free():
    rng = np.random.default_rng(3)
    n = 45
    h = _hermitian(rng, n)
    band, _ = model.stage1_band(h)
    ref = np.linalg.eigvalsh(h)
    d_seq, e_seq, _ = model.stage2_tridiag(band)
    for stagger, waves in ((2, 10 ** 6), (2, 3), (3, 2)):
        d, e, _ = model.stage2_pipelined(band, stagger, waves)
        assert np.abs(model.tridiag_eigvals(d, e) - ref).max() < 1e-12 * n
        assert np.allclose(d, d_seq, atol=1e-12) and np.allclose(e, e_seq, atol=1e-12)  # the same algorithm, reordered
    with pytest.raises(AssertionError, match="share cells"):
        model.stage2_pipelined(band, 1)


def test_pass_chain_keeps_the_addition_order_of_the_partner_products():
    """csrc/tbk_eig_band.hip replaces the per-step barrier of the tile pass by "wave w waits for wave w + 1" wherever
    `na >= 2 NW`; tools/check_pass_chain.py restates the visit schedule: no block may get two writers the rule leaves unordered."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("check_pass_chain", os.path.join(ROOT, "tools", "check_pass_chain.py"))
    chain = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chain)
    unordered_somewhere = False
    for na in range(1, 41):
        for n_waves in (2, 4, 8):
            for members in (1, 2, 4):
                # (with the split last round: every tile of the triangle exactly once, operands fetched before use)
                assert not chain.coverage_errors(na, n_waves * members), (na, n_waves, members)
                bad = chain.violations(na, n_waves, members)
                if chain.chain_allowed(na, n_waves):
                    assert not bad, (na, n_waves, members, bad[:2])
                unordered_somewhere |= bool(bad)
    assert unordered_somewhere  # (the check can fail: eight waves on 5 - 11 blocks are unordered, and keep the barrier)
