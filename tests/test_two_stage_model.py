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


def _band_eig_error(h, panel_qr):
    """Scaled eigenvalue error of the band the model's first stage leaves with `panel_qr` as its panel factorisation."""
    n = len(h)
    saved = model.panel_qr
    model.panel_qr = panel_qr
    try:
        band, _ = model.stage1_band(h)
    finally:
        model.panel_qr = saved
    hb = np.zeros((n, n), dtype=complex)
    for i in range(n):
        for dd in range(min(model.B, n - 1 - i) + 1):
            hb[i, i + dd] = band[i, dd]
            hb[i + dd, i] = np.conj(band[i, dd])
    ref = np.linalg.eigvalsh(h)
    return np.abs(np.linalg.eigvalsh(hb) - ref).max() / max(1e-300, np.abs(ref).max())


@pytest.mark.parametrize("n", [40, 97])
def test_gram_panel_qr_is_as_accurate_as_the_sequential_one_and_restarts_where_it_must(n):
    """Round 5 (csrc/tbk_eig_band.hip, TBK_PANEL_GRAM): all reflectors of a panel from ONE Gram matrix.  The remaining norm
    of a column is a difference of Gram sums; a column where it has cancelled below 1/64 of the full norm ends the round and a
    fresh Gram matrix is formed (model.panel_qr_gram).  Random matrices never need a second round; the structured matrices of
    tests/test_gpu_parity.py::test_eigensolver_structured_matrices that make columns (nearly) dependent -- graded, rank one,
    two equal blocks, rank 3 + noise, a tridiagonal one -- do, and stay as accurate as with one round of sums per reflector;
    WITHOUT the threshold the same matrices lose all accuracy."""
    rng = np.random.default_rng(100 + n)
    rand = _hermitian(rng, n)
    graded = rand * np.outer(10.0 ** -np.arange(n) / max(1, n // 8), np.ones(n))
    blk = np.zeros((n, n), dtype=complex)
    h = n // 2
    blk[:h, :h] = rand[:h, :h]
    blk[h:, h:] = rand[:n - h, :n - h]
    low = rand[:, :3] @ rand[:, :3].conj().T
    cases = {  # name: (matrix, must the QR take extra rounds?)
        "random": (rand, False), "tiny": (rand * 1e-30, False), "huge": (rand * 1e30, False),
        "diagonal": (np.diag(rng.standard_normal(n)).astype(complex), False),
        "identity": (np.eye(n, dtype=complex) * 0.75, False),
        "graded": ((graded + graded.conj().T) / 2, True),
        "imag_offdiag": (np.diag(np.arange(n, dtype=float)) + 1j * (np.eye(n, k=1) - np.eye(n, k=-1)), True),
        "two_equal_blocks": (blk, None),  # (restarts at some sizes only)
        "rank_one": (np.outer(rand[:, 0], rand[:, 0].conj()), True),
        "rank3_plus_1e-9": (low + 1e-9 * rand, True),
    }
    dup = rand.copy()  # exactly dependent columns inside the first panels (every fourth row / column copies its neighbour)
    for q in range(1, n - 1, 4):
        dup[q + 1, :] = dup[q, :]
        dup[:, q + 1] = dup[:, q]
    cases["duplicate_columns"] = ((dup + dup.conj().T) / 2, True)
    worst_without = 0.0
    for name, (mat, restarts) in cases.items():
        sequential = _band_eig_error(mat, model.panel_qr)
        model.GRAM_STATS.update(rounds=0, panels=0, columns=0)
        gram = _band_eig_error(mat, model.panel_qr_gram)
        stats = dict(model.GRAM_STATS)
        assert gram <= max(4.0 * sequential, 2e-14), (name, gram, sequential)
        if restarts:
            assert stats["rounds"] > stats["panels"], (name, stats)
        elif restarts is not None:
            assert stats["rounds"] <= stats["panels"], (name, stats)  # (panels without a row below the diagonal: no round)
        if restarts:
            worst_without = max(worst_without, _band_eig_error(mat, lambda y: model.panel_qr_gram(y, thresh=0.0)))
    assert worst_without > 1e-9  # the threshold is part of the algorithm


def test_gram_panel_qr_returns_the_factors_of_the_sequential_one():
    """Same V, tau and R to rounding on a well-conditioned panel; a panel with an exactly dependent column takes a second
    round and still reproduces Q R."""
    rng = np.random.default_rng(8)
    y = rng.standard_normal((50, model.B)) + 1j * rng.standard_normal((50, model.B))
    v1, t1, r1 = model.panel_qr(y)
    v2, t2, r2 = model.panel_qr_gram(y)
    assert np.abs(v1 - v2).max() < 1e-13 and np.abs(t1 - t2).max() < 1e-13 and np.abs(r1 - r2).max() < 1e-12
    y[:, 5] = y[:, 1] * (0.3 - 0.2j) + y[:, 2]  # exactly dependent
    model.GRAM_STATS.update(rounds=0, panels=0, columns=0)
    v, tau, r = model.panel_qr_gram(y)
    assert model.GRAM_STATS["rounds"] >= 2
    q = np.eye(50, dtype=complex)
    for c in range(model.B):
        q = q @ (np.eye(50) - tau[c] * np.outer(v[:, c], v[:, c].conj()))
    full = np.zeros((50, model.B), dtype=complex)
    full[:model.B] = np.triu(r)
    assert np.abs(q @ full - y).max() < 1e-12 and np.abs(q.conj().T @ q - np.eye(50)).max() < 1e-13


@pytest.mark.parametrize("n, slots, window", [(40, 4, 80), (100, 4, 80), (150, 4, 79), (133, 3, 64), (200, 8, 140), (97, 2, 52)])
def test_second_stage_in_a_cyclic_window_equals_the_pipelined_chase(n, slots, window):
    """``stage2_window``: the working diagonals of the second stage in a cyclic window of columns in front of a backing store
    (csrc/tbk_eig_band.hip, band_chase4w_kernel: 512 columns of LDS in front of global memory, for more orbitals than the LDS
    holds).  The model carries an occupancy tag per window column and asserts that every access finds ITS column and that a
    column only enters a free cell; the tridiagonal equals the pipelined chase's bit for bit (the same sweeps, another
    schedule), and its spectrum is the matrix' (scipy's eigvalsh at _tb_model.py:1149)."""
    rng = np.random.default_rng(n)
    m = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    h = (m + m.conj().T) / 2
    band, _ = model.stage1_band(h)
    d0, e0, _ = model.stage2_pipelined(band, 2, slots)
    d1, e1, ticks = model.stage2_window(band, slots, window)
    assert np.array_equal(d0, d1) and np.array_equal(e0, e1)
    assert model.stage2_window.peak_columns <= window
    assert np.abs(model.tridiag_eigvals(d1, e1) - np.linalg.eigvalsh(h)).max() < 1e-12 * n
    assert ticks > 0


def test_cyclic_window_of_the_kernel_s_size_holds_the_sweeps_in_flight():
    """The kernel's parameters (32 sweep slots, 512 window columns) on a band matrix wider than the window: no access misses,
    no column enters an occupied cell, at most 512 columns are ever held."""
    n = 600
    rng = np.random.default_rng(7)
    band = np.zeros((n, model.B + 1), dtype=complex)
    for dd in range(model.B + 1):
        band[: n - dd, dd] = rng.standard_normal(n - dd) + (1j * rng.standard_normal(n - dd) if dd else 0)
    d, e, _ = model.stage2_window(band, 32, 512)
    assert model.stage2_window.peak_columns <= 512
    hb = np.zeros((n, n), dtype=complex)
    for dd in range(model.B + 1):
        idx = np.arange(n - dd)
        hb[idx, idx + dd] = band[: n - dd, dd]
    hb = np.triu(hb) + np.triu(hb, 1).conj().T
    assert np.abs(model.tridiag_eigvals(d, e) - np.linalg.eigvalsh(hb)).max() < 1e-12 * n


@pytest.mark.parametrize("slots, window, sizes", [
    (16, 272, list(range(258, 520))),
    (32, 512, list(range(513, 1100)) + list(range(1100, 4097, 37)) + [2047, 2048, 2049, 3071, 3072, 3073, 4089, 4095, 4096]),
    (4, 80, list(range(30, 200))),
    (2, 52, list(range(30, 120))),
])
def test_window_trackers_of_the_kernel_find_the_columns_the_table_names(slots, window, sizes):
    """band_chase4w_kernel does not hold a table of which column enters or leaves the window at which tick: three running
    trackers find them (``window_plan_by_trackers`` is that code).  For every orbital count of both instantiations -- <4, 272>
    up to 512 orbitals, <8, 512> above -- and two small parameter sets where generation boundaries coincide far more often,
    the trackers name exactly the columns of ``stage2_window``'s table, tick by tick (round 5: at n = 1 mod 8 with 16 slots a
    generation's last chunk and the next one's first fall on the same tick, and the first tracker missed one of them)."""
    for n in sizes:
        model.stage2_window(np.zeros((n, model.B + 1), dtype=complex), slots, window, plan_only=True)
        table = model.stage2_window.plan
        fetch, evict = model.window_plan_by_trackers(n, slots, window)
        assert fetch == table[0], n
        assert evict == table[1], n
