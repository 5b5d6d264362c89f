"""
oracle/tbk_oracle.py -- CPU restatement of the TBmodels k-space evaluation path.  TEST INFRASTRUCTURE.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  ``tbmodels_amd`` never imports it and has no CPU
fallback -- without ``libtbk.so`` and a GPU the product path raises.

What it restates (reference = Z2PackDev/TBmodels 1.4.4, pure NumPy/SciPy, file:line under
``/root/reference/``):

* :func:`hamilton`  -- ``Model.hamilton``, ``src/tbmodels/_tb_model.py:1076-1132``
* :func:`eigenval`  -- ``Model.eigenval``, ``src/tbmodels/_tb_model.py:1134-1150``
* :func:`kdotp_hamilton` / :func:`kdotp_eigenval` -- ``KdotpModel.hamilton`` / ``eigenval``,
  ``src/tbmodels/kdotp.py:51-100``
* :func:`construct_kdotp` -- ``Model.construct_kdotp``, ``src/tbmodels/_tb_model.py:942-982``

The model is taken as packed arrays (``R int (n_r, dim)``, ``hop complex128 (n_r, N, N)`` holding the
half-space blocks with the ``R = 0`` block halved, exactly what ``model.hop`` holds after
``_tb_model.py:175-218``) instead of the reference's dict, so the same arrays can be handed to the
HIP library.  The arithmetic is the reference's: one broadcast multiply-accumulate per stored R into
an ``(NK, N, N)`` complex128 array, ``+ h.c.``, optional convention-1 orbital phases, then
``scipy.linalg.eigvalsh`` per k-point (LAPACK ``zheevr``; SciPy is the reference's own third-party
dependency for this step, ``pyproject.toml:39``).

PARITY PINNED: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which ``tools/gen_golden.py`` produced by importing the unmodified reference
in the build container and which embed the reference's own stored goldens
(``tests/samples/cli_eigenvals/silicon_eigenvals.hdf5`` at 1e-10, ``tests/regression_data/test_hamilton``,
``test_eigenval``, ``test_wannier``).  Agreement demanded there: <= 1e-12.
"""

import itertools
import math

import numpy as np
import scipy.linalg as la


def _as_k_array(k, dim):
    """
    Argument normalisation of ``_tb_model.py:1103-1108``: a 1-D input (or a bare scalar for a
    1-D model) is ONE k-point, a 2-D input is a batch.  Returns ``(k (NK, dim) float64, single)``.
    """
    k_array = np.array(k, ndmin=1, dtype=float)
    single = k_array.ndim == 1
    if single:
        k_array = k_array.reshape(1, -1)
    if k_array.ndim != 2 or k_array.shape[1] != dim:
        raise ValueError("k has shape {} but the model has dimension {}".format(k_array.shape, dim))
    return k_array, single


def check_convention(convention):
    """``_tb_model.py:1097-1102``: anything but the integers 1 and 2 is a ValueError, before any work."""
    if convention not in [1, 2]:
        raise ValueError("Invalid value '{}' for 'convention': must be either '1' or '2'".format(convention))


def hamilton(r_vec, hop, k, convention=2, pos=None, n_orb=None):
    """
    H(k) for one k-point or a batch (``Model.hamilton``, ``_tb_model.py:1076-1132``).

    ``A[k] = sum_R exp(2 pi i k.R) hop[R]`` accumulated R by R (``:1111-1122``), then
    ``H = A + A^H`` (``:1123``); for ``convention == 1`` each element is multiplied by
    ``conj(e_i) e_j`` with ``e_p = exp(2 pi i k.pos_p)`` (``:1124-1128``).
    Returns ``(N, N)`` for a single point and ``(NK, N, N)`` for a batch (``:1130-1132``).
    """
    check_convention(convention)
    r_vec = np.asarray(r_vec)
    hop = np.asarray(hop, dtype=np.complex128)
    if n_orb is None:
        n_orb = hop.shape[1] if hop.ndim == 3 and hop.shape[0] else len(pos)
    if r_vec.ndim == 2 and r_vec.shape[0]:
        dim = r_vec.shape[1]
    elif pos is not None:
        dim = np.shape(pos)[1]
    else:  # empty model: take the dimension from k itself
        dim = np.array(k, ndmin=2).shape[-1]
    k_array, single = _as_k_array(k, dim)
    n_k = k_array.shape[0]
    ham = np.zeros((n_k, n_orb, n_orb), dtype=np.complex128)
    scratch = np.empty_like(ham)
    for idx in range(len(r_vec)):
        phase = np.exp(2j * np.pi * (k_array @ r_vec[idx].astype(float)))
        np.multiply(phase[:, None, None], hop[idx][None, :, :], out=scratch)
        ham += scratch
    ham += ham.conj().transpose(0, 2, 1)
    if convention == 1:
        pos = np.zeros((n_orb, dim)) if pos is None else np.asarray(pos, dtype=float)
        orb_phase = np.exp(2j * np.pi * (k_array @ pos.T))  # (NK, N)
        ham = orb_phase.conj()[:, :, None] * ham * orb_phase[:, None, :]
    return ham[0] if single else ham


def eigenval(r_vec, hop, k, n_orb=None, pos=None):
    """
    Eigenvalues at one k-point (1-D array) or a batch (a Python list of 1-D arrays, like the
    reference): ``Model.eigenval``, ``_tb_model.py:1134-1150``; convention 2 Hamiltonian,
    ``scipy.linalg.eigvalsh`` with its defaults per matrix.
    """
    ham = hamilton(r_vec, hop, k, convention=2, pos=pos, n_orb=n_orb)
    if ham.ndim == 3:
        return [la.eigvalsh(h) for h in ham]
    return la.eigvalsh(ham)


# ---------------------------------------------------------------------------------------------
# k.p models
# ---------------------------------------------------------------------------------------------
def kdotp_hamilton(powers, coeffs, k):
    """
    ``KdotpModel.hamilton`` (``src/tbmodels/kdotp.py:51-82``):
    ``H(k) = sum_p prod_d k_d^{p_d} * coeffs[p]``; ``powers int (n_p, dim)``, ``coeffs (n_p, N, N)``.
    """
    powers = np.asarray(powers)
    coeffs = np.asarray(coeffs, dtype=np.complex128)
    k_array, single = _as_k_array(k, powers.shape[1])
    ham = np.zeros((k_array.shape[0],) + coeffs.shape[1:], dtype=np.complex128)
    for p_vec, mat in zip(powers, coeffs):
        monomial = np.prod(k_array ** p_vec, axis=-1)
        ham += monomial[:, None, None] * mat[None, :, :]
    return ham[0] if single else ham


def kdotp_eigenval(powers, coeffs, k):
    """``KdotpModel.eigenval`` (``src/tbmodels/kdotp.py:84-100``)."""
    ham = kdotp_hamilton(powers, coeffs, k)
    if ham.ndim == 3:
        return [la.eigvalsh(h) for h in ham]
    return la.eigvalsh(ham)


def construct_kdotp(r_vec, hop, k0, order):
    """
    Taylor coefficients of H(k) around ``k0`` up to total degree ``order``
    (``Model.construct_kdotp``, ``_tb_model.py:942-982``)::

        C[p] = (2 pi i)^{|p|} / prod_d p_d!  *  ( sum_R prod_d R_d^{p_d} e^{2 pi i k0.R} hop[R]  +  h.c.-partner )

    where the Hermitian partner of a stored block is the ``-R`` block ``hop[R]^H``, whose monomial
    carries the sign ``(-1)^{|p|}``.  Returns ``(powers int64 (n_p, dim), coeffs complex128 (n_p, N, N))``
    with the powers in lexicographically sorted order.
    """
    r_vec = np.asarray(r_vec)
    hop = np.asarray(hop, dtype=np.complex128)
    k0 = np.asarray(k0, dtype=float)
    dim = r_vec.shape[1]
    n_orb = hop.shape[1]
    powers = sorted(p for p in itertools.product(range(order + 1), repeat=dim) if sum(p) <= order)
    coeffs = np.zeros((len(powers), n_orb, n_orb), dtype=np.complex128)
    for p_idx, p_vec in enumerate(powers):
        deg = sum(p_vec)
        prefactor = (2j * np.pi) ** deg / np.prod([math.factorial(x) for x in p_vec])
        acc = np.zeros((n_orb, n_orb), dtype=np.complex128)
        for idx in range(len(r_vec)):
            r_f = r_vec[idx].astype(float)
            mono = np.prod(r_f ** np.array(p_vec))
            phase = np.exp(2j * np.pi * np.dot(k0, r_f))
            acc += mono * phase * hop[idx]
            acc += (-1) ** deg * mono * np.conj(phase) * hop[idx].conj().T
        coeffs[p_idx] = prefactor * acc
    return np.array(powers, dtype=np.int64).reshape(len(powers), dim), coeffs
