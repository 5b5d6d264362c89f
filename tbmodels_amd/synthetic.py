"""
Deterministic synthetic tight-binding models for the parity tests and ``bench.py``.

This is the generator that SURVEY.md section 8(d) prescribes for the BASELINE.json configs
(there is no dataset to download, and the reference ships no model of the headline size):

* lattice vectors: enumerate ``[-12, 12]^dim``, keep ``R = 0`` and every vector whose first
  non-zero component is positive (the half-space the reference stores after
  ``/root/reference/src/tbmodels/_tb_model.py:247-298``), order by ``(|R|^2, lexicographic)``
  and take the first ``n_r``;
* dense hoppings: ``hop[R] = s * (A + iB)`` with ``A, B ~ N(0, 1)`` and ``s = 1/sqrt(n_r * n_orb)``;
  the ``R = 0`` block is replaced by ``(M + M^H) / 4`` -- half of a Hermitian on-site block, which is
  how the reference keeps it (``_tb_model.py:268``);
* sparse hoppings: per ``R``, ``round(fill * n_orb^2)`` distinct entries, values as above with
  ``s = 1/sqrt(n_r * fill * n_orb)``, row-major sorted (CSR);
* ``pos = rng.random((n_orb, dim))``.

Only ``numpy`` is imported, and nothing here needs Python >= 3.10, so that ``tools/gen_golden.py``
can run this module next to the imported reference under the container's conda python3.9.
"""

import itertools

import numpy as np

R_BOX = 12

#: seeds of SURVEY.md section 8(d): model seed = MODEL_SEED + config index, k-points K_SEED.
MODEL_SEED = 20240601
K_SEED = 12345


def half_space_vectors(n_r, dim=3, box=R_BOX):
    """First ``n_r`` half-space lattice vectors, ``int32 (n_r, dim)``, sorted by (|R|^2, lexicographic)."""
    rng_1d = range(-box, box + 1)
    cand = []
    for vec in itertools.product(rng_1d, repeat=dim):
        first = next((x for x in vec if x != 0), 0)
        if first >= 0:
            cand.append(vec)
    cand.sort(key=lambda v: (sum(x * x for x in v), v))
    if n_r > len(cand):
        raise ValueError("only {} half-space vectors in the [-{b},{b}]^{d} box".format(len(cand), b=box, d=dim))
    return np.array(cand[:n_r], dtype=np.int32).reshape(n_r, dim)


def dense_model_arrays(n_orb, n_r, seed, dim=3):
    """
    Packed dense model: ``(R int32 (n_r, dim), hop complex128 (n_r, n_orb, n_orb), pos float64 (n_orb, dim))``.
    ``R[0]`` is the zero vector (it sorts first).
    """
    rng = np.random.default_rng(seed)
    r_vec = half_space_vectors(n_r, dim=dim)
    scale = 1.0 / np.sqrt(n_r * n_orb)
    hop = np.empty((n_r, n_orb, n_orb), dtype=np.complex128)
    # one R at a time: bounded scratch and an R-by-R stream that does not depend on n_r
    for idx in range(n_r):
        re_part = rng.standard_normal((n_orb, n_orb))
        im_part = rng.standard_normal((n_orb, n_orb))
        hop[idx] = scale * (re_part + 1j * im_part)
    if n_r > 0:
        assert not r_vec[0].any()
        hop[0] = (hop[0] + hop[0].conj().T) / 4.0
    pos = rng.random((n_orb, dim))
    return r_vec, hop, pos


def csr_model_arrays(n_orb, n_r, seed, fill=0.02, dim=3):
    """
    Packed sparse model in concatenated-CSR form::

        R       int32      (n_r, dim)
        r_ptr   int64      (n_r + 1,)      entries of lattice vector r are [r_ptr[r], r_ptr[r+1])
        row     int32      (nnz,)          row-major sorted inside each R
        col     int32      (nnz,)
        val     complex128 (nnz,)
        pos     float64    (n_orb, dim)

    The ``R = 0`` block is symmetrised like the dense one (so it carries up to twice the entries).
    """
    rng = np.random.default_rng(seed)
    r_vec = half_space_vectors(n_r, dim=dim)
    nnz_r = int(round(fill * n_orb * n_orb))
    scale = 1.0 / np.sqrt(n_r * fill * n_orb)
    rows, cols, vals, r_ptr = [], [], [], [0]
    for idx in range(n_r):
        flat = np.sort(rng.permutation(n_orb * n_orb)[:nnz_r])
        v = scale * (rng.standard_normal(nnz_r) + 1j * rng.standard_normal(nnz_r))
        if idx == 0:
            dense = np.zeros((n_orb, n_orb), dtype=np.complex128)
            dense.reshape(-1)[flat] = v
            dense = (dense + dense.conj().T) / 4.0
            flat = np.flatnonzero(dense.reshape(-1))
            v = dense.reshape(-1)[flat]
        rows.append((flat // n_orb).astype(np.int32))
        cols.append((flat % n_orb).astype(np.int32))
        vals.append(v)
        r_ptr.append(r_ptr[-1] + len(flat))
    pos = rng.random((n_orb, dim))
    return (
        r_vec,
        np.array(r_ptr, dtype=np.int64),
        np.concatenate(rows) if rows else np.zeros(0, np.int32),
        np.concatenate(cols) if cols else np.zeros(0, np.int32),
        np.concatenate(vals) if vals else np.zeros(0, np.complex128),
        pos,
    )


def csr_to_dense(n_orb, r_ptr, row, col, val):
    """Densify concatenated-CSR hoppings into ``complex128 (n_r, n_orb, n_orb)`` (small cases only)."""
    n_r = len(r_ptr) - 1
    hop = np.zeros((n_r, n_orb, n_orb), dtype=np.complex128)
    for idx in range(n_r):
        sl = slice(r_ptr[idx], r_ptr[idx + 1])
        np.add.at(hop[idx], (row[sl], col[sl]), val[sl])
    return hop


def random_kpoints(n_k, dim=3, seed=K_SEED):
    """``default_rng(seed).random((n_k, dim))`` -- the k-point list of configs 2, 3 and 5."""
    return np.random.default_rng(seed).random((n_k, dim))


def uniform_grid(n_per_dim, dim=3):
    """Uniform ``n_per_dim^dim`` grid on ``[0, 1)^dim``, ``indexing='ij'`` (configs 1 and 4)."""
    return grid_slab(n_per_dim, 0, n_per_dim**dim, dim=dim)


def grid_slab(n_per_dim, start, stop, dim=3):
    """
    Rows ``[start, stop)`` of :func:`uniform_grid` without materialising the whole grid
    (each rank of a sharded run builds only its own contiguous slab).
    """
    axis = np.linspace(0.0, 1.0, n_per_dim, endpoint=False)
    idx = np.arange(start, stop, dtype=np.int64)
    out = np.empty((len(idx), dim), dtype=np.float64)
    rem = idx
    for d in range(dim - 1, -1, -1):
        out[:, d] = axis[rem % n_per_dim]
        rem = rem // n_per_dim
    return out
