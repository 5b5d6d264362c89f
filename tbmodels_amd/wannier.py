"""
Wannier90 output files -> half-space hopping blocks, for ``Model.from_wannier_files``.

Reference behaviour: ``Model.from_wannier_files`` with ``_read_hr`` / ``_read_wsvec`` / ``_read_xyz`` /
``_read_win`` (``/root/reference/src/tbmodels/_tb_model.py:399-441``, ``:565-852``), which stream one Python tuple
per matrix element through generators into ``from_hop_list``.  This is the producer of every realistic
model, so it is rebuilt array-wise: the ``*_hr.dat`` body is parsed in one
``numpy`` call into ``(nrpts * N^2, 7)``, degeneracy weights and the ``h_cutoff`` filter are applied as
array operations, the ``*_wsvec.dat`` shifts are expanded with ``np.repeat`` and everything is scattered into
dense ``(n_R, N, N)`` blocks with one ``np.add.at`` -- the result goes to ``Model(hop=...)`` unchanged, so
the Hermiticity check, half-space reduction and ``R = 0`` halving stay where the reference has them.
"""

import itertools
import re

import numpy as np

__all__ = ("read_hr", "read_wsvec", "read_xyz", "read_win", "hop_blocks_from_wannier", "positions_from_xyz")


class WannierParseError(ValueError):
    """Malformed Wannier90 file (the reference raises TbmodelsException / ValueError / AssertionError here)."""


def read_hr(path, ignore_orbital_order=False):
    """
    Parse ``*_hr.dat``.  Returns ``(num_wann, R int64 (nrpts*N^2, 3), row, col, value complex)`` with the
    degeneracy of every lattice point already divided out (``_tb_model.py:399-441``).
    """
    with open(path, encoding="utf-8") as handle:
        handle.readline()  # date line
        num_wann = int(handle.readline())
        nrpts = int(handle.readline())
        degeneracy = []
        for _ in range(-(-nrpts // 15)):
            degeneracy.extend(int(x) for x in handle.readline().split())
        if len(degeneracy) != nrpts:
            raise WannierParseError("expected {} degeneracy weights, found {}".format(nrpts, len(degeneracy)))
        body = np.array(handle.read().split(), dtype=np.float64)
    if body.size % 7:
        raise WannierParseError("the hopping section of '{}' is not a table of 7 columns".format(path))
    table = body.reshape(-1, 7)
    n_sq = num_wann * num_wann
    r_vec = table[:, :3].astype(np.int64)
    row = table[:, 3].astype(np.int64) - 1
    col = table[:, 4].astype(np.int64) - 1
    index = np.arange(len(table))
    if not ignore_orbital_order:
        # the reference's consistency test as written there: first index runs fastest
        bad = (row != index % num_wann) & (col == (index % n_sq) // num_wann)
        if bad.any():
            raise ValueError("Inconsistent orbital numbers in entry {} of '{}'".format(int(np.flatnonzero(bad)[0]), path))
    weights = np.array(degeneracy, dtype=np.float64)[index // n_sq]
    value = (table[:, 5] + 1j * table[:, 6]) / weights
    return num_wann, r_vec, row, col, value


def read_wsvec(path):
    """
    Parse ``*_wsvec.dat`` (written with ``use_ws_distance``): ``{(o1, o2, R): int array (n_T, 3)}``
    (``_tb_model.py:769-793``).
    """
    with open(path, encoding="utf-8") as handle:
        lines = handle.read().splitlines()
    if not lines:
        raise WannierParseError("The 'wsvec' iterator is empty.")
    shifts = {}
    pos = 1  # first line is a comment
    n_lines = len(lines)
    while pos < n_lines:
        head = lines[pos].split()
        if not head:
            pos += 1
            continue
        *r_vec, o_1, o_2 = (int(x) for x in head)
        try:
            count = int(lines[pos + 1])
            block = [tuple(int(x) for x in lines[pos + 2 + t].split()) for t in range(count)]
        except (IndexError, ValueError) as exc:
            raise WannierParseError("Incomplete wsvec iterator.") from exc
        shifts[(o_1 - 1, o_2 - 1, tuple(r_vec))] = np.array(block, dtype=np.int64).reshape(count, len(r_vec))
        pos += 2 + count
    return shifts


def read_xyz(path):
    """``*_centres.xyz``: ``(wannier centres (n, 3), atom positions (m, 3))`` in cartesian coordinates (``:795-812``)."""
    with open(path, encoding="utf-8") as handle:
        count = int(handle.readline())
        handle.readline()
        centres, atoms = [], []
        for line in handle:
            fields = line.split()
            if not fields:
                continue
            (centres if fields[0] == "X" else atoms).append([float(x) for x in fields[1:4]])
    if len(centres) + len(atoms) != count:
        raise WannierParseError("'{}' announces {} entries but holds {}".format(path, count, len(centres) + len(atoms)))
    return np.array(centres, dtype=float).reshape(-1, 3), np.array(atoms, dtype=float).reshape(-1, 3)


_SPLIT = re.compile("[\t :=]+")


def read_win(path):
    """
    The parts of a ``*.win`` file the model needs: ``{"length_unit": ..., "unit_cell_cart": (3, 3) in Angstrom}``
    plus every other key as raw text (``_tb_model.py:814-852``).
    """
    with open(path, encoding="utf-8") as handle:
        raw = handle.read().splitlines()
    lines = []
    for line in raw:
        line = line.split("!")[0].split("#")[0].strip().lower()
        if line:
            lines.append(line)
    mapping = {}
    it = iter(lines)
    for line in it:
        if line.startswith("begin"):
            key = _SPLIT.split(line[5:].strip(" :="), 1)[0]
            block = []
            for inner in it:
                if inner.startswith("end"):
                    if _SPLIT.split(inner[3:].strip(" :="), 1)[0] != key:
                        raise WannierParseError("block '{}' is closed by '{}'".format(key, inner))
                    break
                block.append(inner)
            mapping[key] = block
        else:
            parts = _SPLIT.split(line, 1)
            mapping[parts[0]] = parts[1] if len(parts) > 1 else ""
    unit = mapping.get("length_unit", "ang")
    unit = unit.strip().lower() if isinstance(unit, str) else "ang"
    mapping["length_unit"] = unit
    if "unit_cell_cart" in mapping:
        block = mapping["unit_cell_cart"]
        cell_unit = unit
        if len(block) == 4:
            cell_unit, block = block[0], block[1:]
        cell = np.array([[float(x) for x in _SPLIT.split(row)] for row in block], dtype=float).reshape(3, 3)
        if cell_unit == "bohr":
            cell = cell * 0.52917721092
        mapping["unit_cell_cart"] = cell
    return mapping


def positions_from_xyz(xyz_file, uc, pos_kind="wannier", distance_ratio_threshold=3.0):
    """
    Orbital positions in reduced coordinates from the Wannier centres (``pos_kind='wannier'``) or the nearest
    atom (``'nearest_atom'``), ``_tb_model.py:627-675``.
    """
    centres, atoms = read_xyz(xyz_file)
    uc = np.asarray(uc, dtype=float)
    if pos_kind == "wannier":
        cart = centres
    elif pos_kind == "nearest_atom":
        if distance_ratio_threshold < 1:
            raise ValueError("Invalid value for 'distance_ratio_threshold': must be >= 1.")
        neighbours = np.array(list(itertools.product([-1, 0, 1], repeat=3)), dtype=float)
        cart = np.empty_like(centres)
        for idx, centre in enumerate(centres):
            base = np.floor(np.linalg.solve(uc.T, centre))
            images = (atoms[:, None, :] + ((base + neighbours) @ uc)[None, :, :]).reshape(-1, 3)
            dist = np.linalg.norm(images - centre, axis=1)
            two = np.argpartition(dist, 2)[:2]
            nearest, second = dist[two]
            if second / nearest < distance_ratio_threshold:
                raise WannierParseError(
                    "The ratio ({:.3f}) between the nearest ({:.3f}) and second-nearest ({:.3f}) atomic position "
                    "is less than 'distance_ratio_threshold' ({}).".format(
                        second / nearest, nearest, second, distance_ratio_threshold
                    )
                )
            cart[idx] = images[two[0]]
    else:
        raise ValueError("Invalid value '{}' for 'pos_kind', must be 'wannier' or 'nearest_atom'".format(pos_kind))
    return np.linalg.solve(uc.T, cart.T).T


def hop_blocks_from_wannier(hr_file, wsvec_file=None, h_cutoff=0.0, ignore_orbital_order=False):
    """
    ``(num_wann, {R tuple: (N, N) complex})`` with both ``R`` and ``-R`` present, as ``*_hr.dat`` lists them
    (``contains_cc=True`` input for ``Model``).  With a wsvec file every element is split evenly over its
    ``R + T`` images (``_tb_model.py:683-713``).
    """
    num_wann, r_vec, row, col, value = read_hr(hr_file, ignore_orbital_order=ignore_orbital_order)
    keep = np.abs(value) > h_cutoff
    r_vec, row, col, value = r_vec[keep], row[keep], col[keep], value[keep]
    if wsvec_file is not None:
        shifts = read_wsvec(wsvec_file)
        try:
            per_entry = [shifts[(int(a), int(b), tuple(int(x) for x in r))] for a, b, r in zip(row, col, r_vec)]
        except KeyError as exc:
            raise KeyError(exc.args[0]) from None  # the reference's _async_parse ends in the same KeyError
        counts = np.array([len(t) for t in per_entry], dtype=np.int64)
        all_shifts = np.concatenate(per_entry) if per_entry else np.zeros((0, 3), dtype=np.int64)
        r_vec = np.repeat(r_vec, counts, axis=0) + all_shifts
        row = np.repeat(row, counts)
        col = np.repeat(col, counts)
        value = np.repeat(value / counts, counts)
    if len(value) == 0:
        return num_wann, {}
    uniq, inverse = np.unique(r_vec, axis=0, return_inverse=True)
    blocks = np.zeros((len(uniq), num_wann, num_wann), dtype=np.complex128)
    np.add.at(blocks, (np.asarray(inverse).reshape(-1), row, col), value)
    return num_wann, {tuple(int(x) for x in r): blocks[idx] for idx, r in enumerate(uniq)}
