"""
``Model`` -- the host-side mirror of ``tbmodels.Model`` for the k-space evaluation path.

The reference has no plugin boundary: the hot path is two bound methods,
``Model.hamilton(k, convention=2)`` (``/root/reference/src/tbmodels/_tb_model.py:1076-1132``) and
``Model.eigenval(k)`` (``:1134-1150``), reading ``self.hop``, ``self.size``, ``self.dim``,
``self.pos`` and ``self._sparse``.  This class keeps that surface -- same constructor keywords, same
``hop`` storage convention (half-space lattice vectors, ``R = 0`` block halved; ``:175-218``,
``:247-298``), same ``add_hop`` / ``add_on_site`` / ``set_sparse`` mutators, same argument and return
conventions, same exceptions -- and evaluates both methods on the GPU through ``libtbk.so``
(``include/tbk.h``).  Nothing here computes H(k) on the CPU; without the library and a device the
two methods raise.

Device state is a cache of ``self.hop``: it is re-validated against a content fingerprint on every
call (``add_hop``, ``model.hop[R] += ...`` and ``set_sparse`` all mutate in place), dropped on pickling
and rebuilt lazily.
"""

import collections as co
import ctypes
import os
import threading
import warnings
import zlib

import numpy as np

from . import _lib, _outbuf
from ._sparse_matrix import csr as _csr

try:  # fast content hash for the staging fingerprint; zlib is the fallback
    import xxhash as _xxhash
except ImportError:  # pragma: no cover
    _xxhash = None

__all__ = ("Model",)


def _devices_from_env():
    """``TBK_DEVICES=0,1,...`` (several GPUs behind one model), else ``TBK_DEVICE=i``, else device 0."""
    listed = os.environ.get("TBK_DEVICES", "").strip()
    if listed:
        return [int(part) for part in listed.replace(";", ",").split(",") if part.strip() != ""]
    return [int(os.environ.get("TBK_DEVICE", "0"))]


def _first_nonzero(vec):
    for x in vec:
        if x != 0:
            return x
    return 0


def _hash_bytes(running, array):
    data = np.ascontiguousarray(array)
    if _xxhash is not None:
        running.update(data.view(np.uint8).reshape(-1).data)
        return running
    return zlib.adler32(data.view(np.uint8).reshape(-1).data, running)


class _HopDict(co.defaultdict):
    """
    ``Model.hop``: a ``defaultdict`` (as in the reference, ``_tb_model.py:206``) that remembers whether a
    matrix may have been handed to outside code.

    The staged device copy has to follow in-place edits such as ``model.hop[R] += x``.  Hashing the hopping
    bytes on every call finds them but costs a pass over the model (1.3 ms for 512 matrices of 64 x 64) --
    more than a single-k ``hamilton`` call.  So: the model's own mutators (``add_hop``, ``add_on_site``,
    ``set_sparse``) go through the raw ``dict`` methods and bump ``version``; ANY outside access that can
    reach a matrix (``[]``, ``get``, ``values``, ``items``, ``pop``, ``update`` ...) sets ``exposed`` for good
    -- a reference handed out once can be written through at any later time -- and ``Model._staged`` falls
    back to the content hash from then on.  Untouched models (from files, from arrays) pay nothing.
    """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.exposed = False
        self.version = 0

    def __reduce__(self):
        # pickle: the items are set through __setitem__ (which marks the new dict exposed) and the state is applied
        # AFTER them -- nobody holds references into a freshly unpickled dict, so it starts clean.  Iterating with
        # dict.items keeps pickling from marking the SOURCE exposed.
        return (type(self), (self.default_factory,), {"exposed": False, "version": 0}, None, iter(dict.items(self)))

    def __copy__(self):  # copy.copy(model.hop): the copy shares every matrix with the source
        self.exposed = True
        new = type(self)(self.default_factory)
        dict.update(new, self)
        new.exposed = True
        return new

    def __deepcopy__(self, memo):  # private copies of the matrices: a clean dict
        import copy as _copy  # pylint: disable=import-outside-toplevel

        new = type(self)(self.default_factory)
        for key, value in dict.items(self):
            dict.__setitem__(new, key, _copy.deepcopy(value, memo))
        return new

    def __iter__(self):
        # overriding __iter__ takes dict(hop), {**hop} and dict.update(other, hop) off CPython's exact-dict fast path:
        # they then go through keys() + __getitem__, which marks the exposure
        return dict.__iter__(self)

    def __or__(self, other):  # hop | {...}: the result holds the same matrix objects
        self.exposed = True
        return co.defaultdict.__or__(self, other)

    def __ror__(self, other):
        self.exposed = True
        return co.defaultdict.__ror__(self, other)

    def __ior__(self, other):  # hop |= {...}: contents change behind the edit counter
        self.exposed = True
        dict.update(self, other)
        return self

    def __getitem__(self, key):
        self.exposed = True
        return super().__getitem__(key)

    def __setitem__(self, key, value):
        self.exposed = True
        super().__setitem__(key, value)

    def __delitem__(self, key):
        self.exposed = True
        super().__delitem__(key)

    def _exposing(name):  # pylint: disable=no-self-argument
        def method(self, *args, **kwargs):
            self.exposed = True
            return getattr(co.defaultdict, name)(self, *args, **kwargs)

        method.__name__ = name
        return method

    get = _exposing("get")
    values = _exposing("values")
    items = _exposing("items")
    pop = _exposing("pop")
    popitem = _exposing("popitem")
    setdefault = _exposing("setdefault")
    update = _exposing("update")
    clear = _exposing("clear")
    copy = _exposing("copy")
    del _exposing

    # --- the model's own access: raw dict methods, edits counted in `version` ---
    def raw_items(self):
        return dict.items(self)

    def raw_values(self):
        return dict.values(self)

    def raw_get(self, key):
        """``self[key]`` with the defaultdict insertion on a miss, without marking the dict exposed."""
        if dict.__contains__(self, key):
            return dict.__getitem__(self, key)
        value = self.default_factory()
        dict.__setitem__(self, key, value)
        self.version += 1
        return value

    def raw_set(self, key, value):
        dict.__setitem__(self, key, value)
        self.version += 1


class Model:
    """
    Tight-binding model with GPU evaluation of ``hamilton`` / ``eigenval``.

    Keyword arguments are those of ``tbmodels.Model`` (``_tb_model.py:89-102``): ``on_site``,
    ``hop`` (dict ``R -> (size, size)`` matrix), ``size``, ``dim``, ``occ``, ``pos``, ``uc``,
    ``contains_cc``, ``cc_check_tolerance``, ``sparse``.
    """

    def __init__(
        self,
        *,
        on_site=None,
        hop=None,
        size=None,
        dim=None,
        occ=None,
        pos=None,
        uc=None,
        contains_cc=True,
        cc_check_tolerance=1e-12,
        sparse=False,
    ):
        hop = {} if hop is None else hop
        self._handles = []
        self._staged_fingerprint = None
        self._pinned = False
        self._call_lock = threading.RLock()
        self.devices = _devices_from_env()

        self.set_sparse(sparse)

        # size and dimension are inferred in the reference's order of precedence (:135-172)
        if size is not None:
            self.size = size
        elif on_site is not None:
            self.size = len(on_site)
        elif pos is not None:
            self.size = len(pos)
        elif hop:
            self.size = next(iter(hop.values())).shape[0]
        else:
            raise ValueError(
                "Empty hoppings dictionary supplied and no size, on-site energies or positions given. "
                "Cannot determine the size of the system."
            )
        if dim is not None:
            self.dim = dim
        elif pos is not None:
            self.dim = len(pos[0])
        elif hop:
            self.dim = len(next(iter(hop.keys())))
        elif uc is not None:
            self.dim = len(uc[0])
        else:
            raise ValueError(
                "No dimension specified and no positions, hoppings, or unit cell are given. "
                "The dimensionality of the system cannot be determined."
            )
        self._zero_vec = tuple([0] * self.dim)
        self.uc = None if uc is None else np.array(uc)

        blocks = {tuple(int(x) for x in key): self._as_dense(value) for key, value in hop.items()}

        if pos is None:
            self.pos = np.zeros((self.size, self.dim))
        else:
            if len(pos) != self.size:
                raise ValueError(
                    "Invalid argument for 'pos': The number of positions must be the same as the size "
                    "(number of orbitals) of the system."
                )
            if any(len(p) != self.dim for p in pos):
                raise ValueError(
                    "Invalid argument for 'pos': The length of each position must be the same as the "
                    "dimensionality of the system."
                )
            pos_arr, blocks = self._fold_into_home_cell(np.array(pos, dtype=float), blocks)
            self.pos = pos_arr

        if contains_cc:
            blocks = self._halve_conjugate_pairs(blocks, cc_check_tolerance)
        else:
            blocks = self._fold_to_half_space(blocks)

        self.hop = _HopDict(self._empty_matrix)
        for key, mat in blocks.items():
            if np.any(mat):
                self.hop.raw_set(key, self._matrix_type(mat))
        if on_site is not None:
            if len(on_site) != self.size:
                raise ValueError(
                    "The number of on-site energies {} does not match the size of the system {}".format(
                        len(on_site), self.size
                    )
                )
            self._hop_add(self._zero_vec, 0.5 * self._matrix_type(np.diag(np.array(on_site, dtype=complex))))

        for mat in self._hop_values():
            if mat.shape != (self.size, self.size):
                raise ValueError(
                    "Hopping matrix of shape {0} found, should be ({1},{1}).".format(mat.shape, self.size)
                )
        for key in self.hop.keys():
            if len(key) != self.dim:
                raise ValueError(
                    "The length of R = {} does not match the dimensionality of the system ({})".format(key, self.dim)
                )
        if self.uc is not None and self.uc.shape != (self.dim, self.dim):
            raise ValueError(
                "Inconsistend dimension of the unit cell: {}, does not match the dimensionality of the "
                "system ({})".format(self.uc.shape, self.dim)
            )
        self.occ = None if occ is None else int(occ)

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def _as_dense(value):
        if hasattr(value, "toarray"):
            return np.asarray(value.toarray(), dtype=complex)
        return np.array(value, dtype=complex)

    def _fold_into_home_cell(self, pos, blocks):
        """
        Orbitals outside ``[0, 1)^dim`` are moved into the home cell and every hopping that touches
        them is re-labelled ``R -> R + shift[col] - shift[row]`` (``_tb_model.py:221-245``).
        """
        shift = np.floor(pos).astype(int)
        if not shift.any():
            return pos, blocks
        moved = co.defaultdict(lambda: np.zeros((self.size, self.size), dtype=complex))
        for key, mat in blocks.items():
            rows, cols = np.nonzero(mat)
            for i, j in zip(rows, cols):
                new_key = tuple(int(x) for x in (np.array(key, dtype=int) + shift[j] - shift[i]))
                moved[new_key][i, j] += mat[i, j]
        return pos % 1, dict(moved)

    @staticmethod
    def _halve_conjugate_pairs(blocks, tolerance):
        """
        ``contains_cc=True`` input lists both ``R`` and ``-R``: check ``hop[-R] == hop[R]^H``, keep the
        average on the half-space and HALF of it at ``R = 0`` (``_tb_model.py:247-279``).
        """
        kept = {}
        bad = []
        for key, mat in blocks.items():
            minus = tuple(-x for x in key)
            partner = blocks[minus].conj().T if minus in blocks else np.zeros_like(mat)
            delta = np.linalg.norm(mat - partner)
            if delta > tolerance:
                bad.append((key, delta))
            mean = (mat + partner) / 2
            lead = _first_nonzero(key)
            if lead > 0:
                kept[key] = mean
            elif lead == 0:
                kept[key] = mean / 2
        if bad:
            bad.sort(key=lambda item: -item[1])
            raise ValueError(
                "The provided hoppings do not correspond to a hermitian Hamiltonian. "
                "hoppings[-R] = hoppings[R].H is not fulfilled for the following values:\n"
                + "\n".join("R={}, delta_norm={}".format(key, delta) for key, delta in bad)
            )
        return kept

    def _fold_to_half_space(self, blocks):
        """``contains_cc=False``: a block at ``-R`` is stored as its conjugate transpose at ``+R`` (:281-298)."""
        folded = {}

        def accumulate(key, mat):
            folded[key] = folded[key] + mat if key in folded else mat.copy()

        for key, mat in blocks.items():
            lead = _first_nonzero(key)
            if lead > 0:
                accumulate(key, mat)
            elif lead < 0:
                accumulate(tuple(-x for x in key), mat.conj().T)
            else:
                accumulate(key, 0.5 * mat + 0.5 * mat.conj().T)
        return folded

    @classmethod
    def from_hop_list(cls, *, hop_list=(), size=None, **kwargs):
        """
        Build a model from ``(t, orbital_1, orbital_2, R)`` terms (``_tb_model.py:332-397``); repeated
        ``(orbital_1, orbital_2, R)`` entries add up.
        """
        if size is None:
            if "on_site" not in kwargs:
                raise ValueError(
                    "No on-site energies and no size given. The size of the system cannot be determined."
                )
            size = len(kwargs["on_site"])
        blocks = {}
        for amplitude, i, j, r_vec in hop_list:
            key = tuple(int(x) for x in r_vec)
            if key not in blocks:
                blocks[key] = np.zeros((size, size), dtype=complex)
            blocks[key][i, j] += amplitude
        return cls(size=size, hop=blocks, **kwargs)

    @classmethod
    def from_packed(cls, r_vec, hop, pos=None, **kwargs):
        """
        Build a model from packed half-space arrays ``R (n_r, dim)`` / ``hop (n_r, N, N)`` (the layout of
        the golden fixtures and of ``tbmodels_amd.synthetic``); equivalent to
        ``Model(hop={R: mat}, contains_cc=False, ...)``.
        """
        r_vec = np.asarray(r_vec)
        hop = np.asarray(hop)
        kwargs.setdefault("dim", r_vec.shape[1] if r_vec.ndim == 2 else None)
        if hop.ndim == 3 and hop.shape[0]:
            kwargs.setdefault("size", hop.shape[1])
        blocks = {tuple(int(x) for x in r): h for r, h in zip(r_vec, hop)}
        return cls(hop=blocks, pos=pos, contains_cc=False, **kwargs)

    @classmethod
    def from_wannier_files(
        cls,
        *,
        hr_file,
        wsvec_file=None,
        xyz_file=None,
        win_file=None,
        h_cutoff=0.0,
        ignore_orbital_order=False,
        pos_kind="wannier",
        distance_ratio_threshold=3.0,
        **kwargs,
    ):
        """
        Build a model from Wannier90 output (``*_hr.dat`` and optionally ``*_wsvec.dat``, ``*_centres.xyz``,
        ``*.win``): same keywords and results as ``tbmodels.Model.from_wannier_files``
        (``_tb_model.py:565-715``); the files are parsed array-wise by :mod:`tbmodels_amd.wannier`.
        """
        from . import wannier  # pylint: disable=import-outside-toplevel

        if win_file is not None:
            if "uc" in kwargs:
                raise ValueError(
                    "Ambiguous unit cell: It can be given either via 'uc' or the 'win_file' keywords, but not both."
                )
            kwargs["uc"] = wannier.read_win(win_file)["unit_cell_cart"]
        if xyz_file is not None:
            if "pos" in kwargs:
                raise ValueError(
                    "Ambiguous orbital positions: The positions can be given either via the 'pos' or the "
                    "'xyz_file' keywords, but not both."
                )
            if "uc" not in kwargs:
                raise ValueError(
                    "Positions cannot be read from .xyz file without unit cell given: Transformation from "
                    "cartesian to reduced coordinates not possible. Specify the unit cell using one of the "
                    "keywords 'uc' or 'win_file'."
                )
            kwargs["pos"] = wannier.positions_from_xyz(
                xyz_file, kwargs["uc"], pos_kind=pos_kind, distance_ratio_threshold=distance_ratio_threshold
            )
        num_wann, blocks = wannier.hop_blocks_from_wannier(
            hr_file, wsvec_file=wsvec_file, h_cutoff=h_cutoff, ignore_orbital_order=ignore_orbital_order
        )
        return cls(size=num_wann, hop=blocks, **kwargs)

    # ------------------------------------------------------------------ mutators (reference API)
    def add_hop(self, overlap, orbital_1, orbital_2, R):
        """
        Add ``<orbital_1, 0| H |orbital_2, R> = overlap``; the conjugate term is implied
        (``_tb_model.py:1153-1215``): a negative-half-space ``R`` is stored conjugated at ``-R`` and an
        ``R = 0`` term is split symmetrically.
        """
        R = tuple(R)
        if len(R) != self.dim:
            raise ValueError(
                "Dimension of R ({}) does not match the model dimension ({})".format(len(R), self.dim)
            )
        overlap = complex(overlap)
        mat = np.zeros((self.size, self.size), dtype=complex)
        lead = _first_nonzero(R)
        if lead == 0:
            mat[orbital_1, orbital_2] += overlap / 2.0
            mat[orbital_2, orbital_1] += overlap.conjugate() / 2.0
        elif lead > 0:
            mat[orbital_1, orbital_2] += overlap
        else:
            R = tuple(-x for x in R)
            mat[orbital_2, orbital_1] += overlap.conjugate()
        self._hop_add(R, self._matrix_type(mat))

    def add_on_site(self, on_site):
        """Add to the on-site energies (``_tb_model.py:1217-1234``)."""
        if self.size != len(on_site):
            raise ValueError(
                "The number of on-site energy terms should be {}, but is {}.".format(self.size, len(on_site))
            )
        for orbital, energy in enumerate(on_site):
            self.add_hop(energy / 2.0, orbital, orbital, self._zero_vec)

    def _empty_matrix(self):
        return self._matrix_type(np.zeros((self.size, self.size), dtype=complex))

    # ``self.hop`` as the model itself reads and edits it (see ``_HopDict``); a plain dict assigned by the
    # caller (``model.hop = {...}``) works too and is simply always content-hashed.
    def _hop_items(self):
        hop = self.hop
        return hop.raw_items() if isinstance(hop, _HopDict) else hop.items()

    def _hop_values(self):
        hop = self.hop
        return hop.raw_values() if isinstance(hop, _HopDict) else hop.values()

    def _hop_set(self, key, value):
        hop = self.hop
        if isinstance(hop, _HopDict):
            hop.raw_set(key, value)
        else:
            hop[key] = value

    def _hop_add(self, key, mat):
        hop = self.hop
        if isinstance(hop, _HopDict):
            hop.raw_set(key, hop.raw_get(key) + mat)
        else:
            hop[key] = hop[key] + mat if key in hop else self._empty_matrix() + mat

    def set_sparse(self, sparse=True):
        """Switch the storage of ``hop`` between dense arrays and CSR (``_tb_model.py:1294-1321``)."""
        if getattr(self, "_sparse", None) == sparse:
            return
        self._sparse = sparse
        self._matrix_type = _csr if sparse else np.array
        if hasattr(self, "hop"):
            for key, mat in list(self._hop_items()):
                self._hop_set(key, self._matrix_type(self._as_dense(mat) if sparse else np.array(mat)))

    def _array_cast(self, mat):
        return np.array(mat) if self._sparse else mat

    # ------------------------------------------------------------------ pickling
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_handles"] = []
        state["_staged_fingerprint"] = None
        state.pop("_call_lock", None)
        state.pop("_handle_array_cache", None)  # (a ctypes array of device handles: neither picklable nor valid elsewhere)
        return state

    def __setstate__(self, state):
        # (the hop dict's exposure flag / edit counter are NOT touched here: copy.copy(model) shares the dict with
        # the original, whose handed-out references stay live; a pickled dict resets itself in _HopDict.__reduce__)
        state = dict(state)
        state.pop("_handle", None)  # states written before a model could sit on several devices
        legacy_device = state.pop("device", None)
        self.__dict__.update(state)
        self._handles = []
        if "_devices" not in state:
            self._devices = [int(legacy_device)] if legacy_device is not None else _devices_from_env()
        self._call_lock = threading.RLock()

    def __del__(self):
        self._drop_staging()

    # ------------------------------------------------------------------ staging
    @property
    def device(self):
        """The (first) GPU this model evaluates on; assigning an index makes it the only one."""
        return self.devices[0]

    @device.setter
    def device(self, index):
        self.devices = [int(index)]

    @property
    def devices(self):
        """
        GPU indices this model is staged on.  With more than one, ``hamilton`` / ``eigenval`` cut the k list into
        contiguous slabs, one per entry, and every device fills its rows of the result (``tbk_eigenval_multi``): the
        same single call of a single process as in the reference (``_tb_model.py:1134-1150``), no launcher.  Default:
        ``TBK_DEVICES=0,1,...`` (or ``TBK_DEVICE=i``, or device 0).  An index may repeat (several staged copies on one GPU).
        """
        return list(self._devices)

    @devices.setter
    def devices(self, indices):
        indices = [int(i) for i in indices]
        if not indices or any(i < 0 for i in indices):
            raise ValueError("devices must be a non-empty list of GPU indices, got {!r}".format(indices))
        if indices != getattr(self, "_devices", None):
            self._drop_staging()
        self._devices = indices

    @property
    def _handle(self):
        return self._handles[0] if self._handles else None

    def _drop_staging(self):
        handles, self._handles = getattr(self, "_handles", []), []
        for handle in handles:
            try:
                _lib.lib().tbk_model_destroy(handle)
            except Exception:  # pylint: disable=broad-except  # interpreter shutdown
                pass
        self._staged_fingerprint = None

    def pin_staging(self, pinned=True):
        """
        Skip the per-call content check of ``self.hop`` (it costs one pass over the hopping bytes).
        The caller promises not to mutate the model while pinned.
        """
        self._pinned = bool(pinned)

    def _staging_key(self):
        """What the staged copy is valid for: the edit counter while no matrix has left the model (exact, and
        no pass over the bytes), the content fingerprint afterwards."""
        hop = self.hop
        if isinstance(hop, _HopDict) and not hop.exposed:
            return ("version", hop.version, tuple(self._devices), self.size, self.dim, bool(self._sparse))
        return self._fingerprint()

    def _fingerprint(self):
        running = _xxhash.xxh3_64() if _xxhash is not None else 1
        meta = [*self._devices, -1, self.size, self.dim, int(self._sparse), len(self.hop)]
        for key, mat in self._hop_items():
            meta.extend(key)
            if self._sparse:
                running = _hash_bytes(running, mat.indptr)
                running = _hash_bytes(running, mat.indices)
                running = _hash_bytes(running, mat.data)
            else:
                running = _hash_bytes(running, mat)
        digest = running.intdigest() if _xxhash is not None else running
        return (tuple(meta), digest)

    def packed_hop(self):
        """
        ``self.hop`` as the arrays the C ABI takes: ``R int32 (n_r, dim)`` plus either
        ``hop complex128 (n_r, N, N)`` (dense) or ``(r_ptr int64, row int32, col int32, val complex128)``.
        """
        stored = list(self._hop_items())
        keys = [key for key, _ in stored]
        r_vec = np.array(keys, dtype=np.int32).reshape(len(keys), self.dim)
        if not self._sparse:
            hop = np.empty((len(keys), self.size, self.size), dtype=np.complex128)
            for idx, (_, mat) in enumerate(stored):
                hop[idx] = mat
            return r_vec, hop
        r_ptr = [0]
        rows, cols, vals = [], [], []
        for _, mat in stored:
            coo = mat.tocoo()
            rows.append(coo.row.astype(np.int32))
            cols.append(coo.col.astype(np.int32))
            vals.append(coo.data.astype(np.complex128))
            r_ptr.append(r_ptr[-1] + coo.nnz)
        cat = lambda parts, dtype: np.ascontiguousarray(np.concatenate(parts) if parts else np.zeros(0, dtype), dtype)
        return r_vec, (np.array(r_ptr, dtype=np.int64), cat(rows, np.int32), cat(cols, np.int32), cat(vals, np.complex128))

    def _staged(self):
        """The ``tbk_model*`` on the first device for the current contents of ``self.hop`` (re-staged when they changed)."""
        return self._staged_all()[0]

    def _staged_all(self):
        """One ``tbk_model*`` per entry of ``self.devices`` for the current contents of ``self.hop``."""
        if self._handles and self._pinned:
            return self._handles
        fingerprint = self._staging_key()
        if self._handles and fingerprint == self._staged_fingerprint:
            return self._handles
        self._drop_staging()
        lib = _lib.lib()
        r_vec, payload = self.packed_hop()
        handles = []
        try:
            for device in self._devices:  # the hoppings are replicated: every device holds the whole model
                handle = ctypes.c_void_p()
                if self._sparse:
                    r_ptr, row, col, val = payload
                    status = lib.tbk_model_create_csr(
                        device, self.dim, self.size, len(r_vec), _lib.ptr(r_vec), _lib.ptr(r_ptr), _lib.ptr(row),
                        _lib.ptr(col), _lib.ptr(val), ctypes.byref(handle),
                    )
                else:
                    status = lib.tbk_model_create_dense(
                        device, self.dim, self.size, len(r_vec), _lib.ptr(r_vec), _lib.ptr(payload), ctypes.byref(handle)
                    )
                _lib.check(status)
                handles.append(handle)
        except Exception:
            for handle in handles:
                lib.tbk_model_destroy(handle)
            raise
        self._handles = handles
        self._staged_fingerprint = fingerprint
        return handles

    def _handle_array(self):
        handles = self._staged_all()
        cached = getattr(self, "_handle_array_cache", None)
        if cached is None or cached[0] is not handles:  # (the list object changes whenever the model is re-staged)
            cached = (handles, (ctypes.c_void_p * len(handles))(*[h.value for h in handles]))
            self._handle_array_cache = cached
        return cached[1], len(handles)

    def set_option(self, option, value):
        """Forward a ``TBK_OPT_*`` option to the staged model (see ``include/tbk.h``)."""
        with self._call_lock:
            for handle in self._staged_all():
                _lib.check(_lib.lib().tbk_model_set_option(handle, option, int(value)))

    # ------------------------------------------------------------------ the hot path
    def _k_array(self, k):
        """``np.array(k, ndmin=1)``; 1-D (or scalar) means one k-point (``_tb_model.py:1103-1108``)."""
        k_array = np.asarray(k)  # np.array(k, ndmin=1) without its copy: k is only read
        if k_array.ndim == 0:
            k_array = k_array.reshape(1)
        single = k_array.ndim == 1
        if single:
            k_array = k_array.reshape((1, -1))
        k_array = np.ascontiguousarray(k_array, dtype=np.float64)
        if k_array.ndim != 2 or k_array.shape[1] != self.dim:
            # the reference fails inside np.dot(k_array, R) with a shape ValueError
            raise ValueError(
                "shapes {} and ({},) not aligned: k-point dimension does not match the model".format(
                    k_array.shape, self.dim
                )
            )
        return k_array, single

    def hamilton(self, k, convention=2):
        """
        The Hamilton matrix at one k-point (returns ``(size, size)``) or a list of k-points
        (``(NK, size, size)``), complex128; ``convention`` 1 or 2 as in PythTB
        (``_tb_model.py:1076-1132``).
        """
        if convention not in [1, 2]:
            raise ValueError(
                "Invalid value '{}' for 'convention': must be either '1' or '2'".format(convention)
            )
        k_array, single = self._k_array(k)
        n_k = k_array.shape[0]
        out = _outbuf.empty((n_k, self.size, self.size), np.complex128)
        pos = np.ascontiguousarray(self.pos, dtype=np.float64) if convention == 1 else None
        with self._call_lock:  # (re)staging and the call are one step for other threads (ctypes drops the GIL)
            handles, n_handles = self._handle_array()
            _lib.check(
                _lib.lib().tbk_hamilton_multi(handles, n_handles, _lib.ptr(k_array), n_k, int(convention), _lib.ptr(pos),
                                              _lib.ptr(out))
            )
        return out[0] if single else out

    def eigenval(self, k):
        """
        Ascending eigenvalues at one k-point (1-D array) or a list of k-points (a list of 1-D arrays,
        like the reference: ``_tb_model.py:1134-1150``).
        """
        out = self.eigenval_array(k)
        return out if out.ndim == 1 else list(out)

    def eigenval_array(self, k):
        """
        ``eigenval`` without the list: one ``(NK, N)`` array for a list of k-points (``(N,)`` for one k-point).  Not in
        the reference; building the list of row views costs ~80 ns per k-point, more than the GPU work for small
        models (500 000 k-points of an 8-orbital model: 4.6 ms of kernels, 40 ms of ``list(out)``).
        """
        k_array, single = self._k_array(k)
        n_k = k_array.shape[0]
        out = _outbuf.empty((n_k, self.size), np.float64)
        with self._call_lock:
            # NaN / Inf in k or in the hoppings reach the eigenvalues; the library checks those on the device and
            # returns TBK_ERR_NOT_FINITE -> ValueError, scipy.linalg.eigvalsh(check_finite=True)'s answer to the
            # non-finite Hamiltonian (two np.isfinite passes here cost as much as the kernels for small models)
            handles, n_handles = self._handle_array()
            _lib.check(_lib.lib().tbk_eigenval_multi(handles, n_handles, _lib.ptr(k_array), n_k, _lib.ptr(out)))
        return out[0] if single else out

    def construct_kdotp(self, k, order):
        """
        k.p model around the k-point ``k``: the Taylor expansion of H(k) (convention 2) up to total power
        ``order``, evaluated on the GPU (``Model.construct_kdotp``, ``_tb_model.py:942-982``).
        """
        import itertools  # pylint: disable=import-outside-toplevel
        import math  # pylint: disable=import-outside-toplevel

        from .kdotp import KdotpModel  # pylint: disable=import-outside-toplevel

        if order < 0:
            raise ValueError("The order for the k.p model must be positive.")
        k0 = np.ascontiguousarray(np.array(k, ndmin=1), dtype=np.float64)
        if k0.shape != (self.dim,):
            raise ValueError("k has shape {} but the model has dimension {}".format(k0.shape, self.dim))
        powers = [p for p in itertools.product(range(order + 1), repeat=self.dim) if sum(p) <= order]
        pw = np.array(powers, dtype=np.int32).reshape(len(powers), self.dim)
        pref = np.array(
            [(2j * np.pi) ** sum(p) / np.prod([math.factorial(x) for x in p]) for p in powers], dtype=np.complex128
        )
        source = self
        if self._sparse:  # the derivative kernel reads the dense staged operand
            r_vec, _ = self.packed_hop()
            dense = np.stack([np.array(m) for m in self._hop_values()]) if len(self.hop) else np.zeros((0, self.size, self.size))
            source = Model.from_packed(r_vec, dense, size=self.size, dim=self.dim)
            source.device = self.device
        coeffs = np.empty((len(powers), self.size, self.size), dtype=np.complex128)
        with source._call_lock:
            _lib.check(
                _lib.lib().tbk_kdotp_coefficients(
                    source._staged(), _lib.ptr(k0), len(powers), _lib.ptr(pw), _lib.ptr(pref), _lib.ptr(coeffs)
                )
            )
        return KdotpModel(taylor_coefficients={p: coeffs[i] for i, p in enumerate(powers)})

    # ------------------------------------------------------------------ HDF5 wire format
    @classmethod
    def from_hdf5_file(cls, hdf5_file, **kwargs):
        """
        Load a model stored by ``to_hdf5_file`` -- of this package or of the reference
        (``_tb_model.py:984-1011``).  Explicit keyword arguments take precedence over the file's.
        """
        from . import hdf5_lite  # pylint: disable=import-outside-toplevel

        tree = hdf5_lite.read(hdf5_file)
        if "type_tag" not in tree:
            warnings.warn(
                "The loaded file '{}' is stored in an outdated format. Consider loading and storing the "
                "file to update it.".format(hdf5_file),
                DeprecationWarning,
            )
        return cls.from_hdf5(tree, **kwargs)

    @classmethod
    def from_hdf5(cls, tree, **kwargs):
        """Build a model from the (nested dict) content of a model file (``_tb_model.py:1013-1036``)."""
        group = tree.get("tb_model", tree)  # a development version wrote a top-level 'tb_model' group
        new_kwargs = {"hop": {}}
        for key in ("uc", "occ", "size", "dim", "pos", "sparse"):
            if key in group:
                value = group[key]
                new_kwargs[key] = value.item() if isinstance(value, np.generic) else value
        if "hop" not in kwargs:
            sparse = bool(new_kwargs.get("sparse", False))
            for entry in group.get("hop", {}).values():
                r_vec = tuple(int(x) for x in entry["R"])
                if sparse:
                    new_kwargs["hop"][r_vec] = _csr(
                        (entry["data"], entry["indices"], entry["indptr"]),
                        shape=tuple(int(x) for x in entry["shape"]),
                    )
                else:
                    new_kwargs["hop"][r_vec] = np.array(entry["mat"])
            new_kwargs["contains_cc"] = False
        new_kwargs.update(kwargs)
        return cls(**new_kwargs)

    def to_hdf5(self):
        """The model as the nested dict ``hdf5_lite.write`` stores (``_tb_model.py:1038-1058``)."""
        tree = {"type_tag": "tbmodels.model"}
        if self.uc is not None:
            tree["uc"] = np.asarray(self.uc, dtype=float)
        if self.occ is not None:
            tree["occ"] = np.int64(self.occ)
        tree["size"] = np.int64(self.size)
        tree["dim"] = np.int64(self.dim)
        tree["pos"] = np.asarray(self.pos, dtype=float)
        tree["sparse"] = bool(self._sparse)
        hop = {}
        for i, (r_vec, mat) in enumerate(self._hop_items()):
            entry = {"R": np.array(r_vec, dtype=np.int64)}
            if self._sparse:
                mat = _csr(mat)
                entry["data"] = np.asarray(mat.data, dtype=complex)
                entry["indices"] = np.asarray(mat.indices, dtype=np.int32)
                entry["indptr"] = np.asarray(mat.indptr, dtype=np.int32)
                entry["shape"] = np.array(mat.shape, dtype=np.int64)
            else:
                entry["mat"] = np.asarray(mat, dtype=complex)
            hop[str(i)] = entry
        tree["hop"] = hop
        return tree

    def to_hdf5_file(self, hdf5_file):
        """Save the model to an HDF5 file the reference's ``Model.from_hdf5_file`` / ``io.load`` read."""
        from . import hdf5_lite  # pylint: disable=import-outside-toplevel

        hdf5_lite.write(hdf5_file, self.to_hdf5())

    # ------------------------------------------------------------------ introspection
    def timing(self, reset=True):
        """Per-stage HIP-event times of the staged model: ``{stage: (ms, launches)}`` (needs TBK_OPT_TIMING)."""
        ms = (ctypes.c_double * _lib.TBK_T_COUNT)()
        launches = (ctypes.c_int64 * _lib.TBK_T_COUNT)()
        _lib.check(_lib.lib().tbk_get_timing(self._staged(), ms, launches, int(bool(reset))))
        return {name: (ms[i], launches[i]) for i, name in enumerate(_lib.STAGE_NAMES)}

    def __repr__(self):
        return "tbmodels_amd.Model(hop=<{} matrices>, size={}, dim={}, sparse={})".format(
            len(self.hop), self.size, self.dim, self._sparse
        )
