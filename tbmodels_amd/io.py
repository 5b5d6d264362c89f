"""
Loading and saving in the reference's HDF5 wire formats (`src/tbmodels/io.py:17-40`), and the two
``bands_inspect`` containers the ``eigenvals`` command exchanges (`src/tbmodels/_cli.py:243-262`).

Files are read and written with :mod:`tbmodels_amd.hdf5_lite` (the target image has no ``h5py``); the
layouts are the ones ``fsc.hdf5_io`` produces, so files travel between the two packages:

=================================  ==================================================================
``type_tag``                       content
=================================  ==================================================================
``tbmodels.model``                 ``uc``, ``occ``, ``size``, ``dim``, ``pos``, ``sparse``, ``hop/<i>/{R, mat}`` or
                                   ``hop/<i>/{R, data, indices, indptr, shape}`` (`_tb_model.py:1038-1058`)
``[bands_inspect.]kpoints_explicit``  ``kpoints`` ``(NK, dim)`` float64
``bands_inspect.eigenvals_data``   ``kpoints_obj`` (a k-points object), ``eigenvals`` ``(NK, N)`` float64
``tbmodels.kdotp_model``           ``taylor_coefficients`` (``kdotp.py:19-36``; dict of power tuple -> matrix)
(no tag, ``hop`` present)          legacy model file: loaded with a ``DeprecationWarning`` (`io.py:30-40`)
=================================  ==================================================================
"""

import warnings

import numpy as np

from . import hdf5_lite

__all__ = ("load", "save", "KpointsExplicit", "EigenvalsData")


class KpointsExplicit:
    """An explicit list of k-points in reduced coordinates (``bands_inspect.kpoints.KpointsExplicit``)."""

    def __init__(self, kpoints):
        kpoints = np.array(kpoints, dtype=float)
        if kpoints.ndim != 2:
            raise ValueError("kpoints must be a list of k-points (2D array), got shape {}".format(kpoints.shape))
        self.kpoints = kpoints

    @property
    def kpoints_explicit(self):
        return self.kpoints

    def to_hdf5(self):
        return {"type_tag": "bands_inspect.kpoints_explicit", "kpoints": self.kpoints}

    @classmethod
    def from_hdf5(cls, tree):
        return cls(tree["kpoints"])


class EigenvalsData:
    """Eigenvalues on a list of k-points (``bands_inspect.eigenvals.EigenvalsData``)."""

    def __init__(self, *, kpoints, eigenvals):
        if not isinstance(kpoints, KpointsExplicit):
            kpoints = KpointsExplicit(kpoints)
        eigenvals = np.array(eigenvals, dtype=float)
        if len(kpoints.kpoints) != len(eigenvals):
            raise ValueError(
                "Number of kpoints ({}) does not match the number of eigenvalue lists ({})".format(
                    len(kpoints.kpoints), len(eigenvals)
                )
            )
        self.kpoints = kpoints
        self.eigenvals = eigenvals

    @classmethod
    def from_eigenval_function(cls, *, kpoints, eigenval_function, listable=False):
        """``listable=True``: ONE call with the whole k list (what the ``eigenvals`` command does)."""
        if not isinstance(kpoints, KpointsExplicit):
            kpoints = KpointsExplicit(kpoints)
        if listable:
            eigenvals = eigenval_function(kpoints.kpoints_explicit)
        else:
            eigenvals = [eigenval_function(k) for k in kpoints.kpoints_explicit]
        return cls(kpoints=kpoints, eigenvals=eigenvals)

    def to_hdf5(self):
        return {
            "type_tag": "bands_inspect.eigenvals_data",
            "kpoints_obj": self.kpoints.to_hdf5(),
            "eigenvals": self.eigenvals,
        }

    @classmethod
    def from_hdf5(cls, tree):
        return cls(kpoints=_decode(tree["kpoints_obj"]), eigenvals=tree["eigenvals"])


def _decode(tree):
    from ._model import Model  # pylint: disable=import-outside-toplevel

    tag = tree.get("type_tag") if isinstance(tree, dict) else None
    if tag == "tbmodels.model":
        return Model.from_hdf5(tree)
    if tag == "tbmodels.kdotp_model":
        from .kdotp import KdotpModel  # pylint: disable=import-outside-toplevel

        return KdotpModel.from_hdf5(tree)
    if tag in ("kpoints_explicit", "bands_inspect.kpoints_explicit"):
        return KpointsExplicit.from_hdf5(tree)
    if tag in ("eigenvals_data", "bands_inspect.eigenvals_data"):
        return EigenvalsData.from_hdf5(tree)
    if tag is None and isinstance(tree, dict) and ("hop" in tree or "tb_model" in tree):
        warnings.warn(
            "The loaded file is stored in an outdated format. Consider loading and storing the file to update it.",
            DeprecationWarning,
        )
        return Model.from_hdf5(tree)
    raise ValueError("cannot decode HDF5 content with type_tag {!r}".format(tag))


def load(file_path):
    """Load a model, a k-point list or an eigenvalue table from an HDF5 file."""
    return _decode(hdf5_lite.read(file_path))


def save(obj, file_path):
    """Save an object with a ``to_hdf5()`` tree (``Model``, ``KpointsExplicit``, ``EigenvalsData``)."""
    hdf5_lite.write(file_path, obj.to_hdf5())
