"""
tbmodels_amd -- MI355X-native k-space evaluation for tight-binding models.

Drop-in for the hot path of Z2PackDev/TBmodels: ``Model.hamilton(k, convention=2)`` and
``Model.eigenval(k)`` (and the same two methods of ``KdotpModel``), evaluated on gfx950 through the
C ABI of ``include/tbk.h``.  See DESIGN.md for the path and its kernels, INTEGRATION.md for the
binding a TBmodels maintainer would add.
"""

from ._model import Model
from .kdotp import KdotpModel
from . import synthetic
from . import io

__version__ = "0.1.0"

__all__ = ("Model", "KdotpModel", "synthetic", "io")
