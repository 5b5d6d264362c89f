"""
Array-convertible scipy sparse wrapper.

The reference keeps sparse hoppings as ``scipy.sparse`` matrices that turn into ``ndarray`` under
``np.array(...)`` (``/root/reference/src/tbmodels/_sparse_matrix.py:14-37``); ``Model.hop`` values of a
sparse model are instances of such a type, and callers rely on ``np.array(model.hop[R])``.
"""

import scipy.sparse as sp


class csr(sp.csr_matrix):  # pylint: disable=invalid-name
    """CSR matrix whose ``np.array(x)`` is the dense matrix."""

    def __array__(self, dtype=None, copy=None):  # numpy >= 2 passes dtype/copy
        dense = self.toarray()
        return dense if dtype is None else dense.astype(dtype, copy=False)

    def transpose(self, axes=None, copy=False):
        return type(self)(sp.csr_matrix(self).transpose(axes=axes, copy=copy))

    def conjugate(self, copy=True):
        return type(self)(sp.csr_matrix(self).conjugate(copy=copy))
