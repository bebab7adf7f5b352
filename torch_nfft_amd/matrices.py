"""Matrix-free kernel matrices on top of ``nfft_fastsum``: the public classes of the reference's
``torch_nfft/matrices.py`` (``GramMatrix``, ``AdjacencyMatrix``; same constructor arguments, ``@`` / ``.T`` /
``row_sums`` / ``column_sums`` / ``to_dense`` / ``is_symmetric``), written from the operator algebra:

    Gram       K x        with K[i, j] = kernel(source_j - target_i), one fastsum per product
    adjacency  A = L (K + c I) R     with diagonal L, R chosen by the normalisation
               y = A x                         no shift
               y = S x - A x   /   S x + A x   Laplacian / signless Laplacian, S = D (unnormalised) or I (normalised)

where D = diag((K + c I) 1) are the node degrees.  Normalisations: "sym" L = R = D^-1/2; "left" (alias "rw")
L = D^-1; "right" R = D^-1.  The transpose of an adjacency operator swaps L and R and keeps everything else, so it
is built from the parts without another degree computation.

Two defects of the reference are not reproduced: its ``GramMatrix.is_symmetric`` compares ``sources`` with itself
(matrices.py:65) and its shift step reads an undefined name (matrices.py:149).
"""
import warnings

import torch

from .nfft import nfft_fastsum

_NORMALIZATIONS = {"none": "none", "sym": "sym", "left": "left", "rw": "left", "right": "right"}
_SHIFT_SIGN = {"none": 0, "laplacian": -1, "signless": +1}


def _scale_rows(diag, x):
    """diag(diag) @ x for x of shape [n, *cols]; ``diag`` None is the identity."""
    if diag is None:
        return x
    return diag.reshape((-1,) + (1,) * (x.dim() - 1)) * x


class AbstractMatrix:
    """A linear map given by its action ``apply`` on ``[n_in, *cols]`` tensors.  ``shape`` follows the reference:
    ``(number of sources, number of targets)`` = (input length, output length)."""

    def __init__(self, shape, device):
        self.shape = tuple(shape)
        self.device = device

    # -- to be provided by subclasses
    def apply(self, x):
        raise NotImplementedError()

    def is_symmetric(self):
        return False

    def transpose(self):
        if not self.is_symmetric():
            raise NotImplementedError()
        return self

    # -- derived
    def __matmul__(self, x):
        return self.apply(x)

    @property
    def T(self):
        return self.transpose()

    def _probe(self, make):
        return self.apply(make(self.shape[0], device=self.device))

    def row_sums(self):
        return self._probe(torch.ones)

    def column_sums(self):
        return self.transpose().row_sums()

    def to_dense(self):
        return self._probe(torch.eye)


class GramMatrix(AbstractMatrix):
    def __init__(self, coeffs, sources, targets=None, source_batch=None, target_batch=None, /, batch=None, cutoff=3):
        if batch is not None:
            source_batch, target_batch = batch, batch
        elif targets is None:
            target_batch = source_batch
        if targets is None:
            targets = sources
        self.coeffs, self.cutoff = coeffs, cutoff
        self.sources, self.source_batch = sources, source_batch
        self.targets, self.target_batch = targets, target_batch
        super().__init__((sources.shape[0], targets.shape[0]), sources.device)

    def apply(self, x):
        return nfft_fastsum(x, self.coeffs, self.sources, self.targets, self.source_batch, self.target_batch,
                            cutoff=self.cutoff)

    def is_symmetric(self):
        return self.targets is self.sources and self.target_batch is self.source_batch

    def transpose(self):
        if self.is_symmetric():
            return self
        return GramMatrix(self.coeffs, self.targets, self.sources, self.target_batch, self.source_batch,
                          cutoff=self.cutoff)


class AdjacencyMatrix(AbstractMatrix):
    def __init__(self, gram_matrix, diagonal_offset=0, normalization=None, shift=None, degree_threshold=0):
        if not gram_matrix.is_symmetric():
            raise ValueError("The underlying Gram matrix of an AdjacencyMatrix must be symmetric")
        kind = "none" if normalization is None else str(normalization).lower()
        if kind not in _NORMALIZATIONS:
            raise ValueError(f"Unknown AdjacencyMatrix normalization type: {kind}")
        kind = _NORMALIZATIONS[kind]
        shift = "none" if shift is None else str(shift).lower()
        if shift not in _SHIFT_SIGN:
            raise ValueError(f"Unknown AdjacencyMatrix shift type: {shift}")
        super().__init__(gram_matrix.shape, gram_matrix.device)
        self.gram_matrix = gram_matrix
        self.diagonal_offset = diagonal_offset
        self.normalization = kind
        self.shift = shift
        self._left = self._right = self._shift_diag = None  # None = identity
        if kind == "none" and shift == "none":
            return
        degrees = gram_matrix.row_sums() + diagonal_offset
        if kind == "none":
            self._shift_diag = degrees  # unnormalised (signless) Laplacian: D -+ W
            return
        isolated = degrees < degree_threshold
        n_isolated = int(isolated.sum())
        if n_isolated:
            warnings.warn(f"AdjacencyMatrix with normalization: {n_isolated} out of {degrees.numel()} node degrees "
                          f"are smaller than the threshold {degree_threshold:.4g}", RuntimeWarning, stacklevel=2)
            degrees = torch.where(isolated, torch.full_like(degrees, float("inf")), degrees)  # their rows become 0
        if kind == "sym":
            self._left = self._right = degrees.rsqrt()
        elif kind == "left":
            self._left = degrees.reciprocal()
        else:
            self._right = degrees.reciprocal()

    # names the reference exposes (matrices.py:120-128, 130-151): the degree vector and its inverse powers, the two
    # scaling steps and the shift step.  (normalization "rw" is stored as "left", the same operator D^-1 W.)
    @property
    def degrees(self):
        return self._shift_diag

    @property
    def d_inv(self):
        if self.normalization == "left":
            return self._left
        return self._right if self.normalization == "right" else None

    @property
    def d_inv_sqrt(self):
        return self._left if self.normalization == "sym" else None

    def apply_shift(self, x, y):
        """y = A x -> x-dependent shift: D x -+ y (unnormalised) or x -+ y (normalised); ``shift`` "none": y."""
        sign = _SHIFT_SIGN[self.shift]
        if sign == 0:
            return y
        base = _scale_rows(self._shift_diag, x)
        return base + y if sign > 0 else base - y

    def apply_left_normalization(self, x):
        return _scale_rows(self._left, x)

    def apply_right_normalization(self, x):
        return _scale_rows(self._right, x)

    def apply(self, x):
        z = _scale_rows(self._right, x)
        y = self.gram_matrix.apply(z)
        if self.diagonal_offset != 0:
            y = y + self.diagonal_offset * z
        return self.apply_shift(x, _scale_rows(self._left, y))

    def is_symmetric(self):
        return self._left is self._right  # both None, or the same D^-1/2

    def transpose(self):
        if self.is_symmetric():
            return self
        other = object.__new__(AdjacencyMatrix)
        other.__dict__.update(self.__dict__)
        other._left, other._right = self._right, self._left
        other.normalization = "right" if self.normalization == "left" else "left"
        return other
