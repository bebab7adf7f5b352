"""Matrix-free linear operators on top of ``nfft_fastsum`` (reference: ``torch_nfft/matrices.py``).

Same classes, methods and defaults.  Two defects of the reference are not reproduced: ``GramMatrix.is_symmetric``
compares ``sources`` with itself (matrices.py:65) and ``AdjacencyMatrix.apply_shift`` reads an undefined name
(matrices.py:149)."""
import warnings

import torch

from .nfft import nfft_fastsum


class AbstractMatrix:
    def __init__(self, shape, device):
        self.shape = shape
        self.device = device

    def apply(self, x):
        raise NotImplementedError()

    def __matmul__(self, x):
        return self.apply(x)

    def is_symmetric(self):
        return False

    def transpose(self):
        if self.is_symmetric():
            return self
        raise NotImplementedError()

    @property
    def T(self):
        return self.transpose()

    def row_sums(self):
        # the operator maps R^(#sources) -> R^(#targets); shape follows the reference: (sources, targets)
        return self.apply(torch.ones(self.shape[0], device=self.device))

    def column_sums(self):
        return self.T.row_sums()

    def to_dense(self):
        return self.apply(torch.eye(self.shape[0], device=self.device))


class GramMatrix(AbstractMatrix):
    """K[i, j] = kernel(source_j - target_i), applied with nfft_fastsum (matrices.py:41-69)."""

    def __init__(self, coeffs, sources, targets=None, source_batch=None, target_batch=None, /, batch=None, cutoff=3):
        if targets is None:
            targets = sources
            target_batch = source_batch
        if batch is not None:
            source_batch = batch
            target_batch = batch
        super().__init__((sources.size(0), targets.size(0)), sources.device)
        self.coeffs = coeffs
        self.sources = sources
        self.targets = targets
        self.source_batch = source_batch
        self.target_batch = target_batch
        self.cutoff = cutoff

    def apply(self, x):
        return nfft_fastsum(x, self.coeffs, self.sources, self.targets, self.source_batch, self.target_batch,
                            cutoff=self.cutoff)

    def is_symmetric(self):
        return self.sources is self.targets and self.source_batch is self.target_batch

    def transpose(self):
        if self.is_symmetric():
            return self
        return GramMatrix(self.coeffs, self.targets, self.sources, self.target_batch, self.source_batch,
                          cutoff=self.cutoff)


class AdjacencyMatrix(AbstractMatrix):
    """Graph adjacency operator W = K + offset*I of a symmetric Gram matrix with optional degree normalisation
    ("sym", "left"/"rw", "right") and Laplacian / signless-Laplacian shift (matrices.py:73-175)."""

    def __init__(self, gram_matrix, diagonal_offset=0, normalization=None, shift=None, degree_threshold=0):
        if not gram_matrix.is_symmetric():
            raise ValueError("The underlying Gram matrix of an AdjacencyMatrix must be symmetric")
        super().__init__(gram_matrix.shape, gram_matrix.device)
        self.gram_matrix = gram_matrix
        self.diagonal_offset = diagonal_offset
        normalization = "none" if normalization is None else normalization.lower()
        if normalization == "rw":
            normalization = "left"
        if normalization not in ("none", "sym", "left", "right"):
            raise ValueError(f"Unknown AdjacencyMatrix normalization type: {normalization}")
        self.normalization = normalization
        shift = "none" if shift is None else shift.lower()
        if shift not in ("none", "laplacian", "signless"):
            raise ValueError(f"Unknown AdjacencyMatrix shift type: {shift}")
        self.shift = shift

        if shift != "none" or normalization != "none":
            degrees = gram_matrix.row_sums()
            if diagonal_offset != 0:
                degrees = degrees + diagonal_offset
            if normalization != "none":
                small = degrees < degree_threshold
                if torch.any(small):
                    warnings.warn("AdjacencyMatrix with normalization: {} out of {} node degrees are smaller than "
                                  "the threshold {:.4g}".format(int(small.sum()), degrees.numel(), degree_threshold),
                                  RuntimeWarning, stacklevel=2)
                    degrees = degrees.masked_fill(small, float("inf"))
                if normalization == "sym":
                    self.d_inv_sqrt = torch.rsqrt(degrees)
                else:
                    self.d_inv = 1 / degrees
            else:
                self.degrees = degrees

    @staticmethod
    def _rows(v, x):
        return v[(...,) + (None,) * (x.dim() - 1)] * x

    def apply_left_normalization(self, x):
        if self.normalization == "sym":
            return self._rows(self.d_inv_sqrt, x)
        if self.normalization == "left":
            return self._rows(self.d_inv, x)
        return x

    def apply_right_normalization(self, x):
        if self.normalization == "sym":
            return self._rows(self.d_inv_sqrt, x)
        if self.normalization == "right":
            return self._rows(self.d_inv, x)
        return x

    def apply_shift(self, x, y):
        if self.shift == "none":
            return y
        if self.normalization == "none":
            x = self._rows(self.degrees, x)
        if self.shift == "signless":
            return x + y
        return x - y

    def apply(self, x):
        Dx = self.apply_right_normalization(x)
        y = self.gram_matrix @ Dx
        if self.diagonal_offset != 0:
            y = y + self.diagonal_offset * Dx
        y = self.apply_left_normalization(y)
        return self.apply_shift(x, y)

    def is_symmetric(self):
        return self.normalization not in ("left", "right")

    def transpose(self):
        if self.normalization in ("left", "right"):
            # no normalization / shift arguments: they would trigger another degree computation
            t = AdjacencyMatrix(self.gram_matrix, self.diagonal_offset, normalization=None, shift=None)
            t.normalization = "right" if self.normalization == "left" else "left"
            t.shift = self.shift
            t.d_inv = self.d_inv
            return t
        return self
