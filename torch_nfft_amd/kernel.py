"""``GaussianKernel``: fast Gram / adjacency matrices of a Gaussian kernel (reference: ``torch_nfft/kernel.py``).

Two modes, as in the reference: with a known bound on the point norms the points are scaled by a fixed factor and
the matrix belongs to ``exp(-|z|^2 / sigma^2)``; without one every point set is scaled by its own radius rho and
the matrix belongs to ``exp(-|z|^2 / (rho sigma)^2)``."""
import math

from .coeffs import gaussian_analytic_coeffs, gaussian_interpolated_coeffs
from .matrices import AdjacencyMatrix, GramMatrix
from .utils import scale_points_by_norm, shift_points_by_center


class GaussianKernel:
    def __init__(self, sigma, dim=3, bandwidth=16, cutoff=3, shift_by_center=True, max_euclidean_norm=None,
                 max_infinity_norm=None, analytic=False, reg_degree=-1, reg_width=0.0):
        self.cutoff = cutoff
        self.shift_by_center = shift_by_center
        self.scale_by_norm = None
        # points are mapped into a ball / cube of this radius inside the torus [-1/2, 1/2)^d (kernel.py:77)
        self.factor = 0.25 - 0.5 * reg_width
        if reg_degree < 0:
            radius = max_infinity_norm or max_euclidean_norm
            fallback = "infinity"
        else:
            radius = max_euclidean_norm
            if radius is None and max_infinity_norm is not None:
                radius = max_infinity_norm * math.sqrt(dim)
            fallback = "euclidean"
        if radius is None:
            self.scale_by_norm = fallback
        else:
            self.factor /= radius
        if analytic:
            self.coeffs = gaussian_analytic_coeffs(self.factor * sigma, dim, bandwidth)
        else:
            self.coeffs = gaussian_interpolated_coeffs(self.factor * sigma, dim, bandwidth, reg_degree, reg_width)

    def gram_matrix(self, sources, targets=None, source_batch=None, target_batch=None, /, batch=None):
        if batch is not None:
            source_batch = batch
            target_batch = batch
        same = targets is None
        if self.shift_by_center:
            sources, targets = shift_points_by_center(sources, targets, source_batch, target_batch)
        if self.scale_by_norm is not None:
            sources, targets = scale_points_by_norm(sources, targets, source_batch, target_batch,
                                                    factor=self.factor, norm=self.scale_by_norm)
        else:
            sources = self.factor * sources
            if targets is not None:
                targets = self.factor * targets
        if same:
            targets = None
        return GramMatrix(self.coeffs, sources, targets, source_batch, target_batch, cutoff=self.cutoff)

    def __call__(self, *args, **kwargs):
        return self.gram_matrix(*args, **kwargs)

    def adjacency_matrix(self, sources, batch=None, loop_weight=1, normalization=None, shift=None, degree_threshold=0):
        return AdjacencyMatrix(self.gram_matrix(sources, batch=batch), diagonal_offset=loop_weight - 1,
                               normalization=normalization, shift=shift, degree_threshold=degree_threshold)
