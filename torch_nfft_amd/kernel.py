"""``GaussianKernel``: fast Gram / adjacency matrices of a Gaussian kernel (reference: ``torch_nfft/kernel.py``).

Two modes, as in the reference: with a known bound on the point norms the points are scaled by a fixed factor and
the matrix belongs to ``exp(-|z|^2 / sigma^2)``; without one every point set is scaled by its own radius rho and
the matrix belongs to ``exp(-|z|^2 / (rho sigma)^2)``."""
import math

from .coeffs import gaussian_analytic_coeffs, gaussian_interpolated_coeffs
from .matrices import AdjacencyMatrix, GramMatrix
from .utils import compute_points_center, compute_points_radius


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

    def _into_torus(self, sets):
        """The affine map p -> s_b (p - c_b) that takes the point sets of a problem into the ball the kernel coefficients
        were computed for.  ``sets`` = [(points, batch), ...] (sources, and targets when they are separate); centre c_b and
        scale s_b are those of point set b over ALL its points, sources and targets alike (kernel.py:99-117)."""
        points = [p for p, _ in sets]
        batches = [b for _, b in sets]
        tgt, tgt_b = (points[1], batches[1]) if len(sets) == 2 else (None, None)

        def per_set(value, b):  # a per-set quantity [B, ...] or a scalar, laid out along the points of batch vector b
            return value if b is None else value[b]

        if self.shift_by_center:
            c = compute_points_center(points[0], tgt, batches[0], tgt_b)
            points = [p - per_set(c, b) for p, b in zip(points, batches)]
        if self.scale_by_norm is None:
            return [self.factor * p for p in points]
        r = compute_points_radius(points[0], points[1] if tgt is not None else None, batches[0], tgt_b,
                                  norm=self.scale_by_norm)
        s = self.factor / r
        return [p * (s if b is None else s[b].unsqueeze(-1)) for p, b in zip(points, batches)]

    def gram_matrix(self, sources, targets=None, source_batch=None, target_batch=None, /, batch=None):
        """K[i, j] = kernel(source_j - target_i) as a matrix-free operator; ``targets=None``: the symmetric matrix of
        one point set (the GramMatrix then shares one point plan between its two halves)."""
        if batch is not None:
            source_batch = target_batch = batch
        sets = [(sources, source_batch)]
        if targets is not None:
            sets.append((targets, target_batch))
        mapped = self._into_torus(sets)
        return GramMatrix(self.coeffs, mapped[0], mapped[1] if targets is not None else None, source_batch,
                          target_batch, cutoff=self.cutoff)

    def __call__(self, *args, **kwargs):
        return self.gram_matrix(*args, **kwargs)

    def adjacency_matrix(self, sources, batch=None, loop_weight=1, normalization=None, shift=None, degree_threshold=0):
        """Adjacency / Laplacian operator of the kernel graph on one point set; self-loops weigh ``loop_weight``
        (the kernel's own diagonal is 1, so the Gram diagonal is offset by ``loop_weight - 1``)."""
        gram = self.gram_matrix(sources, None, batch, batch)
        return AdjacencyMatrix(gram, diagonal_offset=loop_weight - 1, normalization=normalization, shift=shift,
                               degree_threshold=degree_threshold)
