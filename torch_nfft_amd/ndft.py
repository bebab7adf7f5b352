"""Exact (slow) discrete transforms in plain torch, on any device -- the ground truth the fast operators are
compared with.  Same names, signatures and conventions as the reference's ``torch_nfft/ndft.py`` (frequencies
``k in [-N/2, N/2)^d`` stored at index ``k + N/2``; ``e^{+}`` for the adjoint, ``e^{-}`` for the forward transform;
trailing dimensions of ``x`` are independent columns), written for this package from those definitions:

* the d-dimensional phase ``exp(2 pi i k.p)`` is the product of d one-dimensional phase tables, so a chunk of points
  costs ``O(chunk * N * d)`` exponentials and one einsum instead of a dense ``N^d x n`` exponential;
* points are processed in chunks, so memory is ``O(N^d * chunk)`` -- the reference's formulation holds
  ``N^d * n`` complex numbers at once.

These functions are validation helpers: nothing in the fast path calls them.
"""
import math

import torch

_CHUNK = 2048


def _axis_phases(pos, N, sign):
    """List of d tensors [n, N]: exp(sign 2 pi i k pos[:, a]) for k = -N/2 .. N/2-1 (evaluated in float64)."""
    k = torch.arange(-(N // 2), N - N // 2, dtype=torch.float64, device=pos.device)
    ang = (2.0 * math.pi * sign) * pos.to(torch.float64).unsqueeze(-1) * k  # [n, d, N]
    return [torch.polar(torch.ones_like(ang[:, a]), ang[:, a]) for a in range(pos.shape[1])]


def _sets(batch, n):
    """[(set index, row selector)] -- one entry (0, all rows) without a batch vector."""
    if batch is None:
        return 1, [(0, slice(0, n))]
    B = int(batch.max().item()) + 1 if n > 0 else 1
    return B, [(b, (batch == b).nonzero(as_tuple=True)[0]) for b in range(B)]


_ADJ = {1: "ia,ic->ac", 2: "ia,ib,ic->abc", 3: "ia,ib,id,ic->abdc"}
_FWD = {1: "ia,ac->ic", 2: "ia,ib,abc->ic", 3: "ia,ib,id,abdc->ic"}


def ndft_adjoint(x, pos, batch=None, N=16):
    """y[b, k + N/2, ...] = sum_{i in set b} x[i, ...] exp(+2 pi i k.pos[i]);  returns ``[B, N, .., N, *cols]`` complex64."""
    n, d = pos.shape
    cols = tuple(x.shape[1:])
    xc = x.reshape(n, -1).to(torch.complex128)
    B, sets = _sets(batch, n)
    y = torch.zeros((B,) + (N,) * d + (xc.shape[1],), dtype=torch.complex128, device=pos.device)
    for b, rows in sets:
        p, v = pos[rows], xc[rows]
        for s in range(0, p.shape[0], _CHUNK):
            ph = _axis_phases(p[s:s + _CHUNK], N, +1.0)
            y[b] += torch.einsum(_ADJ[d], *ph, v[s:s + _CHUNK])
    return y.reshape((B,) + (N,) * d + cols).to(torch.complex64)


def ndft_forward(x, pos, batch=None):
    """y[i, ...] = sum_k x[batch[i], k + N/2, ...] exp(-2 pi i k.pos[i]);  returns ``[n, *cols]`` complex64."""
    n, d = pos.shape
    N = x.shape[1]
    cols = tuple(x.shape[1 + d:])
    xc = x.reshape((x.shape[0],) + (N,) * d + (-1,)).to(torch.complex128)
    y = torch.zeros((n, xc.shape[-1]), dtype=torch.complex128, device=pos.device)
    _, sets = _sets(batch, n)
    for b, rows in sets:
        p = pos[rows]
        out = []
        for s in range(0, p.shape[0], _CHUNK):
            ph = _axis_phases(p[s:s + _CHUNK], N, -1.0)
            out.append(torch.einsum(_FWD[d], *ph, xc[b]))
        if out:
            y[rows] = torch.cat(out, 0)
    return y.reshape((n,) + cols).to(torch.complex64)


def _resolve(sources, targets, source_batch, target_batch, batch):
    if batch is not None:
        source_batch = target_batch = batch
    if targets is None:
        targets, target_batch = sources, source_batch
    return targets, source_batch, target_batch


def ndft_fastsum(x, coeffs, sources, targets=None, source_batch=None, target_batch=None, batch=None, N=16):
    """Exact counterpart of ``nfft_fastsum``: forward_T(coeffs * adjoint_S(x)), real for real ``x``.
    (``N`` is accepted for signature compatibility; the bandwidth is ``coeffs.size(0)``.)"""
    targets, source_batch, target_batch = _resolve(sources, targets, source_batch, target_batch, batch)
    d = sources.shape[1]
    yhat = ndft_adjoint(x, sources, source_batch, N=coeffs.shape[0])
    w = coeffs.to(torch.complex64).reshape((1,) + tuple(coeffs.shape) + (1,) * (yhat.dim() - 1 - d))
    y = ndft_forward(yhat * w, targets, target_batch)
    return y if x.is_complex() else y.real


def _blocks(fn, sources, targets, source_batch, target_batch):
    if source_batch is None:
        return fn(sources, targets)
    B = int(source_batch.max().item()) + 1
    return torch.block_diag(*[fn(sources[source_batch == b], targets[target_batch == b]) for b in range(B)])


def exact_trigonometric_matrix(coeffs, sources, targets=None, source_batch=None, target_batch=None, /, batch=None):
    """A[i, j] = sum_k coeffs[k + N/2] exp(2 pi i k.(s_j - t_i)), block diagonal over the point sets."""
    targets, source_batch, target_batch = _resolve(sources, targets, source_batch, target_batch, batch)
    d, N = coeffs.dim(), coeffs.shape[0]
    c = coeffs.to(torch.complex128)

    def one(src, tgt):
        # sum_k c_k e^{+2 pi i k.s_j} e^{-2 pi i k.t_i}: contract the target phases with c, then the source phases
        pt = _axis_phases(tgt, N, -1.0)
        ps = _axis_phases(src, N, +1.0)
        if d == 1:
            return torch.einsum("ia,a,ja->ij", pt[0], c, ps[0]).to(torch.complex64)
        if d == 2:
            return torch.einsum("ia,ib,ab,ja,jb->ij", pt[0], pt[1], c, ps[0], ps[1]).to(torch.complex64)
        t = torch.einsum("ia,ib,id,abd->iabd", pt[0], pt[1], pt[2], c)
        return torch.einsum("iabd,ja,jb,jd->ij", t, ps[0], ps[1], ps[2]).to(torch.complex64)

    return _blocks(one, sources, targets, source_batch, target_batch)


def exact_gaussian_matrix(sigma, sources, targets=None, source_batch=None, target_batch=None, batch=None):
    """A[i, j] = exp(-|s_j - t_i|^2 / sigma^2), block diagonal over the point sets."""
    targets, source_batch, target_batch = _resolve(sources, targets, source_batch, target_batch, batch)

    def one(src, tgt):
        return torch.exp(-torch.cdist(tgt, src).square() / sigma ** 2)

    return _blocks(one, sources, targets, source_batch, target_batch)
