"""Exact non-equispaced DFT in float64 (numpy) -- TEST INFRASTRUCTURE ONLY.

Restates the equations of the reference's ground truth,
``/root/reference/torch_nfft/ndft.py``:

* ``ndft_adjoint``  (ndft.py:5-23):  y[b, k+N/2, c] = sum_{i: batch[i]=b} x[i,c] exp(+2 pi i k.pos[i])
* ``ndft_forward``  (ndft.py:26-44): y[i, c] = sum_{k in [-N/2,N/2)^d} x[batch[i], k+N/2, c] exp(-2 pi i k.pos[i])
* ``ndft_fastsum``  (ndft.py:48-62): forward(coeffs * adjoint(x)), real part if x is real

Differences from the reference file (deliberate, behaviour-preserving):
the contraction is chunked over points and done per axis (separable phases
exp(2 pi i k_d pos_d)), so memory is O(chunk * N) instead of 8 * N^d * n
bytes (ndft.py:14,36); any number of trailing columns is handled in one call
(the reference handles exactly the trailing layout of ``x`` too, via tensordot).
"""
import numpy as np

_LETTERS = "pqr"


def _phases(pos, N, sign):
    """[n, N] complex128 per axis: exp(sign * 2 pi i * k * pos[:, d]), k = -N/2..N/2-1 (ndft.py:10-14)."""
    k = np.arange(-N // 2, N // 2, dtype=np.float64)
    return [np.exp(sign * 2j * np.pi * pos[:, d:d + 1].astype(np.float64) * k[None, :])
            for d in range(pos.shape[1])]


def _batch_ranges(batch, n):
    if batch is None:
        return [(0, slice(0, n))]
    batch = np.asarray(batch)
    B = int(batch.max()) + 1 if batch.size else 0  # ndft.py:22
    return [(b, np.nonzero(batch == b)[0]) for b in range(B)]


def ndft_adjoint(x, pos, batch=None, N=16, chunk=4096):
    """x [n, *cols] real|complex, pos [n, d] -> y [B, N, ..., N, *cols] complex128."""
    pos = np.asarray(pos)
    x = np.asarray(x)
    n, d = pos.shape
    cols = x.shape[1:]
    xc = x.reshape(n, -1).astype(np.complex128)
    C = xc.shape[1]
    parts = _batch_ranges(batch, n)
    y = np.zeros((len(parts),) + (N,) * d + (C,), dtype=np.complex128)
    sub = ",".join("i" + _LETTERS[a] for a in range(d))
    expr = "ic," + sub + "->" + _LETTERS[:d] + "c"
    for b, sel in parts:
        pb, xb = pos[sel], xc[sel]
        for s in range(0, pb.shape[0], chunk):
            E = _phases(pb[s:s + chunk], N, +1.0)
            y[b] += np.einsum(expr, xb[s:s + chunk], *E, optimize=True)
    return y.reshape((len(parts),) + (N,) * d + cols)


def ndft_forward(x, pos, batch=None, chunk=4096):
    """x [B, N, ..., N, *cols], pos [n, d] -> y [n, *cols] complex128."""
    pos = np.asarray(pos)
    x = np.asarray(x)
    n, d = pos.shape
    N = x.shape[1]
    cols = x.shape[1 + d:]
    B = x.shape[0]
    xc = x.reshape((B,) + (N,) * d + (-1,)).astype(np.complex128)
    C = xc.shape[-1]
    y = np.zeros((n, C), dtype=np.complex128)
    sub = ",".join("i" + _LETTERS[a] for a in range(d))
    expr = _LETTERS[:d] + "c," + sub + "->ic"
    for b, sel in _batch_ranges(batch, n):
        idx = np.arange(n)[sel]
        for s in range(0, idx.shape[0], chunk):
            ii = idx[s:s + chunk]
            E = _phases(pos[ii], N, -1.0)
            y[ii] = np.einsum(expr, xc[b], *E, optimize=True)
    return y.reshape((n,) + cols)


def ndft_fastsum(x, coeffs, sources, targets=None, source_batch=None, target_batch=None, batch=None):
    """forward_T(coeffs * adjoint_S(x)); real part when x is real (ndft.py:48-62)."""
    if targets is None:
        targets, target_batch = sources, source_batch
    if batch is not None:
        source_batch = target_batch = batch
    x = np.asarray(x)
    coeffs = np.asarray(coeffs)
    N = coeffs.shape[0]
    d = coeffs.ndim
    y = ndft_adjoint(x, sources, source_batch, N=N)
    y = y * coeffs.reshape((1,) + coeffs.shape + (1,) * (y.ndim - 1 - d))
    y = ndft_forward(y, targets, target_batch)
    return y if np.iscomplexobj(x) else y.real


def ndft_adjoint_subset(x, pos, freqs):
    """Exact adjoint at selected frequency multi-indices only (size-independent spot check).

    freqs: int array [q, d] of signed frequencies k in [-N/2, N/2).  Returns [q, C] complex128.
    """
    pos = np.asarray(pos, dtype=np.float64)
    x = np.asarray(x)
    xc = x.reshape(pos.shape[0], -1).astype(np.complex128)
    out = np.zeros((freqs.shape[0], xc.shape[1]), dtype=np.complex128)
    for s in range(0, pos.shape[0], 1 << 16):
        ph = np.exp(2j * np.pi * (freqs.astype(np.float64) @ pos[s:s + (1 << 16)].T))
        out += ph @ xc[s:s + (1 << 16)]
    return out
